#!/bin/bash
# Same-box, interleaved whole-edit A/B of three trees (profiles/r06e_ab_whole_edit_r05_vs_r06_same_box.txt):
#   r05     the round-5 tree (b8ffadd)
#   r06pre  round 6 before the timestep rows were hoisted out of the loops (its kernels are round 5's)
#   r06     the working tree (timestep embedding + every block's Linear(swish(temb)) of ALL steps in two launches per edit)
# The old trees are materialised next to the repo before the visit (git-ignored, removed afterwards):
#   mkdir -p gpurun_ab/r05 && git archive b8ffadd -- asyrp_official_amd bench.py oracle include __graft_entry__.py | tar -x -C gpurun_ab/r05
#   (cd gpurun_ab/r05 && mkdir -p profiles && python -m asyrp_official_amd.build)          (same for r06pre from its commit)
#   gpurun --timeout 1200 -- 'bash scripts/ab_r06.sh'
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06e; mkdir -p $OUT
one() {  # one <dir> <label> <i> <extra args>
  (cd $1 && timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $4 2>>$OUT/err.txt | tail -1) > $OUT/$2_$3.json
  python - "$OUT/$2_$3.json" "$2" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("%-8s images/s %.3f  ms/step %.1f  main %.1f TFLOP/s  invariance %s" % (sys.argv[2], r["value"], r["ms_per_step"], r["roofline"]["achieved"], r["parity_check"]["batch_invariance_bitwise"]))
PY
}
for i in 1 2 3; do
  one $GRAFT_REPO_ROOT/gpurun_ab/r05 r05 $i ""
  one $GRAFT_REPO_ROOT/gpurun_ab/r06pre r06pre $i "--no-other-configs"
  one $GRAFT_REPO_ROOT r06 $i "--no-other-configs"
done
