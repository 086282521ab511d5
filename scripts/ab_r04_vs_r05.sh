#!/bin/bash
# Same-box, interleaved whole-edit A/B of the round-4 tree against the working tree (profiles/r05j_ab_whole_edit_r04_vs_r05_same_box.txt).
# The old tree is materialised next to the repo before the visit (it is git-ignored and removed afterwards):
#   mkdir -p gpurun_ab/r04 && git archive 8e52766 -- asyrp_official_amd bench.py oracle include __graft_entry__.py | tar -x -C gpurun_ab/r04
#   (cd gpurun_ab/r04 && mkdir -p profiles && python -m asyrp_official_amd.build)
#   gpurun --timeout 900 -- 'bash scripts/ab_r04_vs_r05.sh'
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05j; mkdir -p $OUT
one() {  # one <dir> <label> <i>
  (cd $1 && timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>>$OUT/err.txt | tail -1) > $OUT/$2_$3.json
  python - "$OUT/$2_$3.json" "$2" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("%-8s images/s %.3f  ms/step %.1f  main %.1f TFLOP/s  invariance %s" % (sys.argv[2], r["value"], r["ms_per_step"], r["roofline"]["achieved"], r["parity_check"]["batch_invariance_bitwise"]))
PY
}
for i in 1 2 3; do one $GRAFT_REPO_ROOT/gpurun_ab/r04 r04 $i; one $GRAFT_REPO_ROOT r05 $i; done
