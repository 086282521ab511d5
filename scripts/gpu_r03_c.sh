#!/bin/bash
# Round 3, visit C: GPU tests (split-plane attention), same-box A/Bs of the whole edit through the library's switches
# (ASYRP_ATTN=old, ASYRP_POLYPHASE=0), B=1 kernel-time / wall-time ratio under rocprofv3.  usage: scripts/gpu_r03_c.sh <tag>
set -u
TAG=${1:-r03c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $OUT/pytest.log
tail -6 $OUT/pytest.log
B="--steps 2 --warmup 1 --no-cpu-baseline --no-parity-check"
for rnd in 1 2; do
  (timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_default_$rnd.json
  (ASYRP_ATTN=old timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_attn_old_$rnd.json
  (ASYRP_POLYPHASE=0 timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/ab_polyphase_off_$rnd.json
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/ab_*.json")):
    try:
        r = json.load(open(f))
        att = r.get("roofline_attention", {})
        print(f.split("/")[-1], "images/s %.3f" % r["value"], "dual step ms %.2f" % r["phase_ms_per_step"]["generation_step_t>=t_edit(dual decoder)"],
              "attention TF %.1f frac %.3f share %.4f" % (att.get("achieved", 0), att.get("frac", 0), att.get("share_of_step", 0)), att.get("kernels"))
    except Exception as e:
        print(f, "ERR", e)
PY
# B=1: kernel time vs wall time
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b1 -o trace -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 5 --warmup 1 --no-cpu-baseline --no-parity-check --no-kernel-events > $OUT/bench_b1_under_rocprof.json 2> $OUT/prof_b1.err)
find $OUT/prof_b1 -name '*kernel_trace*' -size +1M -delete 2>/dev/null
python - <<PY
import csv, glob, json
f = glob.glob("$OUT/prof_b1/**/*kernel_stats.csv", recursive=True)
r = json.loads(open("$OUT/bench_b1_under_rocprof.json").read().strip().splitlines()[-1])
tot = sum(float(x["TotalDurationNs"]) for x in csv.DictReader(open(f[0]))) if f else 0
n_edits = r["steps"] + r["warmup"] + 9.0 / 79.0          # + the 9 phase-timing steps
print("B=1: ms per edit (wall) %.1f; kernel time per edit %.1f ms -> kernel/wall %.3f" % (r["ms_per_step"], tot / 1e6 / n_edits, tot / 1e6 / n_edits / r["ms_per_step"]))
PY
tail -3 $OUT/*.err
