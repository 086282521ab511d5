#!/bin/bash
# prep branch, visit f: 2-way K split for the 32 x 32 layers on the 256-pixel form (ASYRP_SPLITK32=1), same-box A/B
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4prep_f
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/.wt/r4prep
B="--steps 1 --warmup 0 --no-cpu-baseline --no-parity-check"
(timeout 40 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/sk32_off.json
(ASYRP_SPLITK32=1 timeout 40 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/sk32_on.json
python - <<PY
import json
for n in ("sk32_off", "sk32_on"):
    r = json.load(open("$OUT/%s.json" % n))
    print(n, "images/s %.3f" % r["value"], [(x["kernel"][-34:], x["launches_per_step"], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r["kernel_families"][:3]])
PY
