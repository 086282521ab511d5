#!/bin/bash
# round 4, visit a: merged binary (main + next/round4-prep): GPU suite, short bench, A/B of the two unmeasured switches
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04a
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > $OUT/pytest.log
tail -3 $OUT/pytest.log
B="--steps 2 --warmup 1 --no-cpu-baseline --no-parity-check"
(timeout 100 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/base.json
(ASYRP_SPLITK32=1 timeout 100 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/sk32_on.json
(timeout 100 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/base2.json
(timeout 100 python bench.py --config afhq $B 2>> $OUT/ab.err | tail -1) > $OUT/afhq_off.json
(ASYRP_CONV_OUT6=1 timeout 100 python bench.py --config afhq $B 2>> $OUT/ab.err | tail -1) > $OUT/afhq_out6.json
python - <<PY
import json
for n in ("base", "sk32_on", "base2", "afhq_off", "afhq_out6"):
    try:
        r = json.load(open("$OUT/%s.json" % n))
        print(n, "images/s %.3f" % r["value"], [(x["kernel"][-34:], x["launches_per_step"], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r["kernel_families"][:8]])
    except Exception as e:
        print(n, "FAILED", e)
PY
tail -5 $OUT/ab.err
