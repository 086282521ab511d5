#!/bin/bash
# round 4, visit c: float4 epilogue of the K32 kernel: GPU suite, phase stamps, short bench
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04c
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 300 python scripts/k32_phases.py 32 all > $OUT/k32_phases.txt 2> $OUT/err.txt
cat $OUT/k32_phases.txt; tail -3 $OUT/err.txt
B="--steps 2 --warmup 1 --no-cpu-baseline"
(timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/base.json
python - <<PY
import json
for n in ("base",):
    try:
        r = json.load(open("$OUT/%s.json" % n))
        print(n, "images/s %.3f" % r["value"], [(x["kernel"][-34:], x["launches_per_step"], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r["kernel_families"][:12]])
        print(json.dumps(r.get("parity_check"))[:600])
    except Exception as e:
        print(n, "FAILED", e)
PY
tail -5 $OUT/ab.err
