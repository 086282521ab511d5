#!/bin/bash
# round 4, visit g: rolling fragment schedule of the K32 matrix step (inline-asm LDS reads, counted waits): tests, phases, bench
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04g
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -q -x 2>&1 | tail -15) > $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 300 python scripts/k32_phases.py 32 all > $OUT/k32_phases.txt 2> $OUT/err.txt
grep -v "^    CUs\|^    workgroup start" $OUT/k32_phases.txt; tail -3 $OUT/err.txt
timeout 200 python scripts/conv_bench.py 32 k32abl > $OUT/k32abl.txt 2>> $OUT/err.txt
cat $OUT/k32abl.txt
B="--steps 2 --warmup 1 --no-cpu-baseline"
(timeout 200 python bench.py $B 2>> $OUT/ab.err | tail -1) > $OUT/base.json
python - <<PY
import json
for n in ("base",):
    try:
        r = json.load(open("$OUT/%s.json" % n))
        print(n, "images/s %.3f" % r["value"], [(x["kernel"][-34:], x["launches_per_step"], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r["kernel_families"][:12]])
        print(json.dumps(r.get("parity_check"))[:400])
    except Exception as e:
        print(n, "FAILED", e)
PY
tail -5 $OUT/ab.err
