#!/bin/bash
# 8-wave tile variants: parity under each, micro A/B, whole-bench A/B.  usage: scripts/gpu_w8.sh <tag>
set -u
TAG=${1:-w8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q > $OUT/pytest_ops.txt 2>&1; tail -2 $OUT/pytest_ops.txt
for v in 6 14; do
  ASYRP_MAIN_TILE=$v timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q > $OUT/pytest_t$v.txt 2>&1; tail -2 $OUT/pytest_t$v.txt
done
ASYRP_MAIN_TILE=15 timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_t15.txt 2>&1; tail -2 $OUT/pytest_t15.txt
timeout 500 python scripts/conv_bench.py 32 > $OUT/conv_bench.txt 2>&1
grep -A 16 "8-wave tiles" $OUT/conv_bench.txt
for v in 0 6 14 15 0 15; do
  ASYRP_MAIN_TILE=$v timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_t$v.json 2>$OUT/bench_t$v.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_t$v.json").read().strip().splitlines()[-1])
    print("tile $v", round(d["value"],3), "img/s", round(d["roofline"]["achieved"],1), "TF dominant;", round(d["roofline"]["all_gemm_tflops"],1), "TF all gemm")
except Exception as e:
    print("tile $v failed", e)
PY
done
