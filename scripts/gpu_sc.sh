#!/bin/bash
# parity of the fused-shortcut path + kernel-trace stats of one bench step (ratio fused-shortcut / plain launches of the main tile)
set -u
TAG=${1:-sc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_prof.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
find $OUT/prof -name '*kernel_trace*' -size +1M -delete 2>/dev/null
head -4 $OUT/prof/trace_kernel_stats.csv | cut -c1-200
python -c "
import json;d=json.loads(open('$OUT/bench_prof.json').read().strip().splitlines()[-1]);print(d['value'],'img/s (profiled)')"
timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'],'img/s', d['roofline']['achieved'], d['roofline']['all_gemm_tflops'])"
