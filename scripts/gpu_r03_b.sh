#!/bin/bash
# Round 3, visit B: full GPU tests (no -x), polyphase A/B micro-benchmark (bench library), whole-edit bench.  usage: scripts/gpu_r03_b.sh <tag>
set -u
TAG=${1:-r03b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $OUT/pytest.log
tail -6 $OUT/pytest.log
(timeout 300 python scripts/conv_bench.py 32 poly 2>&1 | tail -20) > $OUT/ab_polyphase.txt
cat $OUT/ab_polyphase.txt
(timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> $OUT/bench.err | tail -1) > $OUT/bench_b32.json
cut -c1-330 $OUT/bench_b32.json
(timeout 300 python bench.py --conv-math f16 --steps 3 --warmup 1 --no-cpu-baseline 2> $OUT/bench_f16.err | tail -1) > $OUT/bench_f16.json
cut -c1-330 $OUT/bench_f16.json
tail -3 $OUT/*.err
