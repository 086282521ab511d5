#!/bin/bash
# Round 3, visit A: parity tests (incl. the new reference fixtures, sampler kernel, tape ids, fast mode), the bench line with the
# dual-step CPU cross-check, B=1 latency, the fast-mode line and its numerics experiment.  usage: scripts/gpu_r03_a.sh <tag>
set -u
TAG=${1:-r03a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -30) > $OUT/pytest.log
tail -4 $OUT/pytest.log
(timeout 300 python bench.py --steps 2 --warmup 1 2> $OUT/bench.err | tail -1) > $OUT/bench_b32.json
cut -c1-400 $OUT/bench_b32.json
(timeout 200 python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2> $OUT/bench_b1.err | tail -1) > $OUT/bench_b1.json
cut -c1-300 $OUT/bench_b1.json
(timeout 300 python bench.py --conv-math f16 --steps 2 --warmup 1 2> $OUT/bench_f16.err | tail -1) > $OUT/bench_f16.json
cut -c1-400 $OUT/bench_f16.json
(timeout 400 python tests/experiments/fast_mode_numerics.py 2> $OUT/numerics.err | tail -1) > $OUT/fast_mode_numerics.json
cut -c1-600 $OUT/fast_mode_numerics.json
tail -3 $OUT/*.err
