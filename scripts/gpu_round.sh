#!/bin/bash
# One GPU-box visit: parity tests, a bench line, and a rocprofv3 kernel-trace summary of the same command.
# usage: scripts/gpu_round.sh <tag>   (outputs under gpurun_out/<tag>/)
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $OUT/pytest.log
(timeout 400 python bench.py --steps 1 --warmup 1 2> $OUT/bench.err | tail -3) > $OUT/bench.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_prof.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
find $OUT/prof -name '*kernel_trace*' -size +1M -delete 2>/dev/null
ls -R $OUT | head -30
cat $OUT/pytest.log | tail -5; cat $OUT/bench.json
