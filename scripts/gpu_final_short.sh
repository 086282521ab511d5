#!/bin/bash
# Short evidence refresh on one MI355X box: parity tests, the DRIVER's bench command, rocprofv3 kernel stats of that command, smoke.
# (scripts/gpu_final_r02.sh adds PMC traffic, counter calibration and the other configs.)  usage: scripts/gpu_final_short.sh <tag>
set -u
TAG=${1:-final_short}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $OUT/pytest.log
tail -3 $OUT/pytest.log
(timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench_driver_cmd.err | tail -1) > $OUT/bench_driver_cmd.json
cut -c1-300 $OUT/bench_driver_cmd.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check > $OUT/bench_under_rocprof.json 2> $OUT/prof.err)
find $OUT/prof -name '*kernel_trace*' -size +1M -delete 2>/dev/null
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
tail -3 $OUT/smoke.log
find gpurun_out/$TAG -name '*.csv' -size +1M -delete
