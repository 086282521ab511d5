#!/bin/bash
# kernel-trace stats of the other BASELINE configs (one edit step each).  usage: scripts/gpu_cfgprof.sh <tag>
set -u
TAG=${1:-cfg}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in afhq imagenet; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$cfg -o trace -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_$cfg.json 2> $GRAFT_REPO_ROOT/$OUT/prof_$cfg.err)
  find $OUT/prof_$cfg -name '*kernel_trace*' -size +1M -delete 2>/dev/null
  echo "== $cfg"; head -16 $OUT/prof_$cfg/trace_kernel_stats.csv | cut -d, -f1-5 | cut -c1-150
done
tail -c 600 $OUT/bench_afhq.json
