#!/bin/bash
# kernel micro-benchmarks + PMC counters of the main conv tile.  usage: scripts/gpu_kbench.sh <tag>
set -u
TAG=${1:-kb}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/conv_bench.py 32 > $OUT/conv_bench.txt 2>&1
cd /tmp

cd $GRAFT_REPO_ROOT
find $OUT -name '*.csv' -size +2M -delete
cat $OUT/conv_bench.txt
