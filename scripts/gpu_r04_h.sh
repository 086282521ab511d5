#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 200 python scripts/conv_bench.py 32 k32abl 2> $OUT/err.txt | head -28 > $OUT/k32abl.txt
cat $OUT/k32abl.txt; tail -2 $OUT/err.txt
