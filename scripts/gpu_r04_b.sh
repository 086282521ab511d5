#!/bin/bash
# round 4, visit b: main-tile phase stamps + K sweep (fixed cost per tile, measured)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/k32_phases.py 32 all > $OUT/k32_phases.txt 2> $OUT/err.txt
cat $OUT/k32_phases.txt; tail -3 $OUT/err.txt
