#!/bin/bash
# round 4, visit i: compile-time A/B of where a K step requests its weight pieces / halo loads (d = DMA mid-step, a = halo loads mid-step)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rnd in 1 2 3; do
for v in head d0a0 d1a0; do
  ASYRP_BENCH_LIB=$GRAFT_REPO_ROOT/scripts/calib/libbench_$v.so timeout 120 python scripts/k32_layers.py 32 2>/dev/null >> $OUT/layers.txt
done
done
K32_LAYERS_HEADER=1 ASYRP_BENCH_LIB=$GRAFT_REPO_ROOT/scripts/calib/libbench_d0a0.so timeout 120 python scripts/k32_layers.py 32 2>/dev/null | tail -1 >> $OUT/layers.txt
cat $OUT/layers.txt
