#!/bin/bash
# round 4, visit d: run-to-run determinism of one UNet evaluation: this build vs the merged-only build
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for e in "X=1" "PROBE_LIB=$GRAFT_REPO_ROOT/scripts/calib/libasyrp_hip_merged_only.so"; do
  (env $e timeout 120 python scripts/batch_invariance_probe.py 32 2>&1 | grep -v amdgpu.ids) >> $OUT/probe2.txt
done
cat $OUT/probe2.txt
