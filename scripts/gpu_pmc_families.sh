#!/bin/bash
# SQ / GRBM counters of EVERY kernel family over one whole edit on the shipped library: effective clock, matrix-pipe busy fraction,
# where the wave cycles go.  Two passes (counters in their own runs, --kernel-trace only).  usage: scripts/gpu_pmc_families.sh <tag>
set -u
TAG=${1:-pmc_fam}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P=0
for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
  P=$((P+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$P -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-parity-check > $OUT/pass$P.log 2>&1
  tail -n 2 $OUT/pass$P.log | cut -c1-200
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $OUT asyrp_official_amd/libasyrp_hip.so $OUT/pmc_families.json
find $OUT -name '*.csv' -size +1M -delete
