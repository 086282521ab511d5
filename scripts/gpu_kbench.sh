#!/bin/bash
# kernel micro-benchmarks + PMC counters of the main conv tile.  usage: scripts/gpu_kbench.sh <tag>
set -u
TAG=${1:-kb}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/conv_bench.py 32 > $OUT/conv_bench.txt 2>&1
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc1 -o p -- python $GRAFT_REPO_ROOT/scripts/conv_bench.py 32 one > $GRAFT_REPO_ROOT/$OUT/pmc1.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name '*.csv' -size +2M -delete
cat $OUT/conv_bench.txt
