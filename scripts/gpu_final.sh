#!/bin/bash
# End-of-round evidence: parity tests, bench line, rocprofv3 kernel stats of the same command, HBM traffic (PMC) of the
# dominant kernel over a bench step, SQ counters of the main tile on the micro-benchmark.  usage: scripts/gpu_final.sh <tag>
set -u
TAG=${1:-final}
bash scripts/gpu_round.sh $TAG
bash scripts/gpu_traffic.sh ${TAG}_traffic > gpurun_out/$TAG/traffic.log 2>&1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-include-regex 'igemm_f16x3_kernel<asyrp::XCfg<4, (1, 2, 4|2, 2, 2), 3, 1, 2, 1>, true' --output-format csv -d $OUT/pmc_sq -o p -- python $GRAFT_REPO_ROOT/scripts/conv_bench.py 32 one > $OUT/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/conv_bench.py 32 > $OUT/conv_bench.txt 2>&1
find gpurun_out/$TAG gpurun_out/${TAG}_traffic -name '*.csv' -size +1M -delete
cat gpurun_out/${TAG}_traffic/traffic_summary.json
