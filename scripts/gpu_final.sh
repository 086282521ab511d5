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
# other BASELINE configs on one GPU, and a 2-rank dry run of the launcher path (ranks share the GPU, gloo; not a perf number)
for cfg in afhq imagenet; do
  (timeout 300 python bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline 2> $OUT/bench_$cfg.err | tail -1) > $OUT/bench_$cfg.json
done
(ASYRP_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 0 --batch 4 --no-kernel-events 2> $OUT/bench_2rank_dryrun.err | tail -1) > $OUT/bench_2rank_dryrun.json
cat $OUT/bench_afhq.json $OUT/bench_imagenet.json $OUT/bench_2rank_dryrun.json | cut -c1-400
