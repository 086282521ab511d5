#!/bin/bash
# Round-2 evidence on one MI355X box: parity tests, the DRIVER's bench command, rocprofv3 kernel stats of that command (without
# the B=1 parity-check edits and the CPU baseline, which run after the timed region and would dilute the per-kernel averages),
# HBM traffic (PMC, separate passes) of the dominant kernel, counter calibration, the other configs, the self-launch path.
# usage: scripts/gpu_final_r02.sh <tag>
set -u
TAG=${1:-r02final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $OUT/pytest.log
tail -3 $OUT/pytest.log
(timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench_driver_cmd.err | tail -1) > $OUT/bench_driver_cmd.json
cut -c1-300 $OUT/bench_driver_cmd.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check > $OUT/bench_under_rocprof.json 2> $OUT/prof.err)
find $OUT/prof -name '*kernel_trace*' -size +1M -delete 2>/dev/null
ls $OUT/prof
bash scripts/gpu_traffic.sh ${TAG}_traffic > $OUT/traffic.log 2>&1
cat gpurun_out/${TAG}_traffic/traffic_summary.json
bash scripts/gpu_calib_hbm.sh ${TAG}_calib > $OUT/calib.log 2>&1
for cfg in afhq imagenet; do
  (timeout 300 python bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline 2> $OUT/bench_$cfg.err | tail -1) > $OUT/bench_$cfg.json
done
# the launcher path `python bench.py --gpus 2` (self-launch) as a DRY RUN on this 1-GPU box: gloo, ranks share the device
(ASYRP_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 --batch 4 --no-kernel-events --no-cpu-baseline 2> $OUT/bench_selflaunch_2rank_gloo_dryrun.err | tail -1) > $OUT/bench_selflaunch_2rank_gloo_dryrun.json
cut -c1-400 $OUT/bench_afhq.json $OUT/bench_imagenet.json $OUT/bench_selflaunch_2rank_gloo_dryrun.json
find gpurun_out/$TAG gpurun_out/${TAG}_traffic gpurun_out/${TAG}_calib -name '*.csv' -size +1M -delete
