#!/bin/bash
# End-of-round evidence on one MI355X box: parity tests, the DRIVER's bench command, rocprofv3 kernel stats of that command, HBM traffic
# (PMC, separate passes) of EVERY kernel family on the shipped library, the fast-mode line with its own traffic, the other
# BASELINE configs, B=1.  usage: scripts/gpu_evidence.sh <tag>   (profiles/<tag>_* are then copied by hand)
set -u
TAG=${1:-evidence}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $OUT/pytest.log
tail -3 $OUT/pytest.log
# traffic first: bench.py pastes it only when its stamp matches the loaded library
bash scripts/gpu_traffic_families.sh ${TAG}_traffic_f16x3 > $OUT/traffic_f16x3.log 2>&1
cp gpurun_out/${TAG}_traffic_f16x3/traffic_families.json profiles/traffic_families_celeba_b32_f16x3.json 2>/dev/null
bash scripts/gpu_traffic_families.sh ${TAG}_traffic_f16 --conv-math f16 > $OUT/traffic_f16.log 2>&1
cp gpurun_out/${TAG}_traffic_f16/traffic_families.json profiles/traffic_families_celeba_b32_f16.json 2>/dev/null
cp profiles/traffic_families_celeba_b32_*.json $OUT/ 2>/dev/null
tail -12 $OUT/traffic_f16x3.log
(timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench_driver_cmd.err | tail -1) > $OUT/bench_driver_cmd.json
cut -c1-300 $OUT/bench_driver_cmd.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check > $OUT/bench_under_rocprof.json 2> $OUT/prof.err)
find $OUT/prof -name '*kernel_trace*' -size +1M -delete 2>/dev/null
(timeout 400 python bench.py --conv-math f16 --steps 10 --warmup 3 2> $OUT/bench_f16.err | tail -1) > $OUT/bench_fastmode_f16.json
cut -c1-300 $OUT/bench_fastmode_f16.json
for cfg in afhq imagenet church; do
  (timeout 300 python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline 2> $OUT/bench_$cfg.err | tail -1) > $OUT/bench_$cfg.json
  cut -c1-200 $OUT/bench_$cfg.json
done
(timeout 200 python bench.py --batch 1 --steps 5 --warmup 1 --no-cpu-baseline 2> $OUT/bench_b1.err | tail -1) > $OUT/bench_b1.json
cut -c1-200 $OUT/bench_b1.json
(ASYRP_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 --batch 4 --no-kernel-events --no-cpu-baseline 2> $OUT/bench_selflaunch_2rank_gloo_dryrun.err | tail -1) > $OUT/bench_selflaunch_2rank_gloo_dryrun.json
cut -c1-200 $OUT/bench_selflaunch_2rank_gloo_dryrun.json
find gpurun_out/$TAG gpurun_out/${TAG}_traffic_f16x3 gpurun_out/${TAG}_traffic_f16 -name '*.csv' -size +1M -delete
