#!/bin/bash
# Everything that runs on the GPU box, as ONE parameterised script (round 5: the twelve gpu_*.sh of rounds 1-4 folded into it).
#   usage: scripts/gpu.sh <what> <tag> [args]          outputs under gpurun_out/<tag>/
#   e.g.   gpurun --timeout 900 -- 'bash scripts/gpu.sh visit r05x'
# what:
#   visit     [bench args]   the whole GPU suite (-x), then a short bench of the line of record
#   round                    suite + bench + rocprofv3 --kernel-trace --stats of the bench command
#   evidence                 END-OF-ROUND evidence: suite, HBM traffic of every kernel family (PMC, own passes), the DRIVER's bench command,
#                            rocprofv3 kernel stats of it, the fast-mode line, the other BASELINE configs, B = 1, the 2-rank gloo dry run
#   traffic   [bench args]   HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, --kernel-trace only) of every kernel family
#   pmc                      SQ / GRBM counters of every kernel family over one edit (matrix-pipe busy, where the wave cycles go)
#   pmc-k32                  the same for the main tile in its two instruction forms on one layer
#   power                    socket power / sclk (rocm-smi) while the main tile runs back to back
#   ab        "<ENV=..>" ... same-box interleaved A/B of kernel switches on the line of record (PROFILING build: ASYRP_LIBRARY=bench)
#   kbench                   scripts/conv_bench.py micro-benchmarks
#   cfgprof                  rocprofv3 kernel stats of the AFHQ / ImageNet configs
#   calib-hbm                FETCH_SIZE / WRITE_SIZE against known byte counts (scripts/calib/hbm_counters.hip)
#   r5-calib                 round 5: loop-shape / tile-skeleton calibrations, dispatch map, forced stagger, sclk per variant
#   r5-convin-attn           round 5: conv_in MFMA A/B, attention phase stamps, op / UNet tests
set -u
WHAT=${1:?what}; TAG=${2:?tag}; shift 2 || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
LIB=asyrp_official_amd/libasyrp_hip.so

suite() { (timeout 900 python -m pytest tests -m gpu -q "$@" 2>&1 | grep -v "amdgpu.ids" | tail -40) > $OUT/pytest.log; tail -6 $OUT/pytest.log; }
bench_line() {   # bench_line <file> <bench args...>
  local f=$1; shift
  (timeout 600 python bench.py "$@" 2>> $OUT/bench.err | tail -1) > $f
  python - "$f" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    print("images/s %.3f main %.1f TFLOP/s" % (r["value"], r.get("roofline", {}).get("achieved", 0)),
          [(x["kernel"][-30:], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r.get("kernel_families", [])[:8]])
    print(json.dumps(r.get("parity_check"))[:300])
except Exception as e:
    print("bench FAILED", e)
PY
}
kernel_stats() {   # kernel_stats <dir> <bench args...>   rocprofv3 --kernel-trace --stats of a bench command
  local d=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o trace -- python $GRAFT_REPO_ROOT/bench.py "$@" > $d.json 2> $d.err)
  find $d -name '*kernel_trace*' -size +1M -delete 2>/dev/null
}
traffic() {   # traffic <tag> [bench args]   -> gpurun_out/<tag>/traffic_families.json
  local t=$1; shift; local o=$GRAFT_REPO_ROOT/gpurun_out/$t; mkdir -p $o
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $o/$C -o p -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-parity-check --no-other-configs "$@" > $o/$C.log 2>&1)
  done
  python scripts/traffic_summary.py $o $LIB $o/traffic_families.json \
    "bench.py --steps 1 --warmup 0 --no-parity-check $* (one whole edit + the 9 phase-timing steps), every kernel" | head -30
  find $o -name '*.csv' -size +1M -delete
}

case $WHAT in
visit)
  suite -x -s
  bench_line $OUT/bench.json --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs "$@" ;;
round)
  suite
  bench_line $OUT/bench.json --steps 1 --warmup 1
  kernel_stats $OUT/prof --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs ;;
evidence)
  suite
  # traffic first: bench.py pastes it only when its stamp matches the loaded library
  traffic ${TAG}_traffic_f16x3 > $OUT/traffic_f16x3.log 2>&1
  cp gpurun_out/${TAG}_traffic_f16x3/traffic_families.json profiles/traffic_families_celeba_b32_f16x3.json 2>/dev/null
  traffic ${TAG}_traffic_f16 --conv-math f16 > $OUT/traffic_f16.log 2>&1
  cp gpurun_out/${TAG}_traffic_f16/traffic_families.json profiles/traffic_families_celeba_b32_f16.json 2>/dev/null
  cp profiles/traffic_families_celeba_b32_*.json $OUT/ 2>/dev/null
  tail -14 $OUT/traffic_f16x3.log
  bench_line $OUT/bench_driver_cmd.json --steps 20 --warmup 5
  kernel_stats $OUT/prof --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-other-configs
  mv $OUT/prof.json $OUT/bench_under_rocprof.json 2>/dev/null
  bench_line $OUT/bench_fastmode_f16.json --conv-math f16 --steps 10 --warmup 3 --no-other-configs
  for cfg in afhq imagenet church; do bench_line $OUT/bench_$cfg.json --config $cfg --steps 2 --warmup 1 --no-cpu-baseline; done
  bench_line $OUT/bench_b1.json --batch 1 --steps 5 --warmup 1 --no-cpu-baseline
  (ASYRP_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 --batch 4 --no-kernel-events --no-cpu-baseline 2> $OUT/bench_selflaunch_2rank_gloo_dryrun.err | tail -1) > $OUT/bench_selflaunch_2rank_gloo_dryrun.json
  cut -c1-200 $OUT/bench_selflaunch_2rank_gloo_dryrun.json
  find gpurun_out/$TAG gpurun_out/${TAG}_traffic_f16x3 gpurun_out/${TAG}_traffic_f16 -name '*.csv' -size +1M -delete ;;
evidence-core)      # the three files the driver's line is judged with: traffic (PMC), the driver's command, its rocprofv3 kernel stats
  traffic ${TAG}_traffic_f16x3 > $OUT/traffic_f16x3.log 2>&1
  cp gpurun_out/${TAG}_traffic_f16x3/traffic_families.json profiles/traffic_families_celeba_b32_f16x3.json 2>/dev/null
  cp profiles/traffic_families_celeba_b32_f16x3.json $OUT/ 2>/dev/null
  tail -14 $OUT/traffic_f16x3.log
  bench_line $OUT/bench_driver_cmd.json --steps 20 --warmup 5
  kernel_stats $OUT/prof --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-other-configs
  mv $OUT/prof.json $OUT/bench_under_rocprof.json 2>/dev/null
  find gpurun_out/$TAG gpurun_out/${TAG}_traffic_f16x3 -name '*.csv' -size +1M -delete ;;
traffic)
  traffic $TAG "$@" ;;
pmc)
  P=0
  for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
    P=$((P+1))
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$P -o p -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-parity-check --no-other-configs > $OUT/pass$P.log 2>&1)
    tail -n 2 $OUT/pass$P.log | cut -c1-200
  done
  python scripts/pmc_summary.py $OUT $LIB $OUT/pmc_families.json
  find $OUT -name '*.csv' -size +1M -delete ;;
pmc-k32)
  for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    N=$(echo $C | tr ' ' '_')
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex 'igemm_f16x3' --output-format csv -d $OUT/$N -o p -- \
      python $GRAFT_REPO_ROOT/scripts/conv_bench.py 32 one7 > $OUT/$N.log 2>&1)
  done
  python scripts/pmc_summary.py --k32 $OUT $OUT/summary.json
  find $OUT -name '*.csv' -size +1M -delete ;;
power)
  (rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -E "GPU\[" ) > $OUT/idle.txt
  for TILE in 7 6; do
    python - $TILE > $OUT/run_tile$TILE.txt 2>&1 <<'PY' &
import sys, time
sys.path.insert(0, "scripts")
tile = int(sys.argv[1])
sys.argv[1:] = ["32"]          # conv_bench reads its batch from argv
import conv_bench as cb
t0 = time.time()
while time.time() - t0 < 14.0:
    ms, tf = cb.run(256, 128, 128, 128, 3, tile=tile, iters=40)
    print(f"tile {tile}: {ms:.3f} ms {tf:.1f} TFLOP/s", flush=True)
PY
    PID=$!
    sleep 5
    for i in 1 2 3 4; do
      (echo "--- tile $TILE sample $i"; python scripts/gpuclk.py | tail -1; rocm-smi --showpower --showclocks 2>&1 | grep -E "GPU\[") >> $OUT/samples.txt
      sleep 1.5
    done
    wait $PID
  done
  cat $OUT/idle.txt $OUT/samples.txt; for f in $OUT/run_tile7.txt $OUT/run_tile6.txt; do tail -n 2 $f; done ;;
ab)
  export ASYRP_LIBRARY=bench      # the product library reads no ASYRP_* switch
  i=0
  for rnd in 1 2; do
    for e in "X=0" "$@"; do
      i=$((i+1))
      echo "== $e"
      (env $e timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>> $OUT/err.txt | tail -1) > $OUT/run_$i.json
      python - "$OUT/run_$i.json" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    print("images/s %.3f  invariance %s " % (r["value"], r.get("parity_check", {}).get("batch_invariance_bitwise")), [(x["kernel"][-22:], round(x["tflops"], 1), round(x["share_of_step"], 4)) for x in r["kernel_families"][:7]])
except Exception as ex:
    print("FAILED", ex)
PY
    done
  done
  tail -3 $OUT/err.txt | grep -v amdgpu.ids ;;
kbench)
  timeout 300 python scripts/conv_bench.py 32 "$@" > $OUT/conv_bench.txt 2>&1
  cat $OUT/conv_bench.txt ;;
cfgprof)
  for cfg in afhq imagenet; do
    kernel_stats $OUT/prof_$cfg --config $cfg --steps 1 --warmup 0 --no-cpu-baseline
    echo "== $cfg"; head -16 $OUT/prof_$cfg/trace_kernel_stats.csv | cut -d, -f1-5 | cut -c1-150
  done ;;
calib-hbm)
  hipcc --offload-arch=gfx950 -O3 scripts/calib/hbm_counters.hip -o /tmp/hbm_counters || exit 1
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/calib_$C -o p -- /tmp/hbm_counters > $OUT/calib_$C.log 2>&1)
  done
  python scripts/pmc_summary.py --calib-hbm $OUT $OUT/calib_hbm_counters.json ;;
r5-calib)
  python scripts/gpuclk.py > $OUT/gpuclk_probe.txt 2>&1
  (timeout 200 scripts/calib/bin/loop_shapes_r5) > $OUT/calib_loop_shapes_r5.txt 2>&1
  (timeout 300 scripts/calib/bin/tile_shapes_r5) > $OUT/calib_tile_shapes_r5.txt 2>&1
  (TILE_SHAPES_STAGING=1 timeout 300 scripts/calib/bin/tile_shapes_r5) > $OUT/calib_tile_staging_ablations.txt 2>&1
  for M in map stagger clk prod; do (timeout 300 python scripts/k32_phases.py 32 $M) 2>&1 | grep -v amdgpu.ids > $OUT/k32_$M.txt; done
  tail -12 $OUT/calib_tile_shapes_r5.txt; grep -E "^##|per launch|beside" $OUT/k32_stagger.txt; cat $OUT/k32_clk.txt ;;
r5-convin-attn)
  (timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_fast_mode.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -15) > $OUT/pytest_ops_unet.log
  tail -6 $OUT/pytest_ops_unet.log
  (timeout 200 python scripts/attn_phases.py 2>&1 | grep -v amdgpu.ids) > $OUT/attn_phases.txt
  head -14 $OUT/attn_phases.txt
  export ASYRP_LIBRARY=bench
  for M in 1 0; do echo "== ASYRP_CONV_IN_MFMA=$M"; ASYRP_CONV_IN_MFMA=$M bench_line $OUT/bench_cin$M.json --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check; done ;;
*)
  echo "unknown: $WHAT"; exit 2 ;;
esac
