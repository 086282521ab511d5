#!/bin/bash
# Socket power and shader clock while the main tile runs back to back (K32 form, then the 32x32x16 form), sampled with rocm-smi.
# usage: scripts/gpu_power_probe.sh <tag>
set -u
TAG=${1:-power}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
(rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -E "GPU\[0\]" ) > $OUT/idle.txt
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
    (echo "--- tile $TILE sample $i"; rocm-smi --showpower --showclocks 2>&1 | grep -E "GPU\[0\]") >> $OUT/samples.txt
    sleep 1.5
  done
  wait $PID
done
cat $OUT/idle.txt $OUT/samples.txt; tail -2 $OUT/run_tile7.txt $OUT/run_tile6.txt
