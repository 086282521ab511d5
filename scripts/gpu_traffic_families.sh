#!/bin/bash
# HBM traffic (PMC) of EVERY kernel family over one bench step of the shipped library: FETCH_SIZE and WRITE_SIZE in separate
# passes with --kernel-trace only (guide: they cannot share a pass).  usage: scripts/gpu_traffic_families.sh <tag> [bench args]
set -u
TAG=${1:-traffic_fam}
shift || true
ARGS=${*:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-parity-check $ARGS > $OUT/$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scripts/traffic_summary.py $OUT asyrp_official_amd/libasyrp_hip.so $OUT/traffic_families.json \
  "bench.py --steps 1 --warmup 0 --no-parity-check $ARGS (one whole edit + the 9 phase-timing steps), every kernel" | head -30
find $OUT -name '*.csv' -size +1M -delete
