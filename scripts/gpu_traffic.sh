#!/bin/bash
# HBM traffic of the dominant conv kernel over one whole bench step (PMC, separate passes; guide: FETCH_SIZE costs 3 TCC
# slots, WRITE_SIZE 2 -- they cannot share a pass).  usage: scripts/gpu_traffic.sh <tag>
set -u
TAG=${1:-traffic}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
RX='igemm_f16x3_k32_kernel<asyrp::K32Cfg<8, 2>'   # both instantiations (plain and fused shortcut) of the main tile
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "$RX" --output-format csv -d $OUT/$C -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-parity-check > $OUT/$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, json, glob, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/$TAG/%s/*counter_collection.csv" % c)
    if not f: continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c]
    res[c] = {"launches": len(vals), "mean_per_launch_KB": sum(vals) / max(1, len(vals))}
json.dump(res, open("gpurun_out/$TAG/traffic_summary.json", "w"), indent=1)
print(res)
PY
find $OUT -name '*.csv' -size +1M -delete
