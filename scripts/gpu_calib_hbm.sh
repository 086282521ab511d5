#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts (scripts/calib/hbm_counters.hip); separate --pmc passes, kernel-trace only.
# usage: scripts/gpu_calib_hbm.sh <tag>  -> gpurun_out/<tag>/calib_hbm_counters.json
set -u
TAG=${1:-calib}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/scripts/calib/hbm_counters.hip -o /tmp/hbm_counters || exit 1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/calib_$C -o p -- /tmp/hbm_counters > $OUT/calib_$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, json, collections
BYTES = 2 << 30
res = {"bytes_per_launch": BYTES}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/$TAG/calib_%s/*counter_collection.csv" % c)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        kb = sum(v) / len(v)
        res.setdefault(k, {})[c + "_KB"] = kb
        res[k][c + "_reported_over_true"] = kb * 1024.0 / BYTES
json.dump(res, open("gpurun_out/$TAG/calib_hbm_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
