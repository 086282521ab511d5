#!/bin/bash
# SQ / GRBM counters of the main tile in its two forms (32x32x16 = tile 6, 16x16x32 = tile 7) on one layer, one box:
# effective clock (GRBM_GUI_ACTIVE / 8 XCDs / wall) and matrix-pipe busy fraction.  usage: scripts/gpu_pmc_k32.sh <tag>
set -u
TAG=${1:-pmc_k32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex 'igemm_f16x3' --output-format csv -d $OUT/$N -o p -- \
    python $GRAFT_REPO_ROOT/scripts/conv_bench.py 32 one7 > $OUT/$N.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in glob.glob("gpurun_out/$TAG/*/"):
    for f in glob.glob(d + "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = "k32" if "k32" in r["Kernel_Name"] else "w8"
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k = "k32" if "k32" in r["Kernel_Name"] else "w8"
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
out = {}
for k in agg:
    m = {c: sum(v) / len(v) for c, v in agg[k].items()}
    ms = sum(dur[k]) / max(1, len(dur[k]))
    m["mean_ms_under_profiler"] = ms
    if "GRBM_GUI_ACTIVE" in m:
        clk = m["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3)
        m["effective_clock_GHz"] = clk / 1e9
        m["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (m["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
    out[k] = m
json.dump(out, open("gpurun_out/$TAG/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name '*.csv' -size +1M -delete
