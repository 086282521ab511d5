#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes of SQ / GRBM counters (scripts/gpu.sh pmc) per kernel family of one whole edit:
mean launch duration under the profiler, effective clock, matrix-pipe busy fraction, split of the wave cycles.
usage: pmc_summary.py <dir with pass*/> <library.so> <out.json>"""
import collections
import csv
import glob
import hashlib
import json
import sys

from traffic_summary import family


def k32_summary(root, out):
    """scripts/gpu.sh pmc-k32: the main tile in its two instruction forms (16x16x32 = "k32", 32x32x16 = "w8") on one layer."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in glob.glob(f"{root}/*/"):
        for f in glob.glob(d + "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                agg["k32" if "k32" in r["Kernel_Name"] else "w8"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(d + "*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                dur["k32" if "k32" in r["Kernel_Name"] else "w8"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    res = {}
    for k in agg:
        m = {c: sum(v) / len(v) for c, v in agg[k].items()}
        ms = sum(dur[k]) / max(1, len(dur[k]))
        m["mean_ms_under_profiler"] = ms
        if "GRBM_GUI_ACTIVE" in m:
            m["effective_clock_GHz"] = m["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9
            m["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (m["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
        res[k] = m
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


def calib_hbm_summary(root, out):
    """scripts/gpu.sh calib-hbm: FETCH_SIZE / WRITE_SIZE of scripts/calib/hbm_counters.hip against its known byte counts."""
    BYTES = 2 << 30
    res = {"bytes_per_launch": BYTES}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{root}/calib_{c}/*counter_collection.csv")
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            kb = sum(v) / len(v)
            res.setdefault(k, {})[c + "_KB"] = kb
            res[k][c + "_reported_over_true"] = kb * 1024.0 / BYTES
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


def main():
    if sys.argv[1] == "--k32":
        return k32_summary(sys.argv[2], sys.argv[3])
    if sys.argv[1] == "--calib-hbm":
        return calib_hbm_summary(sys.argv[2], sys.argv[3])
    root, lib, out = sys.argv[1:4]
    cnt = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    dur = collections.defaultdict(lambda: [0.0, 0])
    for d in sorted(glob.glob(f"{root}/pass*/")):
        for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = family(r.get("Kernel_Name") or "")
                c = cnt[k][r["Counter_Name"]]
                c[0] += float(r["Counter_Value"])
                c[1] += 1
        for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = family(r.get("Kernel_Name") or "")
                dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                dur[k][1] += 1
    fams = {}
    for k, cs in cnt.items():
        if "asyrp" not in k or not dur[k][1]:
            continue
        m = {c: v[0] / v[1] for c, v in cs.items()}          # per launch
        sec = dur[k][0] / dur[k][1]
        row = {"launches_seen": max(v[1] for v in cs.values()), "mean_launch_us_under_profiler": sec * 1e6}
        if "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8                   # summed over the 8 XCDs
            row["effective_clock_GHz"] = cyc / sec / 1e9
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                row["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)
        w = m.get("SQ_WAVE_CYCLES")
        if w:
            for c, name in (("SQ_WAIT_ANY", "wave_cycles_parked_frac"), ("SQ_WAIT_INST_ANY", "wave_cycles_issue_stall_frac"),
                            ("SQ_ACTIVE_INST_ANY", "wave_cycles_issuing_frac"), ("SQ_ACTIVE_INST_VALU", "wave_cycles_issuing_valu_frac"),
                            ("SQ_ACTIVE_INST_LDS", "wave_cycles_issuing_lds_frac")):
                if c in m:
                    row[name] = m[c] / w
        row["raw_per_launch"] = m
        fams[k] = row
    res = {"library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
           "collection": "rocprofv3 --kernel-trace --pmc <SQ / GRBM counters> over bench.py --steps 1 --warmup 0 (one whole edit), per family; "
                         "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES cycles summed "
                         "over SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs",
           "families": fams}
    json.dump(res, open(out, "w"), indent=1)
    for k, r in sorted(fams.items(), key=lambda kv: -kv[1]["mean_launch_us_under_profiler"] * kv[1]["launches_seen"]):
        print("%-66s n=%5d %8.1f us clk=%s mfma_busy=%s parked=%s stall=%s valu=%s" % (
            k[-66:], r["launches_seen"], r["mean_launch_us_under_profiler"],
            "%.2f" % r["effective_clock_GHz"] if "effective_clock_GHz" in r else "-",
            "%.3f" % r["mfma_busy_frac"] if "mfma_busy_frac" in r else "-",
            "%.2f" % r["wave_cycles_parked_frac"] if "wave_cycles_parked_frac" in r else "-",
            "%.2f" % r["wave_cycles_issue_stall_frac"] if "wave_cycles_issue_stall_frac" in r else "-",
            "%.2f" % r["wave_cycles_issuing_valu_frac"] if "wave_cycles_issuing_valu_frac" in r else "-"))


if __name__ == "__main__":
    main()
