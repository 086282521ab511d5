#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes of SQ / GRBM counters (scripts/gpu_pmc_families.sh) per kernel family of one whole edit:
mean launch duration under the profiler, effective clock, matrix-pipe busy fraction, split of the wave cycles.
usage: pmc_summary.py <dir with pass*/> <library.so> <out.json>"""
import collections
import csv
import glob
import hashlib
import json
import sys

from traffic_summary import family


def main():
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
