#!/usr/bin/env python
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (one directory per counter) into per-kernel-family HBM bytes per
launch, keyed by the kernel names bench.py prints, stamped with the sha256 of the library that ran.
usage: scripts/traffic_summary.py <dir with FETCH_SIZE/ and WRITE_SIZE/> <libasyrp_hip.so> <out.json> <workload description>"""
import collections
import csv
import glob
import hashlib
import json
import re
import sys


def family(name):
    """rocprofv3 kernel name -> the short family name of bench.py's kernel_families (asyrp_official_amd/engine.py variant_name)."""
    m = re.search(r"igemm_f16x3_k32_kernel<asyrp::K32Cfg<([\w, ]+)>", name)
    if m:
        p = [v.strip() for v in m.group(1).split(",")]
        dflt = [None, None, "16", "1", "3", "false"]
        while len(p) > 2 and p[-1] == dflt[len(p) - 1]:
            p.pop()
        return "asyrp::igemm_f16x3_k32_kernel<asyrp::K32Cfg<%s>>" % ", ".join(p)
    m = re.search(r"igemm_f16x3_kernel<asyrp::XCfg<([\d, ]+)>", name)
    if m:
        p = [v.strip() for v in m.group(1).split(",")][:6]
        return "asyrp::igemm_f16x3_kernel<asyrp::XCfg<%s>>" % ", ".join(p)
    m = re.search(r"igemm_f32_kernel<asyrp::TileCfg<([\d, ]+)>", name)
    if m:
        return "asyrp::igemm_f32_kernel<asyrp::TileCfg<%s>>" % ", ".join(v.strip() for v in m.group(1).split(",")[:6])
    m = re.search(r"attn_planes_kernel<(\d+)", name)   # one row per key-tile count (T = 256 -> 2, T = 64 -> 1): the two sites move different bytes
    if m:
        return "asyrp::attn_planes_kernel<%s>" % m.group(1)
    m = re.search(r"asyrp::(\w+)", name)
    return "asyrp::" + m.group(1) if m else name


def main():
    root, lib, out, workload = sys.argv[1:5]
    per = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(f"{root}/{c}/**/*counter_collection.csv", recursive=True)
        for f in files:
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != c:
                    continue
                k = family(r.get("Kernel_Name") or r.get("Kernel Name") or "")
                per[k][c][0] += float(r["Counter_Value"])
                per[k][c][1] += 1
    # unit / gfx950 corrections as calibrated (profiles/r02zz_calib_hbm_counters.json): FETCH_SIZE and WRITE_SIZE count KB;
    # FETCH_SIZE reports 0.5x the bytes of the kernels' 16-B-per-lane reads, WRITE_SIZE 1.0x
    fam = {}
    for k, v in per.items():
        nf, nw = v["FETCH_SIZE"][1], v["WRITE_SIZE"][1]
        if not nf or not nw:
            continue
        fk, wk = v["FETCH_SIZE"][0] / nf, v["WRITE_SIZE"][0] / nw
        fam[k] = {"launches": nf, "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                  "hbm_bytes_per_launch": 1024.0 * (2.0 * fk + 1.0 * wk)}
    res = {"library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(), "workload": workload,
           "collection": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over every kernel "
                         "(scripts/gpu.sh traffic); bytes = 1024 x (2 x FETCH_SIZE + WRITE_SIZE), scales calibrated in "
                         "profiles/r02zz_calib_hbm_counters.json",
           "families": dict(sorted(fam.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["families"].items():
        print(f"{k[-70:]:70s} n={v['launches']:6d} {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch")


if __name__ == "__main__":
    main()
