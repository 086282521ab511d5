#!/usr/bin/env python
"""Phase breakdown of igemm_f16x3_k32_kernel's main tile (profiling library, ablation instantiation): where a workgroup's time
goes, how long a CU slot stays empty between two workgroups, and the fixed cost per tile from a K sweep at a fixed grid.
usage: scripts/k32_phases.py [B]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from asyrp_official_amd import _lib

lib = _lib.load_bench()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
TILE = int(os.environ.get("K32_TILE", "7"))
F16X3 = _lib.CONV_MATH["f16x3"]


def stamps(H, C0, C1, Cout, abl=32 | 64, res=0, iters=3, tile=TILE):
    nwg = B * ((H + 15) // 16) ** 2 * ((Cout + 127) // 128)
    st = np.zeros((nwg, 8), dtype=np.uint64)
    ms = C.c_float()
    _lib.check(lib.asyrp_op_conv_stamps(0, B, H, H, C0, C1, Cout, 3, 1, 0, 1, res, F16X3, tile, abl, iters, C.byref(ms),
                                        st.ctypes.data_as(C.c_void_p), nwg))
    return ms.value, st


def report(H, C0, C1, Cout, **kw):
    ms, st = stamps(H, C0, C1, Cout, **kw)
    s = st.astype(np.int64)
    fl = 2.0 * B * H * H * Cout * (C0 + C1) * 9
    steps = (C0 + C1) // 16 * 9 // 2
    t0 = s[:, 0].min()
    us = lambda a: a / 100.0
    print(f"== {C0}+{C1}->{Cout} @{H} B={B} {kw}: {ms * 1e3:.1f} us per launch, {fl / (ms * 1e-3) / 1e12:.1f} TFLOP/s, {len(s)} workgroups, {steps} K steps")
    names = ["prologue (start -> first tile staged)", "K loop", "epilogue stores issued", "stats / end", "store drain (end -> vmcnt 0)"]
    segs = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 6)]
    for n, (a, b) in zip(names, segs):
        d = us(s[:, b] - s[:, a])
        print(f"    {n:40s} mean {d.mean():7.2f} us  p10 {np.percentile(d, 10):7.2f}  p50 {np.percentile(d, 50):7.2f}  p90 {np.percentile(d, 90):7.2f}")
    life = us(s[:, 6] - s[:, 0])
    print(f"    workgroup life mean {life.mean():.2f} us; K loop per step {us(s[:, 2] - s[:, 1]).mean() / steps:.3f} us; span {us(s[:, 6].max() - t0):.1f} us")
    # CU slots: key = (xcc, se, sh?, cu) from HW_ID; two workgroups per CU -> pair every start with the latest earlier end on that CU
    hw = st[:, 5]
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
    hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    gaps = []
    per_cu = []
    for k in np.unique(key):
        idx = np.where(key == k)[0]
        per_cu.append(len(idx))
        starts = np.sort(s[idx, 0])
        ends = np.sort(s[idx, 6])
        # the i-th start (i >= 2) fills the slot freed by the (i-2)-th end (two resident workgroups)
        for i in range(2, len(starts)):
            gaps.append(us(starts[i] - ends[i - 2]))
    gaps = np.array(gaps)
    print(f"    CUs seen {len(np.unique(key))}, workgroups per CU min {min(per_cu)} max {max(per_cu)}; slot refill gap (end of a workgroup -> start of the next on "
          f"that CU) mean {gaps.mean():.2f} us p10 {np.percentile(gaps, 10):.2f} p50 {np.percentile(gaps, 50):.2f} p90 {np.percentile(gaps, 90):.2f}")
    first = us(np.sort(s[:, 0]) - t0)
    print(f"    workgroup start times (us) at quantiles 1/6/12/50 %: " + " ".join(f"{first[int(q * (len(first) - 1))]:.1f}" for q in (0.01, 0.06, 0.12, 0.5)))
    return ms, st


def run(H, C0, C1, Cout, tile=TILE, abl=0, iters=8, res=0):
    ms = C.c_float()
    _lib.check(lib.asyrp_op_conv_bench(0, B, H, H, C0, C1, Cout, 3, 1, 0, 1, res, F16X3, tile, abl, iters, C.byref(ms), None))
    return ms.value


def cu_key(st):
    hw = st[:, 5]
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
    hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
    return (((xcc * 8 + ((hwid >> 13) & 0x7)) * 2 + ((hwid >> 12) & 0x1)) * 16 + ((hwid >> 8) & 0xF)), hwid & 0xF, (hwid >> 4) & 3


def overlap_report(st):
    """How much of a workgroup's K loop runs beside its CU partner's K loop (stamps 1..2), beside the partner's other phases, and
    the offset between the starts of the two workgroups resident on a CU (0 = lockstep, half a life = fully staggered)."""
    s = st.astype(np.int64)
    key, _, _ = cu_key(st)
    both, offs = [], []
    for k in np.unique(key):
        idx = np.where(key == k)[0]
        idx = idx[np.argsort(s[idx, 0])]
        iv = [(s[i, 1], s[i, 2]) for i in idx]
        for n, i in enumerate(idx):
            a, b = iv[n]
            cov = 0
            for m in range(max(0, n - 3), min(len(idx), n + 4)):
                if m != n:
                    cov += max(0, min(b, iv[m][1]) - max(a, iv[m][0]))
            both.append(cov / max(1, b - a))
            if n >= 1:
                life = max(1, s[i, 6] - s[i, 0])
                offs.append(((s[i, 0] - s[idx[n - 1], 0]) % life) / life)
    both, offs = np.array(both), np.array(offs)
    print(f"    K-loop time spent beside the partner's K loop: mean {both.mean():.2f} p10 {np.percentile(both, 10):.2f} p90 {np.percentile(both, 90):.2f}; "
          f"start offset to the previous workgroup on the CU / own life: p10 {np.percentile(offs, 10):.2f} p50 {np.percentile(offs, 50):.2f} p90 {np.percentile(offs, 90):.2f}")


def dispatch_map(H=256, C0=128, C1=0, Cout=128):
    """Which linear workgroup ids share a CU in the first dispatch round, and which wave slots (HW_ID.WAVE_ID) they got."""
    ms, st = stamps(H, C0, C1, Cout, abl=32 | 64 | 128)     # stagger mode 3 with 0 us: records the arrival order in slot 7
    key, wave_id, simd = cu_key(st)
    arr = st[:, 7].astype(np.int64)
    n = min(len(st), 768)
    first = {}
    for i in range(n):
        first.setdefault(int(key[i]), []).append((i, int(arr[i]), int(wave_id[i]), int(simd[i])))
    pairs = [v for v in first.values() if len(v) >= 2]
    d = np.array([v[1][0] - v[0][0] for v in pairs])
    print(f"-- dispatch map, first {n} linear ids: {len(first)} CUs; linear-id distance between the first two workgroups of a CU: "
          f"min {d.min()} p50 {int(np.median(d))} max {d.max()}; examples {[v[:3] for v in list(first.values())[:4]]}")
    lt512 = [(a, w) for v in first.values() for (i, a, w, _) in v if i < 512]
    a0 = [w for a, w in lt512 if a == 0]
    a1 = [w for a, w in lt512 if a == 1]
    print(f"   ids < 512: arrival 0 -> wave ids {sorted(set(a0))}, arrival 1 -> wave ids {sorted(set(a1))}, arrivals >= 2: {sum(1 for a, w in lt512 if a >= 2)}; "
          f"ids in [256, 512) with arrival 1: {sum(1 for v in first.values() for (i, a, w, _) in v if 256 <= i < 512 and a == 1)} of 256")


def timed_loop(fn, seconds=2.5):
    from gpuclk import Sampler
    vals = []
    with Sampler() as sm:
        t0 = time.time()
        while time.time() - t0 < seconds:
            vals.append(fn())
    return float(np.median(vals)), sm


if __name__ == "__main__":
    import time
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    what = sys.argv[2] if len(sys.argv) > 2 else "all"
    if what == "map":
        dispatch_map()
        sys.exit(0)
    if what == "stagger":      # ablation instantiation with stamps: second workgroup of every CU delayed once (by arrival order)
        for kw in ({}, {"abl": 32}, {"res": 1}):
            for us_ in (0, 30, 45, 60):
                k = dict(kw)
                base = k.pop("abl", 32 | 64)
                print(f"## stagger {us_} us", flush=True)
                ms, st = report(256, 128, 0, 128, abl=base | ((128 | (us_ << 8)) if us_ else 0), **k)
                overlap_report(st)
        for us_ in (0, 85):
            print(f"## stagger {us_} us", flush=True)
            ms, st = report(256, 128, 128, 128, abl=32 | 64 | ((128 | (us_ << 8)) if us_ else 0))
            overlap_report(st)
        sys.exit(0)
    if what == "clk":          # sclk / power per variant of the same instantiation (is the 2.19 vs 2.48 us/step a clock effect?)
        for name, kw in (("stats", dict(abl=32 | 64)), ("no stats", dict(abl=32)), ("stats + residual", dict(abl=32 | 64, res=1)),
                         ("product kernel, stats", dict(abl=64)), ("product kernel, stats + residual", dict(abl=64, res=1))):
            ms, sm = timed_loop(lambda: run(256, 128, 0, 128, iters=20, **kw))
            print(f"   128->128 @256 {name:34s}: {ms * 1e3:8.1f} us per launch {2.0 * B * 65536 * 128 * 128 * 9 / (ms * 1e-3) / 1e12:6.1f} TFLOP/s; {sm.summary()}", flush=True)
        sys.exit(0)
    if what == "prod":         # product kernel, median of 5 x 12 launches per shape (same-box A/B lines across library builds / environments)
        print("-- product kernel")
        for name, a, kw in (("128->128 @256 stats", (256, 128, 0, 128), dict(abl=64)), ("128->128 @256 plain", (256, 128, 0, 128), {}),
                            ("128->128 @256 stats+res", (256, 128, 0, 128), dict(abl=64, res=1)), ("256->128 @256 stats", (256, 128, 128, 128), dict(abl=64)),
                            ("128->128 @128 stats+res", (128, 128, 0, 128), dict(abl=64, res=1)), ("256->256 @64 stats", (64, 256, 0, 256), dict(abl=64)),
                            ("512->256 @64 stats", (64, 256, 256, 256), dict(abl=64))):
            r = sorted(run(*a, iters=12, **kw) for _ in range(5))[2]
            fl = 2.0 * B * a[0] * a[0] * a[3] * (a[1] + a[2]) * 9
            print(f"   {name:28s}: {r * 1e3:8.1f} us {fl / (r * 1e-3) / 1e12:6.1f} TFLOP/s", flush=True)
        sys.exit(0)
    if what in ("all", "sweep"):
        print(f"-- K sweep at a fixed grid (Cout = 128 @256^2, B={B}, product kernel): time = a + b * steps")
        xs, ys = [], []
        for cin in (32, 64, 96, 128, 192, 256, 384, 512):
            r = sorted(run(256, cin, 0, 128) for _ in range(3))[1]
            steps = cin // 16 * 9 // 2
            fl = 2.0 * B * 65536 * 128 * cin * 9
            xs.append(steps); ys.append(r * 1e3)
            print(f"   Cin {cin:4d} steps {steps:4d}: {r * 1e3:8.1f} us {fl / (r * 1e-3) / 1e12:6.1f} TFLOP/s", flush=True)
        b, a = np.polyfit(xs, ys, 1)
        print(f"   fit: {a:.1f} us + {b:.3f} us/step -> fixed cost = {a / b:.2f} steps per launch of 16 tile rounds; asymptote {2.0 * B * 65536 * 128 * 32 * 9 / 2 / (b * 1e-6) / 1e12 * 2:.1f} TFLOP/s")
        print(f"-- the same at @64^2 Cout = 256 (the 64^2 level: 2 N blocks)")
        xs, ys = [], []
        for cin in (64, 128, 256, 512, 768):
            r = sorted(run(64, cin, 0, 256) for _ in range(3))[1]
            steps = cin // 16 * 9 // 2
            fl = 2.0 * B * 4096 * 256 * cin * 9
            xs.append(steps); ys.append(r * 1e3)
            print(f"   Cin {cin:4d} steps {steps:4d}: {r * 1e3:8.1f} us {fl / (r * 1e-3) / 1e12:6.1f} TFLOP/s", flush=True)
        b, a = np.polyfit(xs, ys, 1)
        print(f"   fit: {a:.1f} us + {b:.3f} us/step -> fixed cost = {a / b:.2f} steps")
    if what in ("all", "stamps"):
        report(256, 128, 0, 128)
        report(256, 128, 0, 128, abl=32)
        report(256, 128, 0, 128, res=1)
        report(256, 128, 128, 128)
        report(64, 256, 0, 256)
