#!/usr/bin/env python
"""Micro-benchmark of single conv configurations through the C ABI (asyrp_op_conv_bench): TFLOP/s per tile/variant,
and timing ablations of the main f16x3 tile.  usage: scripts/conv_bench.py [B]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from asyrp_official_amd import _lib

lib = _lib.load_bench()      # libasyrp_hip_bench.so (python -m asyrp_official_amd.build --bench): product kernels + bench hooks
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def run(H, C0, C1, Cout, k, stride=1, ups=0, pro=1, res=0, math="f16x3", tile=0, abl=0, iters=10, b=None):
    ms = C.c_float()
    b = b or B
    _lib.check(lib.asyrp_op_conv_bench(0, b, H, H, C0, C1, Cout, k, stride, ups, pro, res, _lib.CONV_MATH[math], tile, abl,
                                       iters, C.byref(ms), None))
    Ho = H * (2 if ups else 1) // stride
    fl = 2.0 * b * Ho * Ho * Cout * (C0 + C1) * k * k
    return ms.value, fl / (ms.value * 1e-3) / 1e12


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "one":       # a single configuration, for rocprofv3 --pmc runs
        for tile in (1, 6):
            ms, tf = run(256, 128, 0, 128, 3, tile=tile, iters=5)
            print(f"tile {tile} 128->128 @256 B={B}: {ms:.3f} ms {tf:.1f} TFLOP/s")
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "one7":      # single configuration on the K32 tile and on the 32x32x16 tile (PMC runs)
        for tile in (6, 7):
            ms, tf = run(256, 128, 0, 128, 3, tile=tile, iters=5)
            print(f"tile {tile} 128->128 @256 B={B}: {ms:.3f} ms {tf:.1f} TFLOP/s")
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "k32abl":    # ablations of the K32 tile + tile choice for the 32^2 / 16^2 layers
        print(f"-- ablations of the K32 tile, B={B} (2 = no weight LDS-DMA, 4 = no matrix instructions, 8 = no activation staging)")
        for (H, Ch, C1) in ((256, 128, 0), (256, 128, 128), (64, 256, 0)):
            for abl in (32, 32 | 2, 32 | 8, 32 | 2 | 8, 32 | 4, 32 | 4 | 2 | 8):
                ms, tf = run(H, Ch, C1, Ch, 3, tile=7, abl=abl, iters=6)
                print(f"  {Ch}+{C1}->{Ch} @{H} abl={abl & 31:2d}: {ms:8.3f} ms {tf:7.1f}", flush=True)
        print("-- tile choice for the 32^2 / 16^2 layers: engine's choice (0) vs 128x128 (2) vs the K32 main tile (7), interleaved")
        tiles = (0, 2, 7)
        for (H, Ch, C1, Co) in ((32, 256, 0, 256), (32, 256, 256, 256), (16, 512, 0, 512), (16, 512, 512, 512)):
            r = {t: [] for t in tiles}
            for rnd in range(5):
                for t in tiles:
                    r[t].append(run(H, Ch, C1, Co, 3, tile=t, iters=8)[1])
            med = {t: sorted(r[t])[2] for t in tiles}
            print(f"  {Ch}+{C1}->{Co} @{H}: " + "  ".join(f"t{t} {med[t]:5.1f}" for t in tiles), flush=True)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "k32half":   # 16^2 layers: engine's choice vs 128x128 (2) vs the 128-pixel K32 form (8)
        tiles = (2, 8, 7)
        print(f"-- A/B interleaved, B={B}: 128x128 tile on 32x32x16 (2) vs 128-pixel K32 (8) vs 256-pixel K32 (7)")
        for (H, Ch, C1, Co, kw) in ((16, 512, 0, 512, {}), (16, 512, 512, 512, {}), (16, 512, 0, 512, dict(res=1)),
                                    (32, 256, 0, 256, {}), (32, 256, 256, 256, {}), (8, 512, 0, 512, {})):
            r = {t: [] for t in tiles}
            for rnd in range(5):
                for t in tiles:
                    r[t].append(run(H, Ch, C1, Co, 3, tile=t, iters=10, **kw)[1])
            med = {t: sorted(r[t])[2] for t in tiles}
            print(f"  {Ch}+{C1}->{Co} @{H} {kw}: " + "  ".join(f"t{t} {med[t]:5.1f}" for t in tiles), flush=True)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "k32small":  # 8x8 layers and the stride-2 convs on their K32 forms
        print(f"-- A/B interleaved, B={B}: 8x8 layers: engine's choice without K32 (tile 4 / split-K via tile 0*) vs the 8x8-patch K32 form (9)")
        for (H, Ch, C1, Co, kw) in ((8, 512, 0, 512, {}), (8, 512, 512, 512, {}), (8, 512, 0, 512, dict(res=1))):
            tiles = (0, 4, 9)
            r = {t: [] for t in tiles}
            for rnd in range(5):
                for t in tiles:
                    r[t].append(run(H, Ch, C1, Co, 3, tile=t, iters=10, **kw)[0] * 1e3)
            med = {t: sorted(r[t])[2] for t in tiles}
            print(f"  {Ch}+{C1}->{Co} @{H} {kw}: " + "  ".join(f"t{t} {med[t]:6.1f} us" for t in tiles), flush=True)
        print("-- stride 2 (Downsample, no prologue): 32x32x16 64x128 tile (3) vs the launcher's choice (0 = K32 stride-2 form)")
        for (H, Ch) in ((256, 128), (128, 128), (64, 256), (32, 256), (16, 512)):
            tiles = (3, 0)
            r = {t: [] for t in tiles}
            for rnd in range(5):
                for t in tiles:
                    r[t].append(run(H, Ch, 0, Ch, 3, stride=2, pro=0, tile=t, iters=8)[1])
            med = {t: sorted(r[t])[2] for t in tiles}
            print(f"  {Ch}->{Ch} @{H}->{H // 2}: " + "  ".join(f"t{t} {med[t]:5.1f}" for t in tiles), flush=True)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "poly":      # nearest x2 + 3x3: the 3x3 form over the virtual up-sampling (0) vs polyphase (11)
        print(f"-- A/B interleaved, B={B}: up-sampled 3x3 conv, launcher's 3x3 form (tile 0) vs the polyphase form (tile 11); us per launch")
        for math in ("f16x3", "f16"):
            for (H, Ch, pro) in ((128, 128, 0), (64, 256, 0), (32, 256, 0), (128, 128, 1), (64, 256, 1)):
                tiles = (0, 11)
                r = {t: [] for t in tiles}
                for rnd in range(5):
                    for t in tiles:
                        r[t].append(run(H, Ch, 0, Ch, 3, ups=1, pro=pro, tile=t, math=math, iters=6)[0] * 1e3)
                med = {t: sorted(r[t])[2] for t in tiles}
                print(f"  {math:5s} {Ch}->{Ch} @{H}->{2 * H} prologue={pro}: " + "  ".join(f"t{t} {med[t]:7.1f} us" for t in tiles) +
                      f"   ratio {med[11] / med[0]:.3f}", flush=True)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "small":     # the 16x16 / 8x8 layers on the launcher's own choice; used for the A/Bs recorded under profiles/rd3d_*, rd3e_*
        print(f"-- B={B}, launcher's choice: us per launch, TFLOP/s (median of 5 x 10 launches)")
        for math in ("f16x3", "f16"):
            for (H, Ch, C1, Co, kw) in ((8, 512, 0, 512, {}), (8, 512, 0, 512, dict(res=1)), (16, 512, 0, 512, {}), (16, 512, 512, 512, {}),
                                        (16, 256, 0, 512, {}), (16, 512, 0, 512, dict(res=1))):
                r = sorted(run(H, Ch, C1, Co, 3, math=math, iters=10, **kw) for _ in range(5))[2]
                print(f"  {math:5s} {Ch}+{C1}->{Co} @{H} {kw}: {r[0] * 1e3:7.1f} us {r[1]:6.1f} TFLOP/s", flush=True)
            for (H, Ch) in ((256, 128), (64, 256), (16, 512)):
                r = sorted(run(H, Ch, 0, Ch, 3, stride=2, pro=0, math=math, iters=8) for _ in range(5))[2]
                print(f"  {math:5s} stride 2 {Ch}->{Ch} @{H}->{H // 2}: {r[0] * 1e3:7.1f} us {r[1]:6.1f} TFLOP/s", flush=True)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "quad":      # 8x8 layers: run once with ASYRP_QUAD8=0 and once with the default
        print(f"-- B={B}, ASYRP_QUAD8={os.environ.get('ASYRP_QUAD8', 'default')}: 8x8 layers on the launcher's choice (incl. the reduce launch); us per layer")
        for math in ("f16x3", "f16"):
            for (Ch, C1, Co, kw) in ((512, 0, 512, {}), (512, 0, 512, dict(res=1)), (512, 512, 512, {}), (1024, 0, 1024, {})):
                r = sorted(run(8, Ch, C1, Co, 3, math=math, iters=10, **kw) for _ in range(5))[2]
                print(f"  {math:5s} {Ch}+{C1}->{Co} @8 {kw}: {r[0] * 1e3:7.1f} us {r[1]:6.1f} TFLOP/s", flush=True)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "onebyone":  # 1x1 GEMMs of the attention blocks: tile sweep (2 = 128x128, 3 = 64x128, 4 = 64x64, 1 = 256x128)
        print(f"-- B={B}: 1x1 GEMMs at 16x16 / 8x8, us per launch by tile (0 = launcher's choice)")
        for (H, Ci, Co, pro) in ((16, 512, 1536, 1), (16, 512, 512, 0), (16, 768, 256, 0), (8, 512, 1536, 1), (8, 512, 512, 0), (8, 1024, 512, 0), (32, 512, 1536, 1), (32, 512, 512, 0),
                                 (32, 512, 256, 0), (64, 384, 128, 0), (256, 256, 128, 0)):
            tiles = (0, 1, 2, 3, 4, 15, 16)      # 15 / 16: the barrier-free kernel of gemm1x1.hip (256 / 128 pixels)
            r = {t: [] for t in tiles}
            for rnd in range(5):
                for t in tiles:
                    try:
                        r[t].append(run(H, Ci, 0, Co, 1, pro=pro, res=(0 if pro else 1), tile=t, iters=10)[0] * 1e3)
                    except Exception:
                        r[t].append(float("nan"))
            print(f"  {Ci}->{Co} @{H}: " + "  ".join(f"t{t} {sorted(r[t])[2]:6.1f}" for t in tiles), flush=True)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "ab67":      # interleaved A/B: 8-wave tile on 32x32x16 (6) vs on 16x16x32 (7)
        tiles = (6, 7)
        print(f"-- A/B interleaved, B={B}: 8-wave 256x128 tile on v_mfma_f32_32x32x16_f16 (6) vs v_mfma_f32_16x16x32_f16 (7)")
        for (H, Ch, C1, Co, kw) in ((256, 128, 0, 128, {}), (256, 128, 0, 128, dict(res=1)), (256, 128, 128, 128, {}),
                                    (128, 128, 0, 128, {}), (128, 128, 128, 128, {}), (64, 256, 0, 256, {}),
                                    (64, 256, 256, 256, {}), (128, 128, 0, 128, dict(ups=1, pro=0)), (32, 256, 0, 256, {})):
            r = {t: [] for t in tiles}
            for rnd in range(5):
                for t in tiles:
                    r[t].append(run(H, Ch, C1, Co, 3, tile=t, iters=6, **kw)[1])
            med = {t: sorted(r[t])[2] for t in tiles}
            print(f"  {Ch}+{C1}->{Co} @{H} {kw}: " + "  ".join(f"t{t} {med[t]:5.1f} ({med[t] / med[6]:.3f})" for t in tiles),
                  flush=True)
        sys.exit(0)
    print(f"B={B}")
    layers = [("down.0 conv 128->128 @256", 256, 128, 0, 128, 3, {}),
              ("up.0 conv1 256->128 @256 (concat)", 256, 128, 128, 128, 3, {}),
              ("up.0 nin 1x1 256->128 @256", 256, 128, 128, 128, 1, dict(pro=0)),
              ("conv2+resid 128->128 @256", 256, 128, 0, 128, 3, dict(res=1)),
              ("256->256 @64", 64, 256, 0, 256, 3, {}),
              ("512->256 @64", 64, 256, 256, 256, 3, {}),
              ("128->128 @128", 128, 128, 0, 128, 3, {}),
              ("256->256 @32", 32, 256, 0, 256, 3, {}),
              ("512->512 @16", 16, 512, 0, 512, 3, {}),
              ("1024->512 @16", 16, 512, 512, 512, 3, {}),
              ("512->512 @8", 8, 512, 0, 512, 3, {}),
              ("1024->512 @8", 8, 512, 512, 512, 3, {}),
              ("upsample 128->128 128->256", 128, 128, 0, 128, 3, dict(ups=1, pro=0)),
              ("downsample 128->128 256->128", 256, 128, 0, 128, 3, dict(stride=2, pro=0)),
              ("conv_out 128->3 @256", 256, 128, 0, 3, 3, {}),
              ("conv_in 3->128 @256", 256, 3, 0, 128, 3, dict(pro=0)),
              ("qkv 1x1 512->1536 @16", 16, 512, 0, 1536, 1, dict(pro=1)),
              ]
    for name, H, C0, C1, Co, k, kw in layers:
        for math in ("f16x3",):
            ms, tf = run(H, C0, C1, Co, k, math=math, **kw)
            print(f"{name:40s} {math:6s} {ms:8.3f} ms {tf:7.1f} TFLOP/s", flush=True)
    print("-- 8x8 layers: engine choice (split-K 8 + reduce) vs single-pass 64x64 tile")
    for (Ch, C1) in ((512, 0), (512, 512)):
        for rnd in range(3):
            a = run(8, Ch, C1, 512, 3, tile=0, iters=10)
            b_ = run(8, Ch, C1, 512, 3, tile=4, iters=10)
            print(f"  {Ch}+{C1}->512 @8 r{rnd}: split-K {a[0]*1e3:7.1f} us {a[1]:6.1f} TF   single {b_[0]*1e3:7.1f} us {b_[1]:6.1f} TF", flush=True)
    print("-- A/B interleaved: 4-wave pipelined tile (1) vs 8-wave plain-loop tile (6)")
    tiles = (1, 6)
    for (H, Ch, C1, Co, kw) in ((256, 128, 0, 128, {}), (256, 128, 0, 128, dict(res=1)), (256, 128, 128, 128, {}),
                                (128, 128, 0, 128, {}), (64, 256, 0, 256, {}), (64, 256, 256, 256, {}),
                                (128, 128, 0, 128, dict(ups=1, pro=0)), (32, 256, 0, 256, {})):
        r = {t: [] for t in tiles}
        for rnd in range(5):
            for t in tiles:
                r[t].append(run(H, Ch, C1, Co, 3, tile=t, iters=6, **kw)[1])
        med = {t: sorted(r[t])[2] for t in tiles}
        print(f"  {Ch}+{C1}->{Co} @{H} {kw}: " + "  ".join(f"t{t} {med[t]:5.1f} ({med[t] / med[1]:.3f})" for t in tiles), flush=True)
    print("-- ablations of the 8-wave tile (128->128 @256): abl mask -> TFLOP/s-equivalent (2 = no weight DMA, 8 = no activation loads/staging)")
    for abl in (32, 32 | 8, 32 | 2, 32 | 2 | 8):
        ms, tf = run(256, 128, 0, 128, 3, tile=6, abl=abl, iters=6)
        print(f"  abl={abl & 31:2d}: {ms:8.3f} ms {tf:7.1f}", flush=True)
    print("-- tile sweep, 128->128 @64 and 512->512 @16 (f16x3)")
    for (H, Ch) in ((256, 128), (64, 256), (16, 512), (8, 512), (32, 256)):
        for tile in (1, 2, 3, 4, 6):
            try:
                ms, tf = run(H, Ch, 0, Ch, 3, tile=tile)
                print(f"  {Ch}->{Ch} @{H} tile {tile}: {ms:8.3f} ms {tf:7.1f} TFLOP/s", flush=True)
            except Exception as e:
                print("  tile", tile, "failed", e)
    print("-- ablations of the main tile (128->128 @256, prologue on): abl mask -> ms")
    for abl in (0, 32, 2 | 8 | 32):
        ms, tf = run(256, 128, 0, 128, 3, tile=1, abl=abl)
        print(f"  abl={abl & 31:2d} (profiling build={bool(abl)}): {ms:8.3f} ms {tf:7.1f} TFLOP/s-equivalent", flush=True)
