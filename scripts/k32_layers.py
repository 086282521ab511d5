#!/usr/bin/env python
"""Product-kernel timings of the layers that carry the edit, through the profiling library's bench hook (abl = 0: the product
instantiation).  ASYRP_BENCH_LIB selects an A/B build.  usage: scripts/k32_layers.py [B]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from asyrp_official_amd import _lib

lib = _lib.load_bench()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F16X3 = _lib.CONV_MATH["f16x3"]


def run(H, C0, C1, Cout, res=0, tile=0, iters=6, ups=0, stride=1):
    ms = C.c_float()
    _lib.check(lib.asyrp_op_conv_bench(0, B, H, H, C0, C1, Cout, 3, stride, ups, 1 if stride == 1 else 0, res, F16X3, tile, 0, iters, C.byref(ms), None))
    Ho = H * (2 if ups else 1) // stride
    return 2.0 * B * Ho * Ho * Cout * (C0 + C1) * 9 / (ms.value * 1e-3) / 1e12


LAYERS = [("128->128 @256", dict(H=256, C0=128, C1=0, Cout=128)), ("128->128 @256 +res", dict(H=256, C0=128, C1=0, Cout=128, res=1)),
          ("256->128 @256 cat", dict(H=256, C0=128, C1=128, Cout=128)), ("128->128 @128", dict(H=128, C0=128, C1=0, Cout=128)),
          ("256->256 @64", dict(H=64, C0=256, C1=0, Cout=256)), ("512->256 @64 cat", dict(H=64, C0=256, C1=256, Cout=256)),
          ("256->256 @32", dict(H=32, C0=256, C1=0, Cout=256)), ("512->512 @16", dict(H=16, C0=512, C1=0, Cout=512)),
          ("128->128 @256 s2", dict(H=256, C0=128, C1=0, Cout=128, stride=2))]
tag = os.path.basename(os.environ.get("ASYRP_BENCH_LIB", "default"))
out = []
for name, kw in LAYERS:
    r = sorted(run(**kw) for _ in range(3))[1]
    out.append(f"{r:6.1f}")
print(f"{tag:24s} " + " ".join(out), flush=True)
if os.environ.get("K32_LAYERS_HEADER"):
    print(" " * 25 + " ".join(f"[{n}]" for n, _ in LAYERS))
