#!/usr/bin/env python
"""Phase breakdown of gemm1x1_k32_kernel (profiling library): where a wave's time goes.  usage: scripts/gemm1x1_phases.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from asyrp_official_amd import _lib

lib = _lib.load_bench()
NAMES = ["first loads issued, scale/shift staged", "K loop", "epilogue"]
for (B, H, Ci, Co, pro, np_, tile) in ((32, 16, 512, 1536, 1, 3, 16), (32, 16, 512, 1536, 1, 3, 15), (32, 16, 512, 512, 0, 3, 16), (1, 16, 512, 1536, 1, 3, 16),
                                       (32, 16, 512, 1536, 1, 1, 16), (32, 32, 512, 512, 0, 3, 16), (8, 256, 256, 128, 0, 3, 16)):
    bm = 256 if tile == 15 else 128
    nwave = B * ((H * H + bm - 1) // bm) * ((Co + 127) // 128) * (bm // 32)
    st = np.zeros((nwave, 4), dtype=np.uint64)
    ms = C.c_float()
    _lib.check(lib.asyrp_op_gemm1x1_phases(0, B, H, Ci, Co, pro, np_, tile, 20, C.byref(ms), st.ctypes.data_as(C.c_void_p), None))
    s = st.astype(np.int64)
    d = (s[:, 1:4] - s[:, 0:3]) / 100.0     # us (100 MHz)
    tot = (s[:, 3] - s[:, 0]) / 100.0
    span = (int(s[:, 3].max()) - int(s[:, 0].min())) / 100.0
    fl = 2.0 * B * H * H * Ci * Co
    print(f"B={B} {Ci}->{Co} @{H} pro={pro} NP={np_} tile={tile}: {ms.value * 1e3:.1f} us per launch ({fl / (ms.value * 1e-3) / 1e12:.1f} TFLOP/s); "
          f"{nwave} waves, life mean {tot.mean():.1f} us (min {tot.min():.1f}, max {tot.max():.1f}); first start -> last end {span:.1f} us; "
          f"K steps {Ci // 32}")
    for i, n in enumerate(NAMES):
        print(f"    {n:40s} mean {d[:, i].mean():6.2f} us  min {d[:, i].min():6.2f}  max {d[:, i].max():6.2f}")
    starts = np.sort((s[:, 0] - s[:, 0].min()) / 100.0)
    print("    wave start times (us) at quantiles 10/50/90/100 %:", " ".join(f"{starts[int(q * (len(starts) - 1))]:.1f}" for q in (0.1, 0.5, 0.9, 1.0)))
