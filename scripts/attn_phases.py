#!/usr/bin/env python
"""Phase breakdown of attn_planes_kernel (profiling library): where a workgroup's time goes.  usage: scripts/attn_phases.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from asyrp_official_amd import _lib

lib = _lib.load_bench()
NAMES = ["Q staging", "S^T = K Q^T", "softmax + P -> LDS", "O^T = V^T P^T (last item group)", "store / end"]
for (B, Cc, T, heads, np_) in ((32, 512, 256, 1, 3), (1, 512, 256, 1, 3), (32, 512, 64, 1, 3), (16, 512, 1024, 8, 3), (64, 512, 256, 8, 3), (32, 512, 256, 1, 1)):
    nwg = B * heads * (T // 32)
    st = np.zeros((nwg, 8), dtype=np.uint64)
    ms = C.c_float()
    _lib.check(lib.asyrp_op_attention_phases(0, B, Cc, T, heads, np_, 20, C.byref(ms), st.ctypes.data_as(C.c_void_p), None))
    d = (st[:, 1:6].astype(np.int64) - st[:, 0:5].astype(np.int64)) / 100.0     # us (100 MHz)
    tot = (st[:, 5].astype(np.int64) - st[:, 0].astype(np.int64)) / 100.0
    span = (int(st[:, 5].max()) - int(st[:, 0].min())) / 100.0
    fl = 4.0 * T * T * Cc * B
    print(f"B={B} C={Cc} T={T} heads={heads} NP={np_}: {ms.value * 1e3:.1f} us per launch ({fl / (ms.value * 1e-3) / 1e12:.1f} TFLOP/s); "
          f"workgroup life mean {tot.mean():.1f} us (min {tot.min():.1f}, max {tot.max():.1f}); first start -> last end {span:.1f} us")
    for i, n in enumerate(NAMES):
        print(f"    {n:34s} mean {d[:, i].mean():6.2f} us  max {d[:, i].max():6.2f}")
