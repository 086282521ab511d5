#!/usr/bin/env python
"""One UNet evaluation (plain and dual-decoder Asyrp form) of the benchmarked model at B and alone: bitwise equal rows?
usage: scripts/batch_invariance_probe.py [B]   (env toggles are read by the library: run once per setting)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asyrp_official_amd import _lib
if os.environ.get("PROBE_LIB"):      # A/B against another build of the library
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from bench import celeba_namespace
from asyrp_official_amd import DDPM

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(1234)
dev = torch.device("cuda", 0)
model = DDPM(celeba_namespace(), max_batch=B)
model.setattr_layers(1)
model = model.to(dev).eval()
g = torch.Generator().manual_seed(1234)
x = (2 * torch.rand((B, 3, 256, 256), generator=g) - 1).to(dev)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("ASYRP_"))
for name, kw, tval in (("plain", dict(), 500), ("dual", dict(index=0, t_edit=400, hs_coeff=(1.0, 1.0)), 700)):
    t = torch.full((B,), tval, device=dev)
    outs = model(x, t, **kw)
    outs = outs if isinstance(outs, (tuple, list)) else (outs,)
    for rep in range(2):
        worst = []
        for i in (0, 1, B - 1):
            alone = model(x[i:i + 1].contiguous(), t[i:i + 1], **kw)
            alone = alone if isinstance(alone, (tuple, list)) else (alone,)
            worst.append(max(float((a_[0] - o_[i]).abs().max()) for a_, o_ in zip(alone, outs) if a_ is not None and o_ is not None and torch.is_tensor(a_)))
        again = model(x, t, **kw)
        again = again if isinstance(again, (tuple, list)) else (again,)
        rerun = max(float((a_ - o_).abs().max()) for a_, o_ in zip(again, outs) if torch.is_tensor(a_))
        print(f"[{tag}] {name} rep {rep}: alone-vs-batch max|d| rows (0,1,B-1) = {worst}; batch rerun max|d| = {rerun}", flush=True)
        if rerun > 0:
            d = (again[0] - outs[0]).abs()
            rows = [i for i in range(B) if float(d[i].max()) > 0]
            i = rows[0]
            nz = (d[i] > 0).nonzero()
            print(f"    rows that differ on the rerun: {rows}; row {i}: {len(nz)} of {d[i].numel()} values differ, first at {nz[0].tolist()} last at {nz[-1].tolist()}")
