#!/usr/bin/env python
"""A/B lines of round 2 (run interleaved from scripts/gpu_ab_r02.sh): leader layers on the 8-wave tile (6) and the 16-wave
512x128 experiment (7); the XCD-aware tile map is switched per process through ASYRP_XCD_MAP."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_bench import run  # noqa: E402

tag = "xmap=%s" % os.environ.get("ASYRP_XCD_MAP", "1")
layers = [("128->128 @256", 256, 128, 0, 128, 3, {}), ("256->128 @256 concat", 256, 128, 128, 128, 3, {}),
          ("128->128 @256 +resid", 256, 128, 0, 128, 3, dict(res=1)), ("256->256 @64", 64, 256, 0, 256, 3, {}),
          ("128->128 @128", 128, 128, 0, 128, 3, {}), ("512->256 @64 concat", 64, 256, 256, 256, 3, {})]
for name, H, C0, C1, Co, k, kw in layers:
    line = f"{tag} {name:24s}"
    for tile in (6, 7):
        ms, tf = run(H, C0, C1, Co, k, tile=tile, iters=20, **kw)
        line += f"  tile{tile} {ms:7.3f} ms {tf:6.1f} TF"
    print(line, flush=True)
