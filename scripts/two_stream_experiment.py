"""ONE bounded concurrency experiment — does a second, independent pipeline fill the holes of the
occupancy- / latency-bound section of the UNet (16^2 / 8^2 levels, 1x1 GEMMs, attention, gn_finalize2, split-K reduces: <= 256
workgroups, 5-40 us launches) while the main tile (80 % of the step, power-bound) keeps the chip busy?

  A: 1 engine, batch 32, one stream                      (the line of record)
  B: 2 engines, batch 16 each, two host threads, two streams, the second offset by half a UNet evaluation at the start
Both use the default batch class (nominal_batch = 32: the same tiles), so every image's bits are identical in A and B (checked).
Interleaved A, B, A, B ... on one box; sclk / W sampled from sysfs while each runs.

  python scripts/two_stream_experiment.py [rounds=3] [edits_per_round=2]
"""
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import bench  # noqa: E402
from asyrp_official_amd import DDPM, run_edit  # noqa: E402
from asyrp_official_amd.diffusion_utils import get_beta_schedule  # noqa: E402

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
EDITS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
KW = dict(n_inv=40, n_gen=40, t_0=999, t_edit=500, t_addnoise=0, index=0, hs_coeff=(1.0, 1.0))


def make(B, sd=None):
    torch.manual_seed(1234)
    m = DDPM(bench.celeba_namespace(), max_batch=B)
    m.setattr_layers(1)
    if sd is not None:
        m.load_state_dict(sd)
    return m.cuda().eval()


def clocks():
    try:
        import gpuclk
        return gpuclk.read()
    except Exception:   # noqa: BLE001
        return None


def main():
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    x0 = (2 * torch.rand((32, 3, 256, 256), generator=torch.Generator().manual_seed(1234)) - 1).cuda()
    one = make(32)
    sd = {k: v.clone() for k, v in one.state_dict().items()}
    halves = [make(16, sd), make(16, sd)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ref = run_edit(one, x0, betas, **KW)                         # warm + the bits to compare with
    outs = [None, None]

    def half(i, n):
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                outs[i] = run_edit(halves[i], x0[16 * i:16 * (i + 1)].contiguous(), betas, **KW)

    for i in (0, 1):
        half(i, 1)                                                # warm each engine alone
    torch.cuda.synchronize()
    same = bool(torch.equal(torch.cat(outs), ref))

    def run_a():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(EDITS):
            run_edit(one, x0, betas, **KW)
        torch.cuda.synchronize()
        return 32 * EDITS / (time.perf_counter() - t0)

    def run_b(offset_s):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=half, args=(i, EDITS)) for i in (0, 1)]
        th[0].start()
        time.sleep(offset_s)
        th[1].start()
        [t.join() for t in th]
        torch.cuda.synchronize()
        return 32 * EDITS / (time.perf_counter() - t0)

    res = {"A_1x32": [], "B_2x16_offset_half_unet": [], "B_2x16_no_offset": [], "bits_equal": same, "clk": []}
    for _ in range(ROUNDS):
        res["A_1x32"].append(run_a())
        res["clk"].append(clocks())
        res["B_2x16_offset_half_unet"].append(run_b(0.011))       # half of a ~21 ms B=16 inversion step
        res["B_2x16_no_offset"].append(run_b(0.0))
    for k in ("A_1x32", "B_2x16_offset_half_unet", "B_2x16_no_offset"):
        res[k + "_mean"] = sum(res[k]) / len(res[k])
    res["B_over_A"] = max(res["B_2x16_offset_half_unet_mean"], res["B_2x16_no_offset_mean"]) / res["A_1x32_mean"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
