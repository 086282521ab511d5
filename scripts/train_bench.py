#!/usr/bin/env python
"""Throughput of the DeltaBlock training step on the engine (reference loop diffusion_latent.py:301-354 minus the
CLIP network): CelebA-HQ 256x256 DDPM, seeded random-init weights, the reference's loop shape
    optim.zero_grad(); denoising_step(..., index=0, t_edit=...); loss(x0_t).backward(); optim.step()
timed per step with the stream synchronised on both sides, next to the inference step that computes the same forward.
The reference trains with bs_train = 1 (diffusion_latent.py:1010).  usage: scripts/train_bench.py [B ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from asyrp_official_amd import denoising_step
from oracle import sampler as osamp          # schedule helper only (a benchmark script, like bench.py's cpu_baseline leg)
from oracle.weights import CELEBA, hash_normal
from util_models import hip_model, synthetic

batches = [int(v) for v in sys.argv[1:]] or [1, 4, 8]
b = osamp.beta_schedule().cuda()
out = []
for B in batches:
    m = hip_model(CELEBA, synthetic(CELEBA, 1, seed=1234), 1, max_batch=B)
    for p in m.parameters():
        p.requires_grad = False
    for p in m.layer_0.parameters():
        p.requires_grad = True
    opt = torch.optim.SGD(list(m.layer_0.parameters()), lr=1e-4)
    x = hash_normal("celeba.x", (B, 3, 256, 256), seed=1234).cuda()
    tgt = hash_normal("train.tgt256", (B, 3, 256, 256), seed=5).cuda()
    one = torch.ones(B, device="cuda")
    kw = dict(models=m, logvars=None, b=b, sampling_type="ddim", eta=0.0, index=0, t_edit=500, hs_coeff=(1.0, 1.0))

    def train_step():
        opt.zero_grad()
        _, x0t, _, _ = denoising_step(x, t=one * 768.0, t_next=one * 743.0, **kw)
        torch.nn.functional.l1_loss(x0t, tgt).backward()
        opt.step()

    def infer_step():
        with torch.no_grad():
            denoising_step(x, t=one * 768.0, t_next=one * 743.0, **kw)

    res = {"B": B}
    for name, fn, n in (("train_step_ms", train_step, 10), ("inference_dual_step_ms", infer_step, 10)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / n * 1e3
    res["train_over_inference"] = res["train_step_ms"] / res["inference_dual_step_ms"]
    res["train_steps_per_s"] = 1e3 / res["train_step_ms"]
    res["images_per_s"] = B * 1e3 / res["train_step_ms"]
    print(json.dumps(res), flush=True)
    out.append(res)
    del m, opt
    torch.cuda.empty_cache()
