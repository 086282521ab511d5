"""Deterministic, platform-independent synthetic tensors for parity tests.

No pretrained diffusion weights exist offline (SURVEY.md §7 "hard parts"), so
parity runs on synthetic weights.  To make golden fixtures small the weights are
not stored: every tensor is a pure function of (key, shape) through a splitmix64
hash evaluated with numpy integer arithmetic (bit-identical on any host).

Key names/shapes follow the reference state_dict of ``DDPM``
(/root/reference/models/ddpm/diffusion.py:327-444).
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def hash_uniform(key, shape, lo=-1.0, hi=1.0, seed=0):
    """float32 tensor, uniform in [lo, hi), a pure function of (key, seed, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64(zlib.crc32(key.encode()) + (int(seed) << 32))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)
        z = _splitmix64(_splitmix64(base) ^ idx)
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def hash_normal(key, shape, seed=0):
    """Approximately N(0,1) float32 (sum of 4 uniforms, variance-normalised)."""
    acc = sum(hash_uniform(f"{key}#{i}", shape, -1.0, 1.0, seed).double() for i in range(4))
    return (acc * (3.0 / 4.0) ** 0.5).float()


class DDPMConfig:
    """Hyper-parameters the reference reads from configs/*.yml (celeba.yml:13-25)."""

    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 1, 2, 2, 4, 4), num_res_blocks=2,
                 attn_resolutions=(16,), in_channels=3, resolution=256):
        self.ch, self.out_ch, self.ch_mult = ch, out_ch, tuple(ch_mult)
        self.num_res_blocks, self.attn_resolutions = num_res_blocks, tuple(attn_resolutions)
        self.in_channels, self.resolution = in_channels, resolution

    def as_dict(self):
        return dict(ch=self.ch, out_ch=self.out_ch, ch_mult=list(self.ch_mult),
                    num_res_blocks=self.num_res_blocks, attn_resolutions=list(self.attn_resolutions),
                    in_channels=self.in_channels, resolution=self.resolution)


CELEBA = DDPMConfig()
SMALL = DDPMConfig(ch=32, ch_mult=(1, 2, 2), num_res_blocks=2, attn_resolutions=(16,), resolution=32)


def ddpm_param_shapes(cfg, n_delta=1):
    """Ordered {key: shape} of the reference DDPM state_dict (+ layer_i DeltaBlocks)."""
    sh = OrderedDict()

    def conv(p, cin, cout, k):
        sh[p + ".weight"] = (cout, cin, k, k)
        sh[p + ".bias"] = (cout,)

    def lin(p, cin, cout):
        sh[p + ".weight"] = (cout, cin)
        sh[p + ".bias"] = (cout,)

    def norm(p, c):
        sh[p + ".weight"] = (c,)
        sh[p + ".bias"] = (c,)

    def res(p, cin, cout, temb):
        norm(p + ".norm1", cin); conv(p + ".conv1", cin, cout, 3); lin(p + ".temb_proj", temb, cout)
        norm(p + ".norm2", cout); conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".nin_shortcut", cin, cout, 1)

    def attn(p, c):
        norm(p + ".norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(p + "." + n, c, c, 1)

    ch, temb = cfg.ch, cfg.ch * 4
    nlev = len(cfg.ch_mult)
    in_mult = (1,) + cfg.ch_mult
    lin("temb.dense.0", ch, temb); lin("temb.dense.1", temb, temb)
    conv("conv_in", cfg.in_channels, ch, 3)
    res_now, block_in = cfg.resolution, None
    for i in range(nlev):
        block_in, block_out = ch * in_mult[i], ch * cfg.ch_mult[i]
        for j in range(cfg.num_res_blocks):
            res(f"down.{i}.block.{j}", block_in, block_out, temb)
            block_in = block_out
            if res_now in cfg.attn_resolutions:
                attn(f"down.{i}.attn.{j}", block_in)
        if i != nlev - 1:
            conv(f"down.{i}.downsample.conv", block_in, block_in, 3)
            res_now //= 2
    res("mid.block_1", block_in, block_in, temb); attn("mid.attn_1", block_in)
    res("mid.block_2", block_in, block_in, temb)
    bottleneck = block_in
    up = OrderedDict()
    for i in reversed(range(nlev)):
        block_out, skip_in = ch * cfg.ch_mult[i], ch * cfg.ch_mult[i]
        names = []
        for j in range(cfg.num_res_blocks + 1):
            if j == cfg.num_res_blocks:
                skip_in = ch * in_mult[i]
            names.append(("res", f"up.{i}.block.{j}", block_in + skip_in, block_out))
            block_in = block_out
            if res_now in cfg.attn_resolutions:
                names.append(("attn", f"up.{i}.attn.{j}", block_in, None))
        if i != 0:
            names.append(("conv", f"up.{i}.upsample.conv", block_in, block_in))
            res_now *= 2
        up[i] = names
    for i in range(nlev):  # state_dict order: up.0 first (the reference prepends)
        for kind, p, a, b in up[i]:
            if kind == "res":
                res(p, a, b, temb)
            elif kind == "attn":
                attn(p, a)
            else:
                conv(p, a, b, 3)
    norm("norm_out", block_in); conv("conv_out", block_in, cfg.out_ch, 3)
    for d in range(n_delta):  # DeltaBlock, diffusion.py:228-263 via setattr_layers :433-444
        p = f"layer_{d}"
        conv(p + ".conv1", bottleneck, bottleneck, 1); lin(p + ".temb_proj", temb, bottleneck)
        norm(p + ".norm2", bottleneck); conv(p + ".conv2", bottleneck, bottleneck, 1)
    return sh


def synthetic_state_dict(shapes, seed=0):
    """PyTorch-default-like scales (U(-1/sqrt(fan_in), 1/sqrt(fan_in))); norms near (1, 0).

    A parameter pair whose weight is 1-D is a normalisation (GroupNorm gamma/beta); everything else is a
    conv / linear weight with fan_in = prod(shape[1:]).  No tensor is left at zero, so the reference's
    zero-initialised modules (improved_ddpm/unet.py:252-254,336,657) are exercised too.
    """
    sd = OrderedDict()
    for k, shp in shapes.items():
        wk = k[: -len(".bias")] + ".weight" if k.endswith(".bias") else k
        wshape = shapes.get(wk, shp)
        if len(wshape) == 1:
            if k.endswith(".weight"):
                sd[k] = 1.0 + 0.1 * hash_uniform(k, shp, seed=seed)
            else:
                sd[k] = 0.1 * hash_uniform(k, shp, seed=seed)
            continue
        fan_in = int(np.prod(wshape[1:]))
        bound = 1.0 / fan_in ** 0.5
        sd[k] = hash_uniform(k, shp, -bound, bound, seed=seed)
    return sd
