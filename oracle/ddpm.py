"""Functional PyTorch-CPU restatement of the reference DDPM UNet forward (TEST ORACLE).

Operates on a flat state_dict with the reference's key names, so a pretrained
checkpoint, a shipped Δh checkpoint, or ``oracle.weights.synthetic_state_dict``
can all be used.  Each function cites the reference lines it restates
(/root/reference/models/ddpm/diffusion.py).
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(t, dim):
    """[sin | cos] sinusoidal embedding, divisor half-1 (diffusion.py:42-60)."""
    half = dim // 2
    rate = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -rate)
    arg = t.float()[:, None] * freqs[None, :]
    emb = torch.cat([arg.sin(), arg.cos()], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def swish(x):
    """x * sigmoid(x) (diffusion.py:63-65)."""
    return x * torch.sigmoid(x)


def _gn(x, sd, p):
    """GroupNorm(32, eps=1e-6, affine) (diffusion.py:68-69)."""
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _conv(x, sd, p, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def temb_mlp(sd, t, ch):
    """Linear -> swish -> Linear on the sinusoidal embedding (diffusion.py:477-480)."""
    e = timestep_embedding(t, ch)
    e = F.linear(e, sd["temb.dense.0.weight"], sd["temb.dense.0.bias"])
    return F.linear(swish(e), sd["temb.dense.1.weight"], sd["temb.dense.1.bias"])


def resnet_block(x, temb, sd, p):
    """GN-swish-conv3x3, +Linear(swish(temb)), GN-swish-conv3x3, + (1x1 shortcut of) x
    (diffusion.py:151-170)."""
    h = _conv(swish(_gn(x, sd, p + ".norm1")), sd, p + ".conv1", padding=1)
    h = h + F.linear(swish(temb), sd[p + ".temb_proj.weight"], sd[p + ".temb_proj.bias"])[:, :, None, None]
    h = _conv(swish(_gn(h, sd, p + ".norm2")), sd, p + ".conv2", padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(x, sd, p + ".nin_shortcut")
    return x + h


def attn_block(x, sd, p):
    """Single-head self-attention over H*W tokens, scores scaled by C^-0.5 after the
    product, softmax over keys (diffusion.py:200-225)."""
    b, c, hh, ww = x.shape
    g = _gn(x, sd, p + ".norm")
    q = _conv(g, sd, p + ".q").reshape(b, c, hh * ww)
    k = _conv(g, sd, p + ".k").reshape(b, c, hh * ww)
    v = _conv(g, sd, p + ".v").reshape(b, c, hh * ww)
    w = torch.bmm(q.transpose(1, 2), k) * (int(c) ** (-0.5))      # [b, query, key]
    w = F.softmax(w, dim=2)
    o = torch.bmm(v, w.transpose(1, 2)).reshape(b, c, hh, ww)      # o[c, query] = sum_key v[c,key] w[query,key]
    return x + _conv(o, sd, p + ".proj_out")


def delta_block(x, temb, sd, p):
    """DeltaBlock: conv1x1, (+Linear(swish(temb))), GN, swish, conv1x1 (diffusion.py:250-263)."""
    h = _conv(x, sd, p + ".conv1")
    if temb is not None:
        h = h + F.linear(swish(temb), sd[p + ".temb_proj.weight"], sd[p + ".temb_proj.bias"])[:, :, None, None]
    return _conv(swish(_gn(h, sd, p + ".norm2")), sd, p + ".conv2")


def downsample(x, sd, p):
    """pad right/bottom by one, then valid 3x3 stride-2 conv (diffusion.py:103-107)."""
    return _conv(F.pad(x, (0, 1, 0, 1)), sd, p + ".conv", stride=2)


def upsample(x, sd, p):
    """nearest x2 then 3x3 conv (diffusion.py:83-88)."""
    return _conv(F.interpolate(x, scale_factor=2.0, mode="nearest"), sd, p + ".conv", padding=1)


def encoder(sd, cfg, x, temb):
    """conv_in + down path; returns the skip stack (diffusion.py:485-495)."""
    nlev = len(cfg.ch_mult)
    res = cfg.resolution
    skips = [_conv(x, sd, "conv_in", padding=1)]
    for i in range(nlev):
        for j in range(cfg.num_res_blocks):
            h = resnet_block(skips[-1], temb, sd, f"down.{i}.block.{j}")
            if res in cfg.attn_resolutions:
                h = attn_block(h, sd, f"down.{i}.attn.{j}")
            skips.append(h)
        if i != nlev - 1:
            skips.append(downsample(skips[-1], sd, f"down.{i}.downsample"))
            res //= 2
    return skips


def middle(sd, h, temb):
    """mid.block_1 -> mid.attn_1 -> mid.block_2 (diffusion.py:500-504)."""
    h = resnet_block(h, temb, sd, "mid.block_1")
    h = attn_block(h, sd, "mid.attn_1")
    return resnet_block(h, temb, sd, "mid.block_2")


def decoder(sd, cfg, h, skips, temb):
    """up path over cat(h, skip) + norm_out/swish/conv_out; ``skips`` is read from the
    back without being modified (diffusion.py:546-559 and :564-578 are the same math)."""
    nlev = len(cfg.ch_mult)
    res = cfg.resolution // (2 ** (nlev - 1))
    k = len(skips) - 1
    for i in reversed(range(nlev)):
        for j in range(cfg.num_res_blocks + 1):
            h = resnet_block(torch.cat([h, skips[k]], dim=1), temb, sd, f"up.{i}.block.{j}")
            k -= 1
            if res in cfg.attn_resolutions:
                h = attn_block(h, sd, f"up.{i}.attn.{j}")
        if i != 0:
            h = upsample(h, sd, f"up.{i}.upsample")
            res *= 2
    return _conv(swish(_gn(h, sd, "norm_out")), sd, "conv_out", padding=1)


def slerp(tt, v0, v1):
    """Per-sample spherical interpolation used by the DiffStyle branch (diffusion.py:6-40)."""
    n = v0.shape[0]
    u0 = v0 / v0.reshape(n, -1).norm(dim=1).reshape(n, 1, 1, 1)
    u1 = v1 / v1.reshape(n, -1).norm(dim=1).reshape(n, 1, 1, 1)
    dot = (u0.reshape(n, -1) * u1.reshape(n, -1)).sum(dim=1)
    th0 = torch.acos(dot)
    tht = th0 * tt
    s0 = (torch.sin(th0 - tht) / torch.sin(th0)).reshape(n, 1, 1, 1)
    s1 = (torch.sin(tht) / torch.sin(th0)).reshape(n, 1, 1, 1)
    return s0 * v0 + s1 * v1


def ddpm_forward(sd, cfg, x, t, index=None, t_edit=400, hs_coeff=(1.0, 1.0), delta_h=None,
                 ignore_timestep=False, use_mask=False):
    """DDPM.forward (diffusion.py:473-580): returns (et, et_modified, delta_h, middle_h).

    ``index is None`` -> single decoder, et_modified = delta_h = None.  Otherwise the decoder
    runs on h2 = h*c0 + sum_i DeltaBlock_i(h, temb)*c_{i+1} when t[0] >= t_edit (:510-516),
    on the slerp mix when a delta_h tensor is passed (:518-539), or on h itself (:541-542).
    """
    assert x.shape[2] == x.shape[3] == cfg.resolution
    temb = temb_mlp(sd, t, cfg.ch)
    skips = encoder(sd, cfg, x, temb)
    h = middle(sd, skips[-1], temb)
    et_mod = None
    if index is not None:
        if t[0] >= t_edit:
            if delta_h is None:
                h2 = h * hs_coeff[0]
                for i in range(index + 1):
                    delta_h = delta_block(h, None if ignore_timestep else temb, sd, f"layer_{i}")
                    h2 = h2 + delta_h * hs_coeff[i + 1]
            elif use_mask:
                mask = torch.zeros_like(h)
                mask[:, :, 4:-1, 3:5] = 1.0
                h2 = slerp(1 - hs_coeff[0], h * mask, delta_h * mask) + (1 - mask) * h
            else:
                n = h.shape[0]
                hn = h.reshape(n, -1).norm(dim=1).reshape(n, 1, 1, 1)
                dn = delta_h.reshape(n, -1).norm(dim=1).reshape(n, 1, 1, 1)
                h2 = slerp(1.0 - hs_coeff[0], h, hn * delta_h / dn)
        else:
            h2 = h
        et_mod = decoder(sd, cfg, h2, skips, temb)
    et = decoder(sd, cfg, h, skips, temb)
    return et, et_mod, delta_h, h
