"""Functional PyTorch-CPU restatement of the reference iDDPM / ADM UNet forward (TEST ORACLE).

Follows /root/reference/models/improved_ddpm/unet.py (= models/guided_diffusion/unet.py:437-857, same network) and
models/improved_ddpm/nn.py; operates on a flat state_dict with the reference's key names
(time_embed.*, input_blocks.N.M.*, middle_block.*, output_blocks.N.M.*, out.*, layer_i.*).
"""
import math

import torch
import torch.nn.functional as F

from .ddpm import slerp   # improved_ddpm/unet.py:24-60 is the same function as ddpm/diffusion.py:6-40


class IDDPMConfig:
    """Arguments of create_model (improved_ddpm/script_util.py:45-99) that shape the network."""

    def __init__(self, image_size=256, num_channels=128, num_res_blocks=1, channel_mult=(1, 1, 2, 2, 4, 4),
                 attention_resolutions=(16,), num_head_channels=64, learn_sigma=True, class_cond=False, in_channels=3):
        self.image_size, self.num_channels, self.num_res_blocks = image_size, num_channels, num_res_blocks
        self.channel_mult = tuple(channel_mult)
        self.attention_resolutions = tuple(attention_resolutions)      # spatial sizes ("16" in AFHQ_DICT)
        self.num_head_channels, self.learn_sigma, self.class_cond = num_head_channels, learn_sigma, class_cond
        self.in_channels = in_channels
        self.out_channels = 6 if learn_sigma else 3
        self.attention_ds = tuple(image_size // r for r in self.attention_resolutions)   # script_util.py:77-79


AFHQ = IDDPMConfig()                                                    # AFHQ_DICT / FFHQ / METFACE_DICT / CELEBA_HQ_P2_DICT
IMAGENET = IDDPMConfig(num_channels=256, num_res_blocks=2, attention_resolutions=(32, 16, 8), class_cond=True)
SMALL_I = IDDPMConfig(image_size=32, num_channels=32, num_res_blocks=1, channel_mult=(1, 2, 2),
                      attention_resolutions=(16,), num_head_channels=16)
# the structural features only IMAGENET_DICT uses, at toy size: two ResBlocks per level, attention at several
# resolutions (incl. the bottleneck level), class_cond (unused label_emb in the state_dict)
SMALL_I2 = IDDPMConfig(image_size=32, num_channels=32, num_res_blocks=2, channel_mult=(1, 2, 4),
                       attention_resolutions=(16, 8), num_head_channels=32, class_cond=True)


def block_plan(cfg):
    """The module list UNetModel.__init__ builds (unet.py:527-658) as plain tuples.

    Returns (input_blocks, middle_ch, output_blocks, final_ch); each block is a list of layers
    ("conv", cin, cout) | ("res", cin, cout, mode) with mode in {None, "down", "up"} | ("attn", ch).
    """
    mc = cfg.num_channels
    ch = int(cfg.channel_mult[0] * mc)
    inp = [[("conv", cfg.in_channels, ch)]]
    chans = [ch]
    ds = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [("res", ch, int(mult * mc), None)]
            ch = int(mult * mc)
            if ds in cfg.attention_ds:
                layers.append(("attn", ch))
            inp.append(layers)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            inp.append([("res", ch, ch, "down")])      # resblock_updown=True in every reference dict
            chans.append(ch)
            ds *= 2
    mid = ch
    out = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, int(mc * mult), None)]
            ch = int(mc * mult)
            if ds in cfg.attention_ds:
                layers.append(("attn", ch))
            if level and i == cfg.num_res_blocks:
                layers.append(("res", ch, ch, "up"))
                ds //= 2
            out.append(layers)
    return inp, mid, out, ch


def iddpm_param_shapes(cfg, n_delta=1):
    """Ordered {key: shape} of the reference UNetModel state_dict (+ layer_i DeltaBlocks, unet.py:756-853)."""
    from collections import OrderedDict
    sh = OrderedDict()
    emb = cfg.num_channels * 4

    def conv(p, cin, cout, k):
        sh[p + ".weight"] = (cout, cin, k, k)
        sh[p + ".bias"] = (cout,)

    def norm(p, c):
        sh[p + ".weight"] = (c,)
        sh[p + ".bias"] = (c,)

    def lin(p, cin, cout):
        sh[p + ".weight"] = (cout, cin)
        sh[p + ".bias"] = (cout,)

    def res(p, cin, cout):
        norm(p + ".in_layers.0", cin); conv(p + ".in_layers.2", cin, cout, 3)
        lin(p + ".emb_layers.1", emb, 2 * cout)            # use_scale_shift_norm=True
        norm(p + ".out_layers.0", cout); conv(p + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            conv(p + ".skip_connection", cin, cout, 1)

    def attn(p, c):
        norm(p + ".norm", c)
        sh[p + ".qkv.weight"] = (3 * c, c, 1); sh[p + ".qkv.bias"] = (3 * c,)     # Conv1d
        sh[p + ".proj_out.weight"] = (c, c, 1); sh[p + ".proj_out.bias"] = (c,)

    def layers(prefix, ls):
        for m, l in enumerate(ls):
            p = f"{prefix}.{m}"
            if l[0] == "conv":
                conv(p, l[1], l[2], 3)
            elif l[0] == "res":
                res(p, l[1], l[2])
            else:
                attn(p, l[1])

    inp, mid, out, final = block_plan(cfg)
    lin("time_embed.0", cfg.num_channels, emb); lin("time_embed.2", emb, emb)
    if cfg.class_cond:
        sh["label_emb.weight"] = (1000, emb)
    for n, ls in enumerate(inp):
        layers(f"input_blocks.{n}", ls)
    layers("middle_block", [("res", mid, mid, None), ("attn", mid), ("res", mid, mid, None)])
    for n, ls in enumerate(out):
        layers(f"output_blocks.{n}", ls)
    norm("out.0", final); conv("out.2", final, cfg.out_channels, 3)
    for d in range(n_delta):                              # DeltaBlock: no FiLM (use_scale_shift_norm defaults False)
        p = f"layer_{d}"
        norm(p + ".in_layers.0", mid); conv(p + ".in_layers.2", mid, mid, 1)
        lin(p + ".emb_layers.1", emb, mid)
        norm(p + ".out_layers.0", mid); conv(p + ".out_layers.3", mid, mid, 1)
    return sh


def timestep_embedding(t, dim, max_period=10000):
    """[cos | sin], frequencies exp(-ln(max_period) * i / half) (nn.py:103-121)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(x, sd, p):
    """GroupNorm32(32, C), eps 1e-5, computed in fp32 (nn.py:17-19, 93-100)."""
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def _silu(x):
    return x * torch.sigmoid(x)


def res_block(x, emb, sd, p, mode=None):
    """ResBlock._forward (unet.py:278-298) with use_scale_shift_norm=True."""
    h = _silu(_gn(x, sd, p + ".in_layers.0"))
    if mode == "up":
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif mode == "down":
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    emb_out = F.linear(_silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[:, :, None, None]
    scale, shift = torch.chunk(emb_out, 2, dim=1)
    h = _gn(h, sd, p + ".out_layers.0") * (1 + scale) + shift
    h = F.conv2d(_silu(h), sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def attention_block(x, sd, p, num_head_channels):
    """AttentionBlock._forward + QKVAttentionLegacy (unet.py:341-347, 379-396)."""
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(xf, sd, p + ".norm"), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    n_heads = c // num_head_channels
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(bs, -1, length)
    h = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + h).reshape(b, c, hh, ww)


def delta_block(x, emb, sd, p):
    """DeltaBlock.forward (unet.py:835-853): GN-SiLU-conv1x1, (+Linear(SiLU(emb))), GN-SiLU-conv1x1."""
    h = F.conv2d(_silu(_gn(x, sd, p + ".in_layers.0")), sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"])
    if emb is not None:
        h = h + F.linear(_silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[:, :, None, None]
    return F.conv2d(_silu(_gn(h, sd, p + ".out_layers.0")), sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"])


def _run_layers(h, emb, sd, prefix, ls, cfg):
    for m, l in enumerate(ls):
        p = f"{prefix}.{m}"
        if l[0] == "conv":
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        elif l[0] == "res":
            h = res_block(h, emb, sd, p, l[3])
        else:
            h = attention_block(h, sd, p, cfg.num_head_channels)
    return h


def _decoder(h, hs, emb, sd, out_plan, cfg):
    """output_blocks over cat(h, skip) then out (unet.py:736-750); `hs` is read from the back, not modified."""
    k = len(hs) - 1
    for n, ls in enumerate(out_plan):
        h = _run_layers(torch.cat([h, hs[k]], dim=1), emb, sd, f"output_blocks.{n}", ls, cfg)
        k -= 1
    return F.conv2d(_silu(_gn(h, sd, "out.0")), sd["out.2.weight"], sd["out.2.bias"], padding=1)


def iddpm_forward(sd, cfg, x, timesteps, y=None, index=None, t_edit=400, hs_coeff=(1.0, 1.0), delta_h=None,
                  ignore_timestep=False, use_mask=False):
    """UNetModel.forward (unet.py:676-752): returns (h, h2, delta_h, middle_h); `y` is ignored as in the reference."""
    inp, mid, out_plan, _ = block_plan(cfg)
    emb = timestep_embedding(timesteps, cfg.num_channels)
    emb = F.linear(_silu(F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])),
                   sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    hs = []
    h = x
    for n, ls in enumerate(inp):
        h = _run_layers(h, emb, sd, f"input_blocks.{n}", ls, cfg)
        hs.append(h)
    h = _run_layers(h, emb, sd, "middle_block", [("res", mid, mid, None), ("attn", mid), ("res", mid, mid, None)], cfg)
    middle_h = h
    h2 = None
    if index is not None:
        if timesteps[0] >= t_edit:
            if delta_h is None:                                  # Asyrp, unet.py:702-706
                h2 = h * hs_coeff[0]
                for i in range(index + 1):
                    delta_h = delta_block(h, None if ignore_timestep else emb, sd, f"layer_{i}")
                    h2 = h2 + delta_h * hs_coeff[i + 1]
            elif use_mask:                                       # injected delta_h tensor, masked slerp, unet.py:709-719
                mask = torch.zeros_like(h)
                mask[:, :, 4:-1, 3:5] = 1.0
                h2 = slerp(1 - hs_coeff[0], h * mask, delta_h * mask) + (1 - mask) * h
            else:                                                # norm-matched slerp, unet.py:722-731
                n = h.shape[0]
                hn = h.reshape(n, -1).norm(dim=1).reshape(n, 1, 1, 1)
                dn = delta_h.reshape(n, -1).norm(dim=1).reshape(n, 1, 1, 1)
                h2 = slerp(1.0 - hs_coeff[0], h, hn * delta_h / dn)
        else:
            h2 = h
        h2 = _decoder(h2, hs, emb, sd, out_plan, cfg)
    h = _decoder(h, hs, emb, sd, out_plan, cfg)
    return h, h2, delta_h, middle_h


def make_model(sd, cfg):
    def model(x, t, **kw):
        with torch.no_grad():
            return iddpm_forward(sd, cfg, x, t, **kw)
    return model
