"""`DDPM` — drop-in for the reference's models/ddpm/diffusion.py:327 `DDPM(config)` on MI355X.

Same constructor argument (the YAML-derived namespace), same state_dict key names (so pretrained
UNet checkpoints and the shipped checkpoint/*.pth["0"] DeltaBlocks load unmodified), same
`setattr_layers(n)` / `get_temb(t)` / `forward(x, t, index, t_edit, hs_coeff, delta_h,
ignore_timestep, use_mask)` surface and the same 4-tuple result — but the modules below are
parameter HOLDERS only: all arithmetic runs in the hand-written HIP engine (libasyrp_hip.so).
There is no PyTorch/CPU fallback; calling forward with CPU tensors raises.
"""
import math

import torch
import torch.nn as nn

from . import _lib
from .engine import (AsyrpDeviceError, Engine, alphas_cumprod_from_betas, ddpm_temb_freqs, make_config,
                     param_specs)


class _Holder(nn.Module):
    """A node of the parameter tree; never executed."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: compute runs in the HIP engine via DDPM.forward")


def _attach(root, key, shape):
    parts = key.split(".")
    node = root
    for name in parts[:-1]:
        if name not in node._modules:
            node.add_module(name, _Holder())
        node = node._modules[name]
    node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape, dtype=torch.float32), requires_grad=False))


def _default_init_(sd_items):
    """PyTorch-default-like init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for conv/linear, (1,0) for norms."""
    shapes = {k: tuple(p.shape) for k, p in sd_items}
    with torch.no_grad():
        for k, p in sd_items:
            leaf = k.split(".")[-2]
            if leaf.startswith("norm"):
                p.fill_(1.0 if k.endswith("weight") else 0.0)
                continue
            wshape = shapes[k[:-4] + "weight"] if k.endswith("bias") else shapes[k]
            bound = 1.0 / math.sqrt(max(1, int(torch.tensor(wshape[1:]).prod())))
            p.uniform_(-bound, bound)


def _cfg_get(ns, name, default=None):
    return getattr(ns, name) if hasattr(ns, name) else (ns[name] if isinstance(ns, dict) and name in ns else default)


class DDPM(nn.Module):
    def __init__(self, config, max_batch=64, conv_math="f16x3"):
        super().__init__()
        self.config = config
        self.conv_math = conv_math      # "f16x3" (3 x f16 MFMA, fp32-equivalent, default) or "f32" (fp32 MFMA)
        m, d = _cfg_get(config, "model"), _cfg_get(config, "data")
        self.ch = int(_cfg_get(m, "ch"))
        self.out_ch = int(_cfg_get(m, "out_ch"))
        self.ch_mult = tuple(int(v) for v in _cfg_get(m, "ch_mult"))
        self.num_res_blocks = int(_cfg_get(m, "num_res_blocks"))
        self.attn_resolutions = tuple(int(v) for v in _cfg_get(m, "attn_resolutions"))
        self.in_channels = int(_cfg_get(m, "in_channels"))
        self.resolution = int(_cfg_get(d, "image_size"))
        if not _cfg_get(m, "resamp_with_conv", True):
            raise NotImplementedError("resamp_with_conv=False is not used by any reference config")
        self.temb_ch = self.ch * 4
        self.num_resolutions = len(self.ch_mult)
        self.max_batch = int(max_batch)
        self._n_delta = 0
        self._engine = None
        self._engine_sig = None
        self._uploaded = {}
        for key, shape in param_specs(self._make_cfg(0)):
            _attach(self, key, shape)
        _default_init_(list(self.named_parameters()))

    # ---- reference surface ----------------------------------------------------------------------
    def setattr_layers(self, nums):
        """Add DeltaBlocks layer_0..layer_{nums-1} (models/ddpm/diffusion.py:433-444)."""
        base = {k for k, _ in param_specs(self._make_cfg(0))}
        dev = next(self.parameters()).device
        new = []
        for key, shape in param_specs(self._make_cfg(int(nums))):
            if key in base:
                continue
            top = key.split(".")[0]
            if top in self._modules and key.count(".") == 2 and key.split(".")[1] == "conv1" and key.endswith("weight"):
                del self._modules[top]      # re-created below, as the reference's setattr does
            _attach(self, key, shape)
            new.append(key)
        named = dict(self.named_parameters())
        _default_init_([(k, named[k]) for k in new])
        for k in new:
            named[k].data = named[k].data.to(dev)
        self._n_delta = int(nums)
        self._drop_engine()

    def get_temb(self, t):
        raise NotImplementedError("get_temb is only used by the reference's image_space_noise branch "
                                  "(utils/diffusion_utils.py:62), which is outside the accelerated path")

    def forward(self, x, t, index=None, t_edit=400, hs_coeff=(1.0, 1.0), delta_h=None, ignore_timestep=False,
                use_mask=False):
        assert x.shape[2] == x.shape[3] == self.resolution
        if delta_h is not None:
            raise NotImplementedError("passing a delta_h tensor selects the reference's DiffStyle slerp branch "
                                      "(models/ddpm/diffusion.py:518-539); only the DeltaBlock path is accelerated")
        eng = self._ready_engine(x)
        apply_edit = bool(index is not None and (t[0] >= t_edit))   # the reference's own host sync (:510)
        et, et_mod, dh, mid = eng.unet_forward(x, t, index=index, apply_edit=apply_edit, hs_coeff=hs_coeff,
                                               ignore_timestep=ignore_timestep)
        return et, et_mod, dh, mid

    # ---- engine plumbing ------------------------------------------------------------------------
    def _make_cfg(self, n_delta):
        return make_config(family=_lib.FAMILY_DDPM, resolution=self.resolution, in_channels=self.in_channels,
                           out_channels=self.out_ch, ch=self.ch, ch_mult=self.ch_mult,
                           num_res_blocks=self.num_res_blocks, attn_resolutions=self.attn_resolutions,
                           n_delta=n_delta, conv_math=self.conv_math)

    def _drop_engine(self):
        if self._engine is not None:
            self._engine.close()
        self._engine, self._engine_sig, self._uploaded = None, None, {}

    def set_schedule(self, betas):
        """Hand the beta schedule (the `b` the reference passes to denoising_step) to the engine."""
        self._betas = betas.detach().float().cpu().clone()
        if self._engine is not None:
            self._engine.set_schedule(alphas_cumprod_from_betas(self._betas))

    def engine(self, device=None):
        """The live HIP engine for `device` (created, and parameters re-synchronised, on demand)."""
        if device is None:
            device = next(self.parameters()).device
        device = torch.device(device)
        if device.type != "cuda":
            raise AsyrpDeviceError("DDPM runs only on an MI355X (device type 'cuda' under ROCm); "
                                   "there is no CPU/PyTorch fallback")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        sig = (idx, self._n_delta, self.max_batch)
        if self._engine is None or self._engine_sig != sig:
            self._drop_engine()
            self._engine = Engine(self._make_cfg(self._n_delta), self.max_batch, idx)
            self._engine_sig = sig
            self._engine.set_temb_freqs(ddpm_temb_freqs(self.ch))
            if getattr(self, "_betas", None) is not None:
                self._engine.set_schedule(alphas_cumprod_from_betas(self._betas))
        dirty = False
        for k, p in self.named_parameters():
            stamp = (p.data_ptr(), p._version, tuple(p.shape))
            if self._uploaded.get(k) != stamp:
                self._engine.load_param(k, p)
                self._uploaded[k] = stamp
                dirty = True
        if dirty:
            self._engine.finalize()
        return self._engine

    def _ready_engine(self, x):
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise AsyrpDeviceError("input must live on the GPU: the Asyrp HIP engine has no CPU fallback")
        if x.shape[0] > self.max_batch:
            self.max_batch = int(x.shape[0])
        return self.engine(x.device)
