"""Shared plumbing of the UNet mirrors: parameter holders with the reference's state_dict names + lazy HIP engine."""
import math

import torch
import torch.nn as nn

from .engine import AsyrpDeviceError, Engine, alphas_cumprod_from_betas, param_specs


class _Holder(nn.Module):
    """A node of the parameter tree; never executed."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: compute runs in the HIP engine via the UNet's forward")


def _attach(root, key, shape):
    parts = key.split(".")
    node = root
    for name in parts[:-1]:
        if name not in node._modules:
            node.add_module(name, _Holder())
        node = node._modules[name]
    node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape, dtype=torch.float32), requires_grad=False))


def _default_init_(sd_items):
    """PyTorch-default-like init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for conv/linear, (1,0) for norms (1-D weights)."""
    shapes = {k: tuple(p.shape) for k, p in sd_items}
    with torch.no_grad():
        for k, p in sd_items:
            wshape = shapes.get(k[:-4] + "weight", shapes[k]) if k.endswith("bias") else shapes[k]
            if len(wshape) == 1:
                p.fill_(1.0 if k.endswith("weight") else 0.0)
                continue
            bound = 1.0 / math.sqrt(max(1, int(torch.tensor(wshape[1:]).prod())))
            p.uniform_(-bound, bound)


class HipUNet(nn.Module):
    """Base of `DDPM` and `UNetModel`: owns reference-named parameters, creates/synchronises the HIP engine on demand.
    Subclasses provide `_make_cfg(n_delta)`, `_temb_freqs()` and `resolution`."""

    def _init_params(self, max_batch, conv_math, nominal_batch=0):
        self.max_batch = int(max_batch)
        # batch class of the engine (include/asyrp.h asyrp_config.nominal_batch): 0 = kernels priced at 32 images per GPU (default),
        # 1 / 2 = the small class for single-image serving.  Fixed per engine; results of two classes agree to fp32 rounding.
        self.nominal_batch = int(nominal_batch)
        if self.nominal_batch not in (0, 1, 2, 32):
            raise ValueError(f"nominal_batch must be 0 (= 32, the default class), 1, 2 (the small class) or 32, got {nominal_batch}")
        # "f16x3" (3 x f16 MFMA, fp32-equivalent, default), "f32" (fp32 MFMA), or the fast mode "f16" (ONE f16 MFMA per product:
        # not fp32-equivalent, reported separately with its own error; include/asyrp.h enum asyrp_conv_math)
        self.conv_math = conv_math
        self._n_delta = 0
        self._engine = None
        self._engine_sig = None
        self._uploaded = {}
        for key, shape in param_specs(self._make_cfg(0)):
            _attach(self, key, shape)
        _default_init_(list(self.named_parameters()))

    # ---- reference surface ----------------------------------------------------------------------
    def setattr_layers(self, nums):
        """Add DeltaBlocks layer_0..layer_{nums-1} (models/ddpm/diffusion.py:433-444, models/improved_ddpm/unet.py:756-773)."""
        base = {k for k, _ in param_specs(self._make_cfg(0))}
        dev = next(self.parameters()).device
        new = []
        for key, shape in param_specs(self._make_cfg(int(nums))):
            if key in base:
                continue
            top = key.split(".")[0]
            if top in self._modules and not any(k.startswith(top + ".") for k in new):
                del self._modules[top]      # re-created below, as the reference's setattr does
            _attach(self, key, shape)
            new.append(key)
        named = dict(self.named_parameters())
        _default_init_([(k, named[k]) for k in new])
        for k in new:
            named[k].data = named[k].data.to(dev)
        self._n_delta = int(nums)
        self._drop_engine()

    def _run(self, x, t, index, t_edit, hs_coeff, delta_h, ignore_timestep, use_mask):
        assert x.shape[2] == x.shape[3] == self.resolution
        eng = self._ready_engine(x)
        apply_edit = bool(index is not None and (t[0] >= t_edit))   # the reference's own host sync (:510)
        # a delta_h TENSOR replaces the DeltaBlocks by the per-sample slerp mix (:518-539); use_mask only matters there
        return eng.unet_forward(x, t, index=index, apply_edit=apply_edit, hs_coeff=hs_coeff,
                                ignore_timestep=ignore_timestep, delta_h=delta_h, use_mask=use_mask)

    # ---- engine plumbing ------------------------------------------------------------------------
    def _drop_engine(self):
        if self._engine is not None:
            self._engine.close()
        self._engine, self._engine_sig, self._uploaded = None, None, {}

    def _replicate_for_data_parallel(self):
        """torch.nn.DataParallel over MORE than one device (diffusion_latent.py:591 on a multi-GPU host) replicates the module per
        device with shallow copies: every replica would drive ONE engine handle, bound to one GPU, from its own thread.  Fail loudly;
        the supported multi-GPU form is one process per GPU (sampler.run_edit_sharded).  With a single device
        DataParallel calls the module directly and never gets here."""
        raise AsyrpDeviceError(f"{type(self).__name__} cannot be replicated by torch.nn.DataParallel across several GPUs: the HIP engine "
                               "is bound to one device. Launch one process per GPU (torch.distributed.run) and use "
                               "asyrp_official_amd.run_edit_sharded instead.")

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
            raise AsyrpDeviceError(f"{type(self).__name__} runs only on an MI355X (device type 'cuda' under ROCm); "
                                   "there is no CPU/PyTorch fallback")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        sig = (idx, self._n_delta, self.max_batch, self.nominal_batch)
        if self._engine is None or self._engine_sig != sig:
            self._drop_engine()
            self._engine = Engine(self._make_cfg(self._n_delta), self.max_batch, idx)
            self._engine_sig = sig
            self._engine.set_temb_freqs(self._temb_freqs())
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
