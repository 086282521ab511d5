"""Shared plumbing of the UNet mirrors: parameter holders with the reference's state_dict names + lazy HIP engine."""
import math
import threading

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


class _EngineSlot:
    """The engine of one device: created and re-synchronised under `lock`."""
    __slots__ = ("engine", "sig", "uploaded", "lock")

    def __init__(self):
        self.engine, self.sig, self.uploaded, self.lock = None, None, {}, threading.RLock()


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
        self._slots = {}                        # {device index: _EngineSlot}; shared with DataParallel replicas
        self._slots_lock = threading.RLock()
        for key, shape in param_specs(self._make_cfg(0)):
            _attach(self, key, shape)
        _default_init_(list(self.named_parameters()))

    def __getstate__(self):
        # copy.deepcopy / pickling: engines and locks belong to this process and this object; the copy builds its own on demand
        st = self.__dict__.copy()
        st["_slots"], st["_slots_lock"] = {}, None
        st.pop("_dp_source", None)
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        self._slots, self._slots_lock = {}, threading.RLock()

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
    # One engine per DEVICE, kept in `_slots` on the module the user built (the "source").  torch.nn.DataParallel replicas are
    # shallow copies whose `__dict__["_dp_source"]` names the source: every replica drives the source's engine of ITS device and
    # the engines take their weights from the source's parameters (by tensor version), never from the per-forward broadcast
    # copies DataParallel hands the replicas.
    def _src(self):
        return self.__dict__.get("_dp_source", self)

    @property
    def _engine(self):
        """The engine of the device the parameters live on (None before the first forward): single-device view of `_slots`."""
        src = self._src()
        p = next(src.parameters(), None)
        if p is None or p.device.type != "cuda":
            return None
        slot = src._slots.get(p.device.index if p.device.index is not None else torch.cuda.current_device())
        return slot.engine if slot is not None else None

    def _drop_engine(self):
        src = self._src()
        with src._slots_lock:
            slots, src._slots = list(src._slots.values()), {}
        for slot in slots:
            with slot.lock:
                if slot.engine is not None:
                    with slot.engine.lock:          # not while another thread is inside a call on it
                        slot.engine.close()
                slot.engine, slot.sig, slot.uploaded = None, None, {}

    def _apply(self, fn, *a, **k):
        # model.to(device) / .cuda(): engines of the devices left behind would keep their weights and workspace alive
        before = {p.device for p in self.parameters()}
        out = super()._apply(fn, *a, **k)
        if {p.device for p in self.parameters()} != before and "_slots" in self.__dict__:
            self._drop_engine()
        return out

    def _replicate_for_data_parallel(self):
        """torch.nn.DataParallel (the reference's only multi-GPU form: diffusion_latent.py:179,195,591,1201) replicates the
        module per device on EVERY forward and runs the replicas in threads.  A replica of the mirror is a light proxy: it shares
        the source's per-device engine table, so replica j drives the engine bound to device j (created on first use, weights
        uploaded once and re-synchronised when the SOURCE's parameters change — `model.module.layer_0.load_state_dict(...)`,
        an optimiser step).  The parameter copies DataParallel broadcasts to the replica are never read."""
        replica = super()._replicate_for_data_parallel()
        replica.__dict__["_dp_source"] = self._src()     # via __dict__: Module.__setattr__ would register it as a child
        return replica

    def set_schedule(self, betas):
        """Hand the beta schedule (the `b` the reference passes to denoising_step) to the engine."""
        src = self._src()
        with src._slots_lock:
            src._betas = betas.detach().float().cpu().clone()
            slots = list(src._slots.values())
        for slot in slots:
            with slot.lock:
                if slot.engine is not None:
                    slot.engine.set_schedule(alphas_cumprod_from_betas(src._betas))

    def engine(self, device=None):
        """The live HIP engine for `device` (created, and parameters re-synchronised, on demand)."""
        src = self._src()
        if device is None:
            device = next(src.parameters()).device
        device = torch.device(device)
        if device.type != "cuda":
            raise AsyrpDeviceError(f"{type(self).__name__} runs only on an MI355X (device type 'cuda' under ROCm); "
                                   "there is no CPU/PyTorch fallback")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        with src._slots_lock:
            slot = src._slots.get(idx)
            if slot is None:
                slot = src._slots[idx] = _EngineSlot()
            sig = (idx, src._n_delta, src.max_batch, src.nominal_batch)
        with slot.lock:
            if slot.engine is None or slot.sig != sig:
                if slot.engine is not None:
                    # a larger batch / another DeltaBlock count: the SAME Engine object gets a new native handle, so that references a
                    # caller took earlier (`eng = model.engine()`) stay valid instead of pointing at a destroyed engine
                    with slot.engine.lock:          # not while another thread is inside a call on the old handle
                        slot.engine.close()
                        slot.engine.__init__(src._make_cfg(src._n_delta), sig[2], idx)
                else:
                    slot.engine = Engine(src._make_cfg(src._n_delta), sig[2], idx)
                slot.uploaded = {}
                slot.sig = sig
                slot.engine.set_temb_freqs(src._temb_freqs())
                if getattr(src, "_betas", None) is not None:
                    slot.engine.set_schedule(alphas_cumprod_from_betas(src._betas))
            dirty = False
            for k, p in src.named_parameters():
                stamp = (p.data_ptr(), p._version, tuple(p.shape))
                if slot.uploaded.get(k) != stamp:
                    slot.engine.load_param(k, p)
                    slot.uploaded[k] = stamp
                    dirty = True
            if dirty:
                slot.engine.finalize()
            return slot.engine

    def _ready_engine(self, x):
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise AsyrpDeviceError("input must live on the GPU: the Asyrp HIP engine has no CPU fallback")
        src = self._src()
        with src._slots_lock:
            if x.shape[0] > src.max_batch:
                src.max_batch = int(x.shape[0])
        return self.engine(x.device)
