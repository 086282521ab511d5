"""Thin object wrapper over the C ABI: one Engine = one asyrp_engine on one GPU."""
import ctypes as C
import functools
import math
import threading

import numpy as np
import torch

from . import _lib


def make_config(*, family=_lib.FAMILY_DDPM, resolution, in_channels, out_channels, ch, ch_mult, num_res_blocks,
                attn_resolutions, num_head_channels=0, n_delta=0, conv_math="f16x3", num_classes=0, nominal_batch=0):
    cfg = _lib.AsyrpConfig()
    cfg.family, cfg.resolution, cfg.in_channels, cfg.out_channels = family, resolution, in_channels, out_channels
    cfg.ch, cfg.n_levels, cfg.num_res_blocks = ch, len(ch_mult), num_res_blocks
    for i, m in enumerate(ch_mult):
        cfg.ch_mult[i] = int(m)
    cfg.n_attn = len(attn_resolutions)
    for i, r in enumerate(attn_resolutions):
        cfg.attn_resolutions[i] = int(r)
    cfg.num_head_channels, cfg.n_delta = num_head_channels, n_delta
    cfg.conv_math = _lib.CONV_MATH[conv_math] if isinstance(conv_math, str) else int(conv_math)
    cfg.num_classes = int(num_classes)
    cfg.nominal_batch = int(nominal_batch)      # batch class (include/asyrp.h): 0 = priced at 32 images, 1 / 2 = the small class
    return cfg


def param_specs(cfg):
    """[(state_dict key, shape)] the engine expects for `cfg` — works without a GPU."""
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.asyrp_create(C.byref(h), C.byref(cfg), 1, 0))
    try:
        out = []
        key, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
        for i in range(lib.asyrp_num_params(h)):
            _lib.check(lib.asyrp_param_info(h, i, C.byref(key), shape, C.byref(nd)))
            out.append((key.value.decode(), tuple(int(shape[d]) for d in range(nd.value))))
        return out
    finally:
        lib.asyrp_destroy(h)


def ddpm_temb_freqs(ch):
    """exp(arange(half) * -(ln 1e4/(half-1))) in fp32, the very ops of models/ddpm/diffusion.py:51-54."""
    half = ch // 2
    rate = math.log(10000) / (half - 1)
    return torch.exp(torch.arange(half, dtype=torch.float32) * -rate)


def alphas_cumprod_from_betas(betas):
    """(1 - b).cumprod(0) in fp32 on the CPU, as utils/diffusion_utils.py:67 evaluates it."""
    return (1.0 - betas.detach().float().cpu()).cumprod(dim=0)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _dev_f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise AsyrpDeviceError(f"{name} must be a float32 CUDA(HIP) tensor — the Asyrp engine has no CPU path")
    return t.contiguous()


class AsyrpDeviceError(RuntimeError):
    pass


def _serialised(fn):
    """One call at a time per engine: the engine recycles its workspace on one in-order stream (include/asyrp.h), and ctypes
    releases the GIL, so two host threads sharing an engine (DataParallel replicas on ONE device) must take turns."""
    @functools.wraps(fn)
    def call(self, *a, **k):
        with self.lock:
            return fn(self, *a, **k)
    return call


class Engine:
    def __init__(self, cfg, max_batch, device_index):
        self.lib = _lib.load()
        if not hasattr(self, "lock"):            # re-initialised in place when the owner grows it (HipUNet.engine): keep the lock
            self.lock = threading.RLock()
        self.cfg, self.max_batch, self.device_index = cfg, int(max_batch), int(device_index)
        self.h = C.c_void_p()
        _lib.check(self.lib.asyrp_create(C.byref(self.h), C.byref(cfg), self.max_batch, self.device_index))
        self.out_channels, self.resolution = cfg.out_channels, cfg.resolution
        self.bott_ch = cfg.ch * cfg.ch_mult[cfg.n_levels - 1]      # both families end the encoder at ch * ch_mult[-1]
        self.bott_res = cfg.resolution >> (cfg.n_levels - 1)

    def __repr__(self):
        return (f"Engine(device=cuda:{self.device_index}, max_batch={self.max_batch}, resolution={self.resolution}, "
                f"library={getattr(self.lib, '_asyrp_path', '?')})")

    def close(self):
        if self.h:
            self.lib.asyrp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------------------
    @_serialised
    def load_param(self, key, tensor):
        a = np.ascontiguousarray(tensor.detach().float().cpu().numpy())
        shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        _lib.check(self.lib.asyrp_load_param(self.h, key.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))

    @_serialised
    def set_schedule(self, alphas_cumprod):
        a = np.ascontiguousarray(alphas_cumprod.detach().float().cpu().numpy())
        _lib.check(self.lib.asyrp_set_schedule(self.h, a.ctypes.data_as(C.c_void_p), a.size))

    @_serialised
    def set_temb_freqs(self, freqs):
        a = np.ascontiguousarray(freqs.detach().float().cpu().numpy())
        _lib.check(self.lib.asyrp_set_temb_freqs(self.h, a.ctypes.data_as(C.c_void_p), a.size))

    @_serialised
    def finalize(self):
        _lib.check(self.lib.asyrp_finalize_params(self.h))

    def device_bytes(self):
        return int(self.lib.asyrp_device_bytes(self.h))

    # ---- compute ----------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device_index).cuda_stream)

    def _coeff(self, hs_coeff, index):
        if index is None or index < 0:
            return None, 0
        hs = [float(v) for v in (hs_coeff if isinstance(hs_coeff, (tuple, list)) else (hs_coeff,))]
        arr = (C.c_float * len(hs))(*hs)
        return arr, len(hs)

    def _delta_in(self, delta_h, B):
        if delta_h is None:
            return None
        d = _dev_f32(delta_h, "delta_h")
        want = (B, self.bott_ch, self.bott_res, self.bott_res)
        if tuple(d.shape) != want:
            raise ValueError(f"delta_h must have the bottleneck shape {want}, got {tuple(d.shape)}")
        return d

    def _image(self, x, name, like=None):
        """float32 GPU tensor of shape [B, in_channels, R, R] on this engine's device (the kernels index by these sizes)."""
        x = _dev_f32(x, name)
        want = (self.cfg.in_channels, self.resolution, self.resolution)
        if x.dim() != 4 or tuple(x.shape[1:]) != want:
            raise ValueError(f"{name} must be [B, {want[0]}, {want[1]}, {want[2]}], got {tuple(x.shape)}")
        if like is not None and tuple(x.shape) != tuple(like.shape):
            raise ValueError(f"{name} must have the shape of the image batch {tuple(like.shape)}, got {tuple(x.shape)}")
        if x.shape[0] < 1 or x.shape[0] > self.max_batch:
            raise ValueError(f"batch {x.shape[0]} outside [1, max_batch={self.max_batch}]")
        if x.device.index != self.device_index:
            raise AsyrpDeviceError(f"{name} lives on {x.device}, the engine on cuda:{self.device_index}")
        return x

    @_serialised
    def unet_forward(self, x, t, index=None, apply_edit=False, hs_coeff=(1.0, 1.0), ignore_timestep=False,
                     delta_h=None, use_mask=False):
        x = self._image(x, "x")
        t = _dev_f32(t.float() if isinstance(t, torch.Tensor) else t, "t")
        B = x.shape[0]
        idx = -1 if index is None else int(index)
        R, br, bc = self.resolution, self.bott_res, self.bott_ch
        et = torch.empty((B, self.out_channels, R, R), device=x.device, dtype=torch.float32)
        et_mod = torch.empty_like(et) if idx >= 0 else None
        din = self._delta_in(delta_h, B)
        dh = (torch.empty((B, bc, br, br), device=x.device, dtype=torch.float32)
              if (idx >= 0 and apply_edit and din is None) else None)
        mid = torch.empty((B, bc, br, br), device=x.device, dtype=torch.float32)
        coeff, ncoeff = self._coeff(hs_coeff, idx)
        with torch.cuda.device(self.device_index):
            _lib.check(self.lib.asyrp_unet_forward(self.h, _ptr(x), _ptr(t), B, idx, int(bool(apply_edit)), coeff,
                                                   ncoeff, int(bool(ignore_timestep)), _ptr(din), int(bool(use_mask)),
                                                   _ptr(et), _ptr(et_mod), _ptr(dh), _ptr(mid), self._stream()))
        # with an injected delta_h the reference hands the caller's own tensor back (diffusion.py:580)
        return et, et_mod, (delta_h if delta_h is not None else dh), mid

    @_serialised
    def ddim_step(self, xt, t, t_next, *, eta=0.0, noise=None, learn_sigma=False, index=None, apply_edit=False,
                  hs_coeff=(1.0, 1.0), ignore_timestep=False, dt_lambda=1.0, dt_end=999, delta_h=None, use_mask=False):
        xt = self._image(xt, "xt")
        noise = self._image(noise, "noise", like=xt) if noise is not None else None
        B = xt.shape[0]
        idx = -1 if index is None else int(index)
        br, bc = self.bott_res, self.bott_ch
        xn, x0t = torch.empty_like(xt), torch.empty_like(xt)
        din = self._delta_in(delta_h, B)
        dh = (torch.empty((B, bc, br, br), device=xt.device, dtype=torch.float32)
              if (idx >= 0 and apply_edit and din is None) else None)
        mid = torch.empty((B, bc, br, br), device=xt.device, dtype=torch.float32)
        coeff, ncoeff = self._coeff(hs_coeff, idx)
        with torch.cuda.device(self.device_index):
            _lib.check(self.lib.asyrp_ddim_step(self.h, _ptr(xt), int(t), int(t_next), B, float(eta), _ptr(noise),
                                                int(bool(learn_sigma)), idx, int(bool(apply_edit)), coeff, ncoeff,
                                                int(bool(ignore_timestep)), _ptr(din), int(bool(use_mask)),
                                                float(dt_lambda), int(dt_end), _ptr(xn),
                                                _ptr(x0t), _ptr(dh), _ptr(mid), self._stream()))
        return xn, x0t, (delta_h if delta_h is not None else dh), mid

    @_serialised
    def run_edit(self, x0, seq_inv, seq_gen, *, t_edit, t_addnoise=0, index=0, hs_coeff=(1.0, 1.0),
                 learn_sigma=False, noise=None, want_latent=False):
        x0 = self._image(x0, "x0")
        B = x0.shape[0]
        idx = -1 if index is None else int(index)
        si = (C.c_int32 * max(len(seq_inv), 1))(*[int(v) for v in seq_inv])
        sg = (C.c_int32 * len(seq_gen))(*[int(v) for v in seq_gen])
        if noise is not None:
            noise = _dev_f32(noise, "noise")
            if noise.dim() != 5 or tuple(noise.shape[1:]) != tuple(x0.shape):
                raise ValueError(f"noise must be [n_eta_steps, {', '.join(map(str, x0.shape))}], got {tuple(noise.shape)}")
        n_noise = 0 if noise is None else int(noise.shape[0])
        x_T = torch.empty_like(x0) if want_latent else None
        x_edit = torch.empty_like(x0)
        if idx >= 0 and len(hs_coeff) > 0 and isinstance(hs_coeff[0], (tuple, list)):
            # one tuple per image (a strength sweep as batch entries): [B][index + 2], signalled by a negative count (include/asyrp.h)
            if len(hs_coeff) != B or any(len(hc) != idx + 2 for hc in hs_coeff):
                raise ValueError(f"per-image hs_coeff must be {B} tuples of {idx + 2} coefficients")
            flat = [float(v) for hc in hs_coeff for v in hc]
            coeff, ncoeff = (C.c_float * len(flat))(*flat), -len(flat)
        else:
            coeff, ncoeff = self._coeff(hs_coeff, idx)
        with torch.cuda.device(self.device_index):
            _lib.check(self.lib.asyrp_run_edit(self.h, _ptr(x0), B, si, len(seq_inv), sg, len(seq_gen), int(t_edit),
                                               int(t_addnoise), idx, coeff, ncoeff, int(bool(learn_sigma)),
                                               _ptr(noise), n_noise, _ptr(x_T), _ptr(x_edit), self._stream()))
        return (x_edit, x_T) if want_latent else x_edit

    @_serialised
    def run_inversion(self, x0, seq_inv, *, learn_sigma=False, tap_first=0, tap_count=0, want_x=True, want_x0t=True):
        """DDIM inversion with a per-step read-out (asyrp_run_inversion): returns (x_last, x_tap, x0t_tap), the taps shaped
        [tap_count, B, 3, R, R] (None when not requested / tap_count == 0)."""
        x0 = self._image(x0, "x0")
        B = x0.shape[0]
        si = (C.c_int32 * len(seq_inv))(*[int(v) for v in seq_inv])
        shape = (int(tap_count),) + tuple(x0.shape)
        x_tap = torch.empty(shape, device=x0.device, dtype=torch.float32) if (tap_count and want_x) else None
        x0t_tap = torch.empty(shape, device=x0.device, dtype=torch.float32) if (tap_count and want_x0t) else None
        x_last = torch.empty_like(x0)
        with torch.cuda.device(self.device_index):
            _lib.check(self.lib.asyrp_run_inversion(self.h, _ptr(x0), B, si, len(seq_inv), int(bool(learn_sigma)),
                                                    int(tap_first), int(tap_count), _ptr(x_tap), _ptr(x0t_tap),
                                                    _ptr(x_last), self._stream()))
        return x_last, x_tap, x0t_tap

    # ---- DeltaBlock training step (asyrp_train_forward / asyrp_train_backward) ----------------------
    @_serialised
    def train_forward(self, xt, t, t_next, *, hs_coeff=(1.0, 1.0), ignore_timestep=False, learn_sigma=False):
        """-> (xt_next, x0_t, delta_h, middle_h, tape_id); `tape_id` names the recorded step for train_backward / train_discard."""
        xt = self._image(xt, "xt")
        B = xt.shape[0]
        br, bc = self.bott_res, self.bott_ch
        xn, x0t = torch.empty_like(xt), torch.empty_like(xt)
        dh = torch.empty((B, bc, br, br), device=xt.device, dtype=torch.float32)
        mid = torch.empty((B, bc, br, br), device=xt.device, dtype=torch.float32)
        coeff, ncoeff = self._coeff(hs_coeff, 0)
        tid = C.c_int64(-1)
        with torch.cuda.device(self.device_index):
            _lib.check(self.lib.asyrp_train_forward(self.h, _ptr(xt), int(t), int(t_next), B, int(bool(learn_sigma)), coeff,
                                                    ncoeff, int(bool(ignore_timestep)), _ptr(xn), _ptr(x0t), _ptr(dh),
                                                    _ptr(mid), C.byref(tid), self._stream()))
        self._tape_batch = {int(tid.value): B}
        return xn, x0t, dh, mid, int(tid.value)

    @_serialised
    def train_backward(self, tape_id, d_et_mod, named_shapes):
        """d_et_mod [B,Cout,R,R] for the step `tape_id`; named_shapes: [(state_dict key, shape)] of DeltaBlock parameters ->
        list of gradients (zero-initialised device tensors the engine fills)."""
        d = _dev_f32(d_et_mod, "d_et_mod")
        B = getattr(self, "_tape_batch", {}).get(int(tape_id))
        if B is not None and tuple(d.shape) != (B, self.out_channels, self.resolution, self.resolution):
            raise ValueError(f"d_et_mod must be {(B, self.out_channels, self.resolution, self.resolution)} (the recorded step's "
                             f"eps~), got {tuple(d.shape)}")
        if d.device.index != self.device_index:
            raise AsyrpDeviceError(f"d_et_mod lives on {d.device}, the engine on cuda:{self.device_index}")
        grads = [torch.zeros(tuple(shape), device=d.device, dtype=torch.float32) for _, shape in named_shapes]
        keys = (C.c_char_p * len(grads))(*[k.encode() for k, _ in named_shapes])
        ptrs = (C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        with torch.cuda.device(self.device_index):
            _lib.check(self.lib.asyrp_train_backward(self.h, int(tape_id), _ptr(d), len(grads), keys, ptrs, self._stream()))
        return grads

    def train_discard(self, tape_id=-1):
        """Drop the pending training step `tape_id` (no-op when a later step has replaced it; -1: whatever is pending)."""
        if self.h:
            self.lib.asyrp_train_discard(self.h, int(tape_id))

    @_serialised
    def get_temb(self, t):
        t = _dev_f32(t.float() if isinstance(t, torch.Tensor) else t, "t")
        out = torch.empty((t.shape[0], self.cfg.ch * 4), device=t.device, dtype=torch.float32)
        with torch.cuda.device(self.device_index):
            _lib.check(self.lib.asyrp_get_temb(self.h, _ptr(t), int(t.shape[0]), _ptr(out), self._stream()))
        return out

    # ---- profiling --------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        _lib.check(self.lib.asyrp_profile_enable(self.h, int(bool(on))))

    @staticmethod
    def variant_name(v):
        """Kernel name (as rocprofv3 prints it) and family of a profile variant id (include/asyrp.h)."""
        fam, tid = v // 100000, (v // 1000) % 100
        ks, stride = (v // 100) % 10, (v // 10) % 10
        if fam == 2:
            if v >= 210000:     # the split-plane kernel (round 3): operands arrive as f16 hi/lo planes from the q|k|v projection
                return "attention", "asyrp::attn_planes_kernel (T=%d)" % (v - 210000)
            return "attention", "asyrp::attn_f16x3_kernel (T=%d)" % (v - 200000)
        if fam == 3:
            return "f16x3", "asyrp::conv_out_kernel (Cout=%d)" % (v - 300000)
        if fam == 4:   # conv_in.hip: the one-K-step MFMA form (round 5, default) or the fp32 stencil (ASYRP_CONV_IN_MFMA=0)
            import os
            stencil = os.environ.get("ASYRP_LIBRARY") == "bench" and os.environ.get("ASYRP_CONV_IN_MFMA", "1")[:1] == "0"
            k = "conv_in_kernel" if stencil else "conv_in_mfma_kernel"
            return "f16x3", "asyrp::%s (Cout=%d)" % (k, v - 400000)
        if fam == 1:
            tile = {1: (4, 1, 2, 4), 2: (2, 2, 2, 2), 3: (2, 2, 1, 2), 4: (2, 2, 1, 1), 5: (4, 1, 2, 2), 6: (4, 2, 2, 2),
                    12: (4, 1, 2, 1)}.get(tid, (0,) * 4)
            if tid in (15, 16):                 # gemm1x1.hip: the barrier-free 1x1 kernel (8 / 4 waves)
                return "f16x3", "asyrp::gemm1x1_k32_kernel (WM=%d)" % (4 if tid == 15 else 2)
            if tid in (7, 8, 9, 10, 11, 14):    # the v_mfma_f32_16x16x32_f16 kernels: 256- / 128- / 64-pixel forms, stride 2, polyphase x2, quad 8x8
                return "f16x3", "asyrp::igemm_f16x3_k32_kernel<asyrp::K32Cfg<%s>>" % {7: "8, 2", 8: "8, 4", 9: "8, 8, 8", 10: "8, 8, 16, 2",
                                                                                     11: "8, 2, 16, 1, 2", 14: "8, 2, 16, 1, 3, true"}[tid]
            return "f16x3", "asyrp::igemm_f16x3_kernel<asyrp::XCfg<%d, %d, %d, %d, %d, %d>>" % (tile + (ks, stride))
        tile = {1: (2, 2, 2, 2), 2: (2, 2, 2, 1), 3: (2, 2, 1, 1), 4: (4, 1, 1, 1)}.get(tid, (0,) * 4)
        return "f32", "asyrp::igemm_f32_kernel<asyrp::TileCfg<%d, %d, %d, %d, %d, %d>>" % (tile + (ks, stride))

    def profile_table(self, max_rows=64):
        """Per-kernel-family rows recorded since the last profile_read (does not reset the record)."""
        var = (C.c_int * max_rows)()
        ms, fl, by = (C.c_double * max_rows)(), (C.c_double * max_rows)(), (C.c_double * max_rows)()
        n = (C.c_int64 * max_rows)()
        rows = self.lib.asyrp_profile_table(self.h, max_rows, var, ms, n, fl, by)
        if rows < 0:
            _lib.check(rows)
        out = []
        for i in range(rows):
            fam, name = self.variant_name(var[i])
            out.append(dict(variant=int(var[i]), family=fam, kernel=name, ms=float(ms[i]), launches=int(n[i]),
                            flops=float(fl[i]), bytes=float(by[i])))
        return out

    def profile_read(self):
        """Stats of the dominant implicit-GEMM kernel (and of all GEMM launches) since the last read."""
        var, n = C.c_int(), C.c_int64()
        ms, fl, by, ams, afl = C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_double()
        _lib.check(self.lib.asyrp_profile_read(self.h, C.byref(var), C.byref(ms), C.byref(n), C.byref(fl),
                                               C.byref(by), C.byref(ams), C.byref(afl)))
        v = var.value
        fam_name, name = self.variant_name(v)
        return dict(variant=v, family=fam_name, kernel=name, ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value,
                    all_ms=ams.value, all_flops=afl.value)
