"""Sampler signatures of the reference's vendored `GaussianDiffusion` (third row) on the engine.

The reference carries guided-diffusion's sampler class (models/guided_diffusion/gaussian_diffusion.py; `p_mean_variance`
:232-321, `p_sample` :402-446, `ddim_sample` :544-592, `ddim_reverse_sample` :594-630) but never calls it.  This module keeps
those four call signatures and result dictionaries for scripts written against them, with a different construction:

* every one of the updates is affine in (x, pred_xstart, noise) once the timestep is fixed,
      pred_xstart = clamp?(a*x - b*eps)          sample = p*pred_xstart + q*x + r*noise
  so the float64 schedule is folded ON THE HOST into one coefficient row (a, b, p, q, r, lo, hi, clip) per image
  (`SamplerSchedule.rows`), and
* the tensor arithmetic is ONE elementwise launch of the HIP library (`asyrp_sampler_update`, csrc/kernels.hip) that reads the
  UNet output in place (eps = channels 0..2, learned-variance channels 3..5) - no table gathers, no intermediate tensors.

The Asyrp-modified UNets return a 4-tuple (et, et_modified, delta_h, middle_h); its first element (the un-edited eps, plus the
variance channels of learn_sigma networks) is what the upstream sampler code would see and is what is used here.
There is no CPU path: tensors must live on the GPU, like everywhere else in this package.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .engine import AsyrpDeviceError

VAR_TYPES = ("fixed_small", "fixed_large", "learned_range")
ROW = 8   # floats per coefficient row: a, b, p, q, r, lo, hi, clip


class SamplerSchedule:
    """Float64 diffusion tables and the closed-form coefficient rows derived from them (host only, numpy)."""

    def __init__(self, betas):
        beta = np.asarray(betas, dtype=np.float64).reshape(-1)
        if not ((beta > 0).all() and (beta <= 1).all()):
            raise ValueError("betas must lie in (0, 1]")
        self.beta = beta
        self.T = beta.size
        self.abar = np.cumprod(1.0 - beta)
        self.abar_prev = np.concatenate(([1.0], self.abar[:-1]))
        self.abar_next = np.concatenate((self.abar[1:], [0.0]))
        post = beta * (1.0 - self.abar_prev) / (1.0 - self.abar)            # posterior variance of q(x_{t-1} | x_t, x_0)
        self.log_post = np.log(np.concatenate((post[1:2], post[1:])))       # its log with entry 0 replaced by entry 1
        self.log_large = np.log(np.concatenate((post[1:2], beta[1:])))      # "fixed_large": beta_t, same replacement at 0

    def fixed_log_variance(self, var_type, t):
        return (self.log_large if var_type == "fixed_large" else self.log_post)[t]

    def rows(self, kind, t, *, var_type, clip, eta=0.0, noisy=True):
        """[B, ROW] float32 rows for integer timesteps `t` ([B] array).
        kind: 'posterior' (p_sample / p_mean_variance), 'ddim', 'ddim_reverse'."""
        t = np.asarray(t, dtype=np.int64).reshape(-1)
        ab = self.abar[t]
        a = np.sqrt(1.0 / ab)                     # pred_xstart = a*x - b*eps
        b = np.sqrt(1.0 / ab - 1.0)
        lo = hi = np.zeros_like(ab)
        live = (t != 0).astype(np.float64)        # the samplers add no noise at t == 0
        if kind == "posterior":
            prev = self.abar_prev[t]
            p = self.beta[t] * np.sqrt(prev) / (1.0 - ab)
            q = (1.0 - prev) * np.sqrt(1.0 - self.beta[t]) / (1.0 - ab)
            if var_type == "learned_range":       # per-element log-variance between lo and hi, evaluated by the kernel
                r, lo, hi = live, self.log_post[t], np.log(self.beta[t])
            else:
                r = live * np.exp(0.5 * self.fixed_log_variance(var_type, t))
        elif kind in ("ddim", "ddim_reverse"):
            to = self.abar_prev[t] if kind == "ddim" else self.abar_next[t]
            sigma = np.zeros_like(ab)
            if kind == "ddim" and eta != 0.0:
                sigma = eta * np.sqrt((1.0 - to) / (1.0 - ab)) * np.sqrt(1.0 - ab / to)
            k = np.sqrt(np.maximum(1.0 - to - sigma ** 2, 0.0))
            # sample = sqrt(to)*x0 + k*eps', with eps' = (a*x - x0)/b re-derived from the (possibly clamped) x0
            p = np.sqrt(to) - k / b
            q = k * a / b
            r = sigma * live
        else:
            raise ValueError(kind)
        if not noisy:
            r = np.zeros_like(ab)
        out = np.stack([a, b, p, q, r, lo, hi, np.full_like(ab, 1.0 if clip else 0.0)], axis=1)
        return np.ascontiguousarray(out, dtype=np.float32)


def _gpu_f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise AsyrpDeviceError(f"{name} must be a CUDA(HIP) tensor - the Asyrp engine has no CPU path")
    return t.float().contiguous()


def sampler_update(x, model_out, rows, *, noise=None, want_sample=True, want_xstart=True, want_log_variance=False):
    """One `asyrp_sampler_update` launch on x's device and current stream -> (sample, pred_xstart, log_variance)."""
    x = _gpu_f32(x, "x")
    model_out = _gpu_f32(model_out, "model output")
    B, Cx = x.shape[:2]
    HW = int(np.prod(x.shape[2:]))
    if model_out.shape[0] != B or tuple(model_out.shape[2:]) != tuple(x.shape[2:]):
        raise ValueError(f"model output {tuple(model_out.shape)} does not match x {tuple(x.shape)}")
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    assert rows.shape == (B, ROW)
    if noise is not None:
        noise = _gpu_f32(noise, "noise")
        assert noise.shape == x.shape
    for name, t_ in (("model output", model_out), ("noise", noise)):   # raw pointers are handed to x's device: they must live there
        if t_ is not None and t_.device != x.device:
            raise AsyrpDeviceError(f"{name} is on {t_.device}, x on {x.device}")
    sample = torch.empty_like(x) if want_sample else None
    xstart = torch.empty_like(x) if want_xstart else None
    logvar = torch.empty_like(x) if want_log_variance else None
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    lib = _lib.load()
    with torch.cuda.device(x.device):   # the entry point selects x's device for the calling thread: restore the caller's afterwards
        _lib.check(lib.asyrp_sampler_update(x.device.index, ptr(x), ptr(model_out), int(model_out.shape[1]), B, Cx, HW,
                                            rows.ctypes.data_as(C.c_void_p), ptr(noise), ptr(sample), ptr(xstart), ptr(logvar),
                                            C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    return sample, xstart, logvar


class GaussianDiffusion:
    """Call-compatible with the vendored class for the epsilon-predicting models the reference builds.
    model_var_type: 'fixed_small' | 'fixed_large' | 'learned_range' (learn_sigma networks)."""

    def __init__(self, *, betas, model_mean_type="epsilon", model_var_type="fixed_large", rescale_timesteps=False):
        if model_mean_type != "epsilon":
            raise NotImplementedError("every model of the reference predicts epsilon")
        if model_var_type not in VAR_TYPES:
            raise NotImplementedError(model_var_type)
        self.model_mean_type, self.model_var_type, self.rescale_timesteps = model_mean_type, model_var_type, rescale_timesteps
        self.schedule = SamplerSchedule(betas)
        self.num_timesteps = self.schedule.T
        self.betas, self.alphas_cumprod = self.schedule.beta, self.schedule.abar
        # The schedule tables scripts read off the vendored class (float64 numpy, models/guided_diffusion/gaussian_diffusion.py:143-176),
        # derived here from beta / alpha_bar; the sampling path itself uses SamplerSchedule's coefficient rows, not these.
        ab, be = np.asarray(self.alphas_cumprod, dtype=np.float64), np.asarray(self.betas, dtype=np.float64)
        self.alphas_cumprod_prev = np.concatenate([[1.0], ab[:-1]])
        self.alphas_cumprod_next = np.concatenate([ab[1:], [0.0]])
        self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod = np.sqrt(ab), np.sqrt(1.0 - ab)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - ab)
        self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ab), np.sqrt(1.0 / ab - 1.0)
        self.posterior_variance = be * (1.0 - self.alphas_cumprod_prev) / (1.0 - ab)
        self.posterior_log_variance_clipped = np.log(np.concatenate([self.posterior_variance[1:2], self.posterior_variance[1:]]))
        self.posterior_mean_coef1 = be * np.sqrt(self.alphas_cumprod_prev) / (1.0 - ab)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(1.0 - be) / (1.0 - ab)

    # ---- small helpers of the vendored class (plain torch on the tensors' own device; not on the sampling path) ------------------
    @staticmethod
    def _table(arr, t, like):
        v = torch.from_numpy(np.asarray(arr)).to(device=like.device, dtype=torch.float32)[t.long()]
        return v.view(-1, *([1] * (like.dim() - 1)))

    def _scale_timesteps(self, t):
        return t.float() * (1000.0 / self.num_timesteps) if self.rescale_timesteps else t

    def _predict_xstart_from_eps(self, x_t, t, eps):
        return self._table(self.sqrt_recip_alphas_cumprod, t, x_t) * x_t - self._table(self.sqrt_recipm1_alphas_cumprod, t, x_t) * eps

    def q_sample(self, x_start, t, noise=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        return self._table(self.sqrt_alphas_cumprod, t, x_start) * x_start + self._table(self.sqrt_one_minus_alphas_cumprod, t, x_start) * noise

    def q_posterior_mean_variance(self, x_start, x_t, t):
        mean = self._table(self.posterior_mean_coef1, t, x_t) * x_start + self._table(self.posterior_mean_coef2, t, x_t) * x_t
        var = self._table(self.posterior_variance, t, x_t).expand(x_t.shape)
        logvar = self._table(self.posterior_log_variance_clipped, t, x_t).expand(x_t.shape)
        return mean, var, logvar

    # ---- shared plumbing ------------------------------------------------------------------------------------------------
    def _model_output(self, model, x, t, model_kwargs):
        out = model(x, self._scale_timesteps(t), **(model_kwargs or {}))
        out = out[0] if isinstance(out, (tuple, list)) else out
        want = x.shape[1] * (2 if self.model_var_type == "learned_range" else 1)
        if out.shape[1] != want:
            raise ValueError(f"model returned {out.shape[1]} channels, {self.model_var_type} expects {want}")
        return out

    def _step(self, kind, model, x, t, *, clip_denoised, denoised_fn, model_kwargs, eta=0.0, noise=None, noisy=True,
              want_log_variance=False):
        """pred_xstart and the affine update of `kind` in one launch (two when `denoised_fn` has to see pred_xstart)."""
        tt = t.detach().cpu().numpy()
        assert tt.shape == (x.shape[0],)
        out = self._model_output(model, x, t, model_kwargs)
        rows = self.schedule.rows(kind, tt, var_type=self.model_var_type, clip=clip_denoised, eta=eta, noisy=noisy)
        if noisy and noise is None:
            noise = torch.randn_like(x)     # drawn unconditionally, as the vendored class does (its RNG stream also advances at t == 0)
        if not noisy or not bool((rows[:, 4] != 0).any()):
            noise = None
        if denoised_fn is None:
            return (out,) + sampler_update(x, out, rows, noise=noise, want_log_variance=want_log_variance)
        # x0 first (unclamped), the callback, then the update with x0 handed in as if it were the model output: a = 0, b = -1
        raw = rows.copy()
        raw[:, 7] = 0.0
        _, x0, logvar = sampler_update(x, out, raw, want_sample=False, want_log_variance=want_log_variance)
        x0 = denoised_fn(x0)
        rows2 = rows.copy()
        rows2[:, 0], rows2[:, 1] = 0.0, -1.0
        carrier = x0 if out.shape[1] == x.shape[1] else torch.cat([x0, out[:, x.shape[1]:]], dim=1)
        sample, x0, _ = sampler_update(x, carrier, rows2, noise=noise)
        return out, sample, x0, logvar

    def _variance_maps(self, x, t, logvar):
        if self.model_var_type == "learned_range":
            return torch.exp(logvar), logvar
        lv = torch.from_numpy(self.schedule.fixed_log_variance(self.model_var_type, t.detach().cpu().numpy())).float().to(x.device)
        lv = lv.view(-1, *([1] * (x.dim() - 1))).expand(x.shape)
        return torch.exp(lv), lv

    # ---- the vendored signatures ----------------------------------------------------------------------------------------
    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """:232-321 -> {"mean", "variance", "log_variance", "pred_xstart"}; `t` is an integer tensor [B]."""
        _, mean, x0, logvar = self._step("posterior", model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                         model_kwargs=model_kwargs, noisy=False,
                                         want_log_variance=self.model_var_type == "learned_range")
        var, logvar = self._variance_maps(x, t, logvar)
        return {"mean": mean, "variance": var, "log_variance": logvar, "pred_xstart": x0}

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, noise=None):
        """:402-446 -> {"sample", "pred_xstart"}.  `noise` (extra keyword) stands in for torch.randn_like."""
        if cond_fn is not None:
            raise NotImplementedError("classifier guidance (cond_fn) is not part of the Asyrp path")
        _, sample, x0, _ = self._step("posterior", model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                      model_kwargs=model_kwargs, noise=noise)
        return {"sample": sample, "pred_xstart": x0}

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0, noise=None):
        """:544-592 -> {"sample", "pred_xstart"}."""
        if cond_fn is not None:
            raise NotImplementedError("classifier guidance (cond_fn) is not part of the Asyrp path")
        _, sample, x0, _ = self._step("ddim", model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                      model_kwargs=model_kwargs, eta=float(eta), noise=noise)
        return {"sample": sample, "pred_xstart": x0}

    def ddim_reverse_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None, eta=0.0):
        """:594-630 -> {"sample", "pred_xstart"}: the deterministic x_t -> x_{t+1} step."""
        if eta != 0.0:
            raise ValueError("the reverse ODE step is deterministic (eta must be 0)")
        _, sample, x0, _ = self._step("ddim_reverse", model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                      model_kwargs=model_kwargs, noisy=False)
        return {"sample": sample, "pred_xstart": x0}
