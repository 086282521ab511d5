"""Vendored-sampler signatures of the reference (third row): `GaussianDiffusion.p_sample`, `ddim_sample`,
`ddim_reverse_sample` and `p_mean_variance` of models/guided_diffusion/gaussian_diffusion.py:232-321, 402-446, 544-630
(the same class is vendored under models/improved_ddpm/), as thin wrappers over the engine-backed UNets.

The reference never calls them, and as vendored they cannot drive its own Asyrp-modified UNets: `p_mean_variance` expects
`model(x, t)` to return one tensor (:267) while the modified `forward` returns the 4-tuple (et, et_modified, delta_h, middle_h).
Here the first element (the un-edited eps, with the variance channels when learn_sigma) is used, which is what the upstream
guided-diffusion code these methods come from would see.  Everything but the model call is elementwise arithmetic on
[B,3,R,R] tensors with the reference's float64 numpy tables (:135-171), done in torch on the tensors' device.
"""
import numpy as np
import torch


def _extract_into_tensor(arr, timesteps, broadcast_shape):
    """gaussian_diffusion.py:_extract_into_tensor: float64 table -> float tensor gathered at `timesteps`, broadcastable."""
    res = torch.from_numpy(arr).to(device=timesteps.device)[timesteps].float()
    while len(res.shape) < len(broadcast_shape):
        res = res[..., None]
    return res.expand(broadcast_shape)


class GaussianDiffusion:
    """model_mean_type: 'epsilon' (every reference config).  model_var_type: 'fixed_small' | 'fixed_large' | 'learned_range'
    (learn_sigma=True networks: AFHQ / ImageNet / MetFaces / CelebA-HQ-P2)."""

    def __init__(self, *, betas, model_mean_type="epsilon", model_var_type="fixed_large", rescale_timesteps=False):
        if model_mean_type != "epsilon":
            raise NotImplementedError("every model of the reference predicts epsilon")
        if model_var_type not in ("fixed_small", "fixed_large", "learned_range"):
            raise NotImplementedError(model_var_type)
        self.model_mean_type, self.model_var_type, self.rescale_timesteps = model_mean_type, model_var_type, rescale_timesteps
        betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)

    # ---- pieces of the reference class the three samplers use ----------------------------------------------------------
    def _scale_timesteps(self, t):
        return t.float() * (1000.0 / self.num_timesteps) if self.rescale_timesteps else t

    def _predict_xstart_from_eps(self, x_t, t, eps):
        return (_extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - _extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * eps)

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        return ((_extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - pred_xstart)
                / _extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape))

    def q_posterior_mean_variance(self, x_start, x_t, t):
        mean = (_extract_into_tensor(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + _extract_into_tensor(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return (mean, _extract_into_tensor(self.posterior_variance, t, x_t.shape),
                _extract_into_tensor(self.posterior_log_variance_clipped, t, x_t.shape))

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """gaussian_diffusion.py:232-321.  `t` is an integer tensor [B]."""
        model_kwargs = model_kwargs or {}
        B, C = x.shape[:2]
        assert t.shape == (B,)
        out = model(x, self._scale_timesteps(t), **model_kwargs)
        model_output = out[0] if isinstance(out, (tuple, list)) else out       # Asyrp UNets return (et, et_mod, delta_h, middle_h)
        if self.model_var_type == "learned_range":
            assert model_output.shape == (B, C * 2, *x.shape[2:])
            model_output, model_var_values = torch.split(model_output, C, dim=1)
            min_log = _extract_into_tensor(self.posterior_log_variance_clipped, t, x.shape)
            max_log = _extract_into_tensor(np.log(self.betas), t, x.shape)
            frac = (model_var_values + 1) / 2
            model_log_variance = frac * max_log + (1 - frac) * min_log
            model_variance = torch.exp(model_log_variance)
        else:
            var, logvar = {
                "fixed_large": (np.append(self.posterior_variance[1], self.betas[1:]),
                                np.log(np.append(self.posterior_variance[1], self.betas[1:]))),
                "fixed_small": (self.posterior_variance, self.posterior_log_variance_clipped),
            }[self.model_var_type]
            model_variance = _extract_into_tensor(var, t, x.shape)
            model_log_variance = _extract_into_tensor(logvar, t, x.shape)
        pred_xstart = self._predict_xstart_from_eps(x_t=x, t=t, eps=model_output)
        if denoised_fn is not None:
            pred_xstart = denoised_fn(pred_xstart)
        if clip_denoised:
            pred_xstart = pred_xstart.clamp(-1, 1)
        model_mean, _, _ = self.q_posterior_mean_variance(x_start=pred_xstart, x_t=x, t=t)
        return {"mean": model_mean, "variance": model_variance, "log_variance": model_log_variance, "pred_xstart": pred_xstart}

    # ---- the three vendored sampler signatures -------------------------------------------------------------------------
    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, noise=None):
        """:402-446.  `noise` (extra keyword) replaces torch.randn_like for reproducible comparisons."""
        if cond_fn is not None:
            raise NotImplementedError("classifier guidance (cond_fn) is not part of the Asyrp path")
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        noise = torch.randn_like(x) if noise is None else noise
        nonzero_mask = (t != 0).float().view(-1, *([1] * (len(x.shape) - 1)))
        sample = out["mean"] + nonzero_mask * torch.exp(0.5 * out["log_variance"]) * noise
        return {"sample": sample, "pred_xstart": out["pred_xstart"]}

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0, noise=None):
        """:544-592."""
        if cond_fn is not None:
            raise NotImplementedError("classifier guidance (cond_fn) is not part of the Asyrp path")
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        eps = self._predict_eps_from_xstart(x, t, out["pred_xstart"])
        alpha_bar = _extract_into_tensor(self.alphas_cumprod, t, x.shape)
        alpha_bar_prev = _extract_into_tensor(self.alphas_cumprod_prev, t, x.shape)
        sigma = eta * torch.sqrt((1 - alpha_bar_prev) / (1 - alpha_bar)) * torch.sqrt(1 - alpha_bar / alpha_bar_prev)
        noise = torch.randn_like(x) if noise is None else noise
        mean_pred = out["pred_xstart"] * torch.sqrt(alpha_bar_prev) + torch.sqrt(1 - alpha_bar_prev - sigma ** 2) * eps
        nonzero_mask = (t != 0).float().view(-1, *([1] * (len(x.shape) - 1)))
        return {"sample": mean_pred + nonzero_mask * sigma * noise, "pred_xstart": out["pred_xstart"]}

    def ddim_reverse_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None, eta=0.0):
        """:594-630 (deterministic reverse ODE step x_t -> x_{t+1})."""
        assert eta == 0.0, "Reverse ODE only for deterministic path"
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        eps = ((_extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x.shape) * x - out["pred_xstart"])
               / _extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x.shape))
        alpha_bar_next = _extract_into_tensor(self.alphas_cumprod_next, t, x.shape)
        mean_pred = out["pred_xstart"] * torch.sqrt(alpha_bar_next) + torch.sqrt(1 - alpha_bar_next) * eps
        return {"sample": mean_pred, "pred_xstart": out["pred_xstart"]}
