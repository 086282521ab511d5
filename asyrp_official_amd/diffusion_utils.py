"""Mirror of the reference's utils/diffusion_utils.py (get_beta_schedule :5-9, extract :12-20,
denoising_step :24-104) on top of the HIP engine's fused DDIM step (asyrp_ddim_step)."""
import numpy as np
import torch

from . import data_parallel, training
from ._base import HipUNet


def get_beta_schedule(*, beta_start, beta_end, num_diffusion_timesteps):
    betas = np.linspace(beta_start, beta_end, num_diffusion_timesteps, dtype=np.float64)
    assert betas.shape == (num_diffusion_timesteps,)
    return betas


def extract(a, t, x_shape):
    """Gather coefficients of `a` at integer timesteps `t`, broadcastable to x_shape (:12-20)."""
    bs, = t.shape
    assert x_shape[0] == bs, f"{x_shape[0]}, {t.shape}"
    out = torch.as_tensor(a, dtype=torch.float, device=t.device)[t.long()]
    return out.reshape((bs,) + (1,) * (len(x_shape) - 1))


def _unwrap(models):
    m = models.module if isinstance(models, torch.nn.DataParallel) else models
    if not isinstance(m, HipUNet):
        raise TypeError(f"denoising_step expects an asyrp_official_amd UNet, got {type(m).__name__}")
    return m


def _uniform_int(t, name):
    v = int(t[0].item())           # same host sync the reference performs (diffusion_utils.py:68, diffusion.py:510)
    if t.numel() > 1 and not bool((t == t[0]).all()):
        raise ValueError(f"{name} must hold one timestep for the whole batch (the reference always builds ones(B)*i)")
    return v


def denoising_step(xt, t, t_next, *, models, logvars=None, b, sampling_type='ddim', eta=0.0, learn_sigma=False,
                   index=None, t_edit=0, hs_coeff=(1.0), delta_h=None, use_mask=False, dt_lambda=1,
                   ignore_timestep=False, image_space_noise=0, dt_end=999, warigari=False, noise=None):
    """One DDIM step; returns (xt_next, x0_t, delta_h, middle_h) exactly like the reference.

    Differences, all loud: only sampling_type='ddim' (the 'ddpm' branch of the reference leaves x0_t
    undefined, :73-82); `image_space_noise` is outside the accelerated path; a `delta_h` tensor selects
    the reference's slerp injection (models/ddpm/diffusion.py:518-539) and is handed back unchanged;
    when eta != 0 the Gaussian noise may be passed as `noise=` (bit-parity with a CPU generator),
    otherwise it is drawn on the GPU with torch.randn_like as the reference does.
    """
    if sampling_type != 'ddim':
        raise NotImplementedError("only sampling_type='ddim' is accelerated")
    if type(image_space_noise) != int:
        raise NotImplementedError("image_space_noise selects a DiffStyle branch outside the hot path")
    model = _unwrap(models)
    model.set_schedule(b) if getattr(model, "_betas", None) is None or not torch.equal(
        model._betas, b.detach().float().cpu()) else None
    if not (isinstance(xt, torch.Tensor) and xt.is_cuda):
        model._ready_engine(xt)        # raises AsyrpDeviceError: no CPU path
    ti, tn = _uniform_int(t, "t"), _uniform_int(t_next, "t_next")
    apply_edit = index is not None and ti >= t_edit
    if training.wants_training(model, index, apply_edit):
        # the reference's training loop (diffusion_latent.py:308-321, 349-350): gradients flow to the DeltaBlock only
        if eta != 0 or delta_h is not None or dt_lambda != 1 or (index or 0) != 0:
            raise NotImplementedError("the training step is the eta=0, single-DeltaBlock Asyrp step the reference trains with")
        return training.train_step(model, xt.detach(), ti, tn, hs_coeff=hs_coeff, ignore_timestep=ignore_timestep,
                                   learn_sigma=learn_sigma)
    if eta != 0 and noise is None:
        noise = torch.randn_like(xt)
    step = dict(t=ti, t_next=tn, eta=float(eta), learn_sigma=learn_sigma, index=index, apply_edit=apply_edit, hs_coeff=hs_coeff,
                ignore_timestep=ignore_timestep, dt_lambda=float(dt_lambda), dt_end=int(dt_end), use_mask=use_mask)
    if xt.shape[0] > 1 and data_parallel.wrapper_devices(models):
        # `models` is the reference's DataParallel wrapper over several GPUs (diffusion_latent.py:591): scatter the batch over its
        # device_ids, one fused step per device engine and host thread, gather on output_device (data_parallel.sharded_step)
        return data_parallel.sharded_step(models, model, xt, noise=noise if eta != 0 else None, delta_h=delta_h, **step)
    return model._ready_engine(xt).ddim_step(xt, noise=noise if eta != 0 else None, delta_h=delta_h, **step)
