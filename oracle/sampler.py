"""PyTorch-CPU restatement of the per-step sampler and the two hot loops (TEST ORACLE).

Follows /root/reference/utils/diffusion_utils.py:5-109 (get_beta_schedule, extract,
denoising_step) and /root/reference/diffusion_latent.py:41-46 (schedule), :955-957
(timestep sequence), :1034-1045 (inversion loop), :503-520 (Asyrp generation loop).
"""
import numpy as np
import torch


def beta_schedule(beta_start=1e-4, beta_end=0.02, num_steps=1000):
    """float64 linspace -> fp32 tensor (diffusion_utils.py:5-9, diffusion_latent.py:41-46)."""
    return torch.from_numpy(np.linspace(beta_start, beta_end, num_steps, dtype=np.float64)).float()


def alpha_bar(b):
    """fp32 cumprod of (1-beta), as recomputed on every call at diffusion_utils.py:67."""
    return (1.0 - b).cumprod(dim=0)


def timestep_seq(n_step, t_0=999):
    """seq = int(linspace(0,1,n)*t_0 + 1e-6); seq_next = [-1] + seq[:-1] (diffusion_latent.py:955-957)."""
    seq = [int(s + 1e-6) for s in list(np.linspace(0, 1, n_step) * t_0)]
    return seq, [-1] + seq[:-1]


def _gather(table, t, ndim):
    return table[t.long()].reshape((-1,) + (1,) * (ndim - 1))


def ddim_update(xt, et, et_mod, at, at_next, eta=0.0, noise=None, dt_lambda=1.0, apply_dt=False):
    """x0_t from the (edited) eps, direction from the plain eps (diffusion_utils.py:84-100)."""
    x0_t = (xt - (et if et_mod is None else et_mod) * (1 - at).sqrt()) / at.sqrt()
    if eta == 0:
        xt_next = at_next.sqrt() * x0_t + (1 - at_next).sqrt() * et
    else:
        c1 = eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
        c2 = ((1 - at_next) - c1 ** 2).sqrt()
        xt_next = at_next.sqrt() * x0_t + c2 * et + c1 * noise
    if apply_dt:
        xt_next = at_next.sqrt() * x0_t + (1 - at_next).sqrt() * et * dt_lambda
    return xt_next, x0_t


def denoising_step(xt, t, t_next, *, model, b, eta=0.0, learn_sigma=False, index=None, t_edit=0,
                   hs_coeff=(1.0,), delta_h=None, use_mask=False, dt_lambda=1, ignore_timestep=False,
                   dt_end=999, noise=None):
    """One DDIM step (diffusion_utils.py:24-104, sampling_type='ddim').

    ``model(xt, t, index=..., t_edit=..., hs_coeff=..., delta_h=..., ignore_timestep=..., use_mask=...)``
    returns the reference 4-tuple.  ``noise`` replaces ``torch.randn_like`` (:97) so the
    GPU path can consume the identical tensor.
    """
    et, et_mod, delta_h, middle_h = model(xt, t, index=index, t_edit=t_edit, hs_coeff=hs_coeff,
                                          delta_h=delta_h, ignore_timestep=ignore_timestep, use_mask=use_mask)
    if learn_sigma:
        et = et[:, : et.shape[1] // 2]
        if index is not None:
            et_mod = et_mod[:, : et_mod.shape[1] // 2]
    ab = alpha_bar(b)
    at = _gather(ab, t, xt.dim())
    if t_next.sum() == -t_next.shape[0]:
        at_next = torch.ones_like(at)
    else:
        at_next = _gather(ab, t_next, xt.dim())
    if eta != 0 and noise is None:
        noise = torch.randn_like(xt)
    xt_next, x0_t = ddim_update(xt, et, et_mod if index is not None else None, at, at_next, eta, noise,
                                dt_lambda, bool(dt_lambda != 1 and t[0] >= dt_end))
    return xt_next, x0_t, delta_h, middle_h


def invert(model, x0, b, n_inv=40, t_0=999, learn_sigma=False, record=None):
    """DDIM inversion x0 -> x_T (diffusion_latent.py:1034-1045): 39 steps for n_inv=40."""
    seq, seq_next = timestep_seq(n_inv, t_0)
    x = x0.clone()
    n = x.shape[0]
    for i, j in zip(seq_next[1:], seq[1:]):
        t = torch.ones(n) * i
        tn = torch.ones(n) * j
        x, _, _, _ = denoising_step(x, t, tn, model=model, b=b, eta=0, learn_sigma=learn_sigma)
        if record is not None:
            record.append(x.clone())
    return x


def generate(model, x_T, b, n_gen=40, t_0=999, t_edit=500, t_addnoise=0, hs_coeff=(1.0, 1.0), index=0,
             learn_sigma=False, noises=None, record=None):
    """Asyrp generation x_T -> x_edit (diffusion_latent.py:503-520).

    eta = 1 when t < t_addnoise (:513); ``noises[k]`` is consumed at the k-th such step.
    """
    seq, seq_next = timestep_seq(n_gen, t_0)
    x = x_T.clone()
    n = x.shape[0]
    k = 0
    for i, j in zip(reversed(seq), reversed(seq_next)):
        t = torch.ones(n) * i
        tn = torch.ones(n) * j
        eta = 1.0 if i < t_addnoise else 0.0
        z = None
        if eta != 0:
            z = noises[k]
            k += 1
        x, x0_t, dh, mh = denoising_step(x, t, tn, model=model, b=b, eta=eta, learn_sigma=learn_sigma,
                                         index=index, t_edit=t_edit, hs_coeff=hs_coeff, noise=z)
        if record is not None:
            record.append((x.clone(), x0_t.clone()))
    return x


def make_model(sd, cfg):
    """Bind (state_dict, config) into the ``model(x, t, **kw)`` callable the loops expect."""
    from .ddpm import ddpm_forward

    def model(x, t, **kw):
        with torch.no_grad():
            return ddpm_forward(sd, cfg, x, t, **kw)
    return model
