"""The reference's multi-GPU call shape, `model = torch.nn.DataParallel(model)` (diffusion_latent.py:179,195,591,1201), on
engine-backed UNets.

Three layers, slowest to fastest — all three give the bits of the unwrapped model (an image's result never depends on the images
that share its batch):

  * stock `torch.nn.DataParallel(model)` — works unmodified.  Its `replicate()` still broadcasts every parameter to every device on
    every forward (455 MB per device for the CelebA-HQ UNet: the reference's own waste); the mirror's replicas ignore those copies
    and drive the per-device engines of the source module (`HipUNet._replicate_for_data_parallel`, `_base.py`).
  * `asyrp_official_amd.DataParallel(model)` — the same class with `replicate()` reduced to what an engine-backed module needs: one
    light proxy per device, no parameter broadcast.  A one-word swap at the reference's four `torch.nn.DataParallel(model)` sites.
  * `asyrp_official_amd.denoising_step(..., models=<either wrapper>)` — B2 through the wrapper: the batch is scattered over the
    wrapper's `device_ids`, each chunk runs the FUSED step (`asyrp_ddim_step`) on its device's engine in its own host thread, the
    four results are gathered on `output_device` (`sharded_step` below).  No replicate at all.
  * `asyrp_official_amd.run_edit(<either wrapper>, x0, betas, ...)` — both loops through the wrapper: ONE scatter of x0 (and of the
    eta = 1 noise / per-image coefficient tuples), each chunk runs the whole inversion + generation (`asyrp_run_edit`) on its device's
    engine in its own host thread — one long GIL-free C call per device — and ONE gather of x_edit (`sharded_edit` below).

The fast multi-GPU form stays one process per GPU (`sampler.run_edit_sharded`): no per-step scatter / gather,
no GIL, one RCCL all-gather per edit.
"""
import torch
from torch.nn.parallel import gather, parallel_apply, scatter

from ._base import HipUNet


class DataParallel(torch.nn.DataParallel):
    """`torch.nn.DataParallel` whose replicas of an engine-backed UNet are light proxies (no per-forward parameter broadcast).
    Any other module is replicated by the stock code.

    It also repairs the stock scatter for a batch smaller than the device count: torch replicates non-tensor keyword arguments
    (`index=0`, `hs_coeff=(1.0, 1.0)`, `delta_h=None`, ...) to EVERY device and pads the positional chunks with empty tuples
    (torch/nn/parallel/scatter_gather.py scatter_kwargs), so with the reference's batch-of-one inversion (diffusion_latent.py:1010,
    1038) on two or more visible GPUs the second replica is called without `x` and `t` and raises TypeError — for the reference's own
    DDPM as well, which is why its scripts pin CUDA_VISIBLE_DEVICES to one GPU (script_inference.sh:4,10).  Here the padding is
    dropped: a batch of one runs one replica."""

    def scatter(self, inputs, kwargs, device_ids):
        ins, kws = super().scatter(inputs, kwargs, device_ids)
        if inputs:
            n = sum(1 for i in ins if len(i) > 0)          # the chunks the tensors really produced
            ins, kws = ins[:n], kws[:n]
        return ins, kws

    def replicate(self, module, device_ids):
        if isinstance(module, HipUNet):
            return [module._replicate_for_data_parallel() for _ in device_ids]
        return super().replicate(module, device_ids)


def wrapper_devices(models):
    """device_ids of a DataParallel wrapper that really spans several entries, else None."""
    if isinstance(models, torch.nn.DataParallel) and len(models.device_ids) > 1:
        return list(models.device_ids)
    return None


def sharded_step(models, model, xt, *, noise=None, delta_h=None, **step):
    """One fused DDIM / Asyrp step of the batch `xt`, scattered over `models.device_ids` (DataParallel's own scatter -> threads ->
    gather shape, torch/nn/parallel/data_parallel.py forward).  `step` = the keyword arguments of `Engine.ddim_step` that do not
    depend on the image (timesteps as host ints, eta, index, apply_edit, hs_coeff, ...).  Returns the 4-tuple on
    `models.output_device`; an injected `delta_h` tensor is handed back as the caller's own object, as the reference does."""
    ids = wrapper_devices(models)
    xs = scatter(xt, ids)                              # <= len(ids) chunks along dim 0 (torch.chunk sizes)
    n = len(xs)
    ids = ids[:n]
    nz = scatter(noise, ids) if noise is not None else (None,) * n
    dh = scatter(delta_h, ids) if delta_h is not None else (None,) * n
    if not (len(nz) == n and len(dh) == n):
        raise ValueError("noise / delta_h must have the batch size of xt")

    def run(x, z, d):
        return model._ready_engine(x).ddim_step(x, noise=z, delta_h=d, **step)

    outs = parallel_apply([run] * n, list(zip(xs, nz, dh)), devices=ids)
    xt_next, x0_t, dh_out, mid = gather(outs, models.output_device)
    return xt_next, x0_t, (delta_h if delta_h is not None else dh_out), mid


def unwrap(model):
    """The engine-backed UNet behind a DataParallel wrapper (or the model itself)."""
    m = model.module if isinstance(model, torch.nn.DataParallel) else model
    if not isinstance(m, HipUNet):
        raise TypeError(f"expected an asyrp_official_amd UNet (optionally wrapped in DataParallel), got {type(m).__name__}")
    return m


def sharded_edit(models, model, x0, seq_inv, seq_gen, *, noise=None, hs_coeff=(1.0, 1.0), want_latent=False, **edit):
    """Both loops (`Engine.run_edit`) of the batch `x0`, scattered over `models.device_ids`: one scatter, one whole edit per chunk,
    device and host thread, one gather.  `noise` is [n_eta_steps, B, 3, R, R] (scattered along its batch dimension); per-image
    `hs_coeff` tuples are split with the batch.  Returns x_edit (and x_T) on `models.output_device`."""
    ids = wrapper_devices(models)
    xs = scatter(x0, ids)
    n = len(xs)
    ids = ids[:n]
    nz = scatter(noise, ids, dim=1) if noise is not None else (None,) * n
    per_image = len(hs_coeff) > 0 and isinstance(hs_coeff[0], (tuple, list))
    if per_image and len(hs_coeff) != x0.shape[0]:
        raise ValueError(f"per-image hs_coeff must hold {x0.shape[0]} tuples")
    lo, coeffs = 0, []
    for x in xs:                                                   # the tuples of each chunk's images
        coeffs.append(list(hs_coeff[lo:lo + x.shape[0]]) if per_image else hs_coeff)
        lo += x.shape[0]

    def run(x, z, hc):
        z = z.contiguous() if z is not None else None
        return model._ready_engine(x).run_edit(x.contiguous(), seq_inv, seq_gen, noise=z, hs_coeff=hc, want_latent=want_latent, **edit)

    outs = parallel_apply([run] * n, list(zip(xs, nz, coeffs)), devices=ids)
    return gather(outs, models.output_device)
