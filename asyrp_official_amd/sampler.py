"""The two hot loops of the reference's Asyrp class as one engine call, plus batch sharding.

  invert + generate == diffusion_latent.py:1034-1045 (DDIM inversion) followed by :503-520 (Asyrp
  generation).  Images are independent, so under torch.distributed each rank edits its slice of the
  batch with no communication and the final images are all-gathered once (RCCL over xGMI).
"""
import numpy as np
import torch
import torch.distributed as dist


def timestep_seq(n_step, t_0=999):
    """seq = int(linspace(0,1,n)*t_0 + 1e-6); seq_next = [-1] + seq[:-1]  (diffusion_latent.py:955-957)."""
    seq = [int(s + 1e-6) for s in list(np.linspace(0, 1, n_step) * t_0)]
    return seq, [-1] + seq[:-1]


def shard_bounds(n_items, world_size, rank):
    """Contiguous, balanced slice [lo, hi) of `n_items` for `rank` (first ranks take the remainder)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_shards(local, n_total, group=None):
    """All-gather variable-length batch shards back into the full batch (dim 0), every rank gets it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    ws = dist.get_world_size(group)
    sizes = [shard_bounds(n_total, ws, r) for r in range(ws)]
    cap = max(hi - lo for lo, hi in sizes)
    padded = local.new_zeros((cap,) + tuple(local.shape[1:]))
    padded[: local.shape[0]] = local
    out = [torch.empty_like(padded) for _ in range(ws)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


def count_noise_steps(seq_gen, t_addnoise):
    return sum(1 for t in seq_gen if t < t_addnoise)


@torch.no_grad()
def run_edit(model, x0, betas, *, n_inv=40, n_gen=40, t_0=999, t_edit=500, t_addnoise=0, index=0,
             hs_coeff=(1.0, 1.0), learn_sigma=False, noise=None, want_latent=False, invert=True):
    """x0 [B,3,R,R] (GPU) -> x_edit (and x_T).  `noise` = [n_eta_steps,B,3,R,R] for the eta=1 tail."""
    from . import data_parallel
    wrapper, model = model, data_parallel.unwrap(model)       # `model` may be the reference's DataParallel wrapper (diffusion_latent.py:591)
    model.set_schedule(betas)
    seq_inv = timestep_seq(n_inv, t_0)[0] if invert else []
    seq_gen = timestep_seq(n_gen, t_0)[0]
    need = count_noise_steps(seq_gen, t_addnoise)
    if need and noise is None:
        noise = torch.randn((need,) + tuple(x0.shape), device=x0.device, dtype=torch.float32)
    kw = dict(t_edit=t_edit, t_addnoise=t_addnoise, index=index, hs_coeff=hs_coeff, learn_sigma=learn_sigma, noise=noise,
              want_latent=want_latent)
    if x0.shape[0] > 1 and data_parallel.wrapper_devices(wrapper):
        # several devices behind the wrapper: one scatter, the whole edit per device in its own host thread, one gather
        return data_parallel.sharded_edit(wrapper, model, x0, seq_inv, seq_gen, **kw)
    return model._ready_engine(x0).run_edit(x0, seq_inv, seq_gen, **kw)


@torch.no_grad()
def run_edit_sharded(model, x0_full, betas, *, group=None, noise=None, **kw):
    """Data-parallel edit: this rank processes its slice of x0_full (and of `noise`), then one all-gather."""
    if dist.is_available() and dist.is_initialized():
        ws, rk = dist.get_world_size(group), dist.get_rank(group)
    else:
        ws, rk = 1, 0
    lo, hi = shard_bounds(x0_full.shape[0], ws, rk)
    local_noise = noise[:, lo:hi].contiguous() if noise is not None else None
    local = run_edit(model, x0_full[lo:hi].contiguous(), betas, noise=local_noise, **kw)
    return gather_shards(local, x0_full.shape[0], group)
