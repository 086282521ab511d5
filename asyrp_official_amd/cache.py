"""On-disk formats around the hot path, so the reference's scripts can consume engine output unmodified,
and the editing-strength schedules that sit directly on the two loops (§8f-2).

  latent cache      precomputed/{category}_{mode}_t{t_0}_nim{N}_ninv{n}_pairs.pth = list of [x0, x_rec, x_lat], each [1,3,R,R]
                    (diffusion_latent.py:961-982 naming, :1072 element, :1082 torch.save)
  Δh checkpoint     checkpoint/{exp}_LC_{category}_t{t_0}_ninv{n_inv}_ngen{n_train}_{iter}.pth =
                    {"0": layer_0.state_dict(), ..., "optimizer": ..., "scheduler": ...}    (diffusion_latent.py:393-404, :674-676)
  hs_coeff          (hs_coeff_origin_h, n_train_step / n_test_step * hs_coeff_delta_h), multi-attribute 1/sqrt(k) scaling
                    (diffusion_latent.py:626, :654, :659); --delta_interpolation sweep (:726-755)
  global delta-h    --num_mean_of_delta_hs: per-timestep mean of the DeltaBlock outputs over images, entry 0 = mean over
                    timesteps, stored as checkpoint_latent/{exp}_{n_test}_{n_mean}.pth and re-injected through the
                    `delta_h=` argument of denoising_step (diffusion_latent.py:516, :528-532, :811-831)
"""
import os

import numpy as np
import torch

from .sampler import run_edit


# ---- latent cache -------------------------------------------------------------------------------------------------------
def pairs_path(category, mode, t_0, n_img, n_inv, root="precomputed", class_name=None):
    """File name the reference reads/writes (diffusion_latent.py:961-982; `class_name` only for IMAGENET with a target class)."""
    mid = f"{category}_{class_name}_{mode}" if class_name else f"{category}_{mode}"
    return os.path.join(root, f"{mid}_t{t_0}_nim{n_img}_ninv{n_inv}_pairs.pth")


@torch.no_grad()
def precompute_pairs(model, x0, betas, *, n_inv=40, t_0=999, learn_sigma=False):
    """PHASE A of the reference (`Asyrp.precompute_pairs`, diffusion_latent.py:1034-1072) for a whole batch at once:
    DDIM inversion x0 -> x_lat, then the plain DDIM reconstruction x_lat -> x_rec over the same timesteps.
    Returns the reference's list-of-triples with [1,3,R,R] CPU tensors (it trains / edits with batch entries cat'ed, :789-798)."""
    x_rec, x_lat = run_edit(model, x0, betas, n_inv=n_inv, n_gen=n_inv, t_0=t_0, index=None, learn_sigma=learn_sigma,
                            want_latent=True)
    x0c, xr, xl = x0.detach().cpu(), x_rec.cpu(), x_lat.cpu()
    return [[x0c[i:i + 1].clone(), xr[i:i + 1].clone(), xl[i:i + 1].clone()] for i in range(x0.shape[0])]


@torch.no_grad()
def inversion_trace(model, x0, betas, *, n_inv=40, t_0=999, learn_sigma=False, window=50):
    """Engine half of the reference's LPIPS(t) table builder (`compute_lpips_distance`, diffusion_latent.py:1239-1276): DDIM
    inversion of x0 over n_inv timesteps yielding (t_next, x_{t_next}, x0_t) for EVERY step, so the caller can feed both to
    LPIPS exactly as the reference does (`loss_fn_alex(x, x0)`, `loss_fn_alex(x0_t, x0)`).  The walk is cut into windows of
    `window` steps (asyrp_run_inversion taps) so n_inv = 1000 needs 2 x window x B images of device memory, not 2000 x B."""
    from .data_parallel import unwrap
    from .sampler import timestep_seq
    model = unwrap(model)
    model.set_schedule(betas)
    eng = model._ready_engine(x0)
    seq = timestep_seq(n_inv, t_0)[0]
    x, k = x0, 0
    while k < len(seq) - 1:
        n = min(window, len(seq) - 1 - k)
        sub = seq[k:k + n + 1]
        x, x_tap, x0t_tap = eng.run_inversion(x, sub, learn_sigma=learn_sigma, tap_first=0, tap_count=n)
        for i in range(n):
            yield seq[k + i + 1], x_tap[i], x0t_tap[i]
        k += n


def save_pairs(path, pairs):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save([[t.detach().cpu() for t in triple] for triple in pairs], path)


def load_pairs(path):
    """torch.load(pairs_path, map_location='cpu') as the reference does (diffusion_latent.py:977)."""
    pairs = torch.load(path, map_location=torch.device("cpu"), weights_only=True)   # lists of tensors only: no pickled code
    for triple in pairs:
        if len(triple) != 3 or any(t.dim() != 4 or t.shape[0] != 1 for t in triple):
            raise ValueError(f"{path}: not a list of [x0, x_rec, x_lat] triples of [1,C,H,W] tensors")
    return pairs


def latents_from_pairs(pairs, lo=0, hi=None, device=None):
    """cat the x_lat (and x0) entries lo..hi into batch tensors, as run_test does before save_image (:789-798)."""
    sel = pairs[lo:hi]
    x0 = torch.cat([p[0] for p in sel], dim=0)
    x_lat = torch.cat([p[2] for p in sel], dim=0)
    if device is not None:
        x0, x_lat = x0.to(device), x_lat.to(device)
    return x0, x_lat


# ---- Δh checkpoints -----------------------------------------------------------------------------------------------------
def checkpoint_name(exp, category, t_0, n_inv, n_train_step, it=0, root="checkpoint"):
    """diffusion_latent.py:230-234 / main.py:235 naming."""
    return os.path.join(root, f"{exp}_LC_{category}_t{t_0}_ninv{n_inv}_ngen{n_train_step}_{it}.pth")


def save_delta_checkpoint(model, path, get_h_num=1, optimizer=None, scheduler=None):
    """{"0": layer_0.state_dict(), ..., "optimizer", "scheduler"} (diffusion_latent.py:393-404)."""
    m = model.module if isinstance(model, torch.nn.DataParallel) else model
    dicts = {f"{i}": {k: v.detach().cpu() for k, v in getattr(m, f"layer_{i}").state_dict().items()} for i in range(get_h_num)}
    dicts["optimizer"] = optimizer.state_dict() if optimizer is not None else {}
    dicts["scheduler"] = scheduler.state_dict() if scheduler is not None else {}
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save(dicts, path)


def load_delta_checkpoints(model, paths):
    """layer_i <- torch.load(paths[i])["0"] for each attribute checkpoint (diffusion_latent.py:674-676: one file per DeltaBlock,
    always key "0").  The model must already hold len(paths) DeltaBlocks (`setattr_layers`)."""
    m = model.module if isinstance(model, torch.nn.DataParallel) else model
    for i, p in enumerate(paths):
        sd = torch.load(p, map_location="cpu", weights_only=True)["0"]     # a dict of tensors: no pickled code is executed
        res = getattr(m, f"layer_{i}").load_state_dict(sd)
        if res.missing_keys or res.unexpected_keys:
            raise KeyError(f"{p}: DeltaBlock keys do not match layer_{i}: {res}")


# ---- editing strength ---------------------------------------------------------------------------------------------------
def make_hs_coeff(n_train_step, n_test_step, hs_coeff_delta_h=1.0, hs_coeff_origin_h=1.0, multiple_hs_coeff=None, n_attr=1):
    """hs_coeff tuple the reference hands to denoising_step (diffusion_latent.py:626, :654, :659)."""
    scaling = n_train_step / n_test_step * hs_coeff_delta_h
    if n_attr <= 1 and not multiple_hs_coeff:
        return (1.0 * hs_coeff_origin_h, 1.0 * scaling)
    coeffs = list(multiple_hs_coeff or [])
    coeffs = [float(c) for c in coeffs] + [1.0] * (n_attr - len(coeffs))
    return tuple([1.0 * hs_coeff_origin_h] + [1.0 / n_attr ** 0.5 * scaling * c for c in coeffs])


def delta_interpolation_coeffs(min_delta, max_delta, num_delta, hs_coeff=(1.0, 1.0), multiple_attr=False):
    """--delta_interpolation: the list of hs_coeff tuples the reference builds from the BASE tuple `hs_coeff`
    (= make_hs_coeff(...)), diffusion_latent.py:726-755:
      single attribute (:742-751)   every element of the base tuple times val, then element 0 forced to 1.0;
      --multiple_attr (:728-740)    needs exactly two DeltaBlocks: the num_delta^2 grid (1.0, v1*c1, v2*c2)."""
    vals = np.linspace(min_delta, max_delta, num_delta).tolist()
    base = list(hs_coeff)
    if multiple_attr:
        if len(base) != 3:
            raise ValueError("delta_multiple_attr_interpolation is only supported for get_h_num == 2 (diffusion_latent.py:729)")
        return [(1.0, v1 * base[1], v2 * base[2]) for v1 in vals for v2 in vals]
    out = []
    for v in vals:
        t = [v * e for e in base]
        t[0] = 1.0
        out.append(tuple(t))
    return out


@torch.no_grad()
def edit_sweep(model, x_T, betas, hs_coeffs, batched=True, **kw):
    """Generation (loop B) for every hs_coeff tuple from the same latents, as save_image's outer loop does (:499-534).
    Returns [len(hs_coeffs)] tensors [B,3,R,R].
    batched (round 5): the tuples are independent of one another, so they run as BATCH ENTRIES -- x_T repeated per tuple, one
    coefficient tuple per image (asyrp_run_edit with a per-image table), in chunks of the model's max_batch (<= 128 images) -- instead
    of one engine pass per tuple as the reference loops; every image's bits equal those of the per-tuple pass (tested).  With B
    images and K tuples this is ceil(B K / max_batch) passes instead of K (a 9-point strength sweep of one image: 1 pass, not 9).
    Deterministic generation only: with an eta = 1 tail (t_addnoise > 0, noise supplied or drawn) or want_latent=True the sweep runs
    the reference's one pass per tuple, so seeded runs consume the generator exactly as K separate calls do."""
    hs_coeffs = [tuple(hc) for hc in hs_coeffs]
    B, K = x_T.shape[0], len(hs_coeffs)
    index = kw.get("index", 0)
    cap = min(int(getattr(getattr(model, "module", model), "max_batch", B)), 128)
    if not batched or K <= 1 or index is None or index < 0 or cap < 2 * B or len({len(hc) for hc in hs_coeffs}) != 1 \
            or len(hs_coeffs[0]) != index + 2:
        return [run_edit(model, x_T, betas, invert=False, hs_coeff=hc, **kw) for hc in hs_coeffs]
    from .sampler import count_noise_steps, timestep_seq
    stochastic = count_noise_steps(timestep_seq(kw.get("n_gen", 40), kw.get("t_0", 999))[0], kw.get("t_addnoise", 0)) > 0
    if kw.get("noise") is not None or stochastic or kw.get("want_latent"):
        # keep the reference's per-tuple passes where batching would change what a caller observes: an eta = 1 tail takes / draws
        # its noise per pass (a batched draw would consume the generator in another order), and want_latent makes run_edit
        # return an (x_edit, x_T) pair per pass
        return [run_edit(model, x_T, betas, invert=False, hs_coeff=hc, **kw) for hc in hs_coeffs]
    per_call = cap // B                        # tuples per engine call
    out = []
    for k0 in range(0, K, per_call):
        chunk = hs_coeffs[k0:k0 + per_call]
        xs = x_T.repeat(len(chunk), 1, 1, 1)                                   # [tuple][image] order
        table = [hc for hc in chunk for _ in range(B)]
        if len(chunk) == 1:
            res = run_edit(model, x_T, betas, invert=False, hs_coeff=chunk[0], **kw)
        else:
            res = run_edit(model, xs, betas, invert=False, hs_coeff=table, **kw)
        out.extend(res[i * B:(i + 1) * B] for i in range(len(chunk)))
    return out


# ---- global (mean) delta-h ----------------------------------------------------------------------------------------------
@torch.no_grad()
def generate_stepwise(model, x_T, betas, *, n_gen=40, t_0=999, t_edit=500, t_addnoise=0, index=0, hs_coeff=(1.0, 1.0),
                      learn_sigma=False, delta_h_dict=None, ignore_timesteps=False, collect=None, use_mask=False,
                      dt_lambda=1, noise=None):
    """save_image's generation loop one denoising_step at a time (diffusion_latent.py:503-534), for the two cases the
    fused asyrp_run_edit does not cover:
      collect=<dict>        get_delta_hs: the DeltaBlock output of every step with t >= t_edit is summed into
                            collect[t] (:528-532); the edit itself still uses the DeltaBlocks;
      delta_h_dict=<dict>   the stored delta-h tensors are injected instead of evaluating the DeltaBlocks (:516):
                            delta_h_dict[t] for t >= t_edit, or delta_h_dict[0] at every step with ignore_timesteps.
    `noise` = [n_eta_steps,B,3,R,R] for the eta=1 tail (else drawn on the device).  Returns x_edit."""
    from .diffusion_utils import denoising_step
    from .sampler import timestep_seq
    seq, seq_next = timestep_seq(n_gen, t_0)
    x, B, k = x_T, x_T.shape[0], 0
    for i, j in zip(reversed(seq), reversed(seq_next)):
        t = torch.full((B,), float(i), device=x.device)
        t_next = torch.full((B,), float(j), device=x.device)
        inject = None
        if delta_h_dict is not None and collect is None:
            inject = delta_h_dict[0] if ignore_timesteps else (delta_h_dict[int(i)] if i >= t_edit else None)
            if inject is not None:
                inject = inject.to(x.device).float().expand(B, -1, -1, -1).contiguous()
        eta = 1.0 if i < t_addnoise else 0.0
        nz = None
        if eta and noise is not None:
            nz, k = noise[k], k + 1
        x, _, dh, _ = denoising_step(x, t, t_next, models=model, b=betas, eta=eta, learn_sigma=learn_sigma, index=index,
                                     t_edit=t_edit, hs_coeff=hs_coeff, delta_h=inject, ignore_timestep=ignore_timesteps,
                                     use_mask=use_mask, dt_lambda=dt_lambda, noise=nz)
        if collect is not None and i >= t_edit:
            collect[int(i)] = dh if collect.get(int(i)) is None else collect[int(i)] + dh
    return x


def finish_mean_delta_hs(collect, n_batches, group=None):
    """Turn the per-timestep sums of `n_batches` generate_stepwise(collect=...) calls into the reference's global
    delta-h dictionary (diffusion_latent.py:811-831): every entry divided by the number of batches, and entry 0 = the
    mean over the timesteps that have one (used with --ignore_timesteps).  With a process group the sums and the batch
    count are all-reduced first (one [B,C,8,8] all-reduce per edited timestep), so every rank holds the mean over the
    images of ALL ranks — the only data-path reduction of the whole method."""
    import torch.distributed as dist
    out = dict(collect)
    n = float(n_batches)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        cnt = torch.tensor([n], dtype=torch.float64, device=next(v for v in out.values() if v is not None).device)
        dist.all_reduce(cnt, group=group)
        n = float(cnt.item())
        for k in sorted(k for k, v in out.items() if v is not None):
            out[k] = out[k].clone()
            dist.all_reduce(out[k], group=group)
    for k, v in out.items():
        if v is not None:
            out[k] = v / n
    tot, cnt = None, 0
    for k in out.keys():          # the reference's own accumulation order (dict order = ascending timestep)
        if out[k] is None:
            continue
        tot = out[k].clone() if tot is None else tot + out[k]
        cnt += 1
    if tot is not None:
        out[0] = tot / cnt
    return out


def mean_delta_path(exp_id, n_test_step, num_mean, root="checkpoint_latent"):
    """`checkpoint_latent/{exp_id}_{n_test_step}_{num_mean_of_delta_hs}.pth` (diffusion_latent.py:613,831)."""
    return os.path.join(root, f"{exp_id}_{n_test_step}_{num_mean}.pth")
