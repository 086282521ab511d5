"""On-disk formats around the hot path, so the reference's scripts can consume engine output unmodified,
and the editing-strength schedules that sit directly on the two loops (§8f-2).

  latent cache      precomputed/{category}_{mode}_t{t_0}_nim{N}_ninv{n}_pairs.pth = list of [x0, x_rec, x_lat], each [1,3,R,R]
                    (diffusion_latent.py:961-982 naming, :1072 element, :1082 torch.save)
  Δh checkpoint     checkpoint/{exp}_LC_{category}_t{t_0}_ninv{n_inv}_ngen{n_train}_{iter}.pth =
                    {"0": layer_0.state_dict(), ..., "optimizer": ..., "scheduler": ...}    (diffusion_latent.py:393-404, :674-676)
  hs_coeff          (hs_coeff_origin_h, n_train_step / n_test_step * hs_coeff_delta_h), multi-attribute 1/sqrt(k) scaling
                    (diffusion_latent.py:626, :654, :659); --delta_interpolation sweep (:726-755)
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


def save_pairs(path, pairs):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save([[t.detach().cpu() for t in triple] for triple in pairs], path)


def load_pairs(path):
    """torch.load(pairs_path, map_location='cpu') as the reference does (diffusion_latent.py:977)."""
    pairs = torch.load(path, map_location=torch.device("cpu"), weights_only=False)
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
        sd = torch.load(p, map_location="cpu", weights_only=False)["0"]
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


def delta_interpolation_coeffs(min_delta, max_delta, num_delta, hs_coeff_origin_h=1.0, scaling_factor=1.0):
    """--delta_interpolation: one hs_coeff tuple per strength, linspace(min, max, num) (diffusion_latent.py:726-755)."""
    return [(1.0 * hs_coeff_origin_h, float(d) * scaling_factor) for d in np.linspace(min_delta, max_delta, num_delta)]


@torch.no_grad()
def edit_sweep(model, x_T, betas, hs_coeffs, **kw):
    """Generation (loop B) once per hs_coeff tuple from the same latents, as save_image's outer loop does (:499-534).
    Returns [len(hs_coeffs)] tensors [B,3,R,R].  (The strengths share x_T but not the trajectory: they are independent
    batch entries, so this is `len(hs_coeffs)` engine calls on the already-resident weights.)"""
    return [run_edit(model, x_T, betas, invert=False, hs_coeff=tuple(hc), **kw) for hc in hs_coeffs]
