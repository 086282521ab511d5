"""Shared helpers for the parity tests (tests may use the oracle; the product never does)."""
from argparse import Namespace

import torch

from oracle.weights import ddpm_param_shapes, synthetic_state_dict


def namespace_for(cfg):
    """The YAML-derived namespace the reference hands to DDPM(config) (main.py:311-319)."""
    return Namespace(
        model=Namespace(ch=cfg.ch, out_ch=cfg.out_ch, ch_mult=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks,
                        attn_resolutions=list(cfg.attn_resolutions), dropout=0.0, in_channels=cfg.in_channels,
                        resamp_with_conv=True),
        data=Namespace(image_size=cfg.resolution))


def hip_model(cfg, sd, n_delta, device="cuda", max_batch=8, conv_math="f16x3", nominal_batch=0):
    from asyrp_official_amd import DDPM
    m = DDPM(namespace_for(cfg), max_batch=max_batch, conv_math=conv_math, nominal_batch=nominal_batch)
    m.setattr_layers(n_delta)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.to(device).eval()


def synthetic(cfg, n_delta, seed):
    return synthetic_state_dict(ddpm_param_shapes(cfg, n_delta=n_delta), seed=seed)


def err_stats(got, want, rtol=1e-3, atol=1e-4):
    if type(want).__name__ == "Sampled":      # tests/golden/compact.py: the stored positions of a reference tensor
        got, want = want.take(got.detach().float().cpu()), want.values
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    err = (got - want).abs()
    bad = err > (atol + rtol * want.abs())
    return dict(max_abs=float(err.max()), mean_abs=float(err.mean()), frac_outside=float(bad.float().mean()),
                ref_absmax=float(want.abs().max()))
