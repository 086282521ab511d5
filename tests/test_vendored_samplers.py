"""The vendored sampler signatures: asyrp_official_amd.gaussian_diffusion.GaussianDiffusion against outputs of
the reference's own models/guided_diffusion/gaussian_diffusion.py (tests/golden/vendored_samplers_small.npz).
CPU: the wrapper arithmetic with the reference's recorded model output standing in for the UNet.  GPU: the engine-backed UNets."""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_golden
from oracle.weights import SMALL, hash_normal

BETAS = np.linspace(1e-4, 0.02, 1000, dtype=np.float64)
TIGHT = dict(rtol=1e-5, atol=1e-5)
CASES = (("ddpm", "small.x", 1, "fixed_large"), ("iddpm", "ismall.x", 2, "learned_range"))


def _check(diff, model, x, t, g, pre, dev, tol):
    out = diff.p_mean_variance(model, x, t, clip_denoised=True)
    for k in ("mean", "variance", "log_variance", "pred_xstart"):
        assert_close(out[k], g[f"{pre}.pmv.{k}"], what=f"{pre} p_mean_variance {k}", **tol)
    assert_close(diff.p_mean_variance(model, x, t, clip_denoised=False)["mean"], g[f"{pre}.pmv_noclip.mean"],
                 what=f"{pre} mean (no clip)", **tol)
    ps = diff.p_sample(model, x, t, noise=g[f"{pre}.p_sample.noise"].to(dev))
    assert_close(ps["sample"], g[f"{pre}.p_sample.sample"], what=f"{pre} p_sample", **tol)
    ds = diff.ddim_sample(model, x, t, clip_denoised=False, eta=0.0)
    assert_close(ds["sample"], g[f"{pre}.ddim.sample"], what=f"{pre} ddim_sample", **tol)
    assert_close(ds["pred_xstart"], g[f"{pre}.ddim.pred_xstart"], what=f"{pre} ddim pred_xstart", **tol)
    rs = diff.ddim_reverse_sample(model, x, t, clip_denoised=False, eta=0.0)
    assert_close(rs["sample"], g[f"{pre}.ddim_reverse.sample"], what=f"{pre} ddim_reverse_sample", **tol)


def test_wrapper_arithmetic_on_recorded_model_output():
    from asyrp_official_amd.gaussian_diffusion import GaussianDiffusion
    g = load_golden("vendored_samplers_small.npz")
    for name, xkey, seed, vt in CASES:
        diff = GaussianDiffusion(betas=BETAS, model_var_type=vt)
        x = hash_normal(xkey, (2, 3, 32, 32), seed=seed)
        for tv in (701, 0):
            t = torch.full((2,), tv, dtype=torch.long)
            mo = g[f"{name}.t{tv}.model_out"]
            model = lambda x_, t_, **kw: (mo, None, None, None)      # the 4-tuple the Asyrp UNets return
            _check(diff, model, x, t, g, f"{name}.t{tv}", "cpu", TIGHT)


@pytest.mark.gpu
def test_samplers_drive_the_engine_unets():
    from asyrp_official_amd import UNetModel
    from asyrp_official_amd.gaussian_diffusion import GaussianDiffusion
    from oracle.iddpm import SMALL_I, iddpm_param_shapes
    from oracle.weights import synthetic_state_dict
    from util_models import hip_model, synthetic
    g = load_golden("vendored_samplers_small.npz")
    m_d = hip_model(SMALL, synthetic(SMALL, 2, seed=7), 2)
    cfg = SMALL_I
    m_i = UNetModel(image_size=cfg.image_size, in_channels=3, model_channels=cfg.num_channels, out_channels=cfg.out_channels,
                    num_res_blocks=cfg.num_res_blocks, attention_resolutions=tuple(cfg.attention_ds), dropout=0.0,
                    channel_mult=cfg.channel_mult, num_classes=None, num_heads=4, num_head_channels=cfg.num_head_channels,
                    use_scale_shift_norm=True, resblock_updown=True, max_batch=2)
    m_i.setattr_layers(2)
    m_i.load_state_dict(synthetic_state_dict(iddpm_param_shapes(cfg, n_delta=2), seed=11), strict=True)
    m_i = m_i.cuda().eval()
    for (name, xkey, seed, vt), m in zip(CASES, (m_d, m_i)):
        diff = GaussianDiffusion(betas=BETAS, model_var_type=vt)
        x = hash_normal(xkey, (2, 3, 32, 32), seed=seed).cuda()
        for tv in (701, 0):
            t = torch.full((2,), tv, dtype=torch.long, device="cuda")
            model = lambda x_, t_, **kw: m(x_, t_.float())
            # pred_xstart multiplies an eps difference by sqrt(1/alpha_bar - 1) (up to 13 at t=701): scale the tolerance like the
            # x0_t checks of test_gpu_edit.py
            _check(diff, model, x, t, g, f"{name}.t{tv}", "cuda", dict(rtol=1e-3, atol=2e-3 if tv else 1e-4))
