"""The vendored sampler signatures: asyrp_official_amd.gaussian_diffusion.GaussianDiffusion against outputs of
the reference's own models/guided_diffusion/gaussian_diffusion.py (tests/golden/vendored_samplers_small.npz).
CPU: the host half - the float64 coefficient rows - applied with numpy to the reference's recorded model output.
GPU: the same through the library's kernel (asyrp_sampler_update), and the engine-backed UNets as the model."""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_golden
from oracle.weights import SMALL, hash_normal

BETAS = np.linspace(1e-4, 0.02, 1000, dtype=np.float64)
TIGHT = dict(rtol=1e-5, atol=1e-5)
CASES = (("ddpm", "small.x", 1, "fixed_large"), ("iddpm", "ismall.x", 2, "learned_range"))


def _apply_rows(rows, x, mo, noise=None):
    """numpy statement of what sampler_update_kernel computes from one coefficient row per image (test-side checker)."""
    x, mo = x.numpy().astype(np.float64), mo.numpy().astype(np.float64)
    C = x.shape[1]
    k = rows.astype(np.float64)[:, :, None, None, None]
    a, b, p, q, r, lo, hi, clip = (k[:, i] for i in range(8))
    x0 = a * x - b * mo[:, :C]
    x0 = np.where(clip != 0, np.clip(x0, -1, 1), x0)
    out = p * x0 + q * x
    lv = None
    if mo.shape[1] == 2 * C:
        f = (mo[:, C:] + 1) / 2
        lv = f * hi + (1 - f) * lo
        r = r * np.exp(0.5 * lv)
    if noise is not None:
        out = out + r * noise.numpy().astype(np.float64)
    return torch.from_numpy(out).float(), torch.from_numpy(x0).float(), (torch.from_numpy(lv).float() if lv is not None else None)


def test_coefficient_rows_reproduce_the_reference_on_its_recorded_model_output():
    from asyrp_official_amd.gaussian_diffusion import SamplerSchedule
    g = load_golden("vendored_samplers_small.npz")
    sch = SamplerSchedule(BETAS)
    for name, xkey, seed, vt in CASES:
        x = hash_normal(xkey, (2, 3, 32, 32), seed=seed)
        for tv in (701, 0):
            t = np.full((2,), tv)
            mo, pre = g[f"{name}.t{tv}.model_out"], f"{name}.t{tv}"
            mean, x0, lv = _apply_rows(sch.rows("posterior", t, var_type=vt, clip=True, noisy=False), x, mo)
            assert_close(mean, g[f"{pre}.pmv.mean"], what=f"{pre} mean", **TIGHT)
            assert_close(x0, g[f"{pre}.pmv.pred_xstart"], what=f"{pre} pred_xstart", **TIGHT)
            if vt == "learned_range":
                assert_close(lv, g[f"{pre}.pmv.log_variance"], what=f"{pre} log_variance", **TIGHT)
            else:
                flv = torch.from_numpy(sch.fixed_log_variance(vt, t)).float().view(-1, 1, 1, 1).expand(x.shape)
                assert_close(flv, g[f"{pre}.pmv.log_variance"], what=f"{pre} log_variance", **TIGHT)
                assert_close(torch.exp(flv), g[f"{pre}.pmv.variance"], what=f"{pre} variance", **TIGHT)
            mean, _, _ = _apply_rows(sch.rows("posterior", t, var_type=vt, clip=False, noisy=False), x, mo)
            assert_close(mean, g[f"{pre}.pmv_noclip.mean"], what=f"{pre} mean (no clip)", **TIGHT)
            s, _, _ = _apply_rows(sch.rows("posterior", t, var_type=vt, clip=True), x, mo, g[f"{pre}.p_sample.noise"])
            assert_close(s, g[f"{pre}.p_sample.sample"], what=f"{pre} p_sample", **TIGHT)
            s, x0, _ = _apply_rows(sch.rows("ddim", t, var_type=vt, clip=False, eta=0.0), x, mo)
            assert_close(s, g[f"{pre}.ddim.sample"], what=f"{pre} ddim_sample", **TIGHT)
            assert_close(x0, g[f"{pre}.ddim.pred_xstart"], what=f"{pre} ddim pred_xstart", **TIGHT)
            s, _, _ = _apply_rows(sch.rows("ddim_reverse", t, var_type=vt, clip=False), x, mo)
            assert_close(s, g[f"{pre}.ddim_reverse.sample"], what=f"{pre} ddim_reverse_sample", **TIGHT)


def test_samplers_have_no_cpu_path():
    from asyrp_official_amd.engine import AsyrpDeviceError
    from asyrp_official_amd.gaussian_diffusion import GaussianDiffusion
    diff = GaussianDiffusion(betas=BETAS)
    x = torch.zeros(1, 3, 8, 8)
    with pytest.raises(AsyrpDeviceError):
        diff.ddim_sample(lambda x_, t_: x_, x, torch.zeros(1, dtype=torch.long))


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


@pytest.mark.gpu
def test_kernel_on_recorded_model_output():
    from asyrp_official_amd.gaussian_diffusion import GaussianDiffusion
    g = load_golden("vendored_samplers_small.npz")
    for name, xkey, seed, vt in CASES:
        diff = GaussianDiffusion(betas=BETAS, model_var_type=vt)
        x = hash_normal(xkey, (2, 3, 32, 32), seed=seed).cuda()
        for tv in (701, 0):
            t = torch.full((2,), tv, dtype=torch.long, device="cuda")
            mo = g[f"{name}.t{tv}.model_out"].cuda()
            model = lambda x_, t_, **kw: (mo, None, None, None)      # the 4-tuple the Asyrp UNets return
            _check(diff, model, x, t, g, f"{name}.t{tv}", "cuda", TIGHT)
    # denoised_fn sees pred_xstart before the clamp (:299-304)
    diff = GaussianDiffusion(betas=BETAS, model_var_type="fixed_large")
    x = hash_normal("small.x", (2, 3, 32, 32), seed=1).cuda()
    t = torch.full((2,), 701, dtype=torch.long, device="cuda")
    mo = g["ddpm.t701.model_out"].cuda()
    model = lambda x_, t_, **kw: (mo, None, None, None)
    plain = diff.ddim_sample(model, x, t, clip_denoised=True)
    same = diff.ddim_sample(model, x, t, clip_denoised=True, denoised_fn=lambda v: v)
    assert_close(same["sample"], plain["sample"].cpu(), what="identity denoised_fn", **TIGHT)
    half = diff.ddim_sample(model, x, t, clip_denoised=False, denoised_fn=lambda v: 0.5 * v)
    assert_close(half["pred_xstart"], 0.5 * diff.ddim_sample(model, x, t, clip_denoised=False)["pred_xstart"].cpu(),
                 what="denoised_fn output is pred_xstart", **TIGHT)


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


@pytest.mark.gpu
def test_sampler_kernel_chunks_large_batches_and_eta():
    """asyrp_sampler_update takes its coefficient rows 32 images per launch: a batch of 40 (two chunks, mixed timesteps, eta > 0 with
    supplied noise) equals the rows applied image by image in float64."""
    from asyrp_official_amd.gaussian_diffusion import GaussianDiffusion, SamplerSchedule, sampler_update
    B = 40
    x = hash_normal("samp.x", (B, 3, 16, 16), seed=3)
    mo = hash_normal("samp.mo", (B, 6, 16, 16), seed=4)
    nz = hash_normal("samp.nz", (B, 3, 16, 16), seed=5)
    t = np.array([(37 * i) % 1000 for i in range(B)])
    t[3] = 0
    sch = SamplerSchedule(BETAS)
    for kind, kw in (("ddim", dict(eta=0.7)), ("posterior", {})):
        rows = sch.rows(kind, t, var_type="learned_range", clip=True, **kw)
        got_s, got_x0, got_lv = sampler_update(x.cuda(), mo.cuda(), rows, noise=nz.cuda(), want_log_variance=True)
        want_s, want_x0, want_lv = _apply_rows(rows, x, mo, nz)
        assert_close(got_s, want_s, what=f"{kind} sample, B=40", **TIGHT)
        assert_close(got_x0, want_x0, what=f"{kind} pred_xstart, B=40", **TIGHT)
        assert_close(got_lv, want_lv, what=f"{kind} log_variance, B=40", **TIGHT)
    # the class picks the noise path by itself when eta > 0
    diff = GaussianDiffusion(betas=BETAS, model_var_type="fixed_small")
    tt = torch.from_numpy(t).cuda()
    out = diff.ddim_sample(lambda x_, t_, **k: mo[:, :3].cuda(), x.cuda(), tt, eta=0.5, noise=nz.cuda())
    rows = sch.rows("ddim", t, var_type="fixed_small", clip=True, eta=0.5)
    assert_close(out["sample"], _apply_rows(rows, x, mo[:, :3], nz)[0], what="ddim_sample eta=0.5", **TIGHT)
