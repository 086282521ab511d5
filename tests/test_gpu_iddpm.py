"""iDDPM / ADM family on the HIP engine (models/improved_ddpm/unet.py == models/guided_diffusion/unet.py): B1 forward,
B2 learn_sigma steps and the fused loops against fixtures produced by the reference's own UNetModel and against the oracle."""
import pytest
import torch

from conftest import assert_close, load_golden
from oracle import sampler as osamp
from oracle.iddpm import AFHQ, SMALL_I, iddpm_forward, iddpm_param_shapes, make_model
from oracle.weights import hash_normal, synthetic_state_dict
from util_models import err_stats

pytestmark = pytest.mark.gpu


def hip_iddpm(cfg, sd, n_delta, conv_math="f16x3", max_batch=4):
    from asyrp_official_amd import UNetModel
    m = UNetModel(image_size=cfg.image_size, in_channels=3, model_channels=cfg.num_channels, out_channels=cfg.out_channels,
                  num_res_blocks=cfg.num_res_blocks, attention_resolutions=tuple(cfg.attention_ds), dropout=0.0,
                  channel_mult=cfg.channel_mult, num_classes=(1000 if cfg.class_cond else None), num_heads=4,
                  num_head_channels=cfg.num_head_channels, use_scale_shift_norm=True, resblock_updown=True,
                  max_batch=max_batch, conv_math=conv_math)
    m.setattr_layers(n_delta)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return m.cuda().eval()


@pytest.fixture(scope="module", params=["f16x3", "f32"])
def small(request):
    sd = synthetic_state_dict(iddpm_param_shapes(SMALL_I, n_delta=2), seed=11)
    return hip_iddpm(SMALL_I, sd, 2, conv_math=request.param), sd, hash_normal("ismall.x", (2, 3, 32, 32), seed=2)


@pytest.fixture(scope="module")
def g():
    return load_golden("iddpm_small.npz")


def test_forward_variants(small, g):
    m, _, x = small
    xc = x.cuda()
    t = torch.ones(2, device="cuda") * 701.0
    et, em, dh, mh = m(xc, t)
    assert em is None and dh is None
    assert_close(et, g["fwd_single.et"], what="et")
    assert_close(mh, g["fwd_single.middle_h"], what="middle_h")
    et, em, dh, mh = m(xc, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    for name, got in (("fwd_dual.et", et), ("fwd_dual.et_mod", em), ("fwd_dual.delta_h", dh), ("fwd_dual.middle_h", mh)):
        assert_close(got, g[name], what=name)
    _, em, dh, _ = m(xc, t, index=1, t_edit=500, hs_coeff=(0.9, 0.7, 0.5))
    assert_close(em, g["fwd_multi.et_mod"], what="multi et_mod")
    assert_close(dh, g["fwd_multi.delta_h"], what="multi delta_h")
    _, em, dh, _ = m(xc, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0), ignore_timestep=True)
    assert_close(em, g["fwd_ignoret.et_mod"], what="ignoret et_mod")
    assert_close(dh, g["fwd_ignoret.delta_h"], what="ignoret delta_h")
    et, em, dh, _ = m(xc, torch.ones(2, device="cuda") * 204.0, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    assert dh is None and torch.equal(et, em)
    assert_close(et, g["fwd_noedit.et"], what="noedit et")


def test_y_is_ignored_and_batch_invariance(small):
    m, _, x = small
    xc = x.cuda()
    t2, t1 = torch.ones(2, device="cuda") * 701.0, torch.ones(1, device="cuda") * 701.0
    et2, em2, _, _ = m(xc, t2, y=torch.zeros(2, dtype=torch.long, device="cuda"), index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    et1, em1, _, _ = m(xc[1:2].contiguous(), t1, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    assert torch.equal(et2[1:2], et1) and torch.equal(em2[1:2], em1)


def test_learn_sigma_steps(small, g):
    from asyrp_official_amd import denoising_step
    m, _, x = small
    b = osamp.beta_schedule().cuda()
    xc, one = x.cuda(), torch.ones(2, device="cuda")
    kw = dict(models=m, logvars=None, b=b, sampling_type="ddim", learn_sigma=True)
    xn, x0t, dh, _ = denoising_step(xc, t=one * 0.0, t_next=one * 25.0, eta=0, **kw)
    assert dh is None
    assert_close(xn, g["step_inv.xt_next"], what="inv xt_next")
    assert_close(x0t, g["step_inv.x0_t"], what="inv x0_t")
    ek = dict(index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    xn, x0t, dh, _ = denoising_step(xc, t=one * 701.0, t_next=one * 675.0, eta=0.0, **ek, **kw)
    assert_close(xn, g["step_gen.xt_next"], what="gen xt_next")
    assert_close(x0t, g["step_gen.x0_t"], what="gen x0_t")
    assert_close(dh, g["step_gen.delta_h"], what="gen delta_h")
    xn, x0t, _, _ = denoising_step(xc, t=one * 25.0, t_next=one * 0.0, eta=1.0, noise=g["step_eta.noise"].cuda(), **ek, **kw)
    assert_close(xn, g["step_eta.xt_next"], what="eta xt_next")
    assert_close(x0t, g["step_eta.x0_t"], what="eta x0_t")


def test_teacher_forced_edit_and_fused_loop(small, g):
    """Every step of a 6+6 edit vs the oracle on the GPU's own x_t; the fused loop equals the chain bitwise."""
    from asyrp_official_amd import run_edit
    m, sd, x = small
    b = osamp.beta_schedule()
    model = make_model(sd, SMALL_I)
    eng = m._ready_engine(x.cuda())
    m.set_schedule(b)
    ab = osamp.alpha_bar(b)
    seq, seq_next = osamp.timestep_seq(6)
    xx = x.cuda()
    one = torch.ones(2)
    for i, j in zip(seq_next[1:], seq[1:]):
        xn, x0t, _, _ = eng.ddim_step(xx, i, j, learn_sigma=True)
        w_xn, w_x0, _, _ = osamp.denoising_step(xx.cpu(), one * i, one * j, model=model, b=b, eta=0, learn_sigma=True)
        assert_close(xn, w_xn, what=f"inversion t={i}")
        assert_close(x0t, w_x0, atol=1e-4 * max(1.0, float(ab[i]) ** -0.5), what=f"inversion t={i} x0_t")
        xx = xn
    x_T = xx
    for i, j in zip(reversed(seq), reversed(seq_next)):
        xn, x0t, dh, _ = eng.ddim_step(xx, i, j, learn_sigma=True, index=0, apply_edit=i >= 500, hs_coeff=(1.0, 1.0))
        w_xn, w_x0, w_dh, _ = osamp.denoising_step(xx.cpu(), one * i, one * j, model=model, b=b, eta=0, learn_sigma=True,
                                                   index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        assert_close(xn, w_xn, what=f"generation t={i}")
        assert (dh is None) == (w_dh is None)
        xx = xn
    x_edit, x_T2 = run_edit(m, x.cuda(), b, n_inv=6, n_gen=6, t_edit=500, learn_sigma=True, want_latent=True)
    assert torch.equal(x_T2, x_T) and torch.equal(x_edit, xx)
    st = err_stats(x_T, g["edit.x_T"])
    print("free-running x_T", st, "x_edit", err_stats(x_edit, g["edit.x_edit"]))
    assert st["max_abs"] <= 1e-4 * max(1.0, st["ref_absmax"])


@pytest.mark.parametrize("conv_math", ["f16x3", "f32"])
def test_afhq_full_size_forward(conv_math):
    """i_DDPM('AFHQ') 256x256 (93.6 M params + DeltaBlock), B=1, vs the reference fixture and the oracle."""
    ga = load_golden("iddpm_afhq.npz")
    sd = synthetic_state_dict(iddpm_param_shapes(AFHQ, n_delta=1), seed=4321)
    from asyrp_official_amd import i_DDPM
    m = i_DDPM("AFHQ", max_batch=1, conv_math=conv_math)
    m.setattr_layers(1)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x = hash_normal("afhq.x", (1, 3, 256, 256), seed=4321)
    t = torch.ones(1) * 768.0
    et, em, dh, mh = m(x.cuda(), t.cuda(), index=0, t_edit=444, hs_coeff=(1.0, 1.0))
    for name, got in (("fwd_dual.et", et), ("fwd_dual.et_mod", em), ("fwd_dual.delta_h", dh)):
        print(conv_math, name, err_stats(got, ga[name]))
        assert_close(got, ga[name], what=name)
    et1, _, _, mh1 = m(x.cuda(), t.cuda())
    assert_close(et1, ga["fwd_single.et"], what="single et")
    assert_close(mh1, ga["fwd_single.middle_h"], what="single middle_h")
    from asyrp_official_amd import denoising_step
    b = osamp.beta_schedule().cuda()
    xn, x0t, _, _ = denoising_step(x.cuda(), t=t.cuda(), t_next=torch.ones(1, device="cuda") * 743.0, models=m, logvars=None,
                                   b=b, sampling_type="ddim", eta=0.0, learn_sigma=True, index=0, t_edit=444,
                                   hs_coeff=(1.0, 1.0))
    assert_close(xn, ga["step_gen.xt_next"], what="step xt_next")


def test_imagenet_style_structure_small():
    """Two ResBlocks per level, attention at two resolutions incl. the bottleneck level, class_cond label_emb key."""
    from oracle.iddpm import SMALL_I2
    g2 = load_golden("iddpm_small2.npz")
    sd = synthetic_state_dict(iddpm_param_shapes(SMALL_I2, n_delta=1), seed=13)
    m = hip_iddpm(SMALL_I2, sd, 1)
    x = hash_normal("ismall2.x", (2, 3, 32, 32), seed=3)
    et, em, dh, mh = m(x.cuda(), torch.ones(2, device="cuda") * 555.0, y=torch.tensor([3, 7]).cuda(), index=0, t_edit=500,
                       hs_coeff=(1.0, 0.8))
    for name, got in (("fwd_dual.et", et), ("fwd_dual.et_mod", em), ("fwd_dual.delta_h", dh), ("fwd_dual.middle_h", mh)):
        assert_close(got, g2[name], what=name)


def imagenet_weights():
    """The seeded i_DDPM('IMAGENET') weights + input of tests/golden/make_golden.py:imagenet_state_dict (one CPU generator
    stream, regenerated here; the fixture's probes pin it)."""
    from oracle.iddpm import IMAGENET
    gen = torch.Generator().manual_seed(77)
    sd = {}
    shapes = iddpm_param_shapes(IMAGENET, n_delta=1)
    for k, shp in shapes.items():
        wk = k[:-len(".bias")] + ".weight" if k.endswith(".bias") else k
        ws = shapes.get(wk, shp)
        if len(ws) == 1:
            sd[k] = (1.0 if k.endswith(".weight") else 0.0) + 0.1 * (2 * torch.rand(shp, generator=gen) - 1)
        else:
            fan_in = 1
            for d in ws[1:]:
                fan_in *= d
            sd[k] = (2 * torch.rand(shp, generator=gen) - 1) / fan_in ** 0.5
    x = torch.randn((1, 3, 256, 256), generator=gen)
    return sd, x


def test_imagenet_adm_full_size_forward_vs_reference():
    """BASELINE config 5's model: i_DDPM('IMAGENET') (553.8 M params, 1024-ch bottleneck, attention T=1024/256/64 with 64-ch
    heads), B=1, one dual-decoder forward against the output of the REFERENCE's own i_DDPM('IMAGENET')
    (tests/golden/imagenet_adm.npz, make_golden.py run_imagenet; the oracle is checked against the same fixture on the CPU)."""
    from asyrp_official_amd import i_DDPM
    g = load_golden("imagenet_adm.npz")
    sd, x = imagenet_weights()
    assert torch.equal(x[0, 0, 0, :8], g["probe.x"]) and torch.equal(sd["out.2.weight"].reshape(-1)[:8], g["probe.w"]), \
        "CPU generator stream differs from the fixture's"
    m = i_DDPM("IMAGENET", max_batch=1)
    m.setattr_layers(1)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    t = torch.ones(1) * 700.0
    et, em, dh, mh = m(x.cuda(), t.cuda(), index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    for name, got in (("et", et), ("et_mod", em), ("delta_h", dh), ("middle_h", mh)):
        print(name, err_stats(got, g["fwd_dual." + name]))
        assert_close(got, g["fwd_dual." + name], what=name)


def test_imagenet_adm_full_size_step_vs_reference():
    """BASELINE config 5, one teacher-forced dual-decoder STEP at full size: learn_sigma split of the 6-channel
    head (utils/diffusion_utils.py:47-51) + the DDIM update at the 1024-channel model, t = 700 -> 674, against the reference's own
    denoising_step on i_DDPM('IMAGENET') (tests/golden/imagenet_adm_step.npz, make_golden.py run_imagenet_step)."""
    import os
    from asyrp_official_amd import i_DDPM
    from conftest import GOLDEN
    from oracle import sampler as osamp
    if not os.path.exists(os.path.join(GOLDEN, "imagenet_adm_step.npz")):
        pytest.skip("imagenet_adm_step.npz not generated (tests/golden/make_golden.py --only imagenet_step)")
    g = load_golden("imagenet_adm_step.npz")
    sd, x = imagenet_weights()
    assert torch.equal(x[0, 0, 0, :8], g["probe.x"]) and torch.equal(sd["out.2.weight"].reshape(-1)[:8], g["probe.w"]), \
        "CPU generator stream differs from the fixture's"
    m = i_DDPM("IMAGENET", max_batch=1)
    m.setattr_layers(1)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    b = osamp.beta_schedule()
    m.set_schedule(b)
    eng = m._ready_engine(x.cuda())
    xn, x0t, dh, _ = eng.ddim_step(x.cuda(), 700, 674, apply_edit=True, index=0, hs_coeff=(1.0, 1.0), learn_sigma=True)
    amp = max(1.0, float(osamp.alpha_bar(b)[700]) ** -0.5)
    for name, got, atol in (("xt_next", xn, 1e-4), ("x0_t", x0t, 1e-4 * amp), ("delta_h", dh, 1e-4)):
        print(name, err_stats(got, g["step." + name]))
        assert_close(got, g["step." + name], atol=atol, what="ImageNet-ADM step " + name)
