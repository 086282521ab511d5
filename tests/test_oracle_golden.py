"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz,
produced by tests/golden/make_golden.py from /root/reference)."""
import numpy as np
import torch

from conftest import assert_close
from oracle import sampler
from oracle.ddpm import ddpm_forward
from oracle.weights import CELEBA, SMALL, ddpm_param_shapes, hash_normal, synthetic_state_dict

TIGHT = dict(rtol=1e-5, atol=2e-6)   # same math, same library: only summation-order noise


def _small():
    torch.set_num_threads(1)
    sd = synthetic_state_dict(ddpm_param_shapes(SMALL, n_delta=2), seed=7)
    x = hash_normal("small.x", (2, 3, 32, 32), seed=1)
    return sd, x


def test_hash_inputs_are_stable(golden_small):
    _, x = _small()
    assert torch.equal(x, golden_small["input.x"])


def test_forward_single_and_dual(golden_small):
    sd, x = _small()
    g = golden_small
    t = torch.ones(2) * 701.0
    with torch.no_grad():
        et, em, dh, mh = ddpm_forward(sd, SMALL, x, t)
        assert em is None and dh is None
        assert_close(et, g["fwd_single.et"], what="et", **TIGHT)
        assert_close(mh, g["fwd_single.middle_h"], what="middle_h", **TIGHT)
        et, em, dh, mh = ddpm_forward(sd, SMALL, x, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        assert_close(et, g["fwd_dual.et"], what="et", **TIGHT)
        assert_close(em, g["fwd_dual.et_mod"], what="et_mod", **TIGHT)
        assert_close(dh, g["fwd_dual.delta_h"], what="delta_h", **TIGHT)
        assert_close(mh, g["fwd_dual.middle_h"], what="middle_h", **TIGHT)


def test_forward_variants(golden_small):
    sd, x = _small()
    g = golden_small
    t = torch.ones(2) * 701.0
    with torch.no_grad():
        et, em, dh, _ = ddpm_forward(sd, SMALL, x, t, index=1, t_edit=500, hs_coeff=(0.9, 0.7, 0.5))
        assert_close(em, g["fwd_multi.et_mod"], what="multi et_mod", **TIGHT)
        assert_close(dh, g["fwd_multi.delta_h"], what="multi delta_h", **TIGHT)
        _, em, dh, _ = ddpm_forward(sd, SMALL, x, t, index=0, t_edit=500, ignore_timestep=True)
        assert_close(em, g["fwd_ignoret.et_mod"], what="ignore_timestep et_mod", **TIGHT)
        assert_close(dh, g["fwd_ignoret.delta_h"], what="ignore_timestep delta_h", **TIGHT)
        et, em, dh, _ = ddpm_forward(sd, SMALL, x, torch.ones(2) * 204.0, index=0, t_edit=500)
        assert dh is None and torch.equal(et, em)
        assert_close(et, g["fwd_noedit.et"], what="noedit et", **TIGHT)


def test_steps(golden_small):
    sd, x = _small()
    g = golden_small
    model = sampler.make_model(sd, SMALL)
    b = sampler.beta_schedule()
    one = torch.ones(2)
    xn, x0t, _, _ = sampler.denoising_step(x, one * 0.0, one * 25.0, model=model, b=b, eta=0)
    assert_close(xn, g["step_inv.xt_next"], what="inv xt_next", **TIGHT)
    assert_close(x0t, g["step_inv.x0_t"], what="inv x0_t", **TIGHT)
    kw = dict(model=model, b=b, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    xn, x0t, dh, _ = sampler.denoising_step(x, one * 701.0, one * 675.0, eta=0.0, **kw)
    assert_close(xn, g["step_gen.xt_next"], what="gen xt_next", **TIGHT)
    assert_close(x0t, g["step_gen.x0_t"], what="gen x0_t", **TIGHT)
    assert_close(dh, g["step_gen.delta_h"], what="gen delta_h", **TIGHT)
    xn, x0t, _, _ = sampler.denoising_step(x, one * 25.0, one * 0.0, eta=1.0, noise=g["step_eta.noise"], **kw)
    assert_close(xn, g["step_eta.xt_next"], what="eta xt_next", **TIGHT)
    xn, x0t, _, _ = sampler.denoising_step(x, one * 0.0, one * -1.0, eta=0.0, **kw)
    assert_close(xn, g["step_last.xt_next"], what="last xt_next", **TIGHT)
    assert_close(x0t, g["step_last.x0_t"], what="last x0_t", **TIGHT)
    xn, _, _, _ = sampler.denoising_step(x, one * 701.0, one * 675.0, eta=0.0, dt_lambda=1.05, dt_end=600, **kw)
    assert_close(xn, g["step_dt.xt_next"], what="dt_lambda xt_next", **TIGHT)


def test_whole_edit_loop(golden_small):
    sd, x = _small()
    g = golden_small
    model = sampler.make_model(sd, SMALL)
    b = sampler.beta_schedule()
    x_T = sampler.invert(model, x, b, n_inv=6)
    assert_close(x_T, g["edit.x_T"], what="x_T", rtol=1e-4, atol=1e-5)
    x_e = sampler.generate(model, x_T, b, n_gen=6, t_edit=500, t_addnoise=0)
    assert_close(x_e, g["edit.x_edit"], what="x_edit", rtol=1e-4, atol=1e-5)


def test_timestep_sequence_matches_reference_values():
    seq, nxt = sampler.timestep_seq(40, 999)
    assert seq[:4] == [0, 25, 51, 76] and seq[-2:] == [973, 999] and nxt[0] == -1 and len(seq) == 40
    ab = sampler.alpha_bar(sampler.beta_schedule())
    assert ab.dtype == torch.float32 and abs(float(ab[999]) - 4.0358e-05) < 1e-7


def test_full_size_forward_against_reference(golden_celeba):
    """One 256x256 dual forward of the CelebA-HQ DDPM (114 M params) — ~3 s on 8 cores."""
    torch.set_num_threads(max(1, torch.get_num_threads()))
    sd = synthetic_state_dict(ddpm_param_shapes(CELEBA, n_delta=1), seed=1234)
    x = hash_normal("celeba.x", (1, 3, 256, 256), seed=1234)
    with torch.no_grad():
        et, em, dh, mh = ddpm_forward(sd, CELEBA, x, torch.ones(1) * 768.0, index=0, t_edit=500)
    g = golden_celeba
    loose = dict(rtol=1e-4, atol=1e-5)   # oneDNN blocking differs with thread count
    assert_close(et, g["fwd_dual.et"], what="et", **loose)
    assert_close(em, g["fwd_dual.et_mod"], what="et_mod", **loose)
    assert_close(dh, g["fwd_dual.delta_h"], what="delta_h", **loose)


# ---- iDDPM / ADM family (models/improved_ddpm/unet.py == models/guided_diffusion/unet.py) -------------------------------
ITIGHT = dict(rtol=1e-5, atol=6e-6)   # functional vs module evaluation order (einsum / pooling) noise


def _ismall():
    from oracle.iddpm import SMALL_I, iddpm_param_shapes
    torch.set_num_threads(1)
    sd = synthetic_state_dict(iddpm_param_shapes(SMALL_I, n_delta=2), seed=11)
    return sd, hash_normal("ismall.x", (2, 3, 32, 32), seed=2), SMALL_I


def test_iddpm_forward_variants():
    from conftest import load_golden
    from oracle.iddpm import iddpm_forward
    g = load_golden("iddpm_small.npz")
    sd, x, cfg = _ismall()
    assert torch.equal(x, g["input.x"])
    t = torch.ones(2) * 701.0
    with torch.no_grad():
        et, em, dh, mh = iddpm_forward(sd, cfg, x, t)
        assert em is None and dh is None
        assert_close(et, g["fwd_single.et"], what="et", **ITIGHT)
        assert_close(mh, g["fwd_single.middle_h"], what="middle_h", **ITIGHT)
        et, em, dh, mh = iddpm_forward(sd, cfg, x, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        for name, got in (("fwd_dual.et", et), ("fwd_dual.et_mod", em), ("fwd_dual.delta_h", dh), ("fwd_dual.middle_h", mh)):
            assert_close(got, g[name], what=name, **ITIGHT)
        _, em, dh, _ = iddpm_forward(sd, cfg, x, t, index=1, t_edit=500, hs_coeff=(0.9, 0.7, 0.5))
        assert_close(em, g["fwd_multi.et_mod"], what="multi et_mod", **ITIGHT)
        assert_close(dh, g["fwd_multi.delta_h"], what="multi delta_h", **ITIGHT)
        _, em, dh, _ = iddpm_forward(sd, cfg, x, t, index=0, t_edit=500, ignore_timestep=True)
        assert_close(em, g["fwd_ignoret.et_mod"], what="ignoret et_mod", **ITIGHT)
        assert_close(dh, g["fwd_ignoret.delta_h"], what="ignoret delta_h", **ITIGHT)
        et, em, dh, _ = iddpm_forward(sd, cfg, x, torch.ones(2) * 204.0, index=0, t_edit=500)
        assert dh is None and torch.equal(et, em)
        assert_close(et, g["fwd_noedit.et"], what="noedit et", **ITIGHT)


def test_iddpm_learn_sigma_steps_and_edit_loop():
    from conftest import load_golden
    from oracle.iddpm import make_model
    g = load_golden("iddpm_small.npz")
    sd, x, cfg = _ismall()
    model = make_model(sd, cfg)
    b = sampler.beta_schedule()
    one = torch.ones(2)
    kw = dict(model=model, b=b, learn_sigma=True)
    xn, x0t, _, _ = sampler.denoising_step(x, one * 0.0, one * 25.0, eta=0, **kw)
    assert_close(xn, g["step_inv.xt_next"], what="inv xt_next", **ITIGHT)
    assert_close(x0t, g["step_inv.x0_t"], what="inv x0_t", **ITIGHT)
    ek = dict(index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    xn, x0t, dh, _ = sampler.denoising_step(x, one * 701.0, one * 675.0, eta=0.0, **ek, **kw)
    assert_close(xn, g["step_gen.xt_next"], what="gen xt_next", **ITIGHT)
    assert_close(dh, g["step_gen.delta_h"], what="gen delta_h", **ITIGHT)
    xn, x0t, _, _ = sampler.denoising_step(x, one * 25.0, one * 0.0, eta=1.0, noise=g["step_eta.noise"], **ek, **kw)
    assert_close(xn, g["step_eta.xt_next"], what="eta xt_next", **ITIGHT)
    x_T = sampler.invert(model, x, b, n_inv=6, learn_sigma=True)
    assert_close(x_T, g["edit.x_T"], what="x_T", rtol=1e-4, atol=5e-5)
    x_edit = sampler.generate(model, x_T, b, n_gen=6, t_edit=500, learn_sigma=True)
    st = (x_edit - g["edit.x_edit"]).abs().max() / g["edit.x_edit"].abs().max()
    assert float(st) < 1e-4


def test_iddpm_full_size_afhq_against_reference():
    from conftest import load_golden
    from oracle.iddpm import AFHQ, iddpm_forward, iddpm_param_shapes
    g = load_golden("iddpm_afhq.npz")
    sd = synthetic_state_dict(iddpm_param_shapes(AFHQ, n_delta=1), seed=4321)
    x = hash_normal("afhq.x", (1, 3, 256, 256), seed=4321)
    with torch.no_grad():
        et, em, dh, mh = iddpm_forward(sd, AFHQ, x, torch.ones(1) * 768.0, index=0, t_edit=444, hs_coeff=(1.0, 1.0))
    assert_close(et, g["fwd_dual.et"], what="et", rtol=1e-4, atol=1e-5)
    assert_close(em, g["fwd_dual.et_mod"], what="et_mod", rtol=1e-4, atol=1e-5)
    assert_close(dh, g["fwd_dual.delta_h"], what="delta_h", rtol=1e-4, atol=1e-5)


def test_iddpm_imagenet_style_structure():
    from conftest import load_golden
    from oracle.iddpm import SMALL_I2, iddpm_forward, iddpm_param_shapes
    g = load_golden("iddpm_small2.npz")
    torch.set_num_threads(1)
    sd = synthetic_state_dict(iddpm_param_shapes(SMALL_I2, n_delta=1), seed=13)
    x = hash_normal("ismall2.x", (2, 3, 32, 32), seed=3)
    with torch.no_grad():
        et, em, dh, mh = iddpm_forward(sd, SMALL_I2, x, torch.ones(2) * 555.0, index=0, t_edit=500, hs_coeff=(1.0, 0.8))
    for name, got in (("fwd_dual.et", et), ("fwd_dual.et_mod", em), ("fwd_dual.delta_h", dh), ("fwd_dual.middle_h", mh)):
        assert_close(got, g[name], what=name, **ITIGHT)


def test_injected_delta_h_slerp_branch_both_families():
    """forward / denoising_step with a delta_h TENSOR (diffusion.py:518-539, unet.py:708-731), +/- use_mask."""
    from conftest import load_golden
    from oracle import iddpm as oi
    g = load_golden("slerp_small.npz")
    sd, x = _small()
    isd, ix, icfg = _ismall()
    dh_in = g["input.delta_h"]
    t = torch.ones(2) * 701.0
    b = sampler.beta_schedule()
    cases = (("ddpm", lambda *a, **k: ddpm_forward(sd, SMALL, *a, **k), sampler.make_model(sd, SMALL), x, False, TIGHT),
             ("iddpm", lambda *a, **k: oi.iddpm_forward(isd, icfg, *a, **k), oi.make_model(isd, icfg), ix, True, ITIGHT))
    with torch.no_grad():
        for name, fwd, model, xx, ls, tol in cases:
            for tag, c0, um in (("nomask", 0.7, False), ("mask", 0.7, True), ("nomask_c0", 0.25, False)):
                et, em, dh, mh = fwd(xx, t, index=0, t_edit=500, hs_coeff=(c0, 1.0), delta_h=dh_in, use_mask=um)
                assert dh is dh_in
                assert_close(et, g[f"{name}.{tag}.et"], what=f"{name} {tag} et", **tol)
                assert_close(em, g[f"{name}.{tag}.et_mod"], what=f"{name} {tag} et_mod", **tol)
                assert_close(mh, g[f"{name}.{tag}.middle_h"], what=f"{name} {tag} middle_h", **tol)
            et, em, _, _ = fwd(xx, torch.ones(2) * 204.0, index=0, t_edit=500, hs_coeff=(0.7, 1.0), delta_h=dh_in)
            assert torch.equal(et, em)
            xn, x0t, _, _ = sampler.denoising_step(xx, t, torch.ones(2) * 675.0, model=model, b=b, eta=0.0, index=0, t_edit=500,
                                                   hs_coeff=(0.7, 1.0), delta_h=dh_in, learn_sigma=ls)
            assert_close(xn, g[f"{name}.step.xt_next"], what=f"{name} step xt_next", **tol)
            # x0_t = (xt - et*sqrt(1-at))/sqrt(at) amplifies the forward's summation-order noise by 1/sqrt(at) ~ 14 at t=701
            assert_close(x0t, g[f"{name}.step.x0_t"], what=f"{name} step x0_t", rtol=1e-5, atol=5e-5)


def test_oracle_on_config1_steps_with_shipped_delta_block():
    """The oracle against the reference's own full-size config-1 trajectory (CelebA-HQ 256x256 + the shipped `smiling`
    DeltaBlock): one edited step (t=512, dual decoder) and one eta=1 step, from the reference's x_t."""
    import os
    import pytest
    from conftest import GOLDEN, load_golden
    from oracle.weights import CELEBA
    if not os.path.exists(os.path.join(GOLDEN, "config1_celeba_smiling.npz")):
        pytest.skip("config1 fixture not generated")
    g = load_golden("config1_celeba_smiling.npz")
    sd = synthetic_state_dict(ddpm_param_shapes(CELEBA, n_delta=1), seed=1234)
    for k in list(g):
        if k.startswith("param."):
            sd[k[len("param."):]] = g[k]
    model = sampler.make_model(sd, CELEBA)
    b = sampler.beta_schedule()
    one = torch.ones(1)
    loose = dict(rtol=1e-4, atol=2e-5)      # thread-count dependent summation order at 256x256 (1e-6 per forward)
    xn, x0t, dh, _ = sampler.denoising_step(g["gen512.x_t"], one * 512, one * 486, model=model, b=b, eta=0.0, index=0,
                                            t_edit=500, hs_coeff=(1.0, 1.0))
    assert_close(dh, g["gen512.delta_h"], what="delta_h", **loose)
    assert_close(xn, g["gen512.xt_next"], what="xt_next", **loose)
    assert_close(x0t, g["gen512.x0_t"], what="x0_t", rtol=1e-4, atol=2e-4)
    torch.manual_seed(4321)
    noise = torch.randn(7, 1, 3, 256, 256)
    assert torch.equal(noise[:, 0, 0, 0, :8], g["noise_probe"])
    xn, x0t, _, _ = sampler.denoising_step(g["eta153.x_t"], one * 153, one * 128, model=model, b=b, eta=1.0, index=0,
                                           t_edit=500, hs_coeff=(1.0, 1.0), noise=noise[0])
    assert_close(xn, g["eta153.xt_next"], what="eta xt_next", **loose)


def test_oracle_autograd_matches_reference_autograd():
    """The oracle is differentiable like the reference: DeltaBlock gradients of a fixed functional of (x0_t, xt_next) through
    the oracle's forward equal the reference's own autograd (tests/golden/train_small.npz)."""
    from conftest import load_golden
    g = load_golden("train_small.npz")
    torch.set_num_threads(1)
    sd = synthetic_state_dict(ddpm_param_shapes(SMALL, n_delta=2), seed=7)
    x = hash_normal("small.x", (2, 3, 32, 32), seed=1)
    g1 = hash_normal("train.g_x0t", (2, 3, 32, 32), seed=3)
    g2 = hash_normal("train.g_xtn", (2, 3, 32, 32), seed=4)
    b = sampler.beta_schedule()
    ab = sampler.alpha_bar(b)
    for tag, ign in (("step", False), ("ignoret", True)):
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("layer_0.")}
        et, em, _, _ = ddpm_forward({**sd, **leaves}, SMALL, x, torch.ones(2) * 701.0, index=0, t_edit=500, hs_coeff=(1.0, 0.8),
                                    ignore_timestep=ign)
        xn, x0t = sampler.ddim_update(x, et, em, ab[701], ab[675])
        ((x0t * g1).sum() + (xn * g2).sum()).backward()
        assert_close(x0t, g[f"{tag}.x0_t"], what="x0_t", **TIGHT)
        for k, v in leaves.items():
            want = g[f"{tag}.grad.{k}"]
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            scale = float(want.abs().max())
            assert_close(got, want, rtol=1e-4, atol=1e-5 * max(scale, 1e-30), what=f"{tag} grad {k}")


def test_iddpm_oracle_autograd_matches_reference_autograd():
    from conftest import load_golden
    from oracle.iddpm import SMALL_I, iddpm_forward, iddpm_param_shapes
    g = load_golden("train_small.npz")
    torch.set_num_threads(1)
    sd = synthetic_state_dict(iddpm_param_shapes(SMALL_I, n_delta=2), seed=11)
    x = hash_normal("ismall.x", (2, 3, 32, 32), seed=2)
    g1 = hash_normal("train.g_x0t", (2, 3, 32, 32), seed=3)
    g2 = hash_normal("train.g_xtn", (2, 3, 32, 32), seed=4)
    b = sampler.beta_schedule()
    ab = sampler.alpha_bar(b)
    for tag, ign in (("istep", False), ("iignoret", True)):
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("layer_0.")}
        et, em, _, _ = iddpm_forward({**sd, **leaves}, SMALL_I, x, torch.ones(2) * 701.0, index=0, t_edit=500, hs_coeff=(1.0, 0.8),
                                     ignore_timestep=ign)
        xn, x0t = sampler.ddim_update(x, et[:, :3], em[:, :3], ab[701], ab[675])
        ((x0t * g1).sum() + (xn * g2).sum()).backward()
        assert_close(x0t, g[f"{tag}.x0_t"], what="x0_t", rtol=1e-5, atol=2e-6 * float(g[f"{tag}.x0_t"].abs().max()))
        for k, v in leaves.items():
            want = g[f"{tag}.grad.{k}"]
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            scale = float(want.abs().max())
            assert_close(got, want, rtol=1e-4, atol=1e-5 * max(scale, 1e-30), what=f"{tag} grad {k}")


def test_oracle_on_config4_steps_with_shipped_church_delta_block():
    """The oracle against the reference's own config-4 steps (LSUN-church DDPM + the shipped `church_gothic` DeltaBlock,
    t_edit = 370): the last edited step (384 >= 370) and the first un-edited one."""
    from conftest import load_golden
    g = load_golden("config4_church_gothic.npz")
    sd = synthetic_state_dict(ddpm_param_shapes(CELEBA, n_delta=1), seed=4004)
    for k in list(g):
        if k.startswith("param."):
            sd[k[len("param."):]] = g[k]
    model = sampler.make_model(sd, CELEBA)
    b = sampler.beta_schedule()
    one = torch.ones(1)
    loose = dict(rtol=1e-4, atol=2e-5)
    x = hash_normal("config4.x384", (1, 3, 256, 256), seed=4004)
    xn, x0t, dh, _ = sampler.denoising_step(x, one * 384, one * 358, model=model, b=b, eta=0.0, index=0, t_edit=370,
                                            hs_coeff=(1.0, 1.0))
    assert_close(dh, g["gen384.delta_h"], what="delta_h", **loose)
    assert_close(xn, g["gen384.xt_next"], what="xt_next", **loose)
    x = hash_normal("config4.x358", (1, 3, 256, 256), seed=4004)
    xn, _, dh, _ = sampler.denoising_step(x, one * 358, one * 333, model=model, b=b, eta=0.0, index=0, t_edit=370,
                                          hs_coeff=(1.0, 1.0))
    assert dh is None
    assert_close(xn, g["gen358.xt_next"], what="xt_next below t_edit", **loose)


def test_iddpm_oracle_full_size_imagenet_adm_against_reference():
    """The oracle's iDDPM family at BASELINE config 5's size against the reference's own i_DDPM('IMAGENET') dual forward."""
    from conftest import load_golden
    from oracle.iddpm import IMAGENET, iddpm_forward
    from test_gpu_iddpm import imagenet_weights
    g = load_golden("imagenet_adm.npz")
    sd, x = imagenet_weights()
    assert torch.equal(x[0, 0, 0, :8], g["probe.x"]) and torch.equal(sd["out.2.weight"].reshape(-1)[:8], g["probe.w"])
    with torch.no_grad():
        et, em, dh, mh = iddpm_forward(sd, IMAGENET, x, torch.ones(1) * 700.0, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    for name, got in (("et", et), ("et_mod", em), ("delta_h", dh), ("middle_h", mh)):
        assert_close(got, g["fwd_dual." + name], what=name, rtol=1e-4, atol=2e-5)
