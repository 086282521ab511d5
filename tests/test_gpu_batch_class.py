"""The small batch class (asyrp_config.nominal_batch = 1; round 4): tile shapes and split-K factors priced at a
nominal batch of 1 instead of 32.  It is a property of the ENGINE: on such an engine every call runs the same kernels on an image,
so an image alone still equals its row of a batch bit for bit; against the default class the results agree to fp32 rounding."""
import pytest
import torch

from conftest import assert_close, load_golden
from oracle.weights import CELEBA, SMALL, hash_normal
from util_models import err_stats, hip_model, synthetic

pytestmark = pytest.mark.gpu


def test_small_class_full_size_forward_vs_reference_fixture():
    """CelebA-HQ DDPM 256x256, dual forward at t = 768 on a small-class engine, against the reference's own outputs."""
    sd = synthetic(CELEBA, 1, seed=1234)
    m = hip_model(CELEBA, sd, 1, max_batch=2, nominal_batch=1)
    x = hash_normal("celeba.x", (1, 3, 256, 256), seed=1234)
    t = torch.ones(1) * 768.0
    et, em, dh, mh = m(x.cuda(), t.cuda(), index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    g = load_golden("ddpm_celeba.npz")
    for name, got in (("fwd_dual.et", et), ("fwd_dual.et_mod", em), ("fwd_dual.delta_h", dh)):
        print(name, err_stats(got, g[name]))
        assert_close(got, g[name], what=f"small class {name}")
    et1, _, _, mh1 = m(x.cuda(), t.cuda())
    assert_close(et1, g["fwd_single.et"], what="small class single et")
    assert_close(mh1, g["fwd_single.middle_h"], what="small class single middle_h")
    # the default class on the same input: fp32-rounding apart, not bitwise
    m32 = hip_model(CELEBA, sd, 1, max_batch=2)
    et32, em32, _, _ = m32(x.cuda(), t.cuda(), index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    assert_close(et, et32.cpu(), what="small vs default class et", rtol=1e-4, atol=2e-5)
    assert_close(em, em32.cpu(), what="small vs default class et_mod", rtol=1e-4, atol=2e-5)


def test_small_class_batch_invariance_bitwise():
    """On one small-class engine: image i alone == image i inside a batch of 3, bit for bit (dual forward, full size)."""
    sd = synthetic(CELEBA, 1, seed=1234)
    m = hip_model(CELEBA, sd, 1, max_batch=3, nominal_batch=1)
    x = torch.cat([hash_normal("celeba.x", (1, 3, 256, 256), seed=1234), hash_normal("celeba.x2", (1, 3, 256, 256), seed=5),
                   hash_normal("celeba.x3", (1, 3, 256, 256), seed=6)]).cuda()
    t3, t1 = torch.ones(3, device="cuda") * 768.0, torch.ones(1, device="cuda") * 768.0
    et3, em3, dh3, mh3 = m(x, t3, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    for i in (0, 2):
        et1, em1, dh1, mh1 = m(x[i:i + 1].contiguous(), t1, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        assert torch.equal(et3[i:i + 1], et1) and torch.equal(em3[i:i + 1], em1)
        assert torch.equal(dh3[i:i + 1], dh1) and torch.equal(mh3[i:i + 1], mh1)


def test_small_class_small_unet_vs_reference_fixture():
    """32x32 UNet (attention at 16x16, every block type) on a small-class engine against the reference fixture."""
    g = load_golden("ddpm_small.npz")
    sd = synthetic(SMALL, 2, seed=7)
    m = hip_model(SMALL, sd, 2, nominal_batch=1)
    x = hash_normal("small.x", (2, 3, 32, 32), seed=1)
    t = torch.ones(2, device="cuda") * 701.0
    et, em, dh, mh = m(x.cuda(), t, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    assert_close(et, g["fwd_dual.et"], what="small class et")
    assert_close(em, g["fwd_dual.et_mod"], what="small class et_mod")
    assert_close(dh, g["fwd_dual.delta_h"], what="small class delta_h")
    assert_close(mh, g["fwd_dual.middle_h"], what="small class middle_h")


def test_small_class_iddpm_afhq_full_size_vs_reference_fixture():
    """The iDDPM / ADM family on a small-class engine: i_DDPM('AFHQ') 256x256 (FiLM ResBlocks, up / down ResBlocks, 64-channel
    heads), dual and single forward against the reference's own UNetModel outputs (the recipe of test_afhq_full_size_forward)."""
    from asyrp_official_amd import i_DDPM
    from oracle.iddpm import AFHQ, iddpm_param_shapes
    from oracle.weights import synthetic_state_dict
    ga = load_golden("iddpm_afhq.npz")
    sd = synthetic_state_dict(iddpm_param_shapes(AFHQ, n_delta=1), seed=4321)
    m = i_DDPM("AFHQ", max_batch=1, nominal_batch=1)
    m.setattr_layers(1)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x = hash_normal("afhq.x", (1, 3, 256, 256), seed=4321)
    t = torch.ones(1) * 768.0
    et, em, dh, mh = m(x.cuda(), t.cuda(), index=0, t_edit=444, hs_coeff=(1.0, 1.0))
    for name, got in (("fwd_dual.et", et), ("fwd_dual.et_mod", em), ("fwd_dual.delta_h", dh)):
        print(name, err_stats(got, ga[name]))
        assert_close(got, ga[name], what=f"small class AFHQ {name}")
    et1, _, _, mh1 = m(x.cuda(), t.cuda())
    assert_close(et1, ga["fwd_single.et"], what="small class AFHQ single et")
    assert_close(mh1, ga["fwd_single.middle_h"], what="small class AFHQ single middle_h")
