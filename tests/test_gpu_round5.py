"""Round-5 parity evidence, all against outputs of the REFERENCE itself (tests/golden/make_golden.py
run_config4_full / run_imagenet_traj / run_afhq_b2), plus run-to-run determinism of the paths that had no such test:

  * BASELINE config 4 END TO END: the LSUN-church DDPM + the shipped `church_gothic` DeltaBlock, 39 inversion + 40 Asyrp steps, t_edit = 370;
  * BASELINE config 5, a short FREE-RUNNING trajectory of i_DDPM('IMAGENET') (553.8 M parameters): 3 inversion + 4 Asyrp steps;
  * an iDDPM BATCH (B = 2, two different images) executed by the reference on the whole batch;
  * the AFHQ and ImageNet-ADM forwards, the DeltaBlock training step and the fast mode give the same bits when run again."""
import os

import pytest
import torch

from conftest import GOLDEN, assert_close, load_golden
from oracle import sampler as osamp
from oracle.weights import CELEBA, ddpm_param_shapes, hash_normal, hash_uniform, synthetic_state_dict
from util_models import err_stats, hip_model, synthetic

pytestmark = pytest.mark.gpu


def _need(name):
    if not os.path.exists(os.path.join(GOLDEN, name)):
        pytest.skip(f"{name} not generated (tests/golden/make_golden.py --only ...)")
    return load_golden(name)


def _afhq(max_batch, conv_math="f16x3", with_shipped_delta=True):
    from asyrp_official_amd import i_DDPM
    from oracle.iddpm import AFHQ, iddpm_param_shapes
    sd = synthetic_state_dict(iddpm_param_shapes(AFHQ, n_delta=1), seed=4321)
    if with_shipped_delta:
        g = load_golden("config3_afhq_dog_happy.npz")
        for k in list(g):
            if k.startswith("param."):
                sd[k[len("param."):]] = g[k]
    m = i_DDPM("AFHQ", max_batch=max_batch, conv_math=conv_math)
    m.setattr_layers(1)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return m.cuda().eval()


def test_config4_church_whole_edit_vs_reference():
    """39 + 40 steps free-running on the engine against the reference's x_T / x_edit (diffusion_latent.py:1034-1045, 503-520).  x_T
    (the benign direction) to <= 1e-4 of its elements outside the strict tolerance; x_edit relative to the trajectory scale, as for
    config 1 (the reference does not reproduce itself at 1e-4 on the untamed weights)."""
    from asyrp_official_amd import run_edit
    g = _need("config4_church_full.npz")
    gp = _need("config4_church_gothic.npz")
    sd = synthetic_state_dict(ddpm_param_shapes(CELEBA, n_delta=1), seed=4004)
    for k in list(gp):
        if k.startswith("param."):
            sd[k[len("param."):]] = gp[k]
    m = hip_model(CELEBA, sd, 1, max_batch=1)
    b = osamp.beta_schedule()
    x0 = hash_uniform("config4.x0", (1, 3, 256, 256), seed=4004).cuda()
    x_edit, x_T = run_edit(m, x0, b, n_inv=40, n_gen=40, t_edit=370, t_addnoise=0, want_latent=True)
    st_T, st_e = err_stats(x_T, g["x_T"]), err_stats(x_edit, g["x_edit"])
    print("config4 free-running x_T", st_T)
    print("config4 free-running x_edit", st_e)
    # x_T after 39 free-running steps: |x_T| reaches 5.3 here (config 1: 0.02-scale) and the strict tolerance is met by all but 0-2 of
    # the 196 608 elements depending on the rounding order of the build (max |err| 1.7e-4 ... 2.5e-4 across this round's binaries, each
    # printed above): bounded at 3e-4 of the tensor scale with at most 1e-4 of the elements outside rtol 1e-3 / atol 1e-4
    assert st_T["max_abs"] <= 3e-4 * max(1.0, st_T["ref_absmax"]) and st_T["frac_outside"] <= 1e-4
    assert st_e["max_abs"] <= 3e-4 * max(1.0, st_e["ref_absmax"]) and st_e["frac_outside"] <= 0.02


def test_imagenet_adm_short_trajectory_vs_reference():
    """Config 5 free-running: 3 DDIM inversion steps with learn_sigma, then 4 Asyrp steps (two dual-decoder, two single-decoder,
    the last to t_next = -1) on the reference's own time grid for n_inv = n_gen = 4."""
    from asyrp_official_amd import i_DDPM, run_edit
    from test_gpu_iddpm import imagenet_weights
    g = _need("imagenet_adm_traj.npz")
    sd, _ = imagenet_weights()
    assert torch.equal(sd["out.2.weight"].reshape(-1)[:8], g["probe.w"]), "CPU generator stream differs from the fixture's"
    m = i_DDPM("IMAGENET", max_batch=1)
    m.setattr_layers(1)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    b = osamp.beta_schedule()
    m.set_schedule(b)
    assert [int(v) for v in g["seq"]] == [0, 333, 666, 999]
    x0 = hash_uniform("imagenet.traj.x0", (1, 3, 256, 256), seed=77).cuda()
    x_edit, x_T = run_edit(m, x0, b, n_inv=4, n_gen=4, t_edit=500, learn_sigma=True, want_latent=True)
    st_T, st_e = err_stats(x_T, g["x_T"]), err_stats(x_edit, g["x_edit"])
    print("ImageNet-ADM free-running x_T", st_T)
    print("ImageNet-ADM free-running x_edit", st_e)
    assert st_T["max_abs"] <= 3e-4 * max(1.0, st_T["ref_absmax"]) and st_T["frac_outside"] <= 0.02
    assert st_e["max_abs"] <= 3e-4 * max(1.0, st_e["ref_absmax"]) and st_e["frac_outside"] <= 0.02
    # the first edited step teacher-forced from the reference's own x_T: strict
    eng = m._ready_engine(x0)
    xn, _, dh, _ = eng.ddim_step(g["x_T"].cuda(), 999, 666, apply_edit=True, index=0, hs_coeff=(1.0, 1.0), learn_sigma=True)
    print("ImageNet-ADM t=999 delta_h", err_stats(dh, g["gen999.delta_h"]), "xt_next", err_stats(xn, g["gen999.xt_next"]))
    assert_close(dh, g["gen999.delta_h"], what="ImageNet-ADM t=999 delta_h")
    assert_close(xn, g["gen999.xt_next"], what="ImageNet-ADM t=999 xt_next")


def test_afhq_batch_of_two_pinned_to_the_reference():
    """models/improved_ddpm/unet.py:676-752 on a batch of two DIFFERENT images, executed by the reference on the whole batch."""
    g = _need("iddpm_afhq_b2.npz")
    m = _afhq(2)
    b = osamp.beta_schedule()
    m.set_schedule(b)
    x0 = torch.cat([hash_uniform("afhqb2.x0a", (1, 3, 256, 256), seed=21), hash_uniform("afhqb2.x0b", (1, 3, 256, 256), seed=22)]).cuda()
    xm = torch.cat([hash_normal("afhqb2.xma", (1, 3, 256, 256), seed=23), hash_normal("afhqb2.xmb", (1, 3, 256, 256), seed=24)]).cuda()
    eng = m._ready_engine(x0)
    xn, _, _, _ = eng.ddim_step(x0, 0, 25, learn_sigma=True)
    assert_close(xn, g["inv0.xt_next"], what="AFHQ B=2 inversion 0->25 xt_next")
    xn, _, dh, _ = eng.ddim_step(xm, 768, 742, apply_edit=True, index=0, hs_coeff=(1.0, 1.0), learn_sigma=True)
    print("AFHQ B=2 t=768 delta_h", err_stats(dh, g["gen768.delta_h"]), "xt_next", err_stats(xn, g["gen768.xt_next"]))
    assert_close(dh, g["gen768.delta_h"], what="AFHQ B=2 t=768 delta_h")
    assert_close(xn, g["gen768.xt_next"], what="AFHQ B=2 t=768 xt_next (dual decoder)")
    assert float((g["inv0.xt_next"][0] - g["inv0.xt_next"][1]).abs().max()) > 0.1     # the two rows really are different images


def _same_bits(fn, n=3):
    first = [o.clone() for o in fn() if o is not None]
    for _ in range(n):
        for a_, f_ in zip([o for o in fn() if o is not None], first):
            assert torch.equal(a_, f_), f"run-to-run difference {float((a_.float() - f_.float()).abs().max()):.3e}"


def test_run_to_run_determinism_afhq_and_fast_mode():
    """AFHQ iDDPM (FiLM ResBlocks, pooled down path, 64-channel heads at T = 1024 / 256 / 64) at B = 4, plain and dual forward; the
    CelebA-HQ DDPM in the single-product fast mode at B = 8."""
    m = _afhq(4, with_shipped_delta=False)
    x = hash_normal("determinism.afhq", (4, 3, 256, 256), seed=5).cuda()
    for kw, tval in ((dict(), 500.0), (dict(index=0, t_edit=400, hs_coeff=(1.0, 1.0)), 701.0)):
        t = torch.ones(4, device="cuda") * tval
        _same_bits(lambda: m(x, t, **kw))
    del m
    mf = hip_model(CELEBA, synthetic(CELEBA, 1, seed=11), 1, max_batch=8, conv_math="f16")
    xf = hash_normal("determinism.x", (8, 3, 256, 256), seed=3).cuda()
    tf = torch.ones(8, device="cuda") * 701.0
    _same_bits(lambda: mf(xf, tf, index=0, t_edit=400, hs_coeff=(1.0, 1.0)))


def test_run_to_run_determinism_imagenet_adm():
    from asyrp_official_amd import i_DDPM
    from test_gpu_iddpm import imagenet_weights
    sd, x = imagenet_weights()
    m = i_DDPM("IMAGENET", max_batch=2)
    m.setattr_layers(1)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    xx = torch.cat([x, hash_normal("determinism.adm", (1, 3, 256, 256), seed=9)]).cuda()
    t = torch.ones(2, device="cuda") * 700.0
    _same_bits(lambda: m(xx, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0)), n=2)


def test_run_to_run_determinism_training_step():
    """The DeltaBlock training step at full size: x0_t and every parameter gradient have the same bits when the step is repeated."""
    from asyrp_official_amd import denoising_step
    sd = synthetic(CELEBA, 1, seed=1234)
    m = hip_model(CELEBA, sd, 1, max_batch=2)
    for p in m.parameters():
        p.requires_grad_(False)
    for p in m.layer_0.parameters():
        p.requires_grad_(True)
    x = hash_normal("determinism.train", (2, 3, 256, 256), seed=8).cuda()
    gx = hash_normal("train.g256b", (2, 3, 256, 256), seed=6).cuda()
    b = osamp.beta_schedule().cuda()
    two = torch.ones(2, device="cuda")

    def step():
        for p in m.layer_0.parameters():
            p.grad = None
        _, x0t, _, _ = denoising_step(x, t=two * 768.0, t_next=two * 743.0, models=m, logvars=None, b=b, sampling_type="ddim", eta=0.0,
                                      index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        (x0t * gx).sum().backward()
        return [x0t.detach()] + [p.grad for p in m.layer_0.parameters()]

    _same_bits(step, n=2)


@pytest.mark.parametrize("Cout", [6, 130])
def test_conv3x3_at_16x16_with_cout_not_a_multiple_of_4(Cout):
    """a 16 x 16 layer with Cin = 256 qualifies for the 2-way split-K K32 form by shape, but with Cout % 4 != 0 the K32
    float4 epilogue does not apply; the split factor, the tile and the launched kernel must be decided by ONE predicate (the launch
    used to fall through to a tile that was not the one the split rule promised).  Automatic tile choice, against torch fp32."""
    from test_gpu_ops import TIGHT, hip_conv, ref_conv
    B, Cin, H = 2, 256, 16
    x = hash_normal(f"r5.c16.x.{Cout}", (B, Cin, H, H))
    w = hash_uniform(f"r5.c16.w.{Cout}", (Cout, Cin, 3, 3), -1, 1) / (Cin * 9) ** 0.5
    b = 0.1 * hash_uniform(f"r5.c16.b.{Cout}", (Cout,))
    gn = (1 + 0.1 * hash_uniform("r5.c16.g", (Cin,)), 0.1 * hash_uniform("r5.c16.be", (Cin,)))
    got = hip_conv(x, w, b, gn=gn, silu=True)
    assert_close(got, ref_conv(x, w, b, gn=gn, silu=True), what=f"conv3x3 16x16 Cin=256 Cout={Cout}", **TIGHT)


@pytest.mark.parametrize("math", ["f16x3", "f16"])
@pytest.mark.parametrize("B,C0,C1,Cout,pro,res", [(3, 512, 0, 512, True, True), (2, 512, 0, 1536, True, False), (1, 512, 0, 512, False, True),
                                                   (5, 256, 256, 200, True, True), (4, 1024, 0, 512, True, False)])
def test_1x1_kernel_pair_form_on_8x8_maps(B, C0, C1, Cout, pro, res, math):
    """gemm1x1.hip on 8 x 8 maps (round 5): the two wave rows of a workgroup take two IMAGES -- mid-attention q|k|v / proj_out and
    the DeltaBlock's 1x1 convolutions, which ran on the 64 x 64 implicit-GEMM tile at 35 TFLOP/s.  Odd batches (the last workgroup's
    second image is absent), a concat input, a ragged N tile, per-image GroupNorm rows / channel vectors / residuals, against torch fp32;
    every image alone == the image in the batch bit for bit (whatever its partner and its wave row)."""
    from test_gpu_ops import TIGHT, hip_conv, ref_conv
    x0 = hash_normal(f"g1p.x0.{B}.{C0}", (B, C0, 8, 8))
    x1 = hash_normal(f"g1p.x1.{B}.{C1}", (B, C1, 8, 8)) if C1 else None
    Cin = C0 + C1
    w = hash_uniform(f"g1p.w.{Cin}.{Cout}", (Cout, Cin, 1, 1), -1, 1) / Cin ** 0.5
    b = 0.1 * hash_uniform(f"g1p.b.{Cout}", (Cout,))
    gn = (1 + 0.1 * hash_uniform(f"g1p.g.{Cin}", (Cin,)), 0.1 * hash_uniform(f"g1p.be.{Cin}", (Cin,))) if pro else None
    r = hash_normal(f"g1p.r.{B}.{Cout}", (B, Cout, 8, 8)) if res else None
    ca = hash_normal(f"g1p.ca.{B}.{Cout}", (B, Cout))
    kw = dict(x1=x1, gn=gn, silu=pro and Cout != 1536, residual=r, chan_add=ca)
    got = hip_conv(x0, w, b, math=math, tile=16, **kw)
    want = ref_conv(x0, w, b, **kw)
    if math == "f16x3":
        assert_close(got, want, what="1x1 kernel, pair form", **TIGHT)
        auto = hip_conv(x0, w, b, math=math, **kw)          # (the op hook's automatic tile: the implicit-GEMM tile; the ENGINE routes these layers to the pair form)
        for i in range(B):
            alone = hip_conv(x0[i:i + 1], w, b, x1=None if x1 is None else x1[i:i + 1], gn=gn, silu=kw["silu"],
                             residual=None if r is None else r[i:i + 1], chan_add=ca[i:i + 1], math=math, tile=16)
            assert torch.equal(alone[0], got[i]), f"pair form: image {i} depends on the batch"
        assert_close(auto, want, what="1x1 at 8x8, automatic choice", **TIGHT)
    else:
        err = float((got - want).abs().max())
        assert 1e-6 * float(want.abs().max()) < err <= 4e-3 * float(want.abs().max())


def test_1x1_kernel_pair_form_statistics():
    """GroupNorm partials of the pair form: one statistics row per image, written from the image's own wave row (odd batch)."""
    import ctypes as C
    import torch.nn.functional as F
    from asyrp_official_amd import _lib
    lib = _lib.load()
    _p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    B, Cin, Cout, H, W, offset = 3, 512, 512, 8, 8, 7.0
    x = hash_normal("g1pst.x", (B, Cin, H, W))
    w = hash_uniform("g1pst.w", (Cout, Cin, 1, 1), -1, 1) / Cin ** 0.5
    b = 0.1 * hash_uniform("g1pst.b", (Cout,)) + offset
    gam, bet = 1 + 0.1 * hash_uniform("g1pst.g", (Cout,)), 0.1 * hash_uniform("g1pst.be", (Cout,))
    d = lambda t: t.cuda().contiguous()
    xd, wd, bd, gd, bed = map(d, (x, w, b, gam, bet))
    y = torch.empty((B, Cout, H, W), device="cuda")
    sc, sh = torch.empty((B, Cout), device="cuda"), torch.empty((B, Cout), device="cuda")
    _lib.check(lib.asyrp_op_conv2d_stats(0, _p(xd), Cin, B, H, W, _p(wd), _p(bd), Cout, 1, 16, _p(gd), _p(bed), 1e-6,
                                         _p(y), _p(sc), _p(sh), None))
    torch.cuda.synchronize()
    want_y = F.conv2d(x, w, b)
    assert_close(y.cpu(), want_y, what="conv", rtol=1e-4, atol=2e-5 * offset)
    got_gn = y.cpu() * sc.cpu()[:, :, None, None] + sh.cpu()[:, :, None, None]
    want_gn = F.group_norm(want_y.double(), 32, gam.double(), bet.double(), eps=1e-6).float()
    assert_close(got_gn, want_gn, what="fused GN, pair form", rtol=1e-3, atol=1e-4)


def test_strength_sweep_as_batch_entries_equals_per_tuple_passes():
    """cache.edit_sweep (round 5): the coefficient tuples of an editing-strength sweep (diffusion_latent.py:726-755) run as batch entries
    with ONE TUPLE PER IMAGE (asyrp_run_edit's per-image table) instead of one generation pass per tuple; every image of every tuple
    must equal the per-tuple pass bit for bit -- single attribute (2 coefficients) and two attributes (3), chunking over max_batch."""
    from asyrp_official_amd import cache
    from oracle.weights import SMALL
    for n_delta, index in ((1, 0), (2, 1)):
        sd = synthetic(SMALL, n_delta, seed=21)
        m = hip_model(SMALL, sd, n_delta, max_batch=6)
        b = osamp.beta_schedule()
        x_T = hash_normal(f"sweep.xT.{n_delta}", (2, 3, 32, 32), seed=5).cuda()
        base = cache.make_hs_coeff(40, 6, n_attr=n_delta)
        tuples = cache.delta_interpolation_coeffs(-1.0, 2.0, 4, hs_coeff=base) if n_delta == 1 else \
            cache.delta_interpolation_coeffs(0.0, 1.5, 2, hs_coeff=base, multiple_attr=True)
        kw = dict(n_gen=6, t_edit=400, index=index)
        one_by_one = cache.edit_sweep(m, x_T, b, tuples, batched=False, **kw)
        batched = cache.edit_sweep(m, x_T, b, tuples, batched=True, **kw)       # 4 tuples x 2 images over max_batch 6: two calls
        assert len(batched) == len(tuples) == 4
        for k, (a_, b_) in enumerate(zip(batched, one_by_one)):
            assert a_.shape == x_T.shape and torch.equal(a_, b_), f"tuple {k}: batched sweep differs from the per-tuple pass"
        assert not torch.equal(batched[0], batched[-1])                           # the strengths really differ


def test_per_image_hs_coeff_is_validated():
    from asyrp_official_amd import run_edit
    from oracle.weights import SMALL
    m = hip_model(SMALL, synthetic(SMALL, 1, seed=21), 1, max_batch=4)
    b = osamp.beta_schedule()
    x_T = hash_normal("sweep.bad", (2, 3, 32, 32), seed=5).cuda()
    with pytest.raises(ValueError):
        run_edit(m, x_T, b, invert=False, n_gen=4, hs_coeff=[(1.0, 1.0)] * 3)            # 3 tuples for 2 images
    with pytest.raises(ValueError):
        run_edit(m, x_T, b, invert=False, n_gen=4, hs_coeff=[(1.0, 1.0, 1.0)] * 2)       # wrong tuple length for index 0
