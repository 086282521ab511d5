"""The single-product fast mode (conv_math="f16", SURVEY §7 / §8(d) "bf16 fast mode reported separately").

NOT a parity mode: activations and weights are rounded to f16 and multiplied once (fp32 accumulate), so results differ from the
reference at the 1e-3 level and are checked against (a) a torch emulation of exactly that rounding - which the kernels must match
to accumulation-order noise - and (b) the fp32 reference at the mode's own stated tolerance:
    op level:     |y - y_fp32| <= 4e-3 * max|y_fp32|
    UNet forward: relative L2 error of eps <= 5e-3, max |d eps| <= 2e-2 * max|eps|
The parity modes (f16x3 / f32) keep rtol 1e-3 / atol 1e-4 everywhere else in tests/."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close
from oracle import sampler as osamp
from oracle.weights import SMALL, hash_normal, hash_uniform
from test_gpu_ops import _mk, hip_conv, ref_conv
from util_models import hip_model, synthetic

pytestmark = pytest.mark.gpu


def f16_round(t):
    return t.clamp(-65504, 65504).half().float()


def emulated_conv(x0, w, b, *, x1=None, stride=1, upsample=False, gn=None, silu=False, residual=None):
    """What the NP = 1 kernels compute: conv(rn_f16(act(x)), rn_f16(w * 2^k) / 2^k) with fp32-or-better accumulation."""
    x = x0 if x1 is None else torch.cat([x0, x1], dim=1)
    if gn:
        x = F.group_norm(x, 32, gn[0], gn[1], eps=1e-6)
    if silu:
        x = x * torch.sigmoid(x)
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    sc = 2.0 ** (10 - torch.floor(torch.log2(w.abs().max())))
    wq = f16_round(w * sc) / sc
    xq = f16_round(x)
    k = w.shape[-1]
    if stride == 2:
        y = F.conv2d(F.pad(xq.double(), (0, 1, 0, 1)), wq.double(), b.double(), stride=2)
    else:
        y = F.conv2d(xq.double(), wq.double(), b.double(), padding=k // 2)
    if residual is not None:
        y = y + residual.double()
    return y.float()


def _check(got, emu, exact, what):
    scale = float(exact.abs().max())
    # against the emulation: only accumulation order and the rare f16 ulp flip of an activation (GPU exp vs CPU exp) remain
    assert_close(got, emu, rtol=1e-3, atol=3e-4 * scale, what=f"{what} vs f16 emulation")
    err = float((got - exact).abs().max())
    assert err <= 4e-3 * scale, f"{what}: max err {err:.3e} vs fp32 exceeds the fast mode's tolerance 4e-3 * {scale:.3e}"
    assert err > 1e-6 * scale, f"{what}: suspiciously exact - did the three-product kernel run?"


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8])
def test_conv3x3_every_tile_single_product(tile):
    """Every compiled tile of the f16 family with NP = 1 on a ragged problem (partial tiles on both axes, concat input, GN+SiLU
    prologue, residual)."""
    B, H, W = 2, 40, 24
    x0 = hash_normal("fm.x0", (B, 64, H, W))
    x1 = hash_normal("fm.x1", (B, 32, H, W))
    w = hash_uniform("fm.w", (160, 96, 3, 3), -1, 1) / (96 * 9) ** 0.5
    b = 0.1 * hash_uniform("fm.b", (160,))
    gn = (1 + 0.1 * hash_uniform("fm.g", (96,)), 0.1 * hash_uniform("fm.be", (96,)))
    res = hash_normal("fm.r", (B, 160, H, W))
    kw = dict(x1=x1, gn=gn, silu=True, residual=res)
    got = hip_conv(x0, w, b, math="f16", tile=tile, **kw)
    _check(got, emulated_conv(x0, w, b, **kw), ref_conv(x0, w, b, **kw), f"tile {tile}")


@pytest.mark.parametrize("B,Cin,Cout,H,k,stride,ups", [(2, 128, 128, 32, 3, 1, False), (1, 256, 128, 48, 3, 1, False),
                                                        (2, 512, 512, 8, 3, 1, False), (2, 512, 512, 16, 3, 1, False),
                                                        (2, 128, 128, 32, 3, 2, False), (2, 128, 128, 16, 3, 1, True),
                                                        (2, 512, 1536, 16, 1, 1, False), (2, 3, 128, 32, 3, 1, False),
                                                        (1, 1024, 512, 8, 3, 1, False)])
def test_engine_choice_single_product(B, Cin, Cout, H, k, stride, ups):
    """The launcher's own tile choice (K32 main / 128-pixel / 8x8 / stride-2 forms, 1x1 tiles, conv_in's gather tile, split-K)."""
    x, w, b = _mk(B, Cin, Cout, H, k, f"fm2.{Cin}.{Cout}.{H}.{k}.{stride}.{int(ups)}")
    kw = dict(stride=stride, upsample=ups)
    got = hip_conv(x, w, b, math="f16", **kw)
    _check(got, emulated_conv(x, w, b, **kw), ref_conv(x, w, b, **kw), f"{Cin}->{Cout} @{H} k{k} s{stride}")


def test_conv_out_kernel_single_product():
    B, Cin, H = 2, 128, 32
    x = hash_normal("fm.co.x", (B, Cin, H, H))
    w = hash_uniform("fm.co.w", (3, Cin, 3, 3), -1, 1) / (Cin * 9) ** 0.5
    b = 0.1 * hash_uniform("fm.co.b", (3,))
    gn = (1 + 0.1 * hash_uniform("co.g", (Cin,)), 0.1 * hash_uniform("co.be", (Cin,)))
    got = hip_conv(x, w, b, gn=gn, silu=True, math="f16", tile=13)
    _check(got, emulated_conv(x, w, b, gn=gn, silu=True), ref_conv(x, w, b, gn=gn, silu=True), "conv_out kernel")


def test_attention_single_product():
    from asyrp_official_amd import _lib
    lib = _lib.load()
    B, Cc, T = 2, 128, 256
    qkv = hash_normal("fm.qkv", (B, 3 * Cc, T)).cuda()
    out = torch.empty((B, Cc, T), device="cuda")
    _lib.check(lib.asyrp_op_attention(0, C.c_void_p(qkv.data_ptr()), B, Cc, T, 1, 2, C.c_void_p(out.data_ptr()), None))
    q, k, v = qkv.cpu().double().chunk(3, dim=1)
    w_ = torch.softmax(torch.bmm(q.transpose(1, 2), k) * Cc ** -0.5, dim=2)
    want = torch.bmm(v, w_.transpose(1, 2)).float()
    err = float((out.cpu() - want).abs().max())
    scale = float(want.abs().max())
    print("attention f16: max err", err, "scale", scale)
    assert 1e-7 * scale < err <= 4e-3 * scale


def _rel_l2(a, b):
    return float((a - b).norm() / b.norm())


def test_unet_forward_and_edit_small():
    """Small DDPM UNet: dual forward and a 6+6 edit in the fast mode vs the CPU oracle (fp32): the mode's stated tolerance, and
    bitwise batch invariance (the tile choice does not depend on the mode or the batch)."""
    from asyrp_official_amd import run_edit
    sd = synthetic(SMALL, 1, seed=7)
    m = hip_model(SMALL, sd, 1, conv_math="f16", max_batch=4)
    x = hash_normal("small.x", (2, 3, 32, 32), seed=1)
    t = torch.ones(2) * 701.0
    et, em, dh, mh = m(x.cuda(), t.cuda(), index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    from oracle.ddpm import ddpm_forward
    with torch.no_grad():
        w_et, w_em, w_dh, w_mh = ddpm_forward(sd, SMALL, x, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    for name, got, want in (("et", et, w_et), ("et_mod", em, w_em), ("delta_h", dh, w_dh), ("middle_h", mh, w_mh)):
        r, mx = _rel_l2(got.cpu(), want), float((got.cpu() - want).abs().max())
        print(f"f16 small {name}: rel L2 {r:.3e} max abs {mx:.3e} (ref absmax {float(want.abs().max()):.3e})")
        assert r <= 5e-3 and mx <= 2e-2 * float(want.abs().max())
        assert r > 1e-6, "suspiciously exact: the parity kernels ran instead of the fast mode"
    b = osamp.beta_schedule()
    x4 = hash_normal("fm.edit.x", (4, 3, 32, 32), seed=5).cuda()
    full = run_edit(m, x4, b, n_inv=6, n_gen=6, t_edit=500)
    alone = run_edit(m, x4[2:3].contiguous(), b, n_inv=6, n_gen=6, t_edit=500)
    assert torch.equal(full[2:3], alone), "fast mode lost bitwise batch invariance"
    # training refuses the fast mode loudly
    from asyrp_official_amd import _lib
    eng = m._ready_engine(x4)
    with pytest.raises(_lib.AsyrpError, match="training step needs conv_math"):
        eng.train_forward(x4, 999, 749)


def test_fast_mode_attention_off_the_1x1_kernel_batch_gt_1():
    """attention at 8 x 8 puts the q|k|v projection on the 32x32x16 tile's split-plane epilogue (gemm1x1.hip
    takes HW = 256 only); in the fast mode its lo planes are null, and for every image after the first the epilogue used to form
    null + zo * stride before testing the pointer.  B = 3 in the fast mode: every row equals the image evaluated alone, bit for bit,
    and the result stays within the mode's tolerance of the fp32 oracle."""
    from oracle.ddpm import ddpm_forward
    from oracle.weights import DDPMConfig
    cfg = DDPMConfig(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(8,), resolution=32)
    sd = synthetic(cfg, 1, seed=21)
    x = hash_normal("fm.attn8.x", (3, 3, 32, 32), seed=2)
    t = torch.ones(3) * 701.0
    with torch.no_grad():
        want = ddpm_forward(sd, cfg, x, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    for math in ("f16", "f16x3"):
        m = hip_model(cfg, sd, 1, conv_math=math, max_batch=4)
        outs = m(x.cuda(), t.cuda(), index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        for i in range(3):
            alone = m(x[i:i + 1].cuda().contiguous(), t[i:i + 1].cuda(), index=0, t_edit=500, hs_coeff=(1.0, 1.0))
            for a_, o_ in zip(alone, outs):
                assert torch.equal(a_, o_[i:i + 1]), f"{math}: row {i} of the batch differs from the image alone"
        for name, got, w_ in zip(("et", "et_mod", "delta_h", "middle_h"), outs, want):
            if math == "f16":
                r, mx = _rel_l2(got.cpu(), w_), float((got.cpu() - w_).abs().max())
                assert r <= 5e-3 and mx <= 2e-2 * float(w_.abs().max()), f"fast mode {name}: rel L2 {r:.3e} max {mx:.3e}"
            else:
                assert_close(got, w_, what=f"f16x3 attention at 8x8 {name}")
