"""Op-level parity of the HIP kernels (through the C ABI test hooks) against torch-CPU fp32 —
the same ATen ops the reference dispatches."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close
from oracle.weights import hash_normal, hash_uniform

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


MATHS = ("f16x3", "f32")


def hip_conv(x0, w, b, *, x1=None, stride=1, upsample=False, gn=None, silu=False, chan_add=None, residual=None,
             math="f16x3", tile=0):
    from asyrp_official_amd import _lib
    lib = _lib.load()
    dev = "cuda"
    B, C0, H, W = x0.shape
    Cout, _, k, _ = w.shape
    Ho, Wo = (H * 2, W * 2) if upsample else (H, W)
    if stride == 2:
        Ho, Wo = Ho // 2, Wo // 2
    d = lambda t: None if t is None else t.to(dev).contiguous()
    x0d, x1d, wd, bd, cad, rd = d(x0), d(x1), d(w), d(b), d(chan_add), d(residual)
    gw, gb = (d(gn[0]), d(gn[1])) if gn else (None, None)
    y = torch.empty((B, Cout, Ho, Wo), device=dev)
    _lib.check(lib.asyrp_op_conv2d(0, _p(x0d), C0, _p(x1d), 0 if x1 is None else x1.shape[1], B, H, W, _p(wd), _p(bd),
                                   Cout, k, stride, int(upsample), _p(gw), _p(gb), 1e-6, int(silu), _p(cad), _p(rd),
                                   _p(y), _lib.CONV_MATH[math], int(tile), None))
    torch.cuda.synchronize()
    return y.cpu()


def ref_conv(x0, w, b, *, x1=None, stride=1, upsample=False, gn=None, silu=False, chan_add=None, residual=None):
    x = x0 if x1 is None else torch.cat([x0, x1], dim=1)
    if gn:
        x = F.group_norm(x, 32, gn[0], gn[1], eps=1e-6)
    if silu:
        x = x * torch.sigmoid(x)
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    k = w.shape[-1]
    if stride == 2:
        y = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    else:
        y = F.conv2d(x, w, b, padding=k // 2)
    if chan_add is not None:
        y = y + chan_add[:, :, None, None]
    if residual is not None:
        y = residual + y
    return y


def _mk(B, Cin, Cout, H, k, tag):
    x = hash_normal(f"{tag}.x", (B, Cin, H, H))
    w = hash_uniform(f"{tag}.w", (Cout, Cin, k, k), -1, 1) / (Cin * k * k) ** 0.5
    b = 0.1 * hash_uniform(f"{tag}.b", (Cout,))
    return x, w, b


TIGHT = dict(rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("B,Cin,Cout,H", [(2, 32, 32, 16), (1, 128, 128, 32), (2, 64, 96, 8), (3, 32, 64, 24),
                                          (1, 256, 512, 8), (2, 3, 32, 16), (2, 32, 3, 16), (1, 96, 160, 20)])
@pytest.mark.parametrize("math", MATHS)
def test_conv3x3_plain(B, Cin, Cout, H, math):
    x, w, b = _mk(B, Cin, Cout, H, 3, f"c3.{B}.{Cin}.{Cout}.{H}")
    assert_close(hip_conv(x, w, b, math=math), ref_conv(x, w, b), what="conv3x3", **TIGHT)


def test_f16x3_narrow_tile_for_conv_out():
    """256x32 tile (conv_out: 128 -> 3 / 6 channels at full resolution): forced and as the launcher's own choice."""
    for Cout, tile in ((3, 12), (6, 12), (6, 0)):
        B, Cin, H = 2, 32, 48
        x, w, b = _mk(B, Cin, Cout, H, 3, f"narrow.{Cout}")
        gn = (1 + 0.1 * hash_uniform("narrow.g", (Cin,)), 0.1 * hash_uniform("narrow.be", (Cin,)))
        got = hip_conv(x, w, b, gn=gn, silu=True, tile=tile)
        assert_close(got, ref_conv(x, w, b, gn=gn, silu=True), what=f"conv_out tile {tile} Cout {Cout}", **TIGHT)


@pytest.mark.parametrize("Cout,Cin,H,W", [(3, 128, 32, 32), (2, 64, 32, 32), (3, 32, 40, 24), (1, 256, 16, 48), (3, 128, 256, 256)])
def test_conv_out_taps_in_n_kernel(Cout, Cin, H, W):
    """csrc/conv_out.hip (tile 13): the UNet's last conv with the 9 taps folded into N, GroupNorm + SiLU prologue; partial tiles
    (40x24), ragged output widths, and the production size (the 6-channel iDDPM head stays on the implicit-GEMM tile)."""
    B = 2 if H < 256 else 1
    x = hash_normal(f"co.x.{Cout}.{Cin}.{H}.{W}", (B, Cin, H, W))
    w = hash_uniform(f"co.w.{Cout}.{Cin}", (Cout, Cin, 3, 3), -1, 1) / (Cin * 9) ** 0.5
    b = 0.1 * hash_uniform(f"co.b.{Cout}", (Cout,))
    gn = (1 + 0.1 * hash_uniform("co.g", (Cin,)), 0.1 * hash_uniform("co.be", (Cin,)))
    got = hip_conv(x, w, b, gn=gn, silu=True, tile=13)
    assert_close(got, ref_conv(x, w, b, gn=gn, silu=True), what=f"conv_out kernel Cout={Cout} Cin={Cin} {H}x{W}", **TIGHT)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("k", [1, 3])
def test_f16x3_every_tile_shape(tile, k):
    """Force each compiled tile shape of the f16x3 family (256x128 4-wave, 128x128, 64x128, 64x64, 256x64, 256x128
    8-wave, 256x128 8-wave on the 16x16x32 instruction) on a ragged problem: 40x24 pixels (partial tiles on both axes), 64+32
    concatenated channels, 160 output channels (partial N tile)."""
    if tile in (6, 7, 8, 9) and k == 1:
        pytest.skip("the 8-wave tiles are compiled for 3x3 convolutions only")
    B, H, W = 2, 40, 24
    x0 = hash_normal(f"tile.x0.{k}", (B, 64, H, W))
    x1 = hash_normal(f"tile.x1.{k}", (B, 32, H, W)) * 3.0
    w = hash_uniform(f"tile.w.{k}", (160, 96, k, k), -1, 1) / (96 * k * k) ** 0.5
    b = 0.1 * hash_uniform(f"tile.b.{k}", (160,))
    res = hash_normal(f"tile.r.{k}", (B, 160, H, W))
    ca = hash_normal(f"tile.ca.{k}", (B, 160))
    gn = (1 + 0.1 * hash_uniform("tile.g", (96,)), 0.1 * hash_uniform("tile.be", (96,)))

    def run(**kw):
        from asyrp_official_amd import _lib
        lib = _lib.load()
        d = lambda t: t.cuda().contiguous()
        y = torch.empty((B, 160, H, W), device="cuda")
        a0, a1, wd, bd, g0, g1, cad, rd = map(d, (x0, x1, w, b, gn[0], gn[1], ca, res))
        _lib.check(lib.asyrp_op_conv2d(0, _p(a0), 64, _p(a1), 32, B, H, W, _p(wd), _p(bd), 160, k, 1, 0, _p(g0), _p(g1),
                                       1e-6, 1, _p(cad), _p(rd), _p(y), _lib.MATH_F16X3, tile, None))
        torch.cuda.synchronize()
        return y.cpu()

    x = torch.cat([x0, x1], 1)
    x = F.group_norm(x, 32, gn[0], gn[1], eps=1e-6)
    x = x * torch.sigmoid(x)
    want = res + F.conv2d(x, w, b, padding=k // 2) + ca[:, :, None, None]
    assert_close(run(), want, what=f"tile {tile} k={k}", **TIGHT)


@pytest.mark.parametrize("tile,k,H,W,Cout,offset", [(0, 3, 16, 16, 64, 0.0), (1, 3, 40, 24, 160, 0.0), (2, 1, 40, 24, 96, 0.0),
                                                    (3, 3, 20, 12, 96, 30.0), (4, 3, 8, 8, 32, 0.0), (6, 3, 32, 32, 128, 5.0),
                                                    (7, 3, 32, 32, 128, 5.0), (7, 3, 40, 24, 160, 0.0), (8, 3, 16, 16, 128, 5.0),
                                                    (8, 3, 20, 36, 160, 0.0), (9, 3, 8, 8, 128, 5.0), (9, 3, 20, 12, 160, 0.0)])
def test_fused_groupnorm_statistics_epilogue(tile, k, H, W, Cout, offset):
    """The conv epilogue's per-block {sum, sumsq} partials + finalize == GroupNorm of the conv output (incl. partial
    tiles, 3-channel groups that are not lane-aligned, and a large mean offset that would break a naive fp32 E[x^2]-m^2)."""
    from asyrp_official_amd import _lib
    lib = _lib.load()
    B, Cin = 2, 32
    x = hash_normal(f"st.x.{tile}", (B, Cin, H, W))
    w = hash_uniform(f"st.w.{tile}", (Cout, Cin, k, k), -1, 1) / (Cin * k * k) ** 0.5
    b = 0.1 * hash_uniform(f"st.b.{tile}", (Cout,)) + offset
    gam, bet = 1 + 0.1 * hash_uniform("st.g", (Cout,)), 0.1 * hash_uniform("st.be", (Cout,))
    d = lambda t: t.cuda().contiguous()
    xd, wd, bd, gd, bed = map(d, (x, w, b, gam, bet))
    y = torch.empty((B, Cout, H, W), device="cuda")
    sc, sh = torch.empty((B, Cout), device="cuda"), torch.empty((B, Cout), device="cuda")
    _lib.check(lib.asyrp_op_conv2d_stats(0, _p(xd), Cin, B, H, W, _p(wd), _p(bd), Cout, k, tile, _p(gd), _p(bed), 1e-6,
                                         _p(y), _p(sc), _p(sh), None))
    torch.cuda.synchronize()
    want_y = F.conv2d(x, w, b, padding=k // 2)
    assert_close(y.cpu(), want_y, what="conv", rtol=1e-4, atol=2e-5 * max(1.0, offset))
    got_gn = y.cpu() * sc.cpu()[:, :, None, None] + sh.cpu()[:, :, None, None]
    want_gn = F.group_norm(want_y.double(), 32, gam.double(), bet.double(), eps=1e-6).float()
    assert_close(got_gn, want_gn, what="fused GN", rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("Cin,C0,H,W,Cout,ups", [(32, 32, 16, 16, 128, 0), (64, 32, 48, 32, 128, 0), (160, 96, 20, 36, 192, 0),
                                                 (128, 128, 16, 16, 128, 1), (256, 128, 64, 64, 128, 0)])
def test_f16x3_k32_tile_matches_the_32x32x16_tile(Cin, C0, H, W, Cout, ups):
    """The 8-wave tile on v_mfma_f32_16x16x32_f16 (tile 7: one instruction spans two consecutive (chunk, tap) slices) against the
    fp32 reference AND against the 32x32x16 organisation of the same tile (tile 6), whose products are the same and whose
    summation order differs only inside a K = 32 block: odd and even chunk counts per source, a step that straddles the chunk
    boundary (and the two concatenated sources), partial tiles, the fused nearest-x2 upsample, GroupNorm + SiLU prologue."""
    from asyrp_official_amd import _lib
    lib = _lib.load()
    B, C1 = 2, Cin - C0
    Hs, Ws = (H // 2, W // 2) if ups else (H, W)
    tag = f"k32.{Cin}.{C0}.{H}.{Cout}"
    x0 = hash_normal(tag + ".x0", (B, C0, Hs, Ws))
    x1 = hash_normal(tag + ".x1", (B, C1, Hs, Ws)) * 2.0 if C1 else None
    w = hash_uniform(tag + ".w", (Cout, Cin, 3, 3), -1, 1) / (Cin * 9) ** 0.5
    b = 0.1 * hash_uniform(tag + ".b", (Cout,))
    res = hash_normal(tag + ".r", (B, Cout, H, W))
    ca = hash_normal(tag + ".ca", (B, Cout))
    gn = (1 + 0.1 * hash_uniform(tag + ".g", (Cin,)), 0.1 * hash_uniform(tag + ".be", (Cin,)))
    d = lambda t: None if t is None else t.cuda().contiguous()

    def run(tile):
        y = torch.empty((B, Cout, H, W), device="cuda")
        a0, a1, wd, bd, g0, g1, cad, rd = map(d, (x0, x1, w, b, gn[0], gn[1], ca, res))
        _lib.check(lib.asyrp_op_conv2d(0, _p(a0), C0, _p(a1), C1, B, Hs, Ws, _p(wd), _p(bd), Cout, 3, 1, ups, _p(g0), _p(g1),
                                       1e-6, 1, _p(cad), _p(rd), _p(y), _lib.MATH_F16X3, tile, None))
        torch.cuda.synchronize()
        return y.cpu()

    x = x0 if x1 is None else torch.cat([x0, x1], 1)
    x = F.group_norm(x, 32, gn[0], gn[1], eps=1e-6)
    x = x * torch.sigmoid(x)
    if ups:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    want = res + F.conv2d(x, w, b, padding=1) + ca[:, :, None, None]
    got7, got6, got8 = run(7), run(6), run(8)
    assert_close(got7, want, what=f"tile 7 {tag}", **TIGHT)
    assert_close(got7, got6, what=f"tile 7 vs tile 6 {tag}", rtol=1e-5, atol=2e-6)
    assert torch.equal(got7, run(7)), "the K32 tile must be deterministic"
    # the 128-pixel form of the kernel (8 x 16 patch, waves of 64 pixels x 32 channels) issues the same instructions on the same
    # operands in the same order per accumulator: bit-identical to the 256-pixel form
    assert torch.equal(got8, got7), "tile 8 must equal tile 7 bitwise"
    assert torch.equal(run(9), got7), "tile 9 (8 x 8 patches, 16 channels per wave) must equal tile 7 bitwise"


@pytest.mark.parametrize("B,Ch,C0,C1,Cout,H,W", [(1, 64, 64, 32, 64, 16, 16), (2, 128, 128, 128, 128, 40, 24),
                                                   (1, 32, 48, 0, 160, 20, 36), (2, 64, 48, 48, 192, 36, 20),
                                                   (1, 96, 32, 0, 128, 16, 16), (2, 64, 64, 64, 128, 8, 8)])
def test_fused_shortcut_resblock_tail(B, Ch, C0, C1, Cout, H, W):
    """conv3x3(swish(GN(h))) + nin_shortcut(cat(x0, x1)) in one launch: the shortcut's 1x1 runs as extra single-tap K-chunks
    of the 3x3 conv (partial tiles, two-source concat, Cout not a multiple of the N tile).  Channel counts that are multiples of
    32 run on the 16x16x32 form of the main tile (two raw slices per step; 48+48: a step straddles the two sources), the others
    (48 raw channels; 96 = an odd number of 16-channel chunks is still a multiple of 32) on the 32x32x16 form."""
    from asyrp_official_amd import _lib
    lib = _lib.load()
    tag = f"sc.{Ch}.{C0}.{C1}.{Cout}"
    h = hash_normal(tag + ".h", (B, Ch, H, W)) * 2.0 + 0.3
    x0 = hash_normal(tag + ".x0", (B, C0, H, W))
    x1 = hash_normal(tag + ".x1", (B, C1, H, W)) * 1.5 if C1 else None
    w3 = hash_uniform(tag + ".w3", (Cout, Ch, 3, 3), -1, 1) / (Ch * 9) ** 0.5
    w1 = hash_uniform(tag + ".w1", (Cout, C0 + C1, 1, 1), -1, 1) / (C0 + C1) ** 0.5
    b3, b1 = 0.1 * hash_uniform(tag + ".b3", (Cout,)), 0.1 * hash_uniform(tag + ".b1", (Cout,))
    gw, gb = 1 + 0.1 * hash_uniform(tag + ".g", (Ch,)), 0.1 * hash_uniform(tag + ".be", (Ch,))
    d = lambda t: None if t is None else t.cuda().contiguous()
    hd, x0d, x1d, w3d, w1d, b3d, b1d, gwd, gbd = map(d, (h, x0, x1, w3, w1, b3, b1, gw, gb))
    y = torch.empty((B, Cout, H, W), device="cuda")
    _lib.check(lib.asyrp_op_resblock_tail(0, _p(hd), Ch, _p(x0d), C0, _p(x1d), C1, B, H, W, _p(w3d), _p(b3d), _p(w1d), _p(b1d),
                                          Cout, _p(gwd), _p(gbd), 1e-6, _p(y), None))
    torch.cuda.synchronize()
    a = F.group_norm(h, 32, gw, gb, eps=1e-6)
    a = a * torch.sigmoid(a)
    x = x0 if x1 is None else torch.cat([x0, x1], 1)
    want = F.conv2d(a, w3, b3, padding=1) + F.conv2d(x, w1, b1)
    assert_close(y.cpu(), want, what="fused shortcut", **TIGHT)


@pytest.mark.parametrize("mag", [1.0, 1e-2, 1e-4, 1e-6])
def test_f16x3_small_magnitude_operands_keep_subnormal_lo_terms(mag):
    """x = x_hi + x_lo with f16 terms has relative precision 2^-22 while x_lo is a normal f16 (|x| >= 0.125) and an ABSOLUTE
    precision of 2^-25 below that (x_lo subnormal).  |x| ~ 1e-2 already makes every x_lo subnormal: had the matrix core
    flushed subnormal inputs, the lo terms would vanish (relative error ~2e-4); instead the error must stay within the
    representation bound  2^-24 * sum|w|  (+ fp32 accumulation noise)."""
    B, C, H = 1, 64, 16
    x = hash_normal(f"sub.x.{mag}", (B, C, H, H)) * mag
    w = hash_uniform(f"sub.w.{mag}", (C, C, 3, 3), -1, 1) / 24.0
    b = torch.zeros(C)
    got = hip_conv(x, w, b, math="f16x3")
    want = ref_conv(x.double(), w.double(), b.double())
    err = float((got.double() - want).abs().max())
    bound = 2.0 ** -24 * float(w.abs().sum(dim=(1, 2, 3)).max()) + 3e-6 * float(want.abs().max())
    print("mag", mag, "max err", err, "bound", bound, "rel to max|out|", err / float(want.abs().max()))
    assert err <= bound
    if mag >= 1e-2:
        assert err <= 3e-6 * float(want.abs().max())     # fp32-class: subnormal lo terms were not flushed


def test_f16x3_wide_dynamic_range():
    """Operands spanning many binades (weights 1e-4..1, activations 1e-3..1e2): the two-term f16 split with
    power-of-two pre-scaling must stay fp32-equivalent (no f16 subnormal/overflow loss)."""
    B, C, H = 1, 64, 16
    mag_x = 10.0 ** (hash_uniform("dr.mx", (B, C, H, H), -3, 2))
    x = hash_normal("dr.x", (B, C, H, H)) * mag_x
    mag_w = 10.0 ** (hash_uniform("dr.mw", (C, C, 3, 3), -4, 0))
    w = hash_uniform("dr.w", (C, C, 3, 3), -1, 1) * mag_w / 24.0
    b = torch.zeros(C)
    got, want = hip_conv(x, w, b, math="f16x3"), ref_conv(x.double(), w.double(), b.double()).float()
    assert_close(got, want, what="wide range", rtol=1e-4, atol=1e-5 * float(want.abs().max()))


@pytest.mark.parametrize("B,Cin,Cout,H", [(2, 32, 64, 16), (1, 256, 128, 32), (2, 64, 64, 8), (1, 512, 1536, 8)])
@pytest.mark.parametrize("math", MATHS)
def test_conv1x1(B, Cin, Cout, H, math):
    x, w, b = _mk(B, Cin, Cout, H, 1, f"c1.{B}.{Cin}.{Cout}.{H}")
    assert_close(hip_conv(x, w, b, math=math), ref_conv(x, w, b), what="conv1x1", **TIGHT)


@pytest.mark.parametrize("math", MATHS)
def test_conv3x3_gn_silu_prologue_and_epilogue(math):
    B, Cin, Cout, H = 2, 64, 32, 16
    x, w, b = _mk(B, Cin, Cout, H, 3, "fused")
    gn = (1 + 0.1 * hash_uniform("fused.g", (Cin,)), 0.1 * hash_uniform("fused.be", (Cin,)))
    ca = hash_normal("fused.ca", (B, Cout))
    res = hash_normal("fused.res", (B, Cout, H, H))
    kw = dict(gn=gn, silu=True, chan_add=ca, residual=res)
    assert_close(hip_conv(x, w, b, math=math, **kw), ref_conv(x, w, b, **kw), what="fused resblock conv", **TIGHT)


@pytest.mark.parametrize("math", MATHS)
def test_conv3x3_concat_two_sources_group_straddle(math):
    # 64 + 32 = 96 channels -> 3 per group: groups straddle the two sources (like 512+256 in up.4)
    B, H = 2, 16
    x0 = hash_normal("cat.x0", (B, 64, H, H)) + 0.5
    x1 = 2.0 * hash_normal("cat.x1", (B, 32, H, H)) - 0.3
    _, w, b = _mk(B, 96, 64, H, 3, "cat")
    gn = (1 + 0.1 * hash_uniform("cat.g", (96,)), 0.1 * hash_uniform("cat.be", (96,)))
    kw = dict(x1=x1, gn=gn, silu=True)
    assert_close(hip_conv(x0, w, b, math=math, **kw), ref_conv(x0, w, b, **kw), what="concat conv", **TIGHT)


@pytest.mark.parametrize("math", MATHS)
def test_conv3x3_stride2_asymmetric_pad(math):
    for (B, C, H) in [(2, 32, 16), (1, 128, 32), (1, 64, 8)]:
        x, w, b = _mk(B, C, C, H, 3, f"s2.{C}.{H}")
        assert_close(hip_conv(x, w, b, stride=2, math=math), ref_conv(x, w, b, stride=2), what="downsample conv", **TIGHT)


@pytest.mark.parametrize("B,C,Cout,H,W", [(2, 32, 128, 16, 16), (1, 128, 128, 64, 64), (2, 96, 160, 40, 24), (1, 64, 64, 8, 8),
                                          (1, 48, 64, 16, 16)])
def test_f16x3_stride2_k32_form_matches_the_32x32x16_tile(B, C, Cout, H, W):
    """DDPM Downsample (3x3, stride 2, pad right/bottom) on the 16x16x32 form of the kernel (4 x 16 output patch, 9 x 33 halo, the
    launcher's choice when Cin % 32 == 0) against the fp32 reference and against the 32x32x16 stride-2 tile (tile 3): partial
    patches on both axes, partial N tile, a bias; 48 input channels fall back to the 32x32x16 tile."""
    x = hash_normal(f"s2k.x.{C}.{H}.{W}", (B, C, H, W))
    w = hash_uniform(f"s2k.w.{C}.{Cout}", (Cout, C, 3, 3), -1, 1) / (C * 9) ** 0.5
    b = 0.1 * hash_uniform(f"s2k.b.{Cout}", (Cout,))
    want = ref_conv(x, w, b, stride=2)
    got, old = hip_conv(x, w, b, stride=2), hip_conv(x, w, b, stride=2, tile=3)
    assert_close(got, want, what="stride-2 K32", **TIGHT)
    assert_close(got, old, what="stride-2 K32 vs tile 3", rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("math", MATHS)
def test_conv3x3_nearest_upsample(math):
    for (B, C, H) in [(2, 32, 8), (1, 64, 16), (1, 128, 4)]:
        x, w, b = _mk(B, C, C, H, 3, f"up.{C}.{H}")
        assert_close(hip_conv(x, w, b, upsample=True, math=math), ref_conv(x, w, b, upsample=True), what="upsample conv",
                     **TIGHT)


@pytest.mark.parametrize("math", MATHS)
def test_gn_without_silu_is_the_attention_norm(math):
    B, C, H = 2, 64, 8
    x, w, b = _mk(B, C, 3 * C, H, 1, "qkv")
    gn = (1 + 0.1 * hash_uniform("qkv.g", (C,)), 0.1 * hash_uniform("qkv.be", (C,)))
    assert_close(hip_conv(x, w, b, gn=gn, math=math), ref_conv(x, w, b, gn=gn), what="qkv conv", **TIGHT)


def test_gn_large_mean_offset_is_stable():
    B, C, H = 1, 32, 32
    x, w, b = _mk(B, C, C, H, 3, "off")
    x = x + 30.0     # |mean| >> std: E[x^2]-mean^2 must not lose the variance
    gn = (torch.ones(C), torch.zeros(C))
    assert_close(hip_conv(x, w, b, gn=gn, silu=True), ref_conv(x, w, b, gn=gn, silu=True), what="gn offset",
                 rtol=1e-3, atol=1e-4)


ATT_CASES = [
    # (B, C, T, heads, fused)
    (2, 64, 64, 1, 0), (1, 512, 256, 1, 0), (2, 128, 64, 2, 0), (1, 512, 256, 8, 0),          # unfused fp32-MFMA path
    (2, 64, 64, 1, 1), (1, 512, 256, 1, 1), (2, 512, 64, 1, 1),                               # DDPM AttnBlock: one 512-wide head
    (2, 128, 64, 2, 1), (1, 512, 256, 8, 1), (1, 512, 64, 8, 1),                              # AFHQ iDDPM: 64-channel heads
    (1, 512, 1024, 8, 1), (1, 1024, 256, 16, 1), (1, 1024, 64, 16, 1),                        # ImageNet ADM sites (T = 1024 at 32x32)
    (2, 64, 256, 4, 1), (1, 32, 16, 2, 1),                                                    # toy UNets: 16-channel heads
    (1, 128, 96, 2, 1), (1, 64, 520, 1, 1),                                                   # ragged T: key / query masking
]


@pytest.mark.parametrize("B,C,T,heads,fused", ATT_CASES)
def test_attention(B, C, T, heads, fused):
    from asyrp_official_amd import _lib
    lib = _lib.load()
    qkv = hash_normal(f"att.{B}.{C}.{T}.{heads}", (B, 3 * C, T))
    out = torch.empty((B, C, T), device="cuda")
    qd = qkv.cuda()
    _lib.check(lib.asyrp_op_attention(0, _p(qd), B, C, T, heads, fused, _p(out), None))
    torch.cuda.synchronize()
    if heads == 1:      # models/ddpm/diffusion.py:205-221
        q, k, v = qkv.split(C, dim=1)
        w = torch.bmm(q.transpose(1, 2), k) * (C ** -0.5)
        ref = torch.bmm(v, F.softmax(w, dim=2).transpose(1, 2))
    else:               # QKVAttentionLegacy, models/improved_ddpm/unet.py:379-396
        ch = C // heads
        q, k, v = qkv.reshape(B * heads, ch * 3, T).split(ch, dim=1)
        scale = 1 / (ch ** 0.5) ** 0.5
        w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
        ref = torch.einsum("bts,bcs->bct", F.softmax(w.float(), dim=-1), v).reshape(B, -1, T)
    assert_close(out.cpu(), ref, what="attention", **TIGHT)


@pytest.mark.parametrize("math", ["f16x3", "f16"])
@pytest.mark.parametrize("B,Cin,Cout,H,W,pro", [(2, 64, 160, 40, 24, True), (1, 128, 128, 32, 32, False), (2, 256, 256, 16, 48, True),
                                                 (1, 32, 96, 17, 33, False)])
def test_polyphase_upsample_conv(B, Cin, Cout, H, W, pro, math):
    """Tile 11: nearest x2 + 3x3 (Upsample.conv, models/ddpm/diffusion.py:84-87; ResBlock(up=True).in_layers.2) as four phase-
    collapsed 2x2-tap convolutions on the source grid; partial tiles on both axes, partial N tile, with and without the
    GroupNorm + SiLU prologue, odd source sizes.  f16x3: the parity tolerance; f16: the fast mode's."""
    x = hash_normal(f"pp.x.{Cin}.{H}.{W}", (B, Cin, H, W))
    w = hash_uniform(f"pp.w.{Cin}.{Cout}", (Cout, Cin, 3, 3), -1, 1) / (Cin * 9) ** 0.5
    b = 0.1 * hash_uniform(f"pp.b.{Cout}", (Cout,))
    gn = (1 + 0.1 * hash_uniform("pp.g", (Cin,)), 0.1 * hash_uniform("pp.be", (Cin,))) if pro else None
    got = hip_conv(x, w, b, upsample=True, gn=gn, silu=pro, math=math, tile=11)
    want = ref_conv(x, w, b, upsample=True, gn=gn, silu=pro)
    assert got.shape == (B, Cout, 2 * H, 2 * W)
    if math == "f16x3":
        assert_close(got, want, what="polyphase x2 conv", **TIGHT)
        # and against the 3x3 form over the virtually up-sampled input (the launcher's default for this op hook)
        assert_close(got, hip_conv(x, w, b, upsample=True, gn=gn, silu=pro, math=math), what="polyphase vs 3x3 form", **TIGHT)
    else:
        err = float((got - want).abs().max())
        assert 1e-6 * float(want.abs().max()) < err <= 4e-3 * float(want.abs().max())


def _attention_ref(qkv, B, C, T, heads):
    if heads == 1:
        q, k, v = qkv.double().split(C, dim=1)
        w = torch.bmm(q.transpose(1, 2), k) * (C ** -0.5)
        return torch.bmm(v, F.softmax(w, dim=2).transpose(1, 2)).float()
    ch = C // heads
    q, k, v = qkv.double().reshape(B * heads, ch * 3, T).split(ch, dim=1)
    scale = 1 / (ch ** 0.5) ** 0.5
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    return torch.einsum("bts,bcs->bct", F.softmax(w, dim=-1), v).reshape(B, -1, T).float()


@pytest.mark.parametrize("B,C,T,heads", [(2, 512, 256, 1), (2, 512, 64, 1), (3, 64, 256, 1), (1, 512, 1024, 8), (2, 256, 256, 4),
                                         (1, 1024, 64, 16), (2, 512, 512, 8), (1, 192, 32, 3), (1, 64, 128, 1)])
@pytest.mark.parametrize("fused", [3, 4])
def test_attention_split_plane_kernel(B, C, T, heads, fused):
    """attn_planes_kernel (csrc/attention.hip; the engine's attention since round 3 for the head shapes of the reference's
    configurations: one 512-wide head at T <= 256, 64-wide heads at T <= 1024): q, k as fragment-major f16 hi/lo planes, v
    transposed, both layouts (DDPM q|k|v blocks, legacy per-head [q|k|v]), every key-tile instantiation (T = 32 ... 1024), 1 to
    16 heads.  fused = 3: the fp32-equivalent three-product form at the parity tolerance; 4: the single-product fast mode."""
    from asyrp_official_amd import _lib
    lib = _lib.load()
    qkv = hash_normal(f"att2.{B}.{C}.{T}.{heads}", (B, 3 * C, T))
    out = torch.empty((B, C, T), device="cuda")
    qd = qkv.cuda()
    _lib.check(lib.asyrp_op_attention(0, _p(qd), B, C, T, heads, fused, _p(out), None))
    torch.cuda.synchronize()
    ref = _attention_ref(qkv, B, C, T, heads)
    if fused == 3:
        assert_close(out.cpu(), ref, what="split-plane attention", **TIGHT)
    else:
        err, scale = float((out.cpu() - ref).abs().max()), float(ref.abs().max())
        assert 1e-7 * scale < err <= 4e-3 * scale, (err, scale)


def test_attention_split_plane_kernel_refuses_other_shapes():
    """Head widths other than 64 / 512 (and ragged T) stay on attn_f16x3_kernel: the hook says so instead of computing garbage."""
    from asyrp_official_amd import _lib
    lib = _lib.load()
    for (B, C, T, heads) in ((1, 128, 64, 1), (1, 96, 32, 3), (1, 64, 80, 1), (1, 512, 512, 1)):
        qd = torch.zeros((B, 3 * C, T), device="cuda")
        out = torch.empty((B, C, T), device="cuda")
        with pytest.raises(_lib.AsyrpError, match="not covered"):
            _lib.check(lib.asyrp_op_attention(0, _p(qd), B, C, T, heads, 3, _p(out), None))


@pytest.mark.parametrize("math", ["f16x3", "f16"])
@pytest.mark.parametrize("B,C0,C1,Cout,pro,res", [(5, 512, 0, 512, True, True), (4, 512, 512, 512, True, False), (1, 512, 0, 160, False, False),
                                                   (9, 256, 256, 256, True, True)])
def test_quad_form_8x8_layers(B, C0, C1, Cout, pro, res, math):
    """The launcher's choice for 8 x 8 layers with Cin % 256 == 0: four images per workgroup (2 x 2 arrangement of 8 x 8 patches, each
    with its own zero border), K split 8 ways, fixed-order reduce with bias / residual -- ragged groups (B = 5, 9, 1), concat input,
    per-image GroupNorm prologue, partial N tile."""
    x0 = hash_normal(f"q8.x0.{B}.{C0}", (B, C0, 8, 8))
    x1 = hash_normal(f"q8.x1.{B}.{C1}", (B, C1, 8, 8)) if C1 else None
    Cin = C0 + C1
    w = hash_uniform(f"q8.w.{Cin}.{Cout}", (Cout, Cin, 3, 3), -1, 1) / (Cin * 9) ** 0.5
    b = 0.1 * hash_uniform(f"q8.b.{Cout}", (Cout,))
    gn = (1 + 0.1 * hash_uniform(f"q8.g.{Cin}", (Cin,)), 0.1 * hash_uniform(f"q8.be.{Cin}", (Cin,))) if pro else None
    r = hash_normal(f"q8.r.{B}.{Cout}", (B, Cout, 8, 8)) if res else None
    ca = hash_normal(f"q8.ca.{B}.{Cout}", (B, Cout))
    kw = dict(x1=x1, gn=gn, silu=pro, residual=r, chan_add=ca)
    got = hip_conv(x0, w, b, math=math, **kw)
    want = ref_conv(x0, w, b, **kw)
    if math == "f16x3":
        assert_close(got, want, what="quad 8x8 form", **TIGHT)
        # image i alone == image i inside the batch, bit for bit (its partial sums do not depend on its group)
        i = B - 1
        alone = hip_conv(x0[i:i + 1], w, b, x1=None if x1 is None else x1[i:i + 1], gn=gn, silu=pro,
                         residual=None if r is None else r[i:i + 1], chan_add=ca[i:i + 1], math=math)
        assert torch.equal(alone[0], got[i]), "quad form: result depends on the batch"
    else:
        err = float((got - want).abs().max())
        assert 1e-6 * float(want.abs().max()) < err <= 4e-3 * float(want.abs().max())


@pytest.mark.parametrize("math", ["f16x3", "f16"])
@pytest.mark.parametrize("tile", [15, 16])
@pytest.mark.parametrize("B,C0,C1,Cout,H,W,pro,res", [(2, 512, 0, 1536, 16, 16, True, False), (3, 256, 0, 256, 16, 16, False, True),
                                                      (2, 160, 96, 200, 20, 12, True, True), (1, 64, 0, 24, 8, 8, False, False),
                                                      (2, 128, 64, 128, 40, 24, True, False)])
def test_barrier_free_1x1_kernel(tile, B, C0, C1, Cout, H, W, pro, res, math):
    """gemm1x1.hip (weights in MFMA fragment order, activations converted in registers, no LDS staging): q|k|v-sized, proj_out with
    residual, a concat with a ragged pixel tile and a channel count that is not a multiple of 16, an 8 x 8 image (partial M tile)."""
    x0 = hash_normal(f"g1.x0.{B}.{C0}", (B, C0, H, W))
    x1 = hash_normal(f"g1.x1.{B}.{C1}", (B, C1, H, W)) if C1 else None
    Cin = C0 + C1
    w = hash_uniform(f"g1.w.{Cin}.{Cout}", (Cout, Cin, 1, 1), -1, 1) / Cin ** 0.5
    b = 0.1 * hash_uniform(f"g1.b.{Cout}", (Cout,))
    gn = (1 + 0.1 * hash_uniform(f"g1.g.{Cin}", (Cin,)), 0.1 * hash_uniform(f"g1.be.{Cin}", (Cin,))) if pro else None
    r = hash_normal(f"g1.r.{B}.{Cout}", (B, Cout, H, W)) if res else None
    ca = hash_normal(f"g1.ca.{B}.{Cout}", (B, Cout))
    kw = dict(x1=x1, gn=gn, silu=pro and Cout != 1536, residual=r, chan_add=ca)
    got = hip_conv(x0, w, b, math=math, tile=tile, **kw)
    want = ref_conv(x0, w, b, **kw)
    if math == "f16x3":
        assert_close(got, want, what="1x1 kernel", **TIGHT)
        i = B - 1
        alone = hip_conv(x0[i:i + 1], w, b, x1=None if x1 is None else x1[i:i + 1], gn=gn, silu=kw["silu"],
                         residual=None if r is None else r[i:i + 1], chan_add=ca[i:i + 1], math=math, tile=tile)
        assert torch.equal(alone[0], got[i]), "1x1 kernel: result depends on the batch"
    else:
        err = float((got - want).abs().max())
        assert 1e-6 * float(want.abs().max()) < err <= 4e-3 * float(want.abs().max())


def test_barrier_free_1x1_kernel_refuses_other_shapes():
    from asyrp_official_amd import _lib
    x, w, b = _mk(1, 96, 64, 16, 1, "g1.bad")          # Cin % 64 != 0
    with pytest.raises(_lib.AsyrpError):
        hip_conv(x, w, b, tile=15)
    x, w, b = _mk(1, 64, 64, 16, 3, "g1.bad3")         # 3x3
    with pytest.raises(_lib.AsyrpError):
        hip_conv(x, w, b, tile=16)


@pytest.mark.parametrize("tile,H,W,Cin,Cout,offset", [(15, 16, 16, 64, 128, 5.0), (16, 16, 16, 128, 96, 0.0), (15, 20, 36, 64, 160, 0.0),
                                                     (16, 20, 36, 64, 160, 30.0)])
def test_barrier_free_1x1_kernel_statistics_epilogue(tile, H, W, Cin, Cout, offset):
    """GroupNorm partials of the 1x1 kernel's epilogue (proj_out feeds the next block's norm1) incl. a ragged last pixel tile."""
    from asyrp_official_amd import _lib
    lib = _lib.load()
    B = 2
    x = hash_normal(f"g1st.x.{tile}.{Cin}", (B, Cin, H, W))
    w = hash_uniform(f"g1st.w.{tile}.{Cin}", (Cout, Cin, 1, 1), -1, 1) / Cin ** 0.5
    b = 0.1 * hash_uniform(f"g1st.b.{tile}", (Cout,)) + offset
    gam, bet = 1 + 0.1 * hash_uniform("g1st.g", (Cout,)), 0.1 * hash_uniform("g1st.be", (Cout,))
    d = lambda t: t.cuda().contiguous()
    xd, wd, bd, gd, bed = map(d, (x, w, b, gam, bet))
    y = torch.empty((B, Cout, H, W), device="cuda")
    sc, sh = torch.empty((B, Cout), device="cuda"), torch.empty((B, Cout), device="cuda")
    _lib.check(lib.asyrp_op_conv2d_stats(0, _p(xd), Cin, B, H, W, _p(wd), _p(bd), Cout, 1, tile, _p(gd), _p(bed), 1e-6,
                                         _p(y), _p(sc), _p(sh), None))
    torch.cuda.synchronize()
    want_y = F.conv2d(x, w, b)
    assert_close(y.cpu(), want_y, what="conv", rtol=1e-4, atol=2e-5 * max(1.0, offset))
    got_gn = y.cpu() * sc.cpu()[:, :, None, None] + sh.cpu()[:, :, None, None]
    want_gn = F.group_norm(want_y.double(), 32, gam.double(), bet.double(), eps=1e-6).float()
    assert_close(got_gn, want_gn, what="fused GN", rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("math", ["f16x3", "f16"])
@pytest.mark.parametrize("B,C0,C1,Cout,pro,res", [(2, 512, 0, 512, True, True), (1, 512, 512, 512, True, False), (3, 256, 0, 512, False, False),
                                                   (2, 512, 256, 160, True, True)])
def test_splitk_16x16_layers(B, C0, C1, Cout, pro, res, math):
    """The launcher's choice for 3x3 layers on 16 x 16 maps (round 4): the 128-pixel K32 form with the K range split two ways and a
    fixed-order reduce (bias / timestep vector / residual / GroupNorm partials in the reduce) -- concat input, ragged N tile,
    against the fp32 reference, against the single-pass form (tile 8), and image-alone == image-in-batch bit for bit."""
    x0 = hash_normal(f"sk16.x0.{B}.{C0}", (B, C0, 16, 16))
    x1 = hash_normal(f"sk16.x1.{B}.{C1}", (B, C1, 16, 16)) if C1 else None
    Cin = C0 + C1
    w = hash_uniform(f"sk16.w.{Cin}.{Cout}", (Cout, Cin, 3, 3), -1, 1) / (Cin * 9) ** 0.5
    b = 0.1 * hash_uniform(f"sk16.b.{Cout}", (Cout,))
    gn = (1 + 0.1 * hash_uniform(f"sk16.g.{Cin}", (Cin,)), 0.1 * hash_uniform(f"sk16.be.{Cin}", (Cin,))) if pro else None
    r = hash_normal(f"sk16.r.{B}.{Cout}", (B, Cout, 16, 16)) if res else None
    ca = hash_normal(f"sk16.ca.{B}.{Cout}", (B, Cout))
    kw = dict(x1=x1, gn=gn, silu=pro, residual=r, chan_add=ca)
    got = hip_conv(x0, w, b, math=math, **kw)
    want = ref_conv(x0, w, b, **kw)
    if math == "f16x3":
        assert_close(got, want, what="split-K 16x16", **TIGHT)
        single = hip_conv(x0, w, b, math=math, tile=8, **kw)          # the same form without the split
        assert_close(got, single, what="split-K vs single pass", rtol=1e-5, atol=2e-6)
        i = B - 1
        alone = hip_conv(x0[i:i + 1], w, b, x1=None if x1 is None else x1[i:i + 1], gn=gn, silu=pro,
                         residual=None if r is None else r[i:i + 1], chan_add=ca[i:i + 1], math=math)
        assert torch.equal(alone[0], got[i]), "split-K 16x16: result depends on the batch"
    else:
        err = float((got - want).abs().max())
        assert 1e-6 * float(want.abs().max()) < err <= 4e-3 * float(want.abs().max())


@pytest.mark.parametrize("B,Cout,H,W,wmag", [(2, 128, 32, 32, 1.0), (1, 256, 40, 24, 1.0), (3, 64, 16, 16, 1.0), (1, 128, 256, 256, 1.0),
                                             (2, 128, 32, 32, 900.0), (2, 128, 32, 32, 1e-3)])
def test_first_convolution_stencil_kernel(B, Cout, H, W, wmag):
    """conv_in.hip: the 3 -> Cout first convolution -- ragged patches, 128 / 256 / 64 output channels, the full 256 x 256 size; image
    alone == image in batch bit for bit.  Default form (round 5): ONE K = 32 MFMA step with the two-term f16 operand split.  Its error
    model is per OUTPUT ELEMENT: the dropped x_lo w_lo terms (2^-22 relative per product) plus fp32 accumulation of <= 28 terms, i.e.
    c * sum_k |x_k| |w_k| -- checked against an fp64 convolution with exactly that bound (a looser rtol would hide a dropped
    cross term).  wmag scales the weights: 900 (|w| up to 173: the round-5 kernel's fixed 2^10 pre-scale saturated f16 at |w| >= 64;
    the scale now comes from max|w| at upload) and 1e-3 (the lo planes must not go subnormal)."""
    x = hash_normal(f"cin.x.{B}.{H}", (B, 3, H, W))
    w = wmag * hash_uniform(f"cin.w.{Cout}", (Cout, 3, 3, 3), -1, 1) / 27 ** 0.5
    b = 0.1 * hash_uniform(f"cin.b.{Cout}", (Cout,))
    got = hip_conv(x, w, b, tile=17)
    ref64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    mag = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=1)          # sum |x||w| + |b| per output element
    err = (got.double() - ref64).abs()
    # per product: x and w each truncated at 2^-22 relative by the two-term split, the x_lo w_lo term dropped (2^-22): 3 * 2^-22; fp32
    # accumulation of <= 28 terms adds a few 2^-24 each.  2^-20 covers both forms; a dropped cross term would be 2^-11 (500 x the bound)
    c = 2.0 ** -20
    bound = c * mag + 2.0 ** -23 * ref64.abs()
    bad = err > bound
    assert not bad.any(), (f"conv_in kernel: {int(bad.sum())}/{bad.numel()} outside the per-element bound; max err {float(err.max()):.3e}, "
                           f"max err/bound {float((err / bound).max()):.2f}")
    assert_close(got, ref64.float(), what="conv_in kernel (north-star tolerance)", rtol=1e-3, atol=1e-4 * max(1.0, wmag))
    if B > 1:
        alone = hip_conv(x[B - 1:B], w, b, tile=17)
        assert torch.equal(alone[0], got[B - 1]), "conv_in stencil: result depends on the batch"


def test_first_convolution_stencil_kernel_refuses_other_shapes():
    from asyrp_official_amd import _lib
    x, w, b = _mk(1, 32, 64, 16, 3, "cin.bad")           # 32 input channels
    with pytest.raises(_lib.AsyrpError):
        hip_conv(x, w, b, tile=17)


@pytest.mark.parametrize("H,W,Cout,offset", [(32, 32, 128, 0.0), (40, 24, 128, 30.0), (16, 16, 256, 5.0)])
def test_first_convolution_stencil_kernel_statistics(H, W, Cout, offset):
    """GroupNorm partials of the stencil kernel's epilogue (norm1 of the first block) incl. ragged patches and a large mean."""
    from asyrp_official_amd import _lib
    lib = _lib.load()
    B, Cin = 2, 3
    x = hash_normal(f"cinst.x.{H}", (B, Cin, H, W))
    w = hash_uniform(f"cinst.w.{Cout}", (Cout, Cin, 3, 3), -1, 1) / 27 ** 0.5
    b = 0.1 * hash_uniform(f"cinst.b.{Cout}", (Cout,)) + offset
    gam, bet = 1 + 0.1 * hash_uniform("cinst.g", (Cout,)), 0.1 * hash_uniform("cinst.be", (Cout,))
    d = lambda t: t.cuda().contiguous()
    xd, wd, bd, gd, bed = map(d, (x, w, b, gam, bet))
    y = torch.empty((B, Cout, H, W), device="cuda")
    sc, sh = torch.empty((B, Cout), device="cuda"), torch.empty((B, Cout), device="cuda")
    _lib.check(lib.asyrp_op_conv2d_stats(0, _p(xd), Cin, B, H, W, _p(wd), _p(bd), Cout, 3, 17, _p(gd), _p(bed), 1e-6,
                                         _p(y), _p(sc), _p(sh), None))
    torch.cuda.synchronize()
    want_y = F.conv2d(x, w, b, padding=1)
    assert_close(y.cpu(), want_y, what="conv", rtol=1e-5, atol=4e-6 * max(1.0, offset))
    got_gn = y.cpu() * sc.cpu()[:, :, None, None] + sh.cpu()[:, :, None, None]
    want_gn = F.group_norm(want_y.double(), 32, gam.double(), bet.double(), eps=1e-6).float()
    assert_close(got_gn, want_gn, what="fused GN", rtol=1e-3, atol=1e-4)
