"""CPU model checks of two host/kernel-side schedules that the GPU tests can only observe through their results:

* the K-step / staging schedule of igemm_f16x3_k32_kernel (csrc/conv_f16x3.hip): two consecutive (chunk, tap) slices per step, two
  halo buffers, the rule that decides when the next chunk is staged -- replayed in Python for every chunk count the engine can
  produce, asserting that a step never reads a chunk that is not resident and that a staging pass never overwrites a buffer a
  step of the same barrier interval still reads;
* the skip-half sharing of dual-decoder steps (engine.hip: conv1_shared / skip_plan): on the oracle's ResnetBlock arithmetic
  (models/ddpm/diffusion.py:151-170 restated in oracle/ddpm.py) the conv1 output over cat(h, skip) equals
  conv(h | straddling skip channels) + the shared partial over the clean skip channels, for BOTH decoder inputs, with the channel
  split the engine computes.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle.ddpm import swish
from oracle.weights import hash_normal, hash_uniform


def k32_schedule(nch, nsc=0):
    """Replay of the loop in igemm_f16x3_k32_kernel: yields per step (reads, staged_after) with reads = [(chunk, tap)] * 2."""
    assert nch % 2 == 0
    nsteps3 = nch * 9 // 2
    c0, t0, staged = 0, 0, 0
    resident = {0: 0}            # buffer -> chunk (prologue stages chunk 0 into buffer 0)
    log = []
    for s in range(nsteps3):
        c1, t1 = c0, t0 + 1
        if t1 == 9:
            t1, c1 = 0, c1 + 1
        reads = [(c0, t0), (c1, t1)]
        for (c, _t) in reads:    # operands of this step: the chunk must be resident in buffer c & 1
            assert resident.get(c & 1) == c, f"step {s}: chunk {c} not resident (nch={nch})"
        t0 += 2
        if t0 >= 9:
            t0 -= 9
            c0 += 1
        need = c0 + 1 if t0 == 8 else c0
        if need > staged and need < nch:
            # staged behind this step's matrix passes, before its closing barrier: the buffer must not be one this step reads
            assert all((c & 1) != (need & 1) for (c, _t) in reads), f"step {s}: staging chunk {need} over a buffer in use"
            resident[need & 1] = need
            staged = need
        log.append(reads)
    flat = [ct for reads in log for ct in reads]
    assert flat == [(c, t) for c in range(nch) for t in range(9)], "every (chunk, tap) slice exactly once, in K order"
    assert staged == nch - 1
    return log


@pytest.mark.parametrize("nch", [2, 4, 6, 8, 10, 16, 24, 32, 48, 64, 96, 128])
def test_k32_kstep_schedule_reads_only_resident_chunks(nch):
    log = k32_schedule(nch)
    assert len(log) == nch * 9 // 2
    # a step straddles a chunk boundary exactly once per pair of chunks (slice 8 of an even chunk with slice 0 of the next)
    straddles = [r for r in log if r[0][0] != r[1][0]]
    assert len(straddles) == nch // 2 and all(r[0] == (r[1][0] - 1, 8) and r[1][1] == 0 for r in straddles)


def skip_plan(c_h, c_skip):
    """engine.hip: skip_plan -- channels of the skip tensor that share a GroupNorm(32) group with h, rounded to whole K=32 steps."""
    cin = c_h + c_skip
    if cin % 32 or c_h % 32:
        return None
    gs = cin // 32
    dirty = ((c_h // gs + 1) * gs - c_h) if c_h % gs else 0
    dirty = (dirty + 31) // 32 * 32
    nclean = c_skip - dirty
    return (dirty, nclean) if nclean >= 32 and nclean % 32 == 0 else None


def test_skip_plan_on_every_decoder_concat_of_the_reference_configs():
    # (h channels, skip channels) of the decoder ResnetBlocks: CelebA-HQ / LSUN DDPM, AFHQ iDDPM, ImageNet ADM
    ddpm = [(512, 512), (512, 256), (256, 256), (256, 128), (128, 128)]
    iddpm = [(512, 512), (512, 384), (384, 384), (384, 256), (256, 256), (256, 128), (128, 128)]
    adm = [(1024, 1024), (1024, 512), (512, 512), (512, 256), (256, 256)]
    for c_h, c_s in ddpm + iddpm + adm:
        plan = skip_plan(c_h, c_s)
        assert plan is not None, (c_h, c_s)
        dirty, nclean = plan
        gs = (c_h + c_s) // 32
        # every clean skip channel lives in a group made of skip channels only
        first_clean_group = (c_h + dirty) // gs
        assert first_clean_group * gs >= c_h and dirty + nclean == c_s and (c_h + dirty) % 32 == 0
    assert skip_plan(64, 32) is None          # 96 channels: the straddling group leaves no whole step of clean channels
    assert skip_plan(48, 48) is None          # h not a multiple of 32


@pytest.mark.parametrize("c_h,c_s,cout", [(64, 64, 64), (128, 64, 96), (96, 160, 64)])
def test_shared_skip_partial_reproduces_conv1_for_both_decoder_inputs(c_h, c_s, cout):
    plan = skip_plan(c_h, c_s)
    assert plan is not None
    dirty, _nclean = plan
    B, H = 2, 8
    tag = f"share.{c_h}.{c_s}"
    skip = hash_normal(tag + ".s", (B, c_s, H, H)) * 1.7 + 0.2
    h_a = hash_normal(tag + ".ha", (B, c_h, H, H))                       # decoder pass 1 (h + delta_h)
    h_b = h_a + 0.3 * hash_normal(tag + ".hb", (B, c_h, H, H))           # decoder pass 2 (h)
    cin = c_h + c_s
    gamma, beta = 1 + 0.1 * hash_uniform(tag + ".g", (cin,)), 0.1 * hash_uniform(tag + ".b", (cin,))
    w = hash_uniform(tag + ".w", (cout, cin, 3, 3), -1, 1) / (cin * 9) ** 0.5
    bias = 0.1 * hash_uniform(tag + ".bias", (cout,))
    act = lambda h: swish(F.group_norm(torch.cat([h, skip], 1).double(), 32, gamma.double(), beta.double(), eps=1e-6))
    c_clean = c_h + dirty
    a_a, a_b = act(h_a), act(h_b)
    # the clean skip channels are normalised identically in both passes ...
    assert torch.equal(a_a[:, c_clean:], a_b[:, c_clean:])
    # ... while the straddling group (if any) is not
    if dirty:
        assert not torch.equal(a_a[:, c_h:c_clean], a_b[:, c_h:c_clean])
    part = F.conv2d(a_a[:, c_clean:], w[:, c_clean:].double(), None, padding=1)            # computed once
    for a in (a_a, a_b):
        want = F.conv2d(a, w.double(), bias.double(), padding=1)
        got = F.conv2d(a[:, :c_clean], w[:, :c_clean].double(), bias.double(), padding=1) + part
        assert torch.allclose(got, want, rtol=1e-12, atol=1e-12)


def test_vendored_class_schedule_tables_and_helpers_match_the_reference():
    """The schedule tables / helper methods scripts read off the vendored GaussianDiffusion: same numbers as the
    reference class builds (models/guided_diffusion/gaussian_diffusion.py:143-176, :205-231, :323-341) for the linear schedule."""
    import sys
    import numpy as np
    import torch
    ref_root = "/root/reference"
    import os
    if not os.path.isdir(ref_root):
        import pytest
        pytest.skip("reference not present")
    sys.path.insert(0, ref_root)
    try:
        from models.guided_diffusion import gaussian_diffusion as rgd
    finally:
        sys.path.remove(ref_root)
    from asyrp_official_amd.gaussian_diffusion import GaussianDiffusion
    betas = np.linspace(1e-4, 0.02, 1000, dtype=np.float64)
    ours = GaussianDiffusion(betas=betas, model_var_type="fixed_large")
    ref = rgd.GaussianDiffusion(betas=betas, model_mean_type=rgd.ModelMeanType.EPSILON, model_var_type=rgd.ModelVarType.FIXED_LARGE,
                                loss_type=rgd.LossType.MSE)
    for name in ("alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                 "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                 "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
        np.testing.assert_allclose(getattr(ours, name), getattr(ref, name), rtol=1e-12, atol=0, err_msg=name)
    g = torch.Generator().manual_seed(3)
    x0, xt, eps = (torch.randn((3, 3, 8, 8), generator=g) for _ in range(3))
    t = torch.tensor([0, 500, 999])
    torch.testing.assert_close(ours._predict_xstart_from_eps(xt, t, eps), ref._predict_xstart_from_eps(xt, t, eps), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ours.q_sample(x0, t, noise=eps), ref.q_sample(x0, t, noise=eps), rtol=1e-6, atol=1e-6)
    for a_, b_ in zip(ours.q_posterior_mean_variance(x0, xt, t), ref.q_posterior_mean_variance(x0, xt, t)):
        torch.testing.assert_close(a_, b_.float(), rtol=1e-6, atol=1e-7)
