"""B2 + loop parity: denoising_step mirror and the fused inversion+generation loops."""
import pytest
import torch

from conftest import assert_close
from oracle import sampler as osamp
from oracle.weights import SMALL, hash_normal
from util_models import err_stats, hip_model, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    sd = synthetic(SMALL, 2, seed=7)
    return hip_model(SMALL, sd, 2), sd, hash_normal("small.x", (2, 3, 32, 32), seed=1)


def test_denoising_step_mirror(small, golden_small):
    from asyrp_official_amd import denoising_step
    m, _, x = small
    g = golden_small
    b = osamp.beta_schedule().cuda()
    xc = x.cuda()
    one = torch.ones(2, device="cuda")
    kw = dict(models=m, logvars=None, b=b, sampling_type="ddim")
    xn, x0t, dh, mh = denoising_step(xc, t=one * 0.0, t_next=one * 25.0, eta=0, **kw)
    assert dh is None
    assert_close(xn, g["step_inv.xt_next"], what="inv xt_next")
    assert_close(x0t, g["step_inv.x0_t"], what="inv x0_t")
    ek = dict(index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    xn, x0t, dh, _ = denoising_step(xc, t=one * 701.0, t_next=one * 675.0, eta=0.0, **ek, **kw)
    assert_close(xn, g["step_gen.xt_next"], what="gen xt_next")
    assert_close(x0t, g["step_gen.x0_t"], what="gen x0_t")
    assert_close(dh, g["step_gen.delta_h"], what="gen delta_h")
    xn, x0t, _, _ = denoising_step(xc, t=one * 25.0, t_next=one * 0.0, eta=1.0, noise=g["step_eta.noise"].cuda(),
                                   **ek, **kw)
    assert_close(xn, g["step_eta.xt_next"], what="eta xt_next")
    assert_close(x0t, g["step_eta.x0_t"], what="eta x0_t")
    xn, x0t, _, _ = denoising_step(xc, t=one * 0.0, t_next=one * -1.0, eta=0.0, **ek, **kw)
    assert_close(xn, g["step_last.xt_next"], what="last xt_next")
    assert_close(x0t, g["step_last.x0_t"], what="last x0_t")
    xn, _, _, _ = denoising_step(xc, t=one * 701.0, t_next=one * 675.0, eta=0.0, dt_lambda=1.05, dt_end=600, **ek, **kw)
    assert_close(xn, g["step_dt.xt_next"], what="dt_lambda xt_next")


def test_whole_edit_against_reference_fixture(small, golden_small):
    from asyrp_official_amd import run_edit
    m, _, x = small
    g = golden_small
    b = osamp.beta_schedule()
    x_edit, x_T = run_edit(m, x.cuda(), b, n_inv=6, n_gen=6, t_edit=500, t_addnoise=0, want_latent=True)
    print("x_T", err_stats(x_T, g["edit.x_T"]), "x_edit", err_stats(x_edit, g["edit.x_edit"]))
    assert_close(x_T, g["edit.x_T"], what="x_T")
    assert_close(x_edit, g["edit.x_edit"], what="x_edit")


def test_edit_with_noise_tail_against_oracle(small):
    """eta=1 tail (t < t_addnoise) consuming caller-provided noise, 8+8 steps, vs the oracle loops."""
    from asyrp_official_amd import run_edit
    m, sd, x = small
    b = osamp.beta_schedule()
    model = osamp.make_model(sd, SMALL)
    n = 8
    seq = osamp.timestep_seq(n)[0]
    k = sum(1 for t in seq if t < 300)
    noise = torch.stack([hash_normal(f"tail.{i}", (2, 3, 32, 32)) for i in range(k)])
    x_T = osamp.invert(model, x, b, n_inv=n)
    want = osamp.generate(model, x_T, b, n_gen=n, t_edit=500, t_addnoise=300, noises=list(noise))
    got = run_edit(m, x.cuda(), b, n_inv=n, n_gen=n, t_edit=500, t_addnoise=300, noise=noise.cuda())
    print(err_stats(got, want))
    assert_close(got, want, what="x_edit with noise tail")


def test_generation_only_from_xT(small):
    """The north-star's 'identical x_T/seed' variant: skip inversion, start loop B from x_T."""
    from asyrp_official_amd import run_edit
    m, sd, _ = small
    b = osamp.beta_schedule()
    x_T = hash_normal("xT", (2, 3, 32, 32))
    want = osamp.generate(osamp.make_model(sd, SMALL), x_T, b, n_gen=6, t_edit=500)
    got = run_edit(m, x_T.cuda(), b, n_gen=6, t_edit=500, invert=False)
    assert_close(got, want, what="x_edit from x_T")


def test_sharded_equals_unsharded_bitwise(small):
    """Per-rank slices reproduce the unsharded batch bit-for-bit (images are independent)."""
    from asyrp_official_amd import run_edit
    m, _, x = small
    b = osamp.beta_schedule()
    xs = torch.cat([x, hash_normal("more", (1, 3, 32, 32))]).cuda()
    full = run_edit(m, xs, b, n_inv=4, n_gen=4, t_edit=500)
    parts = [run_edit(m, xs[lo:hi].contiguous(), b, n_inv=4, n_gen=4, t_edit=500) for lo, hi in ((0, 2), (2, 3))]
    assert torch.equal(full, torch.cat(parts))
