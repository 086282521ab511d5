"""the DeltaBlock training step on the engine (diffusion_latent.py:282-354 minus the CLIP network).
Gradients of a loss on (x0_t, xt_next) w.r.t. the DeltaBlock parameters, computed by asyrp_train_backward (transposed
convolutions on the MFMA kernels + GroupNorm/SiLU/attention/upsample backward), against
  (i)  gradients produced by the REFERENCE's own autograd (tests/golden/train_small.npz, make_golden.py run_train_small),
  (ii) autograd through the CPU oracle at the full CelebA-HQ size."""
import pytest
import torch

from conftest import assert_close, load_golden
from oracle import sampler as osamp
from oracle.ddpm import ddpm_forward
from oracle.weights import CELEBA, SMALL, hash_normal
from util_models import hip_model, synthetic

pytestmark = pytest.mark.gpu


def assert_grad_close(got, want, what):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, f"{what}: {tuple(got.shape)} vs {tuple(want.shape)}"
    scale = float(want.abs().max())
    err = (got - want).abs()
    bad = err > (1e-4 * scale + 1e-3 * want.abs())
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} outside rtol 1e-3 / atol 1e-4*max|grad|; max abs err "
                           f"{float(err.max()):.3e}, grad absmax {scale:.3e}")


def _enable_delta_grads(m):
    for p in m.parameters():
        p.requires_grad = False
    for p in m.layer_0.parameters():
        p.requires_grad = True
        p.grad = None


@pytest.mark.parametrize("conv_math", ["f16x3", "f32"])
@pytest.mark.parametrize("tag,ign", [("step", False), ("ignoret", True)])
def test_train_step_gradients_vs_reference_autograd(conv_math, tag, ign):
    from asyrp_official_amd import denoising_step
    g = load_golden("train_small.npz")
    sd = synthetic(SMALL, 2, seed=7)
    m = hip_model(SMALL, sd, 2, conv_math=conv_math)
    _enable_delta_grads(m)
    x = hash_normal("small.x", (2, 3, 32, 32), seed=1).cuda()
    g1 = hash_normal("train.g_x0t", (2, 3, 32, 32), seed=3).cuda()
    g2 = hash_normal("train.g_xtn", (2, 3, 32, 32), seed=4).cuda()
    b = osamp.beta_schedule().cuda()
    one = torch.ones(2, device="cuda")
    kw = dict(models=m, logvars=None, b=b, sampling_type="ddim", eta=0.0, index=0, t_edit=500, hs_coeff=(1.0, 0.8),
              ignore_timestep=ign)
    xn, x0t, dh, mid = denoising_step(x, t=one * 701.0, t_next=one * 675.0, **kw)
    assert x0t.requires_grad and xn.requires_grad and not dh.requires_grad
    assert_close(x0t, g[f"{tag}.x0_t"], what="x0_t")
    assert_close(xn, g[f"{tag}.xt_next"], what="xt_next")
    loss = (x0t * g1).sum() + (xn * g2).sum()
    loss.backward()
    for k, p in m.layer_0.named_parameters():
        if ign and k.startswith("temb_proj."):      # temb=None: the projection is not in the graph, reference autograd leaves None
            assert p.grad is None and not bool(g[f"{tag}.grad.layer_0.{k}"].any()), k
            continue
        assert p.grad is not None, k
        assert_grad_close(p.grad, g[f"{tag}.grad.layer_0.{k}"], f"[{conv_math}/{tag}] d loss / d layer_0.{k}")
    # the training forward runs the inference kernels, except that attention takes the three-launch fp32 form (it keeps
    # the softmax probabilities for the backward pass) instead of the fused f16x3 kernel: equal to summation-order noise
    with torch.no_grad():
        xn2, x0t2, _, _ = denoising_step(x, t=one * 701.0, t_next=one * 675.0, **kw)
    assert_close(xn2, xn.detach(), rtol=1e-4, atol=2e-5, what="training vs inference xt_next")
    assert_close(x0t2, x0t.detach(), rtol=1e-4, atol=2e-5, what="training vs inference x0_t")


def test_sgd_iterations_follow_the_reference_training_loop():
    """Three iterations of the reference's loop shape (zero_grad / step / loss.backward / optim.step, :305-350) with
    torch.optim.SGD on the engine-backed DeltaBlock: every iteration's gradients equal autograd through the CPU oracle
    evaluated at the same (updated) parameters."""
    from asyrp_official_amd import denoising_step
    sd = synthetic(SMALL, 1, seed=11)
    m = hip_model(SMALL, sd, 1)
    _enable_delta_grads(m)
    opt = torch.optim.SGD(list(m.layer_0.parameters()), lr=0.05, weight_decay=0)
    x = hash_normal("train.x", (2, 3, 32, 32), seed=2)
    tgt = hash_normal("train.tgt", (2, 3, 32, 32), seed=5)
    b = osamp.beta_schedule()
    one = torch.ones(2)
    for it, (t, tn) in enumerate(((999, 749), (749, 499), (999, 749))):
        opt.zero_grad()
        _, x0t, _, _ = denoising_step(x.cuda(), t=one.cuda() * t, t_next=one.cuda() * tn, models=m, logvars=None, b=b.cuda(),
                                      sampling_type="ddim", eta=0.0, index=0, t_edit=400, hs_coeff=(1.0, 1.0))
        # (the reference's L1 term, :338, has a sign() gradient: an element of x0_t - target that is within 1e-6 of zero
        # flips between the GPU and the CPU evaluation; a smooth loss keeps the comparison about the backward pass)
        loss = torch.nn.MSELoss()(x0t, tgt.cuda())
        loss.backward()
        # oracle at the current parameters
        cur = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        leaves = {k: cur[k].requires_grad_(True) for k in cur if k.startswith("layer_0.")}
        et, em, _, _ = ddpm_forward({**cur, **leaves}, SMALL, x, one * t, index=0, t_edit=400, hs_coeff=(1.0, 1.0))
        ab = osamp.alpha_bar(b)
        _, o_x0t = osamp.ddim_update(x, et, em, ab[t], ab[tn])
        torch.nn.MSELoss()(o_x0t, tgt).backward()
        for k, p in m.layer_0.named_parameters():
            assert_grad_close(p.grad, leaves["layer_0." + k].grad, f"iteration {it}: layer_0.{k}")
        opt.step()


def test_full_size_celeba_gradients_vs_oracle_autograd():
    """CelebA-HQ 256x256 (the production tiles: 8-wave main tile, fused shortcuts, split-K, 18 decoder ResnetBlocks, 3 attention
    sites), B=1: DeltaBlock gradients vs autograd through the CPU oracle."""
    from asyrp_official_amd import denoising_step
    sd = synthetic(CELEBA, 1, seed=1234)
    m = hip_model(CELEBA, sd, 1, max_batch=1)
    _enable_delta_grads(m)
    x = hash_normal("celeba.x", (1, 3, 256, 256), seed=1234)
    gx = hash_normal("train.g256", (1, 3, 256, 256), seed=6)
    b = osamp.beta_schedule()
    one = torch.ones(1)
    _, x0t, _, _ = denoising_step(x.cuda(), t=one.cuda() * 768.0, t_next=one.cuda() * 743.0, models=m, logvars=None, b=b.cuda(),
                                  sampling_type="ddim", eta=0.0, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    (x0t * gx.cuda()).sum().backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("layer_0.")}
    torch.set_num_threads(16)
    et, em, _, _ = ddpm_forward({**sd, **leaves}, CELEBA, x, one * 768.0, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    ab = osamp.alpha_bar(b)
    _, o_x0t = osamp.ddim_update(x, et, em, ab[768], ab[743])
    (o_x0t * gx).sum().backward()
    for k, p in m.layer_0.named_parameters():
        assert_grad_close(p.grad, leaves["layer_0." + k].grad, f"256x256: layer_0.{k}")


def test_unconsumed_tape_is_released():
    """A differentiable step whose loss is never back-propagated must not pin workspace: the next forward discards the tape."""
    from asyrp_official_amd import denoising_step
    sd = synthetic(SMALL, 1, seed=11)
    m = hip_model(SMALL, sd, 1)
    _enable_delta_grads(m)
    x = hash_normal("train.x", (2, 3, 32, 32), seed=2).cuda()
    b = osamp.beta_schedule().cuda()
    one = torch.ones(2, device="cuda")
    kw = dict(models=m, logvars=None, b=b, sampling_type="ddim", eta=0.0, index=0, t_edit=400, hs_coeff=(1.0, 1.0))
    denoising_step(x, t=one * 999, t_next=one * 749, **kw)
    eng = m._ready_engine(x)
    base = None
    for _ in range(4):
        denoising_step(x, t=one * 999, t_next=one * 749, **kw)
        torch.cuda.synchronize()
        cur = eng.device_bytes()
        assert base is None or cur == base, "workspace grows across abandoned training steps"
        base = cur


@pytest.mark.parametrize("conv_math", ["f16x3", "f32"])
@pytest.mark.parametrize("tag,ign", [("istep", False), ("iignoret", True)])
def test_iddpm_train_step_gradients_vs_reference_autograd(conv_math, tag, ign):
    """The iDDPM / ADM family (models/improved_ddpm/unet.py): learn_sigma output, FiLM ResBlocks incl. the up-sampling ones,
    multi-head legacy attention, the iDDPM DeltaBlock (two GroupNorms, emb_layers) -- gradients vs the reference's autograd."""
    from asyrp_official_amd import denoising_step
    from oracle.iddpm import SMALL_I, iddpm_param_shapes
    from oracle.weights import synthetic_state_dict
    from test_gpu_iddpm import hip_iddpm
    g = load_golden("train_small.npz")
    sd = synthetic_state_dict(iddpm_param_shapes(SMALL_I, n_delta=2), seed=11)
    m = hip_iddpm(SMALL_I, sd, 2, conv_math=conv_math)
    _enable_delta_grads(m)
    x = hash_normal("ismall.x", (2, 3, 32, 32), seed=2).cuda()
    g1 = hash_normal("train.g_x0t", (2, 3, 32, 32), seed=3).cuda()
    g2 = hash_normal("train.g_xtn", (2, 3, 32, 32), seed=4).cuda()
    b = osamp.beta_schedule().cuda()
    one = torch.ones(2, device="cuda")
    xn, x0t, dh, _ = denoising_step(x, t=one * 701.0, t_next=one * 675.0, models=m, logvars=None, b=b, sampling_type="ddim",
                                    eta=0.0, learn_sigma=True, index=0, t_edit=500, hs_coeff=(1.0, 0.8), ignore_timestep=ign)
    assert x0t.requires_grad and tuple(x0t.shape) == (2, 3, 32, 32)
    assert_close(x0t, g[f"{tag}.x0_t"], what="x0_t")
    assert_close(xn, g[f"{tag}.xt_next"], what="xt_next")
    ((x0t * g1).sum() + (xn * g2).sum()).backward()
    for k, p in m.layer_0.named_parameters():
        if ign and k.startswith("emb_layers."):
            assert p.grad is None and not bool(g[f"{tag}.grad.layer_0.{k}"].any()), k
            continue
        assert p.grad is not None, k
        assert_grad_close(p.grad, g[f"{tag}.grad.layer_0.{k}"], f"[iDDPM {conv_math}/{tag}] d loss / d layer_0.{k}")


def test_tape_ids_keep_two_pending_steps_apart():
    """The engine keeps ONE pending step: forward A, forward B, backward(A) must fail loudly instead of back-propagating B's
    activations; backward(B) still works; an abandoned step is dropped when autograd frees its node."""
    from asyrp_official_amd import _lib, denoising_step
    sd = synthetic(SMALL, 1, seed=11)
    m = hip_model(SMALL, sd, 1)
    _enable_delta_grads(m)
    x = hash_normal("train.x", (2, 3, 32, 32), seed=2).cuda()
    b = osamp.beta_schedule().cuda()
    one = torch.ones(2, device="cuda")
    kw = dict(models=m, logvars=None, b=b, sampling_type="ddim", eta=0.0, index=0, t_edit=400, hs_coeff=(1.0, 1.0))
    _, x0_a, _, _ = denoising_step(x, t=one * 999, t_next=one * 749, **kw)
    _, x0_b, _, _ = denoising_step(x * 0.5, t=one * 749, t_next=one * 499, **kw)
    with pytest.raises(_lib.AsyrpError, match="stale tape id"):
        x0_a.sum().backward()
    x0_b.sum().backward()                       # the pending step is B's
    assert all(p.grad is not None for p in m.layer_0.parameters())
    # wrong-shaped gradient and unknown keys are refused before anything is launched
    eng = m._ready_engine(x)
    *_, tid = eng.train_forward(x, 999, 749)
    with pytest.raises(ValueError):
        eng.train_backward(tid, torch.zeros(2, 3, 16, 16, device="cuda"), [("layer_0.conv1.bias", (64,))])
    with pytest.raises(_lib.AsyrpError, match="no gradient for key"):
        eng.train_backward(tid, torch.zeros(2, 3, 32, 32, device="cuda"), [("up.0.block.0.conv1.bias", (32,))])
    # abandoned step: freeing the autograd node releases what the step held, so the inference call that follows can re-use those
    # buffers; while the node is alive they stay pinned and the same call has to allocate more (fresh engines: the pool only grows)
    def bytes_after(release):
        mm = hip_model(SMALL, sd, 1)
        _enable_delta_grads(mm)
        k2 = dict(kw, models=mm)
        out = denoising_step(x, t=one * 999, t_next=one * 749, **k2)
        if release:
            del out
        with torch.no_grad():
            denoising_step(x, t=one * 999, t_next=one * 749, **k2)
        torch.cuda.synchronize()
        return mm._ready_engine(x).device_bytes()
    assert bytes_after(True) < bytes_after(False), "an abandoned training step kept its activations pinned"
