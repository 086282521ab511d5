"""Training step of the DeltaBlock on the HIP engine (reference: diffusion_latent.py:282-354).

The reference trains `layer_0` with
    xt_next, x0_t, _, _ = denoising_step(xt_next.detach(), t=t, t_next=t_next, models=model, ..., index=0, t_edit=..., hs_coeff=...)
    loss = l1_w * L1(x0_t, x0_t_origin) * cosine + clip_w * clip_direction_loss(x0, src, x0_t, trg)
    loss.backward(); optim_ft.step()
with `requires_grad` switched on for the DeltaBlock parameters only (:282-290).  Here the same lines keep working: when
gradients are enabled and a DeltaBlock parameter requires grad, `asyrp_official_amd.denoising_step` routes the step through
`AsyrpTrainStep`, an autograd node whose backward runs the engine's decoder-#2 backward pass (asyrp_train_backward) and hands
autograd the gradients of the DeltaBlock parameters.  The loss (CLIP, L1, id) and the optimiser stay in PyTorch.
"""
import weakref

import torch

from .engine import alphas_cumprod_from_betas


def delta_block_params(model, index=0):
    """[(state_dict key, parameter)] of layer_{index}, in module order."""
    layer = getattr(model, f"layer_{index}")
    return [(f"layer_{index}.{k}", p) for k, p in layer.named_parameters()]


def wants_training(model, index, apply_edit):
    if not (torch.is_grad_enabled() and index is not None and apply_edit):
        return False
    return any(p.requires_grad for _, p in delta_block_params(model, index))


TIMESTEP_PROJECTION = ("temb_proj.", "emb_layers.")     # DDPM / iDDPM DeltaBlock sub-modules fed by the timestep embedding


class AsyrpTrainStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, xt, t, t_next, hs_coeff, ignore_timestep, learn_sigma, *params):
        eng = model._ready_engine(xt)                    # uploads whatever the optimiser changed since the last step
        xt_next, x0_t, dh, mid, tape_id = eng.train_forward(xt, t, t_next, hs_coeff=hs_coeff, ignore_timestep=ignore_timestep,
                                                            learn_sigma=learn_sigma)
        ctx.out_channels = eng.out_channels
        ab = alphas_cumprod_from_betas(model._betas)
        at = float(ab[t])
        at_next = 1.0 if t_next < 0 else float(ab[t_next])
        ctx.eng, ctx.tape_id = eng, tape_id
        # a step whose loss is never back-propagated must not pin its activations: when autograd frees the node, the pending
        # step is dropped (a no-op if the backward consumed it or a later step replaced it)
        weakref.finalize(ctx, eng.train_discard, tape_id)
        named = [(k, tuple(p.shape)) for k, p in delta_block_params(model, 0)]
        # ignore_timestep: the DeltaBlock runs with temb=None (models/ddpm/diffusion.py:253-254), its timestep projection takes
        # no part in the graph and reference autograd leaves those .grad = None
        ctx.skipped = [ignore_timestep and any(s in k for s in TIMESTEP_PROJECTION) for k, _ in named]
        ctx.named_shapes = [ks for ks, skip in zip(named, ctx.skipped) if not skip]
        # x0_t = (xt - et_mod*sqrt(1-at))/sqrt(at);  xt_next = sqrt(at_next)*x0_t + sqrt(1-at_next)*et   (utils/diffusion_utils.py:85-92)
        ctx.k_x0 = -((1.0 - at) ** 0.5) / (at ** 0.5)
        ctx.k_xn = at_next ** 0.5
        ctx.mark_non_differentiable(dh, mid)
        return xt_next, x0_t, dh, mid

    @staticmethod
    def backward(ctx, g_xn, g_x0, _g_dh, _g_mid):
        g = None
        if g_x0 is not None:
            g = g_x0
        if g_xn is not None:
            g = ctx.k_xn * g_xn if g is None else g + ctx.k_xn * g_xn
        if g is None:
            ctx.eng.train_discard(ctx.tape_id)
            return (None,) * 7 + (None,) * len(ctx.skipped)
        d_em = (ctx.k_x0 * g).contiguous()
        if ctx.out_channels != d_em.shape[1]:            # learn_sigma: eps = the first 3 of 6 output channels (diffusion_utils.py:47-51)
            pad = torch.zeros((d_em.shape[0], ctx.out_channels - d_em.shape[1]) + tuple(d_em.shape[2:]), device=d_em.device)
            d_em = torch.cat([d_em, pad], dim=1).contiguous()
        it = iter(ctx.eng.train_backward(ctx.tape_id, d_em, ctx.named_shapes))
        return (None,) * 7 + tuple(None if skip else next(it) for skip in ctx.skipped)


def train_step(model, xt, t, t_next, *, hs_coeff=(1.0, 1.0), ignore_timestep=False, learn_sigma=False):
    """One differentiable Asyrp step: (xt_next, x0_t, delta_h, middle_h) with autograd edges to the DeltaBlock parameters."""
    params = [p for _, p in delta_block_params(model, 0)]
    return AsyrpTrainStep.apply(model, xt, int(t), int(t_next), tuple(float(v) for v in hs_coeff), bool(ignore_timestep),
                                bool(learn_sigma), *params)
