#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE ITSELF (build container only).

Imports /root/reference's own DDPM + denoising_step (they need only torch/numpy),
loads hash-generated weights (oracle.weights — nothing is stored but the outputs),
and records reference outputs as .npz fixtures next to this script:

  ddpm_small.npz   small DDPM (ch=32, 32x32): forwards, single steps, a 6+6-step edit
  ddpm_celeba.npz  full CelebA-HQ DDPM (256x256, B=1): one single + one dual forward,
                   one Asyrp step, DeltaBlock = hash weights
  slerp_small.npz  both small UNets with an injected delta_h tensor (slerp branch, +/- use_mask)
  config1_celeba_smiling.npz   BASELINE config 1 end to end: CelebA-HQ DDPM + the shipped `smiling` DeltaBlock, B=1,
                   full 39 + 40 steps (t_addnoise = 0 and 167), per-step tensors for teacher-forced parity at full size
  config3_afhq_dog_happy.npz   BASELINE config 3 generation phase: AFHQ iDDPM + the shipped `dog_happy` DeltaBlock
  config4_church_gothic.npz    BASELINE config 4's model: LSUN-church DDPM + the shipped `church_gothic` DeltaBlock, t_edit=370,
                   three teacher-forced steps at full size
  imagenet_adm.npz BASELINE config 5's model at full size: i_DDPM('IMAGENET'), B=1, one dual-decoder forward
  config4_church_full.npz / imagenet_adm_traj.npz / iddpm_afhq_b2.npz   (round 5) config 4 end to end (x_T, x_edit of 39 + 40 steps),
                   a free-running 3 + 4-step trajectory of config 5's model, an iDDPM batch of two different images

Run:  python tests/golden/make_golden.py      (needs /root/reference; ~1 min on 8 cores)
"""
import argparse
import os
import sys
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("ASYRP_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

import compact  # noqa: E402  (storage policy: sampled comparison targets, shared DeltaBlock weights)
from oracle.weights import (CELEBA, SMALL, ddpm_param_shapes, hash_normal, hash_uniform,  # noqa: E402
                            synthetic_state_dict)


def ref_model(cfg, sd, n_delta=1):
    from models.ddpm.diffusion import DDPM
    ns = Namespace(
        model=Namespace(ch=cfg.ch, out_ch=cfg.out_ch, ch_mult=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks,
                        attn_resolutions=list(cfg.attn_resolutions), dropout=0.0, in_channels=cfg.in_channels,
                        resamp_with_conv=True),
        data=Namespace(image_size=cfg.resolution))
    m = DDPM(ns)
    m.setattr_layers(n_delta)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.eval()


def run_small(out):
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    torch.set_num_threads(1)
    cfg = SMALL
    sd = synthetic_state_dict(ddpm_param_shapes(cfg, n_delta=2), seed=7)
    m = ref_model(cfg, sd, n_delta=2)
    B = 2
    x = hash_normal("small.x", (B, 3, 32, 32), seed=1)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    logvars = np.zeros(1000)
    g = {}
    with torch.no_grad():
        t = torch.ones(B) * 701.0
        et, _, _, mh = m(x, t)
        g["fwd_single.et"], g["fwd_single.middle_h"] = et, mh
        et, em, dh, mh = m(x, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        g["fwd_dual.et"], g["fwd_dual.et_mod"], g["fwd_dual.delta_h"], g["fwd_dual.middle_h"] = et, em, dh, mh
        et, em, dh, mh = m(x, t, index=1, t_edit=500, hs_coeff=(0.9, 0.7, 0.5))   # two DeltaBlocks (multi-attr)
        g["fwd_multi.et"], g["fwd_multi.et_mod"], g["fwd_multi.delta_h"] = et, em, dh
        et, em, dh, mh = m(x, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0), ignore_timestep=True)
        g["fwd_ignoret.et_mod"], g["fwd_ignoret.delta_h"] = em, dh
        t2 = torch.ones(B) * 204.0
        et, em, dh, mh = m(x, t2, index=0, t_edit=500, hs_coeff=(1.0, 1.0))       # t < t_edit: h2 = h
        assert dh is None and torch.equal(et, em)
        g["fwd_noedit.et"] = et
        # single steps through the reference's denoising_step
        kw = dict(models=m, logvars=logvars, b=betas, sampling_type="ddim")
        xn, x0t, _, _ = denoising_step(x, t=torch.ones(B) * 0.0, t_next=torch.ones(B) * 25.0, eta=0, **kw)
        g["step_inv.xt_next"], g["step_inv.x0_t"] = xn, x0t
        xn, x0t, dh, mh = denoising_step(x, t=t, t_next=torch.ones(B) * 675.0, eta=0.0, index=0, t_edit=500,
                                         hs_coeff=(1.0, 1.0), **kw)
        g["step_gen.xt_next"], g["step_gen.x0_t"], g["step_gen.delta_h"] = xn, x0t, dh
        torch.manual_seed(99)
        z = torch.randn_like(x)
        torch.manual_seed(99)
        xn, x0t, _, _ = denoising_step(x, t=torch.ones(B) * 25.0, t_next=torch.ones(B) * 0.0, eta=1.0, index=0,
                                       t_edit=500, hs_coeff=(1.0, 1.0), **kw)
        g["step_eta.noise"], g["step_eta.xt_next"], g["step_eta.x0_t"] = z, xn, x0t
        xn, x0t, _, _ = denoising_step(x, t=torch.ones(B) * 0.0, t_next=torch.ones(B) * -1.0, eta=0.0, index=0,
                                       t_edit=500, hs_coeff=(1.0, 1.0), **kw)
        g["step_last.xt_next"], g["step_last.x0_t"] = xn, x0t
        xn, x0t, _, _ = denoising_step(x, t=t, t_next=torch.ones(B) * 675.0, eta=0.0, index=0, t_edit=500,
                                       hs_coeff=(1.0, 1.0), dt_lambda=1.05, dt_end=600, **kw)
        g["step_dt.xt_next"] = xn
        # whole edit, 6 inversion + 6 generation steps (same loop shapes as 40/40)
        n_step, t_0, t_edit = 6, 999, 500
        seq = [int(s + 1e-6) for s in list(np.linspace(0, 1, n_step) * t_0)]
        seq_next = [-1] + seq[:-1]
        xx = x.clone()
        for i, j in zip(seq_next[1:], seq[1:]):
            xx, _, _, _ = denoising_step(xx, t=torch.ones(B) * i, t_next=torch.ones(B) * j, eta=0, **kw)
        g["edit.x_T"] = xx.clone()
        for i, j in zip(reversed(seq), reversed(seq_next)):
            xx, x0t, _, _ = denoising_step(xx, t=torch.ones(B) * i, t_next=torch.ones(B) * j, eta=0.0, index=0,
                                           t_edit=t_edit, hs_coeff=(1.0, 1.0), **kw)
        g["edit.x_edit"] = xx
    g["input.x"] = x
    compact.save(out, {k: v.numpy() for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_celeba(out):
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    torch.set_num_threads(os.cpu_count())
    cfg = CELEBA
    sd = synthetic_state_dict(ddpm_param_shapes(cfg, n_delta=1), seed=1234)
    m = ref_model(cfg, sd, n_delta=1)
    x = hash_normal("celeba.x", (1, 3, 256, 256), seed=1234)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    g = {}
    with torch.no_grad():
        t = torch.ones(1) * 768.0
        et, _, _, mh = m(x, t)
        g["fwd_single.et"], g["fwd_single.middle_h"] = et, mh
        et, em, dh, mh = m(x, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        g["fwd_dual.et"], g["fwd_dual.et_mod"], g["fwd_dual.delta_h"] = et, em, dh
        xn, x0t, _, _ = denoising_step(x, t=t, t_next=torch.ones(1) * 743.0, models=m, logvars=np.zeros(1000),
                                       b=betas, sampling_type="ddim", eta=0.0, index=0, t_edit=500,
                                       hs_coeff=(1.0, 1.0))
        g["step_gen.xt_next"], g["step_gen.x0_t"] = xn, x0t
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def ref_iddpm(cfg, sd, n_delta):
    """The reference's own UNetModel, built with the arguments create_model passes (improved_ddpm/script_util.py:81-99)."""
    from models.improved_ddpm.unet import UNetModel
    m = UNetModel(image_size=cfg.image_size, in_channels=3, model_channels=cfg.num_channels, out_channels=cfg.out_channels,
                  num_res_blocks=cfg.num_res_blocks, attention_resolutions=tuple(cfg.attention_ds), dropout=0.0,
                  channel_mult=cfg.channel_mult, num_classes=(1000 if cfg.class_cond else None), use_checkpoint=False,
                  use_fp16=False, num_heads=4, num_head_channels=cfg.num_head_channels, num_heads_upsample=-1,
                  use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=False)
    m.setattr_layers(n_delta)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return m.eval()


def run_iddpm_small(out):
    """Small iDDPM UNet (32x32, 32 base channels, 16-channel heads): forwards, learn_sigma steps, a 6+6-step edit."""
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    from oracle.iddpm import SMALL_I, iddpm_param_shapes
    torch.set_num_threads(1)
    cfg = SMALL_I
    sd = synthetic_state_dict(iddpm_param_shapes(cfg, n_delta=2), seed=11)
    m = ref_iddpm(cfg, sd, 2)
    B = 2
    x = hash_normal("ismall.x", (B, 3, 32, 32), seed=2)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    g = {}
    kw = dict(models=m, logvars=np.zeros(1000), b=betas, sampling_type="ddim", learn_sigma=True)
    with torch.no_grad():
        t = torch.ones(B) * 701.0
        et, _, _, mh = m(x, t)
        g["fwd_single.et"], g["fwd_single.middle_h"] = et, mh
        et, em, dh, mh = m(x, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        g["fwd_dual.et"], g["fwd_dual.et_mod"], g["fwd_dual.delta_h"], g["fwd_dual.middle_h"] = et, em, dh, mh
        et, em, dh, mh = m(x, t, index=1, t_edit=500, hs_coeff=(0.9, 0.7, 0.5))
        g["fwd_multi.et_mod"], g["fwd_multi.delta_h"] = em, dh
        et, em, dh, mh = m(x, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0), ignore_timestep=True)
        g["fwd_ignoret.et_mod"], g["fwd_ignoret.delta_h"] = em, dh
        et, em, dh, mh = m(x, torch.ones(B) * 204.0, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        assert dh is None and torch.equal(et, em)
        g["fwd_noedit.et"] = et
        xn, x0t, _, _ = denoising_step(x, t=torch.ones(B) * 0.0, t_next=torch.ones(B) * 25.0, eta=0, **kw)
        g["step_inv.xt_next"], g["step_inv.x0_t"] = xn, x0t
        xn, x0t, dh, mh = denoising_step(x, t=t, t_next=torch.ones(B) * 675.0, eta=0.0, index=0, t_edit=500,
                                         hs_coeff=(1.0, 1.0), **kw)
        g["step_gen.xt_next"], g["step_gen.x0_t"], g["step_gen.delta_h"] = xn, x0t, dh
        torch.manual_seed(99)
        z = torch.randn_like(x)
        torch.manual_seed(99)
        xn, x0t, _, _ = denoising_step(x, t=torch.ones(B) * 25.0, t_next=torch.ones(B) * 0.0, eta=1.0, index=0,
                                       t_edit=500, hs_coeff=(1.0, 1.0), **kw)
        g["step_eta.noise"], g["step_eta.xt_next"], g["step_eta.x0_t"] = z, xn, x0t
        n_step, t_0, t_edit = 6, 999, 500
        seq = [int(s + 1e-6) for s in list(np.linspace(0, 1, n_step) * t_0)]
        seq_next = [-1] + seq[:-1]
        xx = x.clone()
        for i, j in zip(seq_next[1:], seq[1:]):
            xx, _, _, _ = denoising_step(xx, t=torch.ones(B) * i, t_next=torch.ones(B) * j, eta=0, **kw)
        g["edit.x_T"] = xx.clone()
        for i, j in zip(reversed(seq), reversed(seq_next)):
            xx, x0t, _, _ = denoising_step(xx, t=torch.ones(B) * i, t_next=torch.ones(B) * j, eta=0.0, index=0,
                                           t_edit=t_edit, hs_coeff=(1.0, 1.0), **kw)
        g["edit.x_edit"] = xx
    g["input.x"] = x
    compact.save(out, {k: v.numpy() for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_iddpm_small2(out):
    """ImageNet-style structure at toy size (2 ResBlocks/level, attention at 16 and 8, class_cond label_emb)."""
    from oracle.iddpm import SMALL_I2, iddpm_param_shapes
    torch.set_num_threads(1)
    cfg = SMALL_I2
    sd = synthetic_state_dict(iddpm_param_shapes(cfg, n_delta=1), seed=13)
    m = ref_iddpm(cfg, sd, 1)
    x = hash_normal("ismall2.x", (2, 3, 32, 32), seed=3)
    g = {}
    with torch.no_grad():
        t = torch.ones(2) * 555.0
        et, em, dh, mh = m(x, t, y=torch.tensor([3, 7]), index=0, t_edit=500, hs_coeff=(1.0, 0.8))
        g["fwd_dual.et"], g["fwd_dual.et_mod"], g["fwd_dual.delta_h"], g["fwd_dual.middle_h"] = et, em, dh, mh
    compact.save(out, {k: v.numpy() for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_afhq(out):
    """Full-size AFHQ-dog iDDPM (i_DDPM('AFHQ'), 256x256, B=1): single + dual forward, one learn_sigma Asyrp step.
    Also asserts that guided_Diffusion('MetFACE') (models/guided_diffusion/unet.py) is the same function."""
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    from oracle.iddpm import AFHQ, iddpm_param_shapes
    torch.set_num_threads(os.cpu_count())
    cfg = AFHQ
    sd = synthetic_state_dict(iddpm_param_shapes(cfg, n_delta=1), seed=4321)
    m = ref_iddpm(cfg, sd, 1)
    x = hash_normal("afhq.x", (1, 3, 256, 256), seed=4321)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    g = {}
    with torch.no_grad():
        t = torch.ones(1) * 768.0
        et, _, _, mh = m(x, t)
        g["fwd_single.et"], g["fwd_single.middle_h"] = et, mh
        et, em, dh, mh = m(x, t, index=0, t_edit=444, hs_coeff=(1.0, 1.0))
        g["fwd_dual.et"], g["fwd_dual.et_mod"], g["fwd_dual.delta_h"] = et, em, dh
        xn, x0t, _, _ = denoising_step(x, t=t, t_next=torch.ones(1) * 743.0, models=m, logvars=np.zeros(1000), b=betas,
                                       sampling_type="ddim", eta=0.0, learn_sigma=True, index=0, t_edit=444,
                                       hs_coeff=(1.0, 1.0))
        g["step_gen.xt_next"], g["step_gen.x0_t"] = xn, x0t
        from models.guided_diffusion.script_util import guided_Diffusion
        mg = guided_Diffusion("MetFACE")
        mg.setattr_layers(1)
        mg.load_state_dict(sd, strict=True)
        et_g, em_g, _, _ = mg.eval()(x, t, index=0, t_edit=444, hs_coeff=(1.0, 1.0))
        assert torch.equal(et_g, et) and torch.equal(em_g, em), "guided_diffusion UNet differs from improved_ddpm UNet"
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_slerp(out):
    """The injected-delta_h branch of both UNet families (models/ddpm/diffusion.py:518-539, improved_ddpm/unet.py:708-731):
    a Delta-h TENSOR is passed to forward / denoising_step (how diffusion_latent.py:516 feeds the global mean Delta-h),
    with and without use_mask.  Same small models / inputs as ddpm_small.npz and iddpm_small.npz."""
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    from oracle.iddpm import SMALL_I, iddpm_param_shapes
    torch.set_num_threads(1)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    B = 2
    g = {}
    dh_in = hash_normal("slerp.delta_h", (B, 64, 8, 8), seed=5)
    g["input.delta_h"] = dh_in
    fams = (("ddpm", ref_model(SMALL, synthetic_state_dict(ddpm_param_shapes(SMALL, n_delta=2), seed=7), n_delta=2),
             hash_normal("small.x", (B, 3, 32, 32), seed=1), False),
            ("iddpm", ref_iddpm(SMALL_I, synthetic_state_dict(iddpm_param_shapes(SMALL_I, n_delta=2), seed=11), 2),
             hash_normal("ismall.x", (B, 3, 32, 32), seed=2), True))
    with torch.no_grad():
        for name, m, x, ls in fams:
            t = torch.ones(B) * 701.0
            for tag, c0, um in (("nomask", 0.7, False), ("mask", 0.7, True), ("nomask_c0", 0.25, False)):
                et, em, dh, mh = m(x, t, index=0, t_edit=500, hs_coeff=(c0, 1.0), delta_h=dh_in, use_mask=um)
                assert dh is dh_in
                g[f"{name}.{tag}.et"], g[f"{name}.{tag}.et_mod"], g[f"{name}.{tag}.middle_h"] = et, em, mh
            et, em, dh, mh = m(x, torch.ones(B) * 204.0, index=0, t_edit=500, hs_coeff=(0.7, 1.0), delta_h=dh_in)
            assert torch.equal(et, em)                                   # below t_edit the tensor is ignored (:541-542)
            xn, x0t, dh, mh = denoising_step(x, t=t, t_next=torch.ones(B) * 675.0, models=m, logvars=np.zeros(1000), b=betas,
                                             sampling_type="ddim", learn_sigma=ls, eta=0.0, index=0, t_edit=500,
                                             hs_coeff=(0.7, 1.0), delta_h=dh_in)
            g[f"{name}.step.xt_next"], g[f"{name}.step.x0_t"] = xn, x0t
    compact.save(out, {k: v.numpy() for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def _seq40():
    seq = [int(s + 1e-6) for s in list(np.linspace(0, 1, 40) * 999)]      # diffusion_latent.py:955-957
    return seq, [-1] + seq[:-1]


def config1_state_dict(tame=1.0):
    """BASELINE.json configs[0]/[1]: CelebA-HQ DDPM with hash base weights (seed 1234) + the SHIPPED `smiling` DeltaBlock
    (checkpoint/smiling_LC_CelebA_HQ_t999_ninv40_ngen40_0.pth["0"], loaded as diffusion_latent.py:674-676 does)."""
    sd = synthetic_state_dict(ddpm_param_shapes(CELEBA, n_delta=1), seed=1234)
    ck = torch.load(os.path.join(REF, "checkpoint", "smiling_LC_CelebA_HQ_t999_ninv40_ngen40_0.pth"), map_location="cpu",
                    weights_only=True)["0"]
    for k, v in ck.items():
        sd["layer_0." + k] = v.float().clone()
    if tame != 1.0:   # see run_config1
        sd["conv_out.weight"] = sd["conv_out.weight"] * tame
        sd["conv_out.bias"] = sd["conv_out.bias"] * tame
    return sd


def run_config1(out, tame=1.0):
    """BASELINE config 1 end to end through the REFERENCE: B=1, full 39 inversion + 40 Asyrp steps, t_edit=500, once with
    t_addnoise=0 and once with t_addnoise=167 consuming stored noise (randn(7,1,3,256,256), manual_seed(4321)).
    Stored: x_T, both x_edit, and (x_t, xt_next, x0_t, delta_h) at chosen steps for teacher-forced GPU parity at full size:
    first / last inversion step, first and last edited generation step (t >= t_edit), first un-edited step, the
    t_next = -1 step, two eta = 1 steps.  x_t of a step is stored only where it is not another stored tensor."""
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    torch.set_num_threads(os.cpu_count())
    sd = config1_state_dict(tame)
    m = ref_model(CELEBA, sd, n_delta=1)
    x0 = hash_uniform("config1.x0", (1, 3, 256, 256), seed=1234)          # an image in [-1, 1)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    seq, seq_next = _seq40()
    kw = dict(models=m, logvars=np.zeros(1000), b=betas, sampling_type="ddim")
    one = torch.ones(1)
    g = {}
    with torch.no_grad():
        x = x0.clone()
        for k, (i, j) in enumerate(zip(seq_next[1:], seq[1:])):
            xin = x
            x, x0t, _, _ = denoising_step(x, t=one * i, t_next=one * j, eta=0, **kw)
            if k == 0:
                g["inv_first.xt_next"], g["inv_first.x0_t"] = x.clone(), x0t.clone()
            if k == len(seq) - 2:
                g["inv_last.x_t"], g["inv_last.x0_t"] = xin.clone(), x0t.clone()
        g["x_T"] = x.clone()
        ek = dict(index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        torch.manual_seed(4321)
        noise = torch.randn(7, 1, 3, 256, 256)
        xs = None
        for i, j in zip(reversed(seq), reversed(seq_next)):
            xin = x
            x, x0t, dh, _ = denoising_step(x, t=one * i, t_next=one * j, eta=0.0, **ek, **kw)
            if i == 999:
                g["gen999.xt_next"], g["gen999.x0_t"], g["gen999.delta_h"] = x.clone(), x0t.clone(), dh.clone()
            if i == 512:
                g["gen512.x_t"], g["gen512.xt_next"], g["gen512.x0_t"] = xin.clone(), x.clone(), x0t.clone()
                g["gen512.delta_h"] = dh.clone()
            if i == 486:      # first step below t_edit: x_t == gen512.xt_next
                assert dh is None
                g["gen486.xt_next"] = x.clone()
            if i == 179:
                xs = x.clone()            # state entering the eta = 1 tail (t = 153 < 167)
            if i == 0:
                g["gen0.x_t"] = xin.clone()
        g["x_edit"] = x.clone()           # == x0_t of the last step (alpha_bar_next = 1)
        # the same generation with the stochastic tail: identical until t = 179, then eta = 1 with the stored noise
        x, k = xs, 0
        for i, j in zip(reversed(seq), reversed(seq_next)):
            if i >= 167:
                continue
            xin = x
            # feed the stored noise through the reference's own torch.randn_like call (utils/diffusion_utils.py:97)
            z = noise[k]
            k += 1
            _orig = torch.randn_like
            torch.randn_like = lambda t, _z=z: _z.clone()
            try:
                x, x0t, _, _ = denoising_step(x, t=one * i, t_next=one * j, eta=1.0, **ek, **kw)
            finally:
                torch.randn_like = _orig
            if i == 153:
                g["eta153.x_t"], g["eta153.xt_next"], g["eta153.x0_t"] = xin.clone(), x.clone(), x0t.clone()
            if i == 0:
                g["eta0.x_t"] = xin.clone()
        assert k == 7
        g["x_edit_noise"] = x.clone()
    g["noise_probe"] = noise[:, 0, 0, 0, :8].clone()     # the test regenerates the noise from the seed and checks this slice
    for k, v in sd.items():                               # the shipped DeltaBlock itself: /root/reference is not on the GPU box
        if k.startswith("layer_0."):
            g["param." + k] = v.clone()
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_config1_tame(out, tame=0.01):
    """Config 1 with conv_out scaled by `tame`: the only change that makes the REFERENCE reproduce itself on a free-running
    39+40 edit (measured with the reference on CPU: a 1-ulp change of x0 moves x_edit by 5.8e-2 at tame=1, 5.8e-2 at 0.25,
    5.2e-4 at 0.05, 5.8e-5 at 0.02), so the engine's free-running x_T / x_edit can be held to rtol 1e-3 / atol 1e-4."""
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    torch.set_num_threads(os.cpu_count())
    sd = config1_state_dict(tame)
    m = ref_model(CELEBA, sd, n_delta=1)
    x = hash_uniform("config1.x0", (1, 3, 256, 256), seed=1234)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    seq, seq_next = _seq40()
    kw = dict(models=m, logvars=np.zeros(1000), b=betas, sampling_type="ddim")
    one = torch.ones(1)
    g = {"tame": torch.tensor(float(tame))}
    with torch.no_grad():
        for i, j in zip(seq_next[1:], seq[1:]):
            x, _, _, _ = denoising_step(x, t=one * i, t_next=one * j, eta=0, **kw)
        g["x_T"] = x.clone()
        for i, j in zip(reversed(seq), reversed(seq_next)):
            x, _, _, _ = denoising_step(x, t=one * i, t_next=one * j, eta=0.0, index=0, t_edit=500, hs_coeff=(1.0, 1.0), **kw)
        g["x_edit"] = x.clone()
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_config3(out):
    """BASELINE config 3 (AFHQ-Dog iDDPM + the SHIPPED `dog_happy` DeltaBlock), generation phase from a seeded x_T through the
    REFERENCE: B=1, 40 Asyrp steps, learn_sigma, t_edit=444 (utils/t_edit_dic.py:5).  Stored: x_edit and steps at t=999,
    t=461 (last edited step), t=435 (first un-edited step), t=0 (t_next=-1)."""
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    from oracle.iddpm import AFHQ, iddpm_param_shapes
    torch.set_num_threads(os.cpu_count())
    sd = synthetic_state_dict(iddpm_param_shapes(AFHQ, n_delta=1), seed=4321)
    ck = torch.load(os.path.join(REF, "checkpoint", "dog_happy_LC_dog_t999_ninv40_ngen40_0.pth"), map_location="cpu",
                    weights_only=True)["0"]
    for k, v in ck.items():
        sd["layer_0." + k] = v.float().clone()
    m = ref_iddpm(AFHQ, sd, 1)
    x = hash_normal("config3.xT", (1, 3, 256, 256), seed=4321)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    seq, seq_next = _seq40()
    kw = dict(models=m, logvars=np.zeros(1000), b=betas, sampling_type="ddim", learn_sigma=True, index=0, t_edit=444,
              hs_coeff=(1.0, 1.0))
    one = torch.ones(1)
    g = {}
    with torch.no_grad():
        for i, j in zip(reversed(seq), reversed(seq_next)):
            xin = x
            x, x0t, dh, _ = denoising_step(x, t=one * i, t_next=one * j, eta=0.0, **kw)
            if i == 999:
                g["gen999.xt_next"], g["gen999.x0_t"], g["gen999.delta_h"] = x.clone(), x0t.clone(), dh.clone()
            if i == 461:
                assert dh is not None
                g["gen461.x_t"], g["gen461.xt_next"], g["gen461.delta_h"] = xin.clone(), x.clone(), dh.clone()
            if i == 435:
                assert dh is None
                g["gen435.xt_next"] = x.clone()
            if i == 0:
                g["gen0.x_t"] = xin.clone()
        g["x_edit"] = x.clone()
    for k, v in sd.items():
        if k.startswith("layer_0."):
            g["param." + k] = v.clone()
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_train_small(out):
    """Gradients of the DeltaBlock through the REFERENCE's own autograd (the training step of diffusion_latent.py:301-354 minus
    the CLIP network): small DDPM, one Asyrp step with grad enabled on layer_0 only (:282-290), a fixed linear functional of
    (x0_t, xt_next) as the loss, so dL/dx0_t and dL/dxt_next are known tensors."""
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    torch.set_num_threads(1)
    cfg = SMALL
    sd = synthetic_state_dict(ddpm_param_shapes(cfg, n_delta=2), seed=7)
    m = ref_model(cfg, sd, n_delta=2)
    B = 2
    x = hash_normal("small.x", (B, 3, 32, 32), seed=1)
    g1 = hash_normal("train.g_x0t", (B, 3, 32, 32), seed=3)
    g2 = hash_normal("train.g_xtn", (B, 3, 32, 32), seed=4)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    g = {}
    for tag, ign in (("step", False), ("ignoret", True)):
        for p_ in m.parameters():
            p_.requires_grad = False
        for p_ in m.layer_0.parameters():
            p_.requires_grad = True
            p_.grad = None
        xn, x0t, _, _ = denoising_step(x, t=torch.ones(B) * 701.0, t_next=torch.ones(B) * 675.0, models=m, logvars=np.zeros(1000),
                                       b=betas, sampling_type="ddim", eta=0.0, index=0, t_edit=500, hs_coeff=(1.0, 0.8),
                                       ignore_timestep=ign)
        loss = (x0t * g1).sum() + (xn * g2).sum()
        loss.backward()
        g[f"{tag}.x0_t"], g[f"{tag}.xt_next"] = x0t.detach().clone(), xn.detach().clone()
        for k, p_ in m.layer_0.named_parameters():
            g[f"{tag}.grad.layer_0.{k}"] = (p_.grad.clone() if p_.grad is not None else torch.zeros_like(p_))
    # the iDDPM family (learn_sigma, FiLM ResBlocks with up-sampling, multi-head legacy attention, iDDPM DeltaBlock)
    from oracle.iddpm import SMALL_I, iddpm_param_shapes
    mi = ref_iddpm(SMALL_I, synthetic_state_dict(iddpm_param_shapes(SMALL_I, n_delta=2), seed=11), 2)
    xi = hash_normal("ismall.x", (B, 3, 32, 32), seed=2)
    for tag, ign in (("istep", False), ("iignoret", True)):
        for p_ in mi.parameters():
            p_.requires_grad = False
        for p_ in mi.layer_0.parameters():
            p_.requires_grad = True
            p_.grad = None
        xn, x0t, _, _ = denoising_step(xi, t=torch.ones(B) * 701.0, t_next=torch.ones(B) * 675.0, models=mi, logvars=np.zeros(1000),
                                       b=betas, sampling_type="ddim", eta=0.0, learn_sigma=True, index=0, t_edit=500,
                                       hs_coeff=(1.0, 0.8), ignore_timestep=ign)
        ((x0t * g1).sum() + (xn * g2).sum()).backward()
        g[f"{tag}.x0_t"], g[f"{tag}.xt_next"] = x0t.detach().clone(), xn.detach().clone()
        for k, p_ in mi.layer_0.named_parameters():
            g[f"{tag}.grad.layer_0.{k}"] = (p_.grad.clone() if p_.grad is not None else torch.zeros_like(p_))
    compact.save(out, {k: v.numpy() for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_vendored_samplers(out):
    """The vendored GaussianDiffusion samplers of the reference (models/guided_diffusion/gaussian_diffusion.py: p_mean_variance,
    p_sample, ddim_sample, ddim_reverse_sample) driven by the reference's small UNets through a first-output adapter (the
    vendored code expects model(x, t) -> tensor).  Deterministic calls only: eta = 0, and p_sample with its noise recorded."""
    from models.guided_diffusion import gaussian_diffusion as gd
    from utils.diffusion_utils import get_beta_schedule
    from oracle.iddpm import SMALL_I, iddpm_param_shapes
    torch.set_num_threads(1)
    betas = get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)
    g = {}
    B = 2
    fams = (("ddpm", ref_model(SMALL, synthetic_state_dict(ddpm_param_shapes(SMALL, n_delta=2), seed=7), n_delta=2),
             hash_normal("small.x", (B, 3, 32, 32), seed=1), gd.ModelVarType.FIXED_LARGE),
            ("iddpm", ref_iddpm(SMALL_I, synthetic_state_dict(iddpm_param_shapes(SMALL_I, n_delta=2), seed=11), 2),
             hash_normal("ismall.x", (B, 3, 32, 32), seed=2), gd.ModelVarType.LEARNED_RANGE))
    with torch.no_grad():
        for name, m, x, vt in fams:
            diff = gd.GaussianDiffusion(betas=betas, model_mean_type=gd.ModelMeanType.EPSILON, model_var_type=vt,
                                        loss_type=gd.LossType.MSE)
            model = lambda x_, t_, **kw: m(x_, t_.float())[0]
            for tv in (701, 0):
                t = torch.full((B,), tv, dtype=torch.long)
                g[f"{name}.t{tv}.model_out"] = model(x, t)
                pm = diff.p_mean_variance(model, x, t, clip_denoised=True)
                for k in ("mean", "variance", "log_variance", "pred_xstart"):
                    g[f"{name}.t{tv}.pmv.{k}"] = pm[k].clone()
                pm = diff.p_mean_variance(model, x, t, clip_denoised=False)
                g[f"{name}.t{tv}.pmv_noclip.mean"] = pm["mean"].clone()
                torch.manual_seed(5)
                z = torch.randn_like(x)
                torch.manual_seed(5)
                ps = diff.p_sample(model, x, t)
                g[f"{name}.t{tv}.p_sample.noise"], g[f"{name}.t{tv}.p_sample.sample"] = z, ps["sample"].clone()
                ds = diff.ddim_sample(model, x, t, clip_denoised=False, eta=0.0)
                g[f"{name}.t{tv}.ddim.sample"], g[f"{name}.t{tv}.ddim.pred_xstart"] = ds["sample"].clone(), ds["pred_xstart"].clone()
                rs = diff.ddim_reverse_sample(model, x, t, clip_denoised=False, eta=0.0)
                g[f"{name}.t{tv}.ddim_reverse.sample"] = rs["sample"].clone()
    compact.save(out, {k: v.numpy() for k, v in g.items()})
    print("wrote", out, sorted(g)[:6], "...", len(g), "tensors")


def imagenet_state_dict():
    """Seeded weights for i_DDPM('IMAGENET') (553.8 M parameters): uniform, fan-in scaled (norm weights around 1), from ONE CPU
    generator stream so the GPU test can regenerate them; the probe stored in the fixture pins that stream."""
    from oracle.iddpm import IMAGENET, iddpm_param_shapes
    gen = torch.Generator().manual_seed(77)
    shapes = iddpm_param_shapes(IMAGENET, n_delta=1)
    sd = {}
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


def run_imagenet(out):
    """BASELINE config 5 at full size through the REFERENCE: i_DDPM('IMAGENET') as the reference's own factory builds it
    (models/improved_ddpm/script_util.py:25-42,105-106), B=1, one dual-decoder forward (index=0, t >= t_edit)."""
    from models.improved_ddpm.script_util import i_DDPM
    torch.set_num_threads(os.cpu_count())
    sd, x = imagenet_state_dict()
    m = i_DDPM("IMAGENET")
    m.setattr_layers(1)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m.eval()
    g = {}
    with torch.no_grad():
        et, em, dh, mh = m(x, torch.ones(1) * 700.0, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
    g["fwd_dual.et"], g["fwd_dual.et_mod"], g["fwd_dual.delta_h"], g["fwd_dual.middle_h"] = et, em, dh, mh
    g["probe.x"] = x[0, 0, 0, :8].clone()
    g["probe.w"] = sd["out.2.weight"].reshape(-1)[:8].clone()
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_config4(out):
    """BASELINE config 4's model through the REFERENCE: the LSUN-church DDPM (configs/church.yml has the CelebA-HQ model block)
    with hash base weights (seed 4004) + the SHIPPED `church_gothic` DeltaBlock
    (checkpoint/church_gothic_LC_church_outdoor_t999_ninv40_ngen40_0.pth["0"]), t_edit = 370 (utils/t_edit_dic.py:3), B=1.
    Three teacher-forced steps of the 40-step generation sequence from seeded x_t: the first step (999 -> 973), the last
    edited step (384 -> 358: 384 >= 370) and the first un-edited one (358 -> 333)."""
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    torch.set_num_threads(os.cpu_count())
    sd = synthetic_state_dict(ddpm_param_shapes(CELEBA, n_delta=1), seed=4004)
    ck = torch.load(os.path.join(REF, "checkpoint", "church_gothic_LC_church_outdoor_t999_ninv40_ngen40_0.pth"),
                    map_location="cpu", weights_only=True)["0"]
    for k, v in ck.items():
        sd["layer_0." + k] = v.float().clone()
    m = ref_model(CELEBA, sd, n_delta=1)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    kw = dict(models=m, logvars=np.zeros(1000), b=betas, sampling_type="ddim", eta=0.0, index=0, t_edit=370, hs_coeff=(1.0, 1.0))
    one = torch.ones(1)
    g = {}
    with torch.no_grad():
        for t, tn in ((999, 973), (384, 358), (358, 333)):
            x = hash_normal(f"config4.x{t}", (1, 3, 256, 256), seed=4004)
            xn, x0t, dh, _ = denoising_step(x, t=one * t, t_next=one * tn, **kw)
            g[f"gen{t}.xt_next"], g[f"gen{t}.x0_t"] = xn.clone(), x0t.clone()
            if t >= 370:
                g[f"gen{t}.delta_h"] = dh.clone()
            else:
                assert dh is None
    for k, v in sd.items():                               # the shipped DeltaBlock itself: /root/reference is not on the GPU box
        if k.startswith("layer_0."):
            g["param." + k] = v.clone()
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_config3_full(out):
    """BASELINE config 3 END TO END through the REFERENCE: AFHQ-Dog iDDPM + the SHIPPED `dog_happy` DeltaBlock,
    B=1, a seeded image -> 39 DDIM inversion steps (diffusion_latent.py:1034-1045, learn_sigma) -> x_T -> 40 Asyrp steps, t_edit=444.
    Stored: x_T, x_edit, and teacher-forced inversion steps at full size (first, a middle one, the last)."""
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    from oracle.iddpm import AFHQ, iddpm_param_shapes
    torch.set_num_threads(os.cpu_count())
    sd = synthetic_state_dict(iddpm_param_shapes(AFHQ, n_delta=1), seed=4321)
    ck = torch.load(os.path.join(REF, "checkpoint", "dog_happy_LC_dog_t999_ninv40_ngen40_0.pth"), map_location="cpu",
                    weights_only=True)["0"]
    for k, v in ck.items():
        sd["layer_0." + k] = v.float().clone()
    m = ref_iddpm(AFHQ, sd, 1)
    x0 = hash_uniform("config3.x0", (1, 3, 256, 256), seed=4321)          # an image in [-1, 1)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    seq, seq_next = _seq40()
    kw = dict(models=m, logvars=np.zeros(1000), b=betas, sampling_type="ddim", learn_sigma=True)
    one = torch.ones(1)
    g = {}
    with torch.no_grad():
        x = x0.clone()
        n = len(seq) - 1
        for k, (i, j) in enumerate(zip(seq_next[1:], seq[1:])):
            xin = x
            x, x0t, _, _ = denoising_step(x, t=one * i, t_next=one * j, eta=0, **kw)
            if k == 0:
                g["inv_first.xt_next"], g["inv_first.x0_t"] = x.clone(), x0t.clone()
            if k == n // 2:
                g["inv_mid.t"] = torch.tensor([float(i), float(j)])
                g["inv_mid.x_t"], g["inv_mid.xt_next"] = xin.clone(), x.clone()
            if k == n - 1:
                g["inv_last.x_t"], g["inv_last.x0_t"] = xin.clone(), x0t.clone()
        g["x_T"] = x.clone()
        for i, j in zip(reversed(seq), reversed(seq_next)):
            x, _, _, _ = denoising_step(x, t=one * i, t_next=one * j, eta=0.0, index=0, t_edit=444, hs_coeff=(1.0, 1.0), **kw)
        g["x_edit"] = x.clone()
    for k, v in sd.items():
        if k.startswith("layer_0."):
            g["param." + k] = v.clone()
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_imagenet_step(out):
    """BASELINE config 5, one teacher-forced dual-decoder STEP at full size through the REFERENCE:
    i_DDPM('IMAGENET') (1024-channel bottleneck), B=1, utils/diffusion_utils.py denoising_step with learn_sigma=True
    (the 6-channel head is split, :47-51) and the DDIM update, t = 700 -> 674 with index=0, t_edit=500."""
    from models.improved_ddpm.script_util import i_DDPM
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    torch.set_num_threads(os.cpu_count())
    sd, x = imagenet_state_dict()
    m = i_DDPM("IMAGENET")
    m.setattr_layers(1)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m.eval()
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    g = {}
    with torch.no_grad():
        xn, x0t, dh, mh = denoising_step(x, t=torch.ones(1) * 700.0, t_next=torch.ones(1) * 674.0, models=m, logvars=np.zeros(1000),
                                         b=betas, sampling_type="ddim", eta=0.0, learn_sigma=True, index=0, t_edit=500,
                                         hs_coeff=(1.0, 1.0))
    g["step.xt_next"], g["step.x0_t"], g["step.delta_h"] = xn, x0t, dh
    g["probe.x"] = x[0, 0, 0, :8].clone()
    g["probe.w"] = sd["out.2.weight"].reshape(-1)[:8].clone()
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_config1_b2(out):
    """A BATCH pinned to the reference directly: BASELINE config 1's model (CelebA-HQ DDPM + the shipped
    `smiling` DeltaBlock), B = 2 with two DIFFERENT images, teacher-forced steps executed by the REFERENCE on the whole batch:
    the first inversion step, a middle inversion step, an edited generation step (dual decoder) and an un-edited one."""
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    torch.set_num_threads(os.cpu_count())
    sd = config1_state_dict()
    m = ref_model(CELEBA, sd, n_delta=1)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    kw = dict(models=m, logvars=np.zeros(1000), b=betas, sampling_type="ddim", eta=0.0)
    two = torch.ones(2)
    g = {}
    with torch.no_grad():
        x0 = torch.cat([hash_uniform("config1b2.x0a", (1, 3, 256, 256), seed=11), hash_uniform("config1b2.x0b", (1, 3, 256, 256), seed=12)])
        xn, x0t, _, _ = denoising_step(x0, t=two * 0, t_next=two * 25, **kw)
        g["inv0.xt_next"], g["inv0.x0_t"] = xn.clone(), x0t.clone()
        xm = torch.cat([hash_normal("config1b2.xma", (1, 3, 256, 256), seed=13), hash_normal("config1b2.xmb", (1, 3, 256, 256), seed=14)])
        xn, _, _, _ = denoising_step(xm, t=two * 512, t_next=two * 537, **kw)
        g["inv512.xt_next"] = xn.clone()
        xn, x0t, dh, _ = denoising_step(xm, t=two * 768, t_next=two * 742, index=0, t_edit=500, hs_coeff=(1.0, 1.0), **kw)
        g["gen768.xt_next"], g["gen768.x0_t"], g["gen768.delta_h"] = xn.clone(), x0t.clone(), dh.clone()
        xn, _, dh, _ = denoising_step(xm, t=two * 307, t_next=two * 281, index=0, t_edit=500, hs_coeff=(1.0, 1.0), **kw)
        assert dh is None
        g["gen307.xt_next"] = xn.clone()
    for k, v in sd.items():
        if k.startswith("layer_0."):
            g["param." + k] = v.clone()
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_config4_full(out):
    """BASELINE config 4's model END TO END through the REFERENCE: the LSUN-church DDPM + the SHIPPED `church_gothic`
    DeltaBlock, B=1, a seeded image -> 39 DDIM inversion steps (diffusion_latent.py:1034-1045) -> x_T -> 40 Asyrp steps with t_edit=370
    (utils/t_edit_dic.py:3; diffusion_latent.py:503-520).  Stored: x_T and x_edit only (the DeltaBlock is in config4_church_gothic.npz,
    the image and the base weights are hash-generated)."""
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    torch.set_num_threads(os.cpu_count())
    sd = synthetic_state_dict(ddpm_param_shapes(CELEBA, n_delta=1), seed=4004)
    ck = torch.load(os.path.join(REF, "checkpoint", "church_gothic_LC_church_outdoor_t999_ninv40_ngen40_0.pth"),
                    map_location="cpu", weights_only=True)["0"]
    for k, v in ck.items():
        sd["layer_0." + k] = v.float().clone()
    m = ref_model(CELEBA, sd, n_delta=1)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    kw = dict(models=m, logvars=np.zeros(1000), b=betas, sampling_type="ddim")
    seq, seq_next = _seq40()
    one = torch.ones(1)
    g = {}
    with torch.no_grad():
        x = hash_uniform("config4.x0", (1, 3, 256, 256), seed=4004)
        for i, j in zip(seq_next[1:], seq[1:]):
            x, _, _, _ = denoising_step(x, t=one * i, t_next=one * j, eta=0, **kw)
        g["x_T"] = x.clone()
        for i, j in zip(reversed(seq), reversed(seq_next)):
            x, _, _, _ = denoising_step(x, t=one * i, t_next=one * j, eta=0.0, index=0, t_edit=370, hs_coeff=(1.0, 1.0), **kw)
        g["x_edit"] = x.clone()
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_imagenet_traj(out):
    """BASELINE config 5, a short FREE-RUNNING trajectory through the REFERENCE: i_DDPM('IMAGENET') (553.8 M
    parameters), B=1, n_inv = n_gen = 4 on the reference's own time grid (np.linspace(0, 1, 4) * 999 = 0, 333, 666, 999): 3 DDIM
    inversion steps with learn_sigma, then 4 Asyrp steps with index=0, t_edit=500 (two dual-decoder steps, two single-decoder steps,
    the last one to t_next = -1).  Stored: x_T, x_edit and the delta_h of the first edited step."""
    from models.improved_ddpm.script_util import i_DDPM
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    torch.set_num_threads(os.cpu_count())
    sd, _ = imagenet_state_dict()
    m = i_DDPM("IMAGENET")
    m.setattr_layers(1)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m.eval()
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    kw = dict(models=m, logvars=np.zeros(1000), b=betas, sampling_type="ddim", learn_sigma=True)
    seq = [int(s + 1e-6) for s in list(np.linspace(0, 1, 4) * 999)]      # diffusion_latent.py:955-957 with n_inv = 4
    seq_next = [-1] + seq[:-1]
    one = torch.ones(1)
    g = {"seq": torch.tensor(seq, dtype=torch.float32)}
    with torch.no_grad():
        x = hash_uniform("imagenet.traj.x0", (1, 3, 256, 256), seed=77)
        for i, j in zip(seq_next[1:], seq[1:]):
            x, _, _, _ = denoising_step(x, t=one * i, t_next=one * j, eta=0, **kw)
        g["x_T"] = x.clone()
        first = True
        for i, j in zip(reversed(seq), reversed(seq_next)):
            x, _, dh, _ = denoising_step(x, t=one * i, t_next=one * j, eta=0.0, index=0, t_edit=500, hs_coeff=(1.0, 1.0), **kw)
            if first:
                g["gen999.delta_h"], g["gen999.xt_next"] = dh.clone(), x.clone()
                first = False
        g["x_edit"] = x.clone()
    g["probe.w"] = sd["out.2.weight"].reshape(-1)[:8].clone()
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_afhq_b2(out):
    """An iDDPM BATCH pinned to the reference directly (models/improved_ddpm/unet.py:676-752): AFHQ-Dog iDDPM + the
    shipped `dog_happy` DeltaBlock, B = 2 with two DIFFERENT images, teacher-forced steps executed by the REFERENCE on the whole batch:
    an inversion step with learn_sigma and an edited (dual-decoder) generation step."""
    from utils.diffusion_utils import denoising_step, get_beta_schedule
    from oracle.iddpm import AFHQ, iddpm_param_shapes
    torch.set_num_threads(os.cpu_count())
    sd = synthetic_state_dict(iddpm_param_shapes(AFHQ, n_delta=1), seed=4321)
    ck = torch.load(os.path.join(REF, "checkpoint", "dog_happy_LC_dog_t999_ninv40_ngen40_0.pth"), map_location="cpu",
                    weights_only=True)["0"]
    for k, v in ck.items():
        sd["layer_0." + k] = v.float().clone()
    m = ref_iddpm(AFHQ, sd, 1)
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    kw = dict(models=m, logvars=np.zeros(1000), b=betas, sampling_type="ddim", eta=0.0, learn_sigma=True)
    two = torch.ones(2)
    g = {}
    with torch.no_grad():
        x0 = torch.cat([hash_uniform("afhqb2.x0a", (1, 3, 256, 256), seed=21), hash_uniform("afhqb2.x0b", (1, 3, 256, 256), seed=22)])
        xn, _, _, _ = denoising_step(x0, t=two * 0, t_next=two * 25, **kw)
        g["inv0.xt_next"] = xn.clone()
        xm = torch.cat([hash_normal("afhqb2.xma", (1, 3, 256, 256), seed=23), hash_normal("afhqb2.xmb", (1, 3, 256, 256), seed=24)])
        xn, _, dh, _ = denoising_step(xm, t=two * 768, t_next=two * 742, index=0, t_edit=444, hs_coeff=(1.0, 1.0), **kw)
        g["gen768.xt_next"], g["gen768.delta_h"] = xn.clone(), dh.clone()
    compact.save(out, {k: v.numpy().astype(np.float32) for k, v in g.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in g.items()})


def run_checkpoint_keys(out):
    """Key names / shapes of the shipped DeltaBlock checkpoints (one per UNet family) -> delta_checkpoint_keys.json."""
    import json
    res = {}
    for f in ("smiling_LC_CelebA_HQ_t999_ninv40_ngen40_0.pth", "dog_happy_LC_dog_t999_ninv40_ngen40_0.pth"):
        sd = torch.load(os.path.join(REF, "checkpoint", f), map_location="cpu", weights_only=True)["0"]
        res[f] = {k: list(v.shape) for k, v in sd.items()}
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["small", "celeba", "keys", "iddpm_small", "iddpm_small2", "afhq", "slerp", "config1", "config1_tame", "config3", "train", "samplers", "imagenet", "config4", "config3_full", "imagenet_step", "config1_b2", "config4_full", "imagenet_traj", "afhq_b2"], default=None)
    a = ap.parse_args()
    if a.only in (None, "keys"):
        run_checkpoint_keys(os.path.join(HERE, "delta_checkpoint_keys.json"))
    if a.only in (None, "slerp"):
        run_slerp(os.path.join(HERE, "slerp_small.npz"))
    if a.only in (None, "iddpm_small"):
        run_iddpm_small(os.path.join(HERE, "iddpm_small.npz"))
    if a.only in (None, "iddpm_small2"):
        run_iddpm_small2(os.path.join(HERE, "iddpm_small2.npz"))
    if a.only in (None, "afhq"):
        run_afhq(os.path.join(HERE, "iddpm_afhq.npz"))
    if a.only in (None, "small"):
        run_small(os.path.join(HERE, "ddpm_small.npz"))
    if a.only in (None, "celeba"):
        run_celeba(os.path.join(HERE, "ddpm_celeba.npz"))
    if a.only in (None, "config1"):
        run_config1(os.path.join(HERE, "config1_celeba_smiling.npz"))
    if a.only in (None, "samplers"):
        run_vendored_samplers(os.path.join(HERE, "vendored_samplers_small.npz"))
    if a.only in (None, "train"):
        run_train_small(os.path.join(HERE, "train_small.npz"))
    if a.only in (None, "config1_tame"):
        run_config1_tame(os.path.join(HERE, "config1_celeba_tame.npz"))
    if a.only in (None, "config3"):
        run_config3(os.path.join(HERE, "config3_afhq_dog_happy.npz"))
    if a.only in (None, "imagenet"):
        run_imagenet(os.path.join(HERE, "imagenet_adm.npz"))
    if a.only in (None, "config4"):
        run_config4(os.path.join(HERE, "config4_church_gothic.npz"))
    if a.only in (None, "config1_b2"):
        run_config1_b2(os.path.join(HERE, "config1_b2_celeba_smiling.npz"))
    if a.only in (None, "imagenet_step"):
        run_imagenet_step(os.path.join(HERE, "imagenet_adm_step.npz"))
    if a.only in (None, "config3_full"):
        run_config3_full(os.path.join(HERE, "config3_afhq_full.npz"))
    if a.only in (None, "config4_full"):
        run_config4_full(os.path.join(HERE, "config4_church_full.npz"))
    if a.only in (None, "afhq_b2"):
        run_afhq_b2(os.path.join(HERE, "iddpm_afhq_b2.npz"))
    if a.only in (None, "imagenet_traj"):
        run_imagenet_traj(os.path.join(HERE, "imagenet_adm_traj.npz"))
