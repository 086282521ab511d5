"""`UNetModel`, `create_model`, `i_DDPM`, `guided_Diffusion` — drop-ins for the reference's iDDPM / ADM family:
models/improved_ddpm/unet.py:437 + script_util.py:5-109 and models/guided_diffusion/unet.py:437 + script_util.py:10-46,173-234
(the two UNetModel classes are the same network; the reference picks one by dataset, diffusion_latent.py:109-120).

Same constructor signature, same state_dict key names (time_embed.*, input_blocks.N.M.*, middle_block.*, output_blocks.N.M.*,
out.*, layer_i.*), same `setattr_layers` and `forward(x, timesteps, y=None, index=None, t_edit=400, hs_coeff=(1.0, 1.0),
delta_h=None, ignore_timestep=False, use_mask=False)` -> (h, h2, delta_h, middle_h).  Modules are parameter holders; the
arithmetic runs in the HIP engine.  Supported = what the reference's arch dicts use: resblock_updown=True,
use_scale_shift_norm=True, num_head_channels > 0, legacy attention order, fp32.
"""
import math

import torch

from . import _lib
from ._base import HipUNet
from .engine import make_config

AFHQ_DICT = dict(attention_resolutions="16", class_cond=False, dropout=0.0, image_size=256, learn_sigma=True,
                 num_channels=128, num_head_channels=64, num_res_blocks=1, resblock_updown=True, use_fp16=False,
                 use_scale_shift_norm=True, num_heads=4, num_heads_upsample=-1, channel_mult="", use_checkpoint=False,
                 use_new_attention_order=False)                       # improved_ddpm/script_util.py:5-22
IMAGENET_DICT = dict(attention_resolutions="32,16,8", class_cond=True, image_size=256, learn_sigma=True, num_channels=256,
                     num_head_channels=64, num_res_blocks=2, resblock_updown=True, use_fp16=False,
                     use_scale_shift_norm=True, dropout=0.0, num_heads=4, num_heads_upsample=-1, channel_mult="",
                     use_checkpoint=False, use_new_attention_order=False)   # improved_ddpm/script_util.py:25-42
METFACE_DICT = dict(AFHQ_DICT)                                          # guided_diffusion/script_util.py:10-27
CELEBA_HQ_P2_DICT = dict(AFHQ_DICT)                                     # guided_diffusion/script_util.py:29-46
NUM_CLASSES = 1000


def iddpm_temb_freqs(dim, max_period=10000):
    """exp(-ln(max_period) * arange(half) / half) in fp32, the ops of models/improved_ddpm/nn.py:113-116."""
    half = dim // 2
    return torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)


class UNetModel(HipUNet):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False, max_batch=64, conv_math="f16x3", nominal_batch=0):
        super().__init__()
        unsupported = []
        if not resblock_updown: unsupported.append("resblock_updown=False")
        if not use_scale_shift_norm: unsupported.append("use_scale_shift_norm=False")
        if num_head_channels is None or num_head_channels <= 0: unsupported.append("num_head_channels=-1")
        if use_new_attention_order: unsupported.append("use_new_attention_order=True")
        if use_fp16: unsupported.append("use_fp16=True")
        if dims != 2: unsupported.append(f"dims={dims}")
        if any(float(m) != int(m) for m in channel_mult): unsupported.append("fractional channel_mult (image_size=512)")
        if unsupported:
            raise NotImplementedError("not used by any arch dict of the reference and not accelerated: " + ", ".join(unsupported))
        self.image_size, self.in_channels, self.model_channels = int(image_size), int(in_channels), int(model_channels)
        self.out_channels, self.num_res_blocks = int(out_channels), int(num_res_blocks)
        self.attention_resolutions = tuple(int(a) for a in attention_resolutions)   # DOWNSAMPLE RATES, as in the reference
        self.channel_mult = tuple(int(m) for m in channel_mult)
        self.num_classes, self.num_head_channels = num_classes, int(num_head_channels)
        self.dropout, self.conv_resample, self.use_checkpoint = dropout, conv_resample, use_checkpoint
        self.dtype = torch.float32
        self.resolution = self.image_size
        self._init_params(max_batch, conv_math, nominal_batch)

    def forward(self, x, timesteps, y=None, index=None, t_edit=400, hs_coeff=(1.0, 1.0), delta_h=None,
                ignore_timestep=False, use_mask=False):
        # `y` is accepted and ignored, exactly as the reference's forward does (models/improved_ddpm/unet.py:676-688)
        return self._run(x, timesteps, index, t_edit, hs_coeff, delta_h, ignore_timestep, use_mask)

    def _make_cfg(self, n_delta):
        return make_config(family=_lib.FAMILY_IDDPM, resolution=self.image_size, in_channels=self.in_channels,
                           out_channels=self.out_channels, ch=self.model_channels, ch_mult=self.channel_mult,
                           num_res_blocks=self.num_res_blocks,
                           attn_resolutions=tuple(self.image_size // ds for ds in self.attention_resolutions),
                           num_head_channels=self.num_head_channels, n_delta=n_delta, conv_math=self.conv_math,
                           num_classes=int(self.num_classes or 0), nominal_batch=self.nominal_batch)

    def _temb_freqs(self):
        return iddpm_temb_freqs(self.model_channels)


def create_model(image_size, num_channels, num_res_blocks, channel_mult="", learn_sigma=False, class_cond=False,
                 use_checkpoint=False, attention_resolutions="16", num_heads=1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0, resblock_updown=False, use_fp16=False,
                 use_new_attention_order=False, **engine_kw):
    """models/improved_ddpm/script_util.py:45-99 (== models/guided_diffusion/script_util.py:180-234)."""
    if channel_mult == "":
        channel_mult = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}.get(image_size)
        if channel_mult is None:
            raise ValueError(f"unsupported image size: {image_size}")
    else:
        channel_mult = tuple(int(ch_mult) for ch_mult in channel_mult.split(","))
    attention_ds = [image_size // int(res) for res in attention_resolutions.split(",")]
    return UNetModel(image_size=image_size, in_channels=3, model_channels=num_channels,
                     out_channels=(3 if not learn_sigma else 6), num_res_blocks=num_res_blocks,
                     attention_resolutions=tuple(attention_ds), dropout=dropout, channel_mult=channel_mult,
                     num_classes=(NUM_CLASSES if class_cond else None), use_checkpoint=use_checkpoint, use_fp16=use_fp16,
                     num_heads=num_heads, num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
                     use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
                     use_new_attention_order=use_new_attention_order, **engine_kw)


def i_DDPM(dataset_name="AFHQ", **engine_kw):
    """models/improved_ddpm/script_util.py:102-109."""
    if dataset_name in ["AFHQ", "FFHQ"]:
        return create_model(**AFHQ_DICT, **engine_kw)
    if dataset_name == "IMAGENET":
        return create_model(**IMAGENET_DICT, **engine_kw)
    raise NotImplementedError(f"i_DDPM: unknown dataset {dataset_name!r}")


def guided_Diffusion(dataset_name="MetFACE", **engine_kw):
    """models/guided_diffusion/script_util.py:173-177."""
    if dataset_name == "MetFACE":
        return create_model(**METFACE_DICT, **engine_kw)
    if dataset_name == "CelebA_HQ_P2":
        return create_model(**CELEBA_HQ_P2_DICT, **engine_kw)
    raise NotImplementedError(f"guided_Diffusion: unknown dataset {dataset_name!r}")
