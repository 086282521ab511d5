"""`DDPM` — drop-in for the reference's models/ddpm/diffusion.py:327 `DDPM(config)` on MI355X.

Same constructor argument (the YAML-derived namespace), same state_dict key names (so pretrained
UNet checkpoints and the shipped checkpoint/*.pth["0"] DeltaBlocks load unmodified), same
`setattr_layers(n)` / `get_temb(t)` / `forward(x, t, index, t_edit, hs_coeff, delta_h,
ignore_timestep, use_mask)` surface and the same 4-tuple result — but the modules below are
parameter HOLDERS only: all arithmetic runs in the hand-written HIP engine (libasyrp_hip.so).
There is no PyTorch/CPU fallback; calling forward with CPU tensors raises.
"""
from . import _lib
from ._base import HipUNet
from .engine import AsyrpDeviceError, ddpm_temb_freqs, make_config  # noqa: F401  (AsyrpDeviceError re-exported)


def _cfg_get(ns, name, default=None):
    return getattr(ns, name) if hasattr(ns, name) else (ns[name] if isinstance(ns, dict) and name in ns else default)


class DDPM(HipUNet):
    def __init__(self, config, max_batch=64, conv_math="f16x3", nominal_batch=0):
        super().__init__()
        self.config = config
        m, d = _cfg_get(config, "model"), _cfg_get(config, "data")
        self.ch = int(_cfg_get(m, "ch"))
        self.out_ch = int(_cfg_get(m, "out_ch"))
        self.ch_mult = tuple(int(v) for v in _cfg_get(m, "ch_mult"))
        self.num_res_blocks = int(_cfg_get(m, "num_res_blocks"))
        self.attn_resolutions = tuple(int(v) for v in _cfg_get(m, "attn_resolutions"))
        self.in_channels = int(_cfg_get(m, "in_channels"))
        self.resolution = int(_cfg_get(d, "image_size"))
        if not _cfg_get(m, "resamp_with_conv", True):
            raise NotImplementedError("resamp_with_conv=False is not used by any reference config")
        self.temb_ch = self.ch * 4
        self.num_resolutions = len(self.ch_mult)
        self._init_params(max_batch, conv_math, nominal_batch)

    def get_temb(self, t):
        """models/ddpm/diffusion.py:464-470: dense1(swish(dense0(get_timestep_embedding(t, ch)))) -> [B, 4*ch]."""
        return self._ready_engine(t).get_temb(t)

    def forward(self, x, t, index=None, t_edit=400, hs_coeff=(1.0, 1.0), delta_h=None, ignore_timestep=False,
                use_mask=False):
        return self._run(x, t, index, t_edit, hs_coeff, delta_h, ignore_timestep, use_mask)

    def _make_cfg(self, n_delta):
        return make_config(family=_lib.FAMILY_DDPM, resolution=self.resolution, in_channels=self.in_channels,
                           out_channels=self.out_ch, ch=self.ch, ch_mult=self.ch_mult,
                           num_res_blocks=self.num_res_blocks, attn_resolutions=self.attn_resolutions,
                           n_delta=n_delta, conv_math=self.conv_math, nominal_batch=self.nominal_batch)

    def _temb_freqs(self):
        return ddpm_temb_freqs(self.ch)
