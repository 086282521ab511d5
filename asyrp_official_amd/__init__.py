"""asyrp_official_amd — MI355X-native (gfx950) Asyrp DDIM sampling engine.

Host-side mirror of the reference interface for the accelerated path only:
  DDPM                      models/ddpm/diffusion.py:327
  UNetModel, i_DDPM, ...    models/improved_ddpm/{unet,script_util}.py, models/guided_diffusion/{unet,script_util}.py
  denoising_step, ...       utils/diffusion_utils.py:5-109
  run_edit / run_edit_sharded   the loops at diffusion_latent.py:1034-1045 and :503-520
All compute is in asyrp_official_amd/libasyrp_hip.so (C ABI: include/asyrp.h).
"""
from .ddpm import DDPM  # noqa: F401
from .improved_ddpm import UNetModel, create_model, guided_Diffusion, i_DDPM  # noqa: F401
from .diffusion_utils import denoising_step, extract, get_beta_schedule  # noqa: F401
from .engine import AsyrpDeviceError, Engine  # noqa: F401
from .sampler import gather_shards, run_edit, run_edit_sharded, shard_bounds, timestep_seq  # noqa: F401
from . import cache  # noqa: F401  (latent-cache / Δh-checkpoint formats, hs_coeff schedules)

__all__ = ["DDPM", "UNetModel", "create_model", "i_DDPM", "guided_Diffusion", "denoising_step", "extract", "get_beta_schedule", "run_edit", "run_edit_sharded",
           "timestep_seq", "shard_bounds", "gather_shards", "Engine", "AsyrpDeviceError"]
