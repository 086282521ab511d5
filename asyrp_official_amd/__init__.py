"""asyrp_official_amd — MI355X-native (gfx950) Asyrp DDIM sampling engine.

Host-side mirror of the reference interface for the accelerated path only:
  DDPM                      models/ddpm/diffusion.py:327
  UNetModel, i_DDPM, ...    models/improved_ddpm/{unet,script_util}.py, models/guided_diffusion/{unet,script_util}.py
  denoising_step, ...       utils/diffusion_utils.py:5-109
  run_edit / run_edit_sharded   the loops at diffusion_latent.py:1034-1045 and :503-520
  training.train_step           the DeltaBlock training step of diffusion_latent.py:301-354 (loss and optimiser stay in PyTorch)
  GaussianDiffusion             p_sample / ddim_sample / ddim_reverse_sample of models/guided_diffusion/gaussian_diffusion.py
All compute is in asyrp_official_amd/libasyrp_hip.so (C ABI: include/asyrp.h).
"""
from .ddpm import DDPM  # noqa: F401
from .improved_ddpm import UNetModel, create_model, guided_Diffusion, i_DDPM  # noqa: F401
from .diffusion_utils import denoising_step, extract, get_beta_schedule  # noqa: F401
from .engine import AsyrpDeviceError, Engine  # noqa: F401
from .data_parallel import DataParallel  # noqa: F401  (torch.nn.DataParallel without the per-forward parameter broadcast)
from .sampler import gather_shards, run_edit, run_edit_sharded, shard_bounds, timestep_seq  # noqa: F401
from . import cache  # noqa: F401  (latent-cache / Δh-checkpoint formats, hs_coeff schedules)
from . import training  # noqa: F401  (DeltaBlock training step: autograd node over asyrp_train_forward / _backward)
from .gaussian_diffusion import GaussianDiffusion  # noqa: F401  (vendored p_sample / ddim_sample / ddim_reverse_sample signatures)

__all__ = ["DDPM", "UNetModel", "create_model", "i_DDPM", "guided_Diffusion", "denoising_step", "extract", "get_beta_schedule", "run_edit", "run_edit_sharded",
           "timestep_seq", "shard_bounds", "gather_shards", "Engine", "AsyrpDeviceError", "GaussianDiffusion", "training", "cache", "DataParallel"]
