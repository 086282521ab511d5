#!/usr/bin/env python
"""bench.py — edited images/sec of the Asyrp hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch: DDIM inversion (n_inv-1 = 39 UNet evaluations)
followed by Asyrp generation (n_gen = 40 evaluations, second decoder for t >= t_edit) of B images of
256x256 through the CelebA-HQ DDPM UNet (configs[1] of BASELINE.json: batch 32 on one MI355X).
Inputs (x0, weights) are resident in HBM before the timed region; synthetic data, seeded random-init
weights of the reference architecture (no checkpoints offline).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
      N > 1 without a launcher: bench.py re-executes itself under torch.distributed.run with N ranks
      (one process per GPU, RCCL), and fails loudly when fewer than N GPUs are visible.
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

Prints ONE JSON line on rank 0 (see README/DESIGN for the fields); weak scaling: every rank edits its
own B images with no data-path collective, then the final images are all-gathered once (RCCL).
"""
import argparse
import json
import os
import socket
import sys
import time
from argparse import Namespace

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, dense matrix peaks: v_mfma_f32_32x32x2_f32 157.3 TFLOP/s; f16/bf16 MFMA ~2500 TFLOP/s
F32_MFMA_PEAK_TFLOPS = 157.3
F16_MFMA_PEAK_TFLOPS = 2500.0
HBM_PEAK_TBS = 8.0
# measured on this chip with UNet-like operands, bare MFMA stream (scripts/calib/mfma_energy.hip -> profiles/r02s_calib_mfma_energy.txt):
# the power envelope caps v_mfma_f32_32x32x16_f16 at 1.62-1.66 PFLOP/s and v_mfma_f32_16x16x32_f16 (the main tile's instruction
# since r02t) at 1.85-1.89 PFLOP/s; 2.44-2.46 with zero operands
F16_MFMA_MEASURED_RANDOM_TFLOPS = {"16x16x32": 1870.0, "32x32x16": 1657.0}
T_EDIT, T_0, N_INV, N_GEN = 500, 999, 40, 40
CHECK_DUAL, CHECK_INV = (999, 973), (0, 25)     # (t, t_next) of the two steps cross-checked against the CPU side

CELEBA = dict(ch=128, out_ch=3, ch_mult=[1, 1, 2, 2, 4, 4], num_res_blocks=2, attn_resolutions=[16],
              in_channels=3, resolution=256)   # /root/reference/configs/celeba.yml:13-25


def celeba_namespace():
    c = CELEBA
    return Namespace(model=Namespace(ch=c["ch"], out_ch=c["out_ch"], ch_mult=c["ch_mult"],
                                     num_res_blocks=c["num_res_blocks"], attn_resolutions=c["attn_resolutions"],
                                     dropout=0.0, in_channels=c["in_channels"], resamp_with_conv=True),
                     data=Namespace(image_size=c["resolution"]))


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N=1 only): the reference itself when it is importable here, else the oracle (its restatement)
# ---------------------------------------------------------------------------------------------------------------------
def _reference_root():
    ref = os.environ.get("ASYRP_REFERENCE", "/root/reference")
    return ref if os.path.isfile(os.path.join(ref, "utils", "diffusion_utils.py")) else None


def _cpu_step_fn(model_cpu_sd, betas, family, learn_sigma):
    """(kind, step) with step(x, t, t_next, **kw) -> (xt_next, x0_t, delta_h, middle_h) on the CPU."""
    import numpy as np
    ref = _reference_root()
    if ref:
        sys.path.insert(0, ref)
        try:
            from utils.diffusion_utils import denoising_step as ref_step
            if family == "ddpm":
                from models.ddpm.diffusion import DDPM as RefDDPM
                m = RefDDPM(celeba_namespace())
            else:
                from models.improved_ddpm.script_util import i_DDPM as ref_iddpm
                m = ref_iddpm("AFHQ" if family == "afhq" else "IMAGENET")
            m.setattr_layers(1)
            m.load_state_dict(model_cpu_sd, strict=True)
            m.eval()

            def step(x, t, t_next, **kw):
                with torch.no_grad():
                    return ref_step(x, t=t, t_next=t_next, models=m, logvars=np.zeros(1000), b=betas, sampling_type="ddim",
                                    learn_sigma=learn_sigma, **kw)
            return "reference", step
        except Exception as e:     # an unimportable reference is not an error of the bench: fall back to the port, say so
            print(f"[bench] reference at {ref} not usable ({type(e).__name__}: {e}); timing the oracle port", file=sys.stderr)
        finally:
            sys.path.remove(ref)
    from oracle import sampler as osamp
    from oracle.weights import DDPMConfig
    if family == "ddpm":
        cfg = DDPMConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in CELEBA.items()})
        model = osamp.make_model(model_cpu_sd, cfg)
    else:
        from oracle import iddpm as oi
        model = oi.make_model(model_cpu_sd, oi.AFHQ if family == "afhq" else oi.IMAGENET)

    def step(x, t, t_next, **kw):
        return osamp.denoising_step(x, t, t_next, model=model, b=betas, learn_sigma=learn_sigma, **kw)
    return "port", step


def cpu_baseline(model_cpu_sd, betas, family="ddpm", learn_sigma=False, x_check=None):
    """The reference's own modules (kind="reference") when /root/reference (or $ASYRP_REFERENCE) is importable on this
    host, else the oracle restatement (kind="port"), timed on a bounded sample: B=1, 4 inversion steps + 4 Asyrp steps
    (dual decoder, as the reference executes them), extrapolated linearly to the 39 + 40 steps of one edit.
    Thread count: the best of {16, 32, 64} threads on one warm forward each (one thread per visible core is far slower on
    the 256-core GPU hosts: 102 s per forward vs 0.93 s at 16 threads, measured r02a).  Also returns the CPU results of one dual-decoder Asyrp step and of the first inversion step on `x_check` for the parity check."""
    kind, step = _cpu_step_fn(model_cpu_sd, betas, family, learn_sigma)
    avail = len(os.sched_getaffinity(0))
    g = torch.Generator().manual_seed(1234)
    x = 2 * torch.rand((1, 3, 256, 256), generator=g) - 1
    one = torch.ones(1)
    tried = {}
    step(x, one * 0.0, one * 25.0, eta=0)           # page in / warm caches
    for n in sorted({min(n, avail) for n in (16, 32, 64)}):
        torch.set_num_threads(n)
        t0 = time.perf_counter()
        step(x, one * 0.0, one * 25.0, eta=0)
        tried[n] = time.perf_counter() - t0
    cores = min(tried, key=tried.get)
    torch.set_num_threads(cores)
    inv_pairs = ((0, 25), (25, 51), (51, 76), (76, 102))
    gen_pairs = ((999, 973), (973, 947), (947, 922), (922, 896))
    t0 = time.perf_counter()
    for (i, j) in inv_pairs:
        x, _, _, _ = step(x, one * float(i), one * float(j), eta=0)
    t_inv = (time.perf_counter() - t0) / len(inv_pairs)
    t0 = time.perf_counter()
    for (i, j) in gen_pairs:
        x, _, _, _ = step(x, one * float(i), one * float(j), eta=0.0, index=0, t_edit=T_EDIT, hs_coeff=(1.0, 1.0))
    t_gen = (time.perf_counter() - t0) / len(gen_pairs)
    per_image = (N_INV - 1) * t_inv + N_GEN * t_gen
    check = None
    if x_check is not None:
        # the CPU side of bench.py's parity_check: a dual-decoder Asyrp step (t >= t_edit, DeltaBlock active, second decoder) and
        # the first inversion step, on the images handed in (rows 0 and B-1 of the GPU batch)
        ones = torch.ones(x_check.shape[0])
        check = {"dual": step(x_check, ones * float(CHECK_DUAL[0]), ones * float(CHECK_DUAL[1]), eta=0.0, index=0, t_edit=T_EDIT,
                              hs_coeff=(1.0, 1.0)),
                 "inversion": step(x_check, ones * float(CHECK_INV[0]), ones * float(CHECK_INV[1]), eta=0)}
    calib = None
    if kind == "port":
        # the port timed against the reference's own modules on ONE host (scripts/cpu_port_vs_reference.py, build container): how far
        # the oracle's rate is from the reference's on the same sample, and whether their outputs differ
        try:
            with open(os.path.join(ROOT, "profiles", "cpu_port_vs_reference_same_host.json")) as f:
                c = json.load(f)
            calib = {"port_over_reference_rate_same_host": c["port_over_reference_rate"],
                     "outputs_bit_identical": c["outputs_bit_identical"],
                     "estimated_reference_value": (1.0 / per_image) / c["port_over_reference_rate"],
                     "source": "profiles/cpu_port_vs_reference_same_host.json (%d threads, build container)" % c["host"]["threads"]}
        except (OSError, KeyError, ValueError):
            calib = None
    res = {"value": 1.0 / per_image, "unit": "images/s", "cores": cores, "kind": kind,
           "sample": f"B=1: {len(inv_pairs)} inversion + {len(gen_pairs)} dual-decoder Asyrp steps timed ({t_inv:.2f} s, "
                     f"{t_gen:.2f} s per step), extrapolated to {N_INV - 1}+{N_GEN} steps; threads chosen by one warm forward each: "
                     + ", ".join(f"{n}: {s:.2f} s" for n, s in tried.items()) + f" (host has {avail} visible cores)"}
    if calib:
        res["port_vs_reference"] = calib
    return res, check


# ---------------------------------------------------------------------------------------------------------------------
BENCH_METRIC = "edited images/sec, CelebA-HQ 256^2 40-step Asyrp, 1/2/4/8 GPU"   # BASELINE.json `metric`


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run` with N ranks on this node."""
    backend = os.environ.get("ASYRP_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        sys.exit(f"[bench] --gpus {n} requested but only {have} GPU(s) visible: refusing to report a {n}-GPU number")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["ASYRP_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(cmd[0], cmd, env)


def paste_traffic(res, tpath, lib_path, dt, steps):
    """HBM traffic per kernel family into a result line.  PMC counters cannot be read inline; the committed rocprofv3 --pmc passes
    of this command (scripts/gpu.sh traffic -> profiles/traffic_families_<config>_b<B>_<math>.json) are reported per launch
    -- but only when they were taken on the library that is loaded now (sha256 stamp): a stale measurement is refused, never pasted.
    dt = seconds of the timed region, steps = its step count (kernel_families carries each family's share of the region and its
    launches per step)."""
    import hashlib
    if not os.path.exists(tpath):
        return
    tr = json.load(open(tpath))
    sha = hashlib.sha256(open(lib_path, "rb").read()).hexdigest()
    if tr.get("library_sha256") != sha:
        res["roofline"]["traffic_note"] = (f"{os.path.basename(tpath)} was measured on another build of the library "
                                           f"(sha256 {tr.get('library_sha256', '?')[:12]} != {sha[:12]}): not reported")
        return
    fams_t = tr["families"]
    hit = fams_t.get(res["roofline"]["kernel"])
    if hit:
        res["roofline"]["traffic"] = hit["hbm_bytes_per_launch"]
        res["roofline"]["traffic_note"] = (f"HBM bytes per launch = 1024 x (2 x FETCH_SIZE + WRITE_SIZE), rocprofv3 --pmc passes "
                                           f"over every launch of one edit on this library build ({os.path.basename(tpath)}); "
                                           "compare with algorithmic_bytes_per_launch")
    for fm in res.get("kernel_families", []):
        key = fm["kernel"].split(" (")[0]
        ht = fams_t.get(key)
        if "attn_planes_kernel (T=" in fm["kernel"]:   # the counter file has one row per key-tile count of the kernel (T = 256 -> 2, 64 -> 1)
            T = int(fm["kernel"].split("(T=")[1].rstrip(")"))
            ht = fams_t.get("asyrp::attn_planes_kernel<%d>" % ((T // 16 + 7) // 8)) or ht
        if ht and fm["launches_per_step"]:
            sec_per_launch = fm["share_of_step"] * dt / steps / fm["launches_per_step"]
            fm["counter_bytes_per_launch"] = ht["hbm_bytes_per_launch"]
            fm["counter_GBps"] = ht["hbm_bytes_per_launch"] / sec_per_launch / 1e9
            fm["counter_frac_of_hbm_peak"] = fm["counter_GBps"] / (HBM_PEAK_TBS * 1e3)


def bench_config(a):
    """(family, learn_sigma, images per GPU, workload string) of `--config`: which BASELINE.json configuration a line is quoted on.
    One function for the real run and the plumbing dry run, so an 8-GPU line cannot be quoted on the wrong configuration."""
    family = {"celeba": "ddpm", "church": "ddpm", "afhq": "afhq", "imagenet": "imagenet"}[a.config]
    B = a.batch or {"celeba": 32, "church": 32, "afhq": 64, "imagenet": 16}[a.config]
    name = {"celeba": "CelebA-HQ DDPM", "church": "LSUN-Church DDPM", "afhq": "AFHQ-Dog iDDPM",
            "imagenet": "ImageNet ADM (improved_ddpm UNet, 256 base ch)"}[a.config]
    baseline = {"celeba": "BASELINE.json configs[1]", "church": "BASELINE.json configs[3] (batch 256 = 8 x 32 per GPU)",
                "afhq": "BASELINE.json configs[2]", "imagenet": "BASELINE.json configs[4] (batch 128 = 8 x 16 per GPU)"}[a.config]
    workload = (f"{name} 256x256, batch={B}/GPU, ninv={N_INV} (39 UNet evals) + ngen={N_GEN} "
                f"Asyrp (t_edit={T_EDIT}: 20 dual-decoder + 20 single-decoder evals), 1 DeltaBlock")
    return family, family != "ddpm", B, workload, baseline


def other_config_sample(name, dev, betas, conv_math, reuse=None):
    """ONE timed edit of another BASELINE configuration (outside the headline's timed region; a 1-edit sample, not a benchmark):
    images/s, the dominant kernel's roofline fraction from the same HIP-event record, and the bitwise batch-invariance flag (the
    last image edited alone == its row of the batch, full 39+40 steps).  `reuse` = (model, engine) when the configuration runs the
    headline's own UNet (configs/church.yml has configs/celeba.yml's model block)."""
    from asyrp_official_amd import i_DDPM, run_edit
    cfg = argparse.Namespace(config=name, batch=0)
    family, learn_sigma, B, workload, baseline_cfg = bench_config(cfg)
    t_build = time.perf_counter()
    if reuse is not None:
        model, eng = reuse
    else:
        torch.manual_seed(1234)
        model = i_DDPM("AFHQ" if family == "afhq" else "IMAGENET", max_batch=B, conv_math=conv_math)
        model.setattr_layers(1)
        model = model.to(dev).eval()
        eng = model.engine(dev)
    model.set_schedule(betas)
    x0 = (2 * torch.rand((B, 3, 256, 256), generator=torch.Generator().manual_seed(4321)) - 1).to(dev)
    kw = dict(t_0=T_0, t_edit=T_EDIT, t_addnoise=0, index=0, hs_coeff=(1.0, 1.0), learn_sigma=learn_sigma)
    run_edit(model, x0, betas, n_inv=3, n_gen=2, **kw)       # untimed: workspace allocation, every kernel of both decoder passes
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    eng.profile_read()
    eng.profile_enable(True)
    t0 = time.perf_counter()
    out = run_edit(model, x0, betas, n_inv=N_INV, n_gen=N_GEN, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.profile_enable(False)
    prof = eng.profile_read()
    alone = run_edit(model, x0[B - 1:].contiguous(), betas, n_inv=N_INV, n_gen=N_GEN, **kw)
    res = {"workload": workload, "baseline_config": baseline_cfg, "batch_per_gpu": B, "sample": "1 edit, timed once (after an untimed 2+2-step pass that allocates the workspace)",
           "images_per_s": B / dt, "ms_per_edit": 1e3 * dt, "finite": bool(torch.isfinite(out).all()),
           "batch_invariance_bitwise": bool(torch.equal(alone[0], out[B - 1])), "setup_s": t_build}
    if prof and prof["launches"]:
        ach = prof["flops"] / (prof["ms"] * 1e-3) / 1e12
        peak = F16_MFMA_PEAK_TFLOPS / (1.0 if conv_math == "f16" else 3.0)
        res["dominant_kernel"] = {"kernel": prof["kernel"], "tflops": ach, "frac_of_mfma_bound": ach / peak,
                                  "share_of_edit": prof["ms"] * 1e-3 / dt, "launches": prof["launches"]}
    if reuse is None:
        model._drop_engine()
        del model, eng
    del x0, out, alone
    torch.cuda.empty_cache()
    return res


def engine_device_index(local_rank, backend, ndev):
    """The device a rank builds its engine on: its LOCAL_RANK under RCCL (one process per GPU); ranks share devices only in the gloo
    dry run on a box with fewer GPUs than ranks."""
    return local_rank if backend == "nccl" else local_rank % max(ndev, 1)


def _setup_ranks(a, need_gpu):
    """Launcher contract: `python bench.py --gpus N` re-executes itself under torch.distributed.run (one rank per GPU); a rank reads
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment and joins the process group (nccl = RCCL; gloo = dry run)."""
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        _self_launch(a.gpus)            # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.exit(f"[bench] --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks: they must agree")
    # ASYRP_BENCH_BACKEND=gloo is a DRY-RUN knob for a box with fewer GPUs than ranks (ranks share devices, the final
    # gather is staged through host memory): it exercises the launcher / barrier / max-over-ranks code, not RCCL
    backend = os.environ.get("ASYRP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count() if need_gpu else 0
    if need_gpu and backend == "nccl" and ndev < world:
        sys.exit(f"[bench] {world} ranks but only {ndev} GPU(s) visible (one process per GPU)")
    dev_index = engine_device_index(local_rank, backend, ndev) if need_gpu else -1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))   # RCCL
        else:
            dist.init_process_group(backend=backend)
    return world, rank, local_rank, backend, dev_index


def _timed_steps(one_step, a, world, backend, dev, sync, between=None):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + device sync on both sides; returns (seconds = MAX over ranks,
    the last step's outputs)."""
    for _ in range(a.warmup):
        one_step()
    if between is not None:
        between()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(a.steps):
        out = one_step()
    sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, out


def _plumbing_dry_run(a):
    """The N-rank skeleton of this file without the engine (see --plumbing-dry-run): the real launcher, rendezvous, seeds, timing
    brackets, shard -> all-gather and JSON line, with `edit(image) = 0.5 * image + mean(image)` on CPU tensors.  Rank 0 rebuilds
    every rank's input from its seed and checks that the gathered batch equals the unsharded computation bit for bit."""
    world, rank, local_rank, backend, _ = _setup_ranks(a, need_gpu=False)
    if world > 1 and backend == "nccl":
        sys.exit("[bench] --plumbing-dry-run runs on CPU tensors: set ASYRP_BENCH_BACKEND=gloo")
    from asyrp_official_amd.sampler import gather_shards
    B = a.batch or 2
    edit = lambda x: 0.5 * x + x.mean(dim=(1, 2, 3), keepdim=True)   # noqa: E731  (any per-image function)
    seed = 1234 + rank
    x0 = 2 * torch.rand((B, 3, 16, 16), generator=torch.Generator().manual_seed(seed)) - 1

    def one_step():
        local = edit(x0)
        return local, (gather_shards(local, B * world) if world > 1 else local)
    dt, (_, full) = _timed_steps(one_step, a, world, backend, "cpu", sync=lambda: None)
    seeds = [seed]
    # the device index every rank WOULD build its engine on in the real (RCCL) run, and the configuration the line would be quoted on
    devs = [engine_device_index(local_rank, "nccl", world)]
    family, learn_sigma, B_real, workload, baseline_cfg = bench_config(argparse.Namespace(config=a.config, batch=0))
    if world > 1:
        got = [None] * world
        dist.all_gather_object(got, (seed, devs[0], int(os.environ.get("LOCAL_RANK", "0"))))
        seeds = [g_[0] for g_ in got]
        devs = [g_[1] for g_ in got]
        assert all(g_[1] == g_[2] for g_ in got), "a rank would build its engine on a device other than its LOCAL_RANK"
    if rank == 0:
        want = torch.cat([edit(2 * torch.rand((B, 3, 16, 16), generator=torch.Generator().manual_seed(sd_)) - 1) for sd_ in seeds])
        res = {"metric": BENCH_METRIC, "value": None, "unit": "images/s", "n_gpus": a.gpus, "world_size": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": 1e3 * dt / max(a.steps, 1), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dry_run": True, "data": "plumbing dry run: CPU tensors, stand-in per-image function, no engine",
               "backend": backend, "rank_seeds": seeds, "images_per_rank": B, "gathered_shape": list(full.shape),
               "engine_device_index_per_rank": devs,
               "config": {"workload": workload, "baseline_config": baseline_cfg, "batch_per_gpu": B_real, "learn_sigma": learn_sigma,
                          "parallelism": f"dp{world} (batch sharded, one all-gather of x_edit)", "world_size": world},
               "gathered_equals_unsharded": bool(torch.equal(full, want))}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default: the config's: 32 / 64 / 16)")
    ap.add_argument("--config", choices=["celeba", "church", "afhq", "imagenet"], default="celeba",
                    help="celeba = BASELINE.json configs[1] (the metric's config, default); church = configs[3] per GPU "
                         "(same DDPM UNet, batch 256 = 8 x 32); afhq = configs[2] (iDDPM, batch 64); imagenet = "
                         "configs[4] per GPU (ADM, batch 128 = 8 x 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket conv launches with HIP events")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the one-edit samples of BASELINE configs 2, 3, 4 that the default 1-GPU celeba line carries in `other_configs`")
    ap.add_argument("--plumbing-dry-run", action="store_true",
                    help="NO engine, NO GPU: run only the N-rank skeleton of this file (self-launch, rendezvous, per-rank seeds, barrier + "
                         "max-over-ranks timing, shard -> all-gather, the JSON line) on CPU tensors with a stand-in per-image function; "
                         "use with ASYRP_BENCH_BACKEND=gloo.  The line says so (`dry_run`) and carries no throughput claim.")
    ap.add_argument("--nominal-batch", type=int, default=0,
                    help="batch class of the engine (asyrp_config.nominal_batch): 0 = kernels priced at 32 images per GPU (default, the "
                         "line of record), 1 = the small class for single-image serving (use with --batch 1)")
    ap.add_argument("--conv-math", choices=["f16x3", "f32", "f16"], default="f16x3",
                    help="f16x3: 3 x f16 MFMA per product, fp32-equivalent (default, the line of record); f32: fp32-input MFMA; "
                         "f16: the FAST mode - one f16 MFMA per product, not fp32-equivalent, reported as a separate line whose "
                         "dtype says so and whose parity_check carries the measured error instead of gating on the parity tolerance")
    a = ap.parse_args()

    if a.plumbing_dry_run:
        return _plumbing_dry_run(a)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    world, rank, local_rank, backend, dev_index = _setup_ranks(a, need_gpu=True)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from asyrp_official_amd import DDPM, i_DDPM, run_edit
    from asyrp_official_amd.diffusion_utils import get_beta_schedule
    from asyrp_official_amd.sampler import gather_shards

    family, learn_sigma, B, workload, baseline_cfg = bench_config(a)
    torch.manual_seed(1234)                     # main.py:301 default seed
    # small multi-rank runs re-edit the WHOLE gathered batch on rank 0 for gather_check: size the engine for it up front (growing
    # max_batch later would re-create the engine under the handles taken below)
    max_b = B * world if (world > 1 and B * world <= 16) else B
    if family == "ddpm":
        model = DDPM(celeba_namespace(), max_batch=max_b, conv_math=a.conv_math, nominal_batch=a.nominal_batch)   # configs/church.yml has the same model block
    else:
        model = i_DDPM("AFHQ" if family == "afhq" else "IMAGENET", max_batch=max_b, conv_math=a.conv_math, nominal_batch=a.nominal_batch)
    model.setattr_layers(1)                     # get_h_num = 1
    cpu_sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    g = torch.Generator().manual_seed(1234 + rank)
    x0_cpu = 2 * torch.rand((B, 3, 256, 256), generator=g) - 1
    x0 = x0_cpu.to(dev)
    eng = model.engine(dev)
    model.set_schedule(betas)
    edit_kw = dict(n_inv=N_INV, n_gen=N_GEN, t_0=T_0, t_edit=T_EDIT, t_addnoise=0, index=0, hs_coeff=(1.0, 1.0),
                   learn_sigma=learn_sigma)

    def one_step():
        local = run_edit(model, x0, betas, **edit_kw)
        full = local
        if world > 1:   # the one collective of the path
            full = gather_shards(local, B * world) if backend == "nccl" else gather_shards(local.cpu(), B * world).to(dev)
        return local, full

    def _enable_profile():
        if not a.no_kernel_events:
            torch.cuda.synchronize()
            eng.profile_read()
            eng.profile_enable(True)
    dt, (out_local, out) = _timed_steps(one_step, a, world, backend, dev, sync=torch.cuda.synchronize, between=_enable_profile)
    prof, table = None, []
    if not a.no_kernel_events:
        eng.profile_enable(False)
        table = eng.profile_table()
        prof = eng.profile_read()
    assert torch.isfinite(out).all(), "non-finite output"
    assert out.shape[0] == B * world

    # ---- evidence that the N ranks did N different shards and that the one collective delivered them (outside the timed region) ----
    gather_check = None
    if world > 1:
        import hashlib
        sha = lambda t_: hashlib.sha256(t_.detach().cpu().contiguous().numpy().tobytes()).hexdigest()   # noqa: E731
        mine = (rank, 1234 + rank, dev_index, os.getpid(), sha(out_local), [sha(out[r * B:(r + 1) * B]) for r in range(world)])
        got = [None] * world
        dist.all_gather_object(got, mine)
        if rank == 0:
            gather_check = {"world_size": world, "rank_seeds": [g_[1] for g_ in got], "engine_device_index_per_rank": [g_[2] for g_ in got],
                            "distinct_processes": len({g_[3] for g_ in got}) == world,
                            # every rank's own result is the slice every OTHER rank received for it
                            "gathered_slices_equal_rank_results_sha256": all(g_[5][r] == got[r][4] for g_ in got for r in range(world)),
                            "shards_differ": len({g_[4] for g_ in got}) == world}
            if B * world <= 16:
                # small runs only: rank 0 rebuilds every rank's input from its seed and edits the whole batch on ITS engine in one call
                xs = torch.cat([2 * torch.rand((B, 3, 256, 256), generator=torch.Generator().manual_seed(sd_)) - 1
                                for sd_ in gather_check["rank_seeds"]]).to(dev)
                whole = run_edit(model, xs, betas, **edit_kw)
                gather_check["gathered_equals_unsharded_bitwise"] = bool(torch.equal(whole, out))
                del xs, whole

    # ---- parity at the benchmarked configuration (outside the timed region) ----------------------------------------
    # (1) batch invariance: images 0 and B-1 edited ALONE (B=1) must equal their rows of the batched result bit for bit
    # (2) (N=1, with the CPU baseline) a dual-decoder Asyrp step with t >= t_edit (DeltaBlock + second decoder + skip sharing) and
    #     the first inversion step, images 0 and B-1 computed inside the full batch, all four outputs vs the CPU reference/oracle
    parity = None
    if not a.no_parity_check:
        diffs = {}
        for i in sorted({0, B - 1}):
            alone = run_edit(model, x0[i:i + 1].contiguous(), betas, **edit_kw)
            diffs[i] = float((alone[0] - out_local[i]).abs().max())
        parity = {"batch_invariance_bitwise": all(v == 0.0 for v in diffs.values()),
                  "images_checked_alone_vs_in_batch": sorted(diffs), "max_abs_diff": max(diffs.values()),
                  "what": f"run_edit of image i alone (B=1) == row i of the B={B} x_edit, full {N_INV - 1}+{N_GEN} steps"}

    # seconds per step of the two loops separately, outside the timed region: 3 fused steps of each kind
    def phase_ms(t, t_next, **kw):
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        for _ in range(3):
            eng.ddim_step(x0, t, t_next, learn_sigma=learn_sigma, **kw)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0_) / 3
    phases = {"inversion_step": phase_ms(486, 512),
              "generation_step_t>=t_edit(dual decoder)": phase_ms(742, 717, index=0, apply_edit=True, hs_coeff=(1.0, 1.0)),
              "generation_step_t<t_edit": phase_ms(256, 230, index=0, apply_edit=False, hs_coeff=(1.0, 1.0))}
    # GPU side of the CPU cross-check, computed INSIDE the full batch: rows 0 and B-1 of a dual-decoder step and of an inversion step
    chk_rows = sorted({0, B - 1})
    gpu_check = None
    if rank == 0 and not a.no_parity_check:
        pick = lambda outs: [o[chk_rows].cpu() if o is not None else None for o in outs]   # noqa: E731
        gpu_check = {"dual": pick(eng.ddim_step(x0, CHECK_DUAL[0], CHECK_DUAL[1], learn_sigma=learn_sigma, index=0, apply_edit=True,
                                                hs_coeff=(1.0, 1.0))),
                     "inversion": pick(eng.ddim_step(x0, CHECK_INV[0], CHECK_INV[1], learn_sigma=learn_sigma))}

    rc = 0
    NPROD = 1.0 if a.conv_math == "f16" else 3.0       # f16 matrix products issued per algorithmic product
    fast = a.conv_math == "f16"
    if rank == 0:
        images = B * world * a.steps
        res = {
            "metric": BENCH_METRIC,
            "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f16x3": "f32 (conv products on f16 MFMA as exact two-term splits, fp32 accumulate)", "f32": "f32",
                      "f16": "f16 (FAST MODE: one f16 MFMA product per term, fp32 accumulate, fp32 activations in HBM; not "
                             "fp32-equivalent - see parity_check for its measured error)"}[a.conv_math], "data": "synthetic (seeded U[-1,1) images, seeded random-init weights)",
            "config": {"workload": workload, "baseline_config": baseline_cfg,
                       "batch_per_gpu": B, "parallelism": f"dp{world} (batch sharded, one all-gather of x_edit)",
                       "launcher": "self (bench.py re-executed under torch.distributed.run)"
                       if os.environ.get("ASYRP_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if world > 1 else "single process"),
                       "collective_backend": (backend if world > 1 else None),
                       # proof that the collective library saw N ranks (a SCALE record is judged on this)
                       "world_size": world,
                       "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version())
                                        if world > 1 and backend == "nccl" else None),
                       # A/B switches read by the library from the environment: a non-default kernel choice can never be
                       # benchmarked silently
                       # (round 5: the PRODUCT library reads no environment; the switches act only when the profiling build is loaded
                       #  in its place with ASYRP_LIBRARY=bench, and `library` says which one produced this line)
                       "library": "profiling build (ASYRP_LIBRARY=bench: A/B switches honoured)" if os.environ.get("ASYRP_LIBRARY") == "bench" else "product",
                       "switches": {k: os.environ.get(k, "default") for k in ("ASYRP_MAIN_TILE", "ASYRP_SKIP_SHARE", "ASYRP_XCD_MAP", "ASYRP_POLYPHASE", "ASYRP_ATTN", "ASYRP_QUAD8", "ASYRP_GEMM1X1", "ASYRP_SPLITK16", "ASYRP_SPLITK32", "ASYRP_CONV_IN", "ASYRP_CONV_IN_MFMA", "ASYRP_CONV_OUT6")}
                       if os.environ.get("ASYRP_LIBRARY") == "bench" else "not read by the product library",
                       "conv_math": a.conv_math, "nominal_batch": a.nominal_batch or 32},
            "phase_ms_per_step": phases,
            # generation only (x_T given, e.g. --load_random_noise): derived from the per-step times above
            "generation_only_images_per_s": B * world / (1e-3 * (20 * phases["generation_step_t>=t_edit(dual decoder)"] +
                                                                  20 * phases["generation_step_t<t_edit"])),
        }
        if prof and prof["launches"]:
            ach = prof["flops"] / (prof["ms"] * 1e-3) / 1e12
            if prof["family"] in ("f16x3", "attention"):
                # the kernel issues NPROD f16 matrix products per algorithmic product (3 in the fp32-equivalent two-term split,
                # 1 in the fast mode): its ceiling for ALGORITHMIC flops is the dense f16 MFMA peak / NPROD
                peak = F16_MFMA_PEAK_TFLOPS / NPROD
                basis = ("dense f16 MFMA 2500 TFLOP/s / 3 matrix instructions per fp32-equivalent product "
                         "(two-term f16 operand split)" if NPROD == 3 else "dense f16 MFMA 2500 TFLOP/s (one product per term)")
            else:
                peak, basis = F32_MFMA_PEAK_TFLOPS, "dense fp32 MFMA (v_mfma_f32_32x32x2_f32)"
            res["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": None, "peak_basis": basis,
                               "x_fp32_mfma_peak": ach / F32_MFMA_PEAK_TFLOPS, "kernel": prof["kernel"],
                               "launches": prof["launches"], "avg_launch_ms": prof["ms"] / prof["launches"],
                               "flops_per_launch": prof["flops"] / prof["launches"],
                               "algorithmic_bytes_per_launch": prof["bytes"] / prof["launches"],
                               "frac_of_measured_mfma_ceiling": (
                                   ach / (F16_MFMA_MEASURED_RANDOM_TFLOPS["16x16x32" if "k32" in prof["kernel"] else "32x32x16"] / NPROD)
                                   if prof["family"] == "f16x3" else None),
                               "all_gemm_tflops": prof["all_flops"] / (prof["all_ms"] * 1e-3) / 1e12,
                               "gemm_time_share_of_step": prof["all_ms"] * 1e-3 / dt}
        # per-kernel-family table (both bounds, the binding one named per family), from the same HIP-event record
        fams = []
        for r in sorted(table, key=lambda r: -r["ms"]):
            if not r["launches"] or r["ms"] <= 0:
                continue
            mf_peak = F32_MFMA_PEAK_TFLOPS if r["family"] == "f32" else F16_MFMA_PEAK_TFLOPS / NPROD
            t_mfma = r["flops"] / (mf_peak * 1e12)
            t_hbm = r["bytes"] / (HBM_PEAK_TBS * 1e12)
            sec = r["ms"] * 1e-3
            fams.append({"kernel": r["kernel"], "launches_per_step": r["launches"] / a.steps,
                         "share_of_step": sec / dt, "tflops": r["flops"] / sec / 1e12, "algorithmic_GBps": r["bytes"] / sec / 1e9,
                         "bound": "mfma" if t_mfma >= t_hbm else "hbm", "frac_of_bound": max(t_mfma, t_hbm) / sec})
        if fams:
            res["kernel_families"] = fams
            att = [r for r in table if r["family"] == "attention" and r["ms"] > 0]
            if att:
                ms_, fl_ = sum(r["ms"] for r in att), sum(r["flops"] for r in att)
                res["roofline_attention"] = {"bound": "mfma", "achieved": fl_ / (ms_ * 1e-3) / 1e12,
                                             "peak": F16_MFMA_PEAK_TFLOPS / NPROD, "unit": "TFLOP/s",
                                             "frac": fl_ / (ms_ * 1e-3) / 1e12 / (F16_MFMA_PEAK_TFLOPS / NPROD),
                                             "f16_mfma_issue_tflops": NPROD * fl_ / (ms_ * 1e-3) / 1e12,
                                             "launches_per_step": sum(r["launches"] for r in att) / a.steps,
                                             "share_of_step": ms_ * 1e-3 / dt,
                                             "kernels": sorted({r["kernel"] for r in att})}
        # HBM traffic per kernel family (see paste_traffic): only a measurement taken on THIS library build is reported
        if "roofline" in res:
            from asyrp_official_amd import _lib
            paste_traffic(res, os.path.join(ROOT, "profiles", f"traffic_families_{a.config}_b{B}_{a.conv_math}.json"), getattr(_lib.load(), "_asyrp_path", _lib.LIB_PATH),
                          dt, a.steps)
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"], cpu_chk = cpu_baseline(cpu_sd, betas, family, learn_sigma,
                                                         x_check=x0_cpu[chk_rows] if gpu_check is not None else None)
            if parity is not None and cpu_chk is not None:
                kind = res["cpu_baseline"]["kind"]
                ab = (1.0 - betas).cumprod(0)
                for which, (t_, tn_) in (("dual", CHECK_DUAL), ("inversion", CHECK_INV)):
                    entry = {"what": f"{'dual-decoder Asyrp step (index=0, t_edit=%d, DeltaBlock active)' % T_EDIT if which == 'dual' else 'inversion step'} "
                                     f"t={t_}->{tn_}, images {chk_rows} computed inside the B={B} batch on the GPU vs the CPU {kind} "
                                     "on the same x; rtol 1e-3 / atol 1e-4 (x0_t: atol x 1/sqrt(alpha_bar_t), it divides eps by that)",
                             "outputs": {}}
                    ok_all = True
                    for name, gv, cv in zip(("xt_next", "x0_t", "delta_h", "middle_h"), gpu_check[which], cpu_chk[which]):
                        if gv is None or cv is None:
                            continue
                        atol = 1e-4 * (float(ab[t_]) ** -0.5 if name == "x0_t" else 1.0)
                        err = (gv - cv).abs()
                        ok = bool((err <= atol + 1e-3 * cv.abs()).all())
                        ok_all &= ok
                        entry["outputs"][name] = {"max_abs_err": float(err.max()), "mean_abs_err": float(err.mean()),
                                                  "ref_abs_max": float(cv.abs().max()), "atol": atol, "within_tolerance": ok,
                                                  # the same comparison at the UNSCALED north-star tolerance (differs for x0_t only)
                                                  "frac_outside_unscaled_tolerance": float((err > 1e-4 + 1e-3 * cv.abs()).float().mean())}
                    entry["within_tolerance"] = ok_all
                    parity[f"{which}_step_vs_cpu_{kind}"] = entry
                    if fast:
                        entry["note"] = "fast mode: the parity tolerance is reported, not required (this line is not the line of record)"
                    elif not ok_all:
                        rc = 1
        if parity is not None:
            res["parity_check"] = parity
            if not parity["batch_invariance_bitwise"]:
                rc = 1
        if gather_check is not None:
            res["gather_check"] = gather_check
            if not (gather_check["gathered_slices_equal_rank_results_sha256"] and gather_check.get("gathered_equals_unsharded_bitwise", True)):
                rc = 1
        if world == 1 and a.config == "celeba" and not a.batch and not a.no_other_configs and a.conv_math != "f32":
            # BASELINE configs 2, 3 (per GPU) and 4 (per GPU): never the headline, never inside its timed region; each a 1-edit sample
            del out, out_local
            oc = {"church": other_config_sample("church", dev, betas, a.conv_math, reuse=(model, eng))}
            model._drop_engine()
            torch.cuda.empty_cache()
            for name in ("afhq", "imagenet"):
                try:
                    oc[name] = other_config_sample(name, dev, betas, a.conv_math)
                except Exception as e:   # noqa: BLE001   a sample must not take the headline line down with it
                    oc[name] = {"error": f"{type(e).__name__}: {e}"}
            res["other_configs"] = oc
            if any(v.get("batch_invariance_bitwise") is False for v in oc.values()):
                rc = 1
        print(json.dumps(res), flush=True)
        if rc:
            print("[bench] PARITY CHECK FAILED (see parity_check in the JSON line)", file=sys.stderr)
    if world > 1:
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
