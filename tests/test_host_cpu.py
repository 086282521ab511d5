"""CPU-side checks of the host logic and of the C-ABI surface (no compute calls: there is no GPU here)."""
import json
import os
import re
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT
from oracle.weights import CELEBA, SMALL, ddpm_param_shapes
from util_models import namespace_for

REF = os.environ.get("ASYRP_REFERENCE", "/root/reference")


def _header_functions(bench_hooks=False):
    """Entry points include/asyrp.h declares for the product library (or, bench_hooks: only those inside #ifdef ASYRP_BENCH_HOOKS)."""
    src = open(os.path.join(ROOT, "include", "asyrp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    hooks = "".join(re.findall(r"#ifdef ASYRP_BENCH_HOOKS(.*?)#endif", src, flags=re.S))
    if not bench_hooks:
        src = re.sub(r"#ifdef ASYRP_BENCH_HOOKS.*?#endif", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asyrp_[a-z0-9_]+)\s*\(", hooks if bench_hooks else src)))


def test_library_exports_every_declared_symbol():
    from asyrp_official_amd import _lib
    lib = _lib.load()
    declared = _header_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/asyrp.h but not exported by libasyrp_hip.so"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared, "ctypes signature table and include/asyrp.h disagree"
    header = open(os.path.join(ROOT, "include", "asyrp.h")).read()
    assert lib.asyrp_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define ASYRP_ABI_VERSION (\d+)", header).group(1))
    # the profiling hooks are NOT in the product library (they live in libasyrp_hip_bench.so, built with -DASYRP_BENCH_HOOKS)
    hooks = _header_functions(bench_hooks=True)
    assert hooks == sorted(_lib.BENCH_SIGS) and hooks
    for name in hooks:
        assert not hasattr(lib, name), f"profiling hook {name} leaked into the product library"


def test_engine_parameter_inventory_equals_reference_state_dict():
    """asyrp_param_info lists exactly the reference's state_dict keys/shapes (pinned via the golden-checked oracle shapes)."""
    from asyrp_official_amd.engine import make_config, param_specs
    for cfg, nd in ((SMALL, 2), (CELEBA, 1)):
        c = make_config(resolution=cfg.resolution, in_channels=cfg.in_channels, out_channels=cfg.out_ch, ch=cfg.ch,
                        ch_mult=cfg.ch_mult, num_res_blocks=cfg.num_res_blocks, attn_resolutions=cfg.attn_resolutions,
                        n_delta=nd)
        got = dict(param_specs(c))
        want = {k: tuple(v) for k, v in ddpm_param_shapes(cfg, n_delta=nd).items()}
        assert got == want


def test_mirror_state_dict_roundtrip_and_delta_block_keys():
    from asyrp_official_amd import DDPM
    m = DDPM(namespace_for(SMALL), max_batch=2)
    m.setattr_layers(1)
    keys = set(m.state_dict().keys())
    assert keys == set(ddpm_param_shapes(SMALL, n_delta=1).keys())
    # key names of a shipped DeltaBlock checkpoint (fixture written by tests/golden/make_golden.py from checkpoint/*.pth)
    fx = json.load(open(os.path.join(GOLDEN, "delta_checkpoint_keys.json")))
    big = DDPM(namespace_for(CELEBA), max_batch=1)
    big.setattr_layers(1)
    layer = dict(big.layer_0.state_dict())
    assert {k: list(v.shape) for k, v in layer.items()} == fx["smiling_LC_CelebA_HQ_t999_ninv40_ngen40_0.pth"]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "checkpoint")), reason="reference checkpoints not on this box")
def test_shipped_delta_checkpoints_load_unmodified():
    """Every DDPM-flavour checkpoint/*.pth["0"] loads into model.layer_0 exactly as diffusion_latent.py:674-676 does."""
    from asyrp_official_amd import DDPM
    m = DDPM(namespace_for(CELEBA), max_batch=1)
    m.setattr_layers(1)
    n = 0
    for f in sorted(os.listdir(os.path.join(REF, "checkpoint"))):
        sd = torch.load(os.path.join(REF, "checkpoint", f), map_location="cpu", weights_only=False)["0"]
        if "conv1.weight" not in sd:      # iDDPM-flavour DeltaBlock (in_layers/out_layers keys): other UNet family
            continue
        res = m.layer_0.load_state_dict(sd)
        assert not res.missing_keys and not res.unexpected_keys
        assert torch.equal(m.state_dict()["layer_0.conv2.weight"], sd["conv2.weight"])
        n += 1
    assert n >= 20


def test_cpu_tensors_fail_loudly_no_fallback():
    from asyrp_official_amd import AsyrpDeviceError, DDPM, denoising_step
    m = DDPM(namespace_for(SMALL), max_batch=2)
    x = torch.zeros(1, 3, 32, 32)
    with pytest.raises(AsyrpDeviceError):
        m(x, torch.ones(1) * 10.0)
    with pytest.raises(AsyrpDeviceError):
        denoising_step(x, torch.ones(1) * 10.0, torch.ones(1) * 5.0, models=m, logvars=None,
                       b=torch.linspace(1e-4, 0.02, 1000))
    with pytest.raises(AsyrpDeviceError):      # the injected-delta_h (slerp) branch has no CPU path either
        m(x, torch.ones(1), index=0, delta_h=torch.zeros(1, 64, 8, 8))


def test_shard_bounds_cover_batch():
    from asyrp_official_amd import shard_bounds
    for n in (1, 5, 32, 33, 256):
        for ws in (1, 2, 3, 8):
            spans = [shard_bounds(n, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from asyrp_official_amd.sampler import gather_shards, shard_bounds
    full = torch.arange(n_total * 3 * 4 * 4, dtype=torch.float32).reshape(n_total, 3, 4, 4)
    lo, hi = shard_bounds(n_total, world, rank)
    local = full[lo:hi] * 2.0 + 1.0            # stand-in for "edit my images": any per-image function
    got = gather_shards(local.contiguous(), n_total)
    torch.save(got, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [4, 5])
def test_two_rank_gloo_shard_and_all_gather(tmp_path, n_total):
    """The N>1 path: contiguous batch shards, no data-path collective, one all-gather (uneven shards padded)."""
    port = _free_port()
    mp.spawn(_rank_main, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    full = torch.arange(n_total * 3 * 4 * 4, dtype=torch.float32).reshape(n_total, 3, 4, 4) * 2.0 + 1.0
    for r in range(2):
        assert torch.equal(torch.load(os.path.join(tmp_path, f"r{r}.pt")), full)


def _mean_rank_main(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from asyrp_official_amd.cache import finish_mean_delta_hs
    g = torch.Generator().manual_seed(5)
    parts = torch.randn((3, 2, 1, 4, 2, 2), generator=g)        # [batch, timestep, ...]: rank 0 owns batch 0-1, rank 1 batch 2
    mine = parts[:2] if rank == 0 else parts[2:]
    collect = {0: None, 499: None, 749: sum(mine[:, 0]), 999: sum(mine[:, 1])}
    torch.save(finish_mean_delta_hs(collect, len(mine)), os.path.join(out_dir, f"m{rank}.pt"))
    dist.destroy_process_group()


def test_global_mean_delta_h_dictionary_single_and_two_rank(tmp_path):
    """finish_mean_delta_hs == the reference's bookkeeping (diffusion_latent.py:811-831); across ranks the per-timestep
    sums are all-reduced so every rank holds the mean over all images (the one data-path reduction of the method)."""
    from asyrp_official_amd.cache import finish_mean_delta_hs, mean_delta_path
    g = torch.Generator().manual_seed(5)
    parts = torch.randn((3, 2, 1, 4, 2, 2), generator=g)
    # the reference's own loop, restated: sums per timestep, /= number of batches, entry 0 = mean over filled timesteps
    ref = {0: None, 499: None, 749: parts[0, 0] + parts[1, 0] + parts[2, 0], 999: parts[0, 1] + parts[1, 1] + parts[2, 1]}
    for k in ref:
        if ref[k] is not None:
            ref[k] = ref[k] / 3
    ref[0] = (ref[749] + ref[999]) / 2
    got = finish_mean_delta_hs({0: None, 499: None, 749: sum(parts[:, 0]), 999: sum(parts[:, 1])}, 3)
    assert got[499] is None and all(torch.equal(got[k], ref[k]) for k in (0, 749, 999))
    mp.spawn(_mean_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        m = torch.load(os.path.join(tmp_path, f"m{r}.pt"))
        assert m[499] is None
        for k in (0, 749, 999):
            torch.testing.assert_close(m[k], ref[k], rtol=1e-6, atol=1e-7)
    assert mean_delta_path("smiling", 40, 20) == "checkpoint_latent/smiling_40_20.pth"


def test_iddpm_parameter_inventory_and_factories():
    """i_DDPM / guided_Diffusion mirrors expose exactly the reference UNetModel state_dict (pinned via make_golden's strict load)."""
    from asyrp_official_amd import guided_Diffusion, i_DDPM
    from oracle.iddpm import AFHQ, iddpm_param_shapes
    want = {k: tuple(v) for k, v in iddpm_param_shapes(AFHQ, n_delta=1).items()}
    for m in (i_DDPM("AFHQ"), i_DDPM("FFHQ"), guided_Diffusion("MetFACE"), guided_Diffusion("CelebA_HQ_P2")):
        m.setattr_layers(1)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == want
    fx = json.load(open(os.path.join(GOLDEN, "delta_checkpoint_keys.json")))
    assert {k: list(v.shape) for k, v in m.layer_0.state_dict().items()} == fx["dog_happy_LC_dog_t999_ninv40_ngen40_0.pth"]
    with pytest.raises(NotImplementedError):
        from asyrp_official_amd import UNetModel
        UNetModel(64, 3, 32, 3, 1, (2,), channel_mult=(1, 2), num_head_channels=16)     # no updown / scale-shift: unused by the reference


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "checkpoint")), reason="reference checkpoints not on this box")
def test_shipped_iddpm_delta_checkpoints_load_unmodified():
    from asyrp_official_amd import i_DDPM
    m = i_DDPM("AFHQ", max_batch=1)
    m.setattr_layers(1)
    n = 0
    for f in sorted(os.listdir(os.path.join(REF, "checkpoint"))):
        sd = torch.load(os.path.join(REF, "checkpoint", f), map_location="cpu", weights_only=False)["0"]
        if "in_layers.2.weight" not in sd:
            continue
        res = m.layer_0.load_state_dict(sd)
        assert not res.missing_keys and not res.unexpected_keys
        n += 1
    assert n >= 5


def test_cache_formats_roundtrip_and_reference_names(tmp_path):
    """§8(f)-1: the latent-cache and Δh-checkpoint files are the reference's own formats and names."""
    from asyrp_official_amd import DDPM, cache
    assert cache.pairs_path("CelebA_HQ", "test", 999, 4, 40) == "precomputed/CelebA_HQ_test_t999_nim4_ninv40_pairs.pth"
    assert cache.checkpoint_name("smiling", "CelebA_HQ", 999, 40, 40) == "checkpoint/smiling_LC_CelebA_HQ_t999_ninv40_ngen40_0.pth"
    pairs = [[torch.randn(1, 3, 8, 8) for _ in range(3)] for _ in range(3)]
    p = os.path.join(tmp_path, "precomputed", "x_pairs.pth")
    cache.save_pairs(p, pairs)
    back = torch.load(p, map_location="cpu", weights_only=False)          # exactly what diffusion_latent.py:977 does
    assert isinstance(back, list) and len(back) == 3 and all(len(t) == 3 for t in back)
    assert all(torch.equal(a, b) for ta, tb in zip(pairs, cache.load_pairs(p)) for a, b in zip(ta, tb))
    x0, x_lat = cache.latents_from_pairs(back, 1, 3)
    assert x0.shape == (2, 3, 8, 8) and torch.equal(x_lat[0], pairs[1][2][0])
    m = DDPM(namespace_for(SMALL), max_batch=1)
    m.setattr_layers(2)
    ck = os.path.join(tmp_path, "checkpoint", "a_0.pth")
    cache.save_delta_checkpoint(m, ck, get_h_num=2)
    d = torch.load(ck, map_location="cpu", weights_only=False)
    assert set(d.keys()) == {"0", "1", "optimizer", "scheduler"}
    assert set(d["0"].keys()) == {"conv1.weight", "conv1.bias", "temb_proj.weight", "temb_proj.bias", "norm2.weight",
                                  "norm2.bias", "conv2.weight", "conv2.bias"}
    m2 = DDPM(namespace_for(SMALL), max_batch=1)
    m2.setattr_layers(2)
    cache.load_delta_checkpoints(m2, [ck, ck])       # multiple_attr: one file per DeltaBlock, key "0" of each (:674-676)
    assert torch.equal(m2.layer_1.conv2.weight, m.layer_0.conv2.weight)


def test_hs_coeff_schedules_match_reference_formulas():
    from asyrp_official_amd import cache
    assert cache.make_hs_coeff(40, 40) == (1.0, 1.0)
    assert cache.make_hs_coeff(40, 20, hs_coeff_delta_h=1.5) == (1.0, 3.0)                     # :626, :659
    hc = cache.make_hs_coeff(40, 40, multiple_hs_coeff=[0.5], n_attr=2)                      # :654
    assert hc[0] == 1.0 and abs(hc[1] - 0.5 / 2 ** 0.5) < 1e-12 and abs(hc[2] - 1 / 2 ** 0.5) < 1e-12
    sw = cache.delta_interpolation_coeffs(-1.0, 1.0, 5)
    assert [c[1] for c in sw] == [-1.0, -0.5, 0.0, 0.5, 1.0] and all(c[0] == 1.0 for c in sw)
    # base tuple with hs_coeff_origin_h != 1 and a scaled delta: every element times val, element 0 forced to 1.0 (:742-751)
    sw = cache.delta_interpolation_coeffs(0.5, 1.5, 3, hs_coeff=cache.make_hs_coeff(40, 20, hs_coeff_delta_h=1.5, hs_coeff_origin_h=0.9))
    assert sw == [(1.0, 1.5), (1.0, 3.0), (1.0, 4.5)]
    # two attributes: the num_delta^2 grid (1.0, v1*c1, v2*c2) (:728-740)
    base = cache.make_hs_coeff(40, 40, n_attr=2)
    grid = cache.delta_interpolation_coeffs(0.0, 1.0, 2, hs_coeff=base, multiple_attr=True)
    assert grid == [(1.0, 0.0, 0.0), (1.0, 0.0, base[2]), (1.0, base[1], 0.0), (1.0, base[1], base[2])]
    with pytest.raises(ValueError):
        cache.delta_interpolation_coeffs(0.0, 1.0, 2, hs_coeff=(1.0, 1.0), multiple_attr=True)


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "utils", "diffusion_utils.py")), reason="reference not on this box")
def test_bench_cpu_baseline_uses_the_reference_when_importable():
    """bench.py's cpu_baseline leg: kind == "reference" (the reference's own DDPM + denoising_step) when /root/reference is
    importable, and that path computes the same step as the oracle port it falls back to on the GPU box."""
    sys.path.insert(0, ROOT)
    import bench
    from asyrp_official_amd import DDPM
    from oracle import sampler as osamp
    torch.manual_seed(1234)
    m = DDPM(bench.celeba_namespace(), max_batch=1)
    m.setattr_layers(1)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    betas = osamp.beta_schedule()
    kind, step = bench._cpu_step_fn(sd, betas, "ddpm", False)
    assert kind == "reference"
    x = 2 * torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(7)) - 1
    one = torch.ones(1)
    torch.set_num_threads(8)
    got = step(x, one * 0.0, one * 25.0, eta=0)[0]
    os.environ["ASYRP_REFERENCE"] = "/nonexistent"
    try:
        kind2, step2 = bench._cpu_step_fn(sd, betas, "ddpm", False)
    finally:
        del os.environ["ASYRP_REFERENCE"]
    assert kind2 == "port"
    want = step2(x, one * 0.0, one * 25.0, eta=0)[0]
    assert torch.allclose(got, want, rtol=1e-5, atol=2e-6)


def test_bench_reports_counter_traffic_only_for_the_loaded_library_build(tmp_path):
    """bench.py pastes the committed rocprofv3 --pmc traffic into a line only when the file's sha256 stamp equals the library
    it is running on; a stale file is refused with a note."""
    import hashlib
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lib = tmp_path / "lib.so"
    lib.write_bytes(b"library build A")
    main = "asyrp::igemm_f16x3_k32_kernel<asyrp::K32Cfg<8, 2>>"
    fams = {main: {"hbm_bytes_per_launch": 1.5e9}, "asyrp::attn_planes_kernel": {"hbm_bytes_per_launch": 6e7}}
    tfile = tmp_path / "traffic.json"

    def line():
        return {"roofline": {"kernel": main, "traffic": None},
                "kernel_families": [{"kernel": main, "launches_per_step": 4000.0, "share_of_step": 0.8},
                                    {"kernel": "asyrp::attn_planes_kernel (T=256)", "launches_per_step": 400.0, "share_of_step": 0.004},
                                    {"kernel": "asyrp::conv_out_kernel (Cout=3)", "launches_per_step": 99.0, "share_of_step": 0.01}]}

    # stamp of another build: nothing is pasted
    tfile.write_text(json.dumps({"library_sha256": hashlib.sha256(b"library build B").hexdigest(), "families": fams}))
    r = line()
    bench.paste_traffic(r, str(tfile), str(lib), dt=8.0, steps=2)
    assert r["roofline"]["traffic"] is None and "another build" in r["roofline"]["traffic_note"]
    assert all("counter_GBps" not in f for f in r["kernel_families"])
    # matching stamp: bytes per launch of the dominant kernel, GB/s per family from this run's own launch durations
    tfile.write_text(json.dumps({"library_sha256": hashlib.sha256(b"library build A").hexdigest(), "families": fams}))
    r = line()
    bench.paste_traffic(r, str(tfile), str(lib), dt=8.0, steps=2)
    assert r["roofline"]["traffic"] == 1.5e9
    sec = 0.8 * 8.0 / 2 / 4000.0                                  # 0.8 of the 8-second region, 2 steps x 4000 launches
    assert abs(r["kernel_families"][0]["counter_GBps"] - 1.5e9 / sec / 1e9) < 1e-6
    assert "counter_GBps" in r["kernel_families"][1]              # "name (T=256)" is keyed by the kernel name
    assert "counter_GBps" not in r["kernel_families"][2]          # no counters for this family in the file: left out, not guessed
    # no file at all: untouched
    r = line()
    bench.paste_traffic(r, str(tmp_path / "missing.json"), str(lib), dt=8.0, steps=2)
    assert r["roofline"]["traffic"] is None and "traffic_note" not in r["roofline"]


def test_bench_eight_rank_plumbing_dry_run_under_gloo():
    """`python bench.py --gpus 8` end to end WITHOUT hardware: the real self-launch (torch.distributed.run, 8
    ranks), rendezvous on 127.0.0.1, per-rank seeds, barrier + max-over-ranks timing and the one all-gather, with the engine replaced by
    a stand-in per-image function on CPU tensors (--plumbing-dry-run), so that the first real 8-GPU run cannot fail on plumbing."""
    import json
    import subprocess
    env = dict(os.environ, ASYRP_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "3",
                        "--plumbing-dry-run"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly ONE JSON line"
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["world_size"] == 8 and res["steps"] == 2 and res["warmup"] == 1
    assert res["dry_run"] is True and res["value"] is None and res["scaling"] == "weak"
    assert len(set(res["rank_seeds"])) == 8 and res["rank_seeds"] == [1234 + r_ for r_ in range(8)]
    assert res["gathered_shape"] == [24, 3, 16, 16] and res["gathered_equals_unsharded"] is True
    assert res["metric"].startswith("edited images/sec")
    # every rank builds its engine on its LOCAL_RANK's device (one process per GPU)
    assert res["engine_device_index_per_rank"] == list(range(8))
    assert res["config"]["batch_per_gpu"] == 32 and "configs[1]" in res["config"]["baseline_config"]


@pytest.mark.parametrize("config,batch,tag,name", [("church", 32, "configs[3]", "LSUN-Church"), ("imagenet", 16, "configs[4]", "ImageNet ADM")])
def test_bench_eight_rank_dry_run_names_the_sharded_baseline_configs(config, batch, tag, name):
    """`--config church --gpus 8` / `--config imagenet --gpus 8` (BASELINE configs 4 and 5: batch 256 / 128 sharded over 8 GPUs) report
    32 / 16 images per GPU and name the configuration, so the first real 8-GPU line cannot be quoted on the wrong one."""
    import json
    import subprocess
    env = dict(os.environ, ASYRP_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--config", config,
                        "--plumbing-dry-run"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 8 and res["engine_device_index_per_rank"] == list(range(8))
    assert res["config"]["batch_per_gpu"] == batch and tag in res["config"]["baseline_config"] and name in res["config"]["workload"]
    assert res["config"]["learn_sigma"] == (config == "imagenet")


def test_batch_class_values_are_validated_before_they_reach_the_library():
    """asyrp_config.nominal_batch accepts the tested classes only (0 = 32, 1, 2, 32): the Python mirror raises before an
    engine is built, and asyrp_create repeats the check (and requires the reserved words to be zero) for any other binding."""
    from asyrp_official_amd import DDPM
    sys.path.insert(0, ROOT)
    import bench
    for ok in (0, 1, 2, 32):
        DDPM(bench.celeba_namespace(), max_batch=1, nominal_batch=ok)
    for bad in (-1, 3, 16, 64, 4096):
        with pytest.raises(ValueError):
            DDPM(bench.celeba_namespace(), max_batch=1, nominal_batch=bad)
    src = open(os.path.join(ROOT, "asyrp_official_amd", "csrc", "engine.hip")).read()
    assert "asyrp_config.reserved must be zero" in src and "nominal_batch must be 0" in src


def test_data_parallel_replicas_are_light_proxies_of_the_source():
    """torch.nn.DataParallel (the reference's only multi-GPU form, diffusion_latent.py:179,195,591,1201) replicates the module per
    device on every forward.  A replica of the mirror must (i) exist — round 5 raised here, which killed the unmodified edit script on
    a multi-GPU node at its first UNet call — (ii) share the SOURCE's per-device engine table and lock, (iii) name
    the source, also when a replica is replicated again, (iv) not be registered as a child module, and the source must stay
    deep-copyable / picklable (engines and locks are per process)."""
    import copy
    import pickle
    from asyrp_official_amd import DDPM, DataParallel, i_DDPM
    from asyrp_official_amd._base import HipUNet
    sys.path.insert(0, ROOT)
    import bench
    for m in (DDPM(bench.celeba_namespace(), max_batch=1), i_DDPM("AFHQ", max_batch=1)):
        m.setattr_layers(1)
        n_children = len(list(m.children()))
        r = m._replicate_for_data_parallel()
        assert isinstance(r, HipUNet) and r is not m and r._src() is m and m._src() is m
        assert r._slots is m._slots and r._slots_lock is m._slots_lock
        assert r._replicate_for_data_parallel()._src() is m
        assert len(list(m.children())) == n_children and "_dp_source" not in m._modules and "_dp_source" not in r._modules
        assert m._engine is None                                   # nothing is created before the first forward (CPU parameters)
        # the reference's post-wrap accesses (diffusion_latent.py:284,288,675) reach the source's parameters
        w = DataParallel(m, device_ids=None) if torch.cuda.is_available() else None
        sd = {k: torch.full_like(v, 0.25) for k, v in m.layer_0.state_dict().items()}
        (w.module if w is not None else m).layer_0.load_state_dict(sd)
        assert all(bool((p == 0.25).all()) for p in r._src().layer_0.parameters())
        m2 = copy.deepcopy(m)
        assert m2._slots == {} and m2._slots is not m._slots and m2._slots_lock is not m._slots_lock
        assert all(torch.equal(a, b) for a, b in zip(m2.layer_0.parameters(), m.layer_0.parameters()))
        m3 = pickle.loads(pickle.dumps(m))
        assert m3._slots == {} and set(m3.state_dict()) == set(m.state_dict())


def test_data_parallel_subclass_skips_the_parameter_broadcast():
    """asyrp_official_amd.DataParallel.replicate: one light proxy per device for an engine-backed module (no broadcast: the stock
    replicate() copies every parameter to every device on every forward), the stock path for anything else."""
    from asyrp_official_amd import DDPM, DataParallel
    from asyrp_official_amd.data_parallel import wrapper_devices
    sys.path.insert(0, ROOT)
    import bench
    m = DDPM(bench.celeba_namespace(), max_batch=1)
    w = DataParallel.__new__(DataParallel)              # the constructor needs visible GPUs; replicate() itself does not
    reps = DataParallel.replicate(w, m, [0, 1, 2])
    assert len(reps) == 3 and all(r._src() is m and r._slots is m._slots for r in reps) and len({id(r) for r in reps}) == 3
    assert all(not r._parameters for r in reps)
    assert wrapper_devices(m) is None


def test_product_library_reads_no_environment_switch():
    """Round 5 hygiene: every ASYRP_* kernel switch lives in the PROFILING build only; the product library has no
    switch name in its image and does not import getenv, so a shell variable cannot change which kernels a product run executes."""
    import subprocess
    from asyrp_official_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"ASYRP_" not in blob, "an environment switch name is compiled into the product library"
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in und


def test_edit_sweep_batches_the_tuples_and_keeps_their_order(monkeypatch):
    """cache.edit_sweep's host logic (round 5) with a stand-in for the engine call: K coefficient tuples x B images run as batch
    entries in [tuple][image] order, chunked by max_batch (<= 128), with one tuple per image; the result list is per tuple, as the
    reference's per-tuple loop returns it (diffusion_latent.py:726-755, 499-534).  Fallbacks: batched=False, a single tuple, a model
    whose max_batch cannot hold two tuples, supplied noise."""
    from asyrp_official_amd import cache

    calls = []

    def fake_run_edit(model, x, betas, *, invert, hs_coeff, **kw):
        calls.append((int(x.shape[0]), hs_coeff))
        per = hs_coeff if isinstance(hs_coeff[0], (tuple, list)) else [hs_coeff] * x.shape[0]
        assert len(per) == x.shape[0]
        return torch.stack([x[i] * float(per[i][1]) + float(per[i][0]) for i in range(x.shape[0])])

    monkeypatch.setattr(cache, "run_edit", fake_run_edit)

    class M:
        max_batch = 6

    x_T = torch.arange(2 * 3 * 4 * 4, dtype=torch.float32).reshape(2, 3, 4, 4)
    tuples = cache.delta_interpolation_coeffs(-1.0, 2.0, 4, hs_coeff=(1.0, 0.5))
    want = [x_T * hc[1] + hc[0] for hc in tuples]
    got = cache.edit_sweep(M(), x_T, None, tuples, n_gen=4)
    assert [c[0] for c in calls] == [6, 2]                     # 3 tuples x 2 images in the first call, the last tuple alone
    assert isinstance(calls[0][1][0], tuple) and len(calls[0][1]) == 6 and calls[0][1][0] == calls[0][1][1] != calls[0][1][2]
    assert len(got) == 4 and all(torch.equal(g, w) for g, w in zip(got, want))
    # per-tuple fallbacks: asked for, supplied noise, an eta = 1 tail whose noise would be DRAWN (a batched draw consumes the
    # generator in another order), want_latent (run_edit then returns a pair per pass)
    for kwargs, n_calls in ((dict(batched=False), 4), (dict(noise=torch.zeros(1)), 4), (dict(t_addnoise=300), 4), (dict(want_latent=True), 4)):
        calls.clear()
        got = cache.edit_sweep(M(), x_T, None, tuples, n_gen=4, **kwargs)
        assert len(calls) == n_calls and all(torch.equal(g, w) for g, w in zip(got, want))
    M.max_batch = 3                                             # cannot hold two tuples of two images: one pass per tuple
    calls.clear()
    got = cache.edit_sweep(M(), x_T, None, tuples, n_gen=4)
    assert len(calls) == 4 and all(torch.equal(g, w) for g, w in zip(got, want))



def test_compact_fixture_storage_roundtrip(tmp_path):
    """tests/golden/compact.py: a fixture written under the storage policy reads back as the same values -- sampled keys at the hashed
    positions (one per run of 16 elements, reproducible from the key alone), whole keys untouched, the shared DeltaBlock weights
    merged back under `param.*` -- and assert_close / err_stats compare a Sampled target at exactly those positions."""
    import numpy as np
    from conftest import assert_close
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import compact
    from util_models import err_stats
    name = "config4_church_gothic.npz"
    rng = np.random.default_rng(0)
    full = {k: rng.standard_normal((1, 3, 32, 32)).astype(np.float32) for k in compact.SAMPLED[name]}
    full["gen999.delta_h"] = rng.standard_normal((1, 8, 4, 4)).astype(np.float32)
    full["param.layer_0.conv1.weight"] = rng.standard_normal((4, 4, 1, 1)).astype(np.float32)
    compact.save(str(tmp_path / name), full)
    assert (tmp_path / compact.PARAM_FILE[name]).exists()
    compact.save(str(tmp_path / name), full)                              # second fixture of the same checkpoint: weights must agree
    g = compact.load(str(tmp_path / name))
    assert torch.equal(g["gen999.delta_h"], torch.from_numpy(full["gen999.delta_h"]))
    assert torch.equal(g["param.layer_0.conv1.weight"], torch.from_numpy(full["param.layer_0.conv1.weight"]))
    for k in compact.SAMPLED[name]:
        s_ = g[k]
        assert isinstance(s_, compact.Sampled) and s_.shape == (1, 3, 32, 32) and s_.values.numel() == 3 * 32 * 32 // 16
        idx = compact.sample_index(3 * 32 * 32, k)
        assert np.array_equal(idx // 16, np.arange(idx.size)) and len({int(i) % 16 for i in idx}) > 8      # one per run, spread
        assert torch.equal(s_.values, torch.from_numpy(full[k].reshape(-1)[idx]))
        t = torch.from_numpy(full[k])
        assert_close(t, s_)
        assert err_stats(t, s_)["max_abs"] == 0.0
        bad = t.clone().reshape(-1)
        bad[int(idx[5])] += 1.0                                           # a stored position: must be seen
        with pytest.raises(AssertionError):
            assert_close(bad.reshape(t.shape), s_)
    with pytest.raises(AssertionError):                                  # other weights under the same checkpoint name: refused
        compact.save(str(tmp_path / name), dict(full, **{"param.layer_0.conv1.weight": full["param.layer_0.conv1.weight"] + 1}))


def test_data_parallel_sharding_bookkeeping_with_stand_in_engines(monkeypatch):
    """data_parallel.sharded_step / sharded_edit on CPU tensors with stand-ins for torch's CUDA scatter / threads / gather and for the
    engine: the batch is cut as torch.chunk cuts it, every chunk gets ITS rows of the noise (batch dimension 1 of the
    [n_eta, B, ...] stack), of an injected delta_h and of a per-image coefficient table, chunks run on the device their wrapper entry
    names, results come back in batch order, an injected delta_h is returned as the caller's own object, want_latent gathers both
    tensors.  (The real wrappers run in tests/test_gpu_data_parallel.py; this is the N > 1 logic the driver can check without a GPU.)"""
    import threading
    from asyrp_official_amd import data_parallel as dp

    def fake_scatter(t, ids, dim=0):
        return tuple(torch.chunk(t, len(ids), dim=dim))

    seen = []

    def fake_parallel_apply(fns, inputs, kwargs_tup=None, devices=None):
        out = [None] * len(fns)

        def work(i):
            seen.append((devices[i], threading.get_ident()))
            out[i] = fns[i](*inputs[i])
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(fns))]
        [t.start() for t in th]
        [t.join() for t in th]
        return out

    def fake_gather(outs, dev):
        first = outs[0]
        if first is None:
            return None
        if isinstance(first, torch.Tensor):
            return torch.cat(outs, 0)
        return type(first)(fake_gather([o[i] for o in outs], dev) for i in range(len(first)))

    monkeypatch.setattr(dp, "scatter", fake_scatter)
    monkeypatch.setattr(dp, "parallel_apply", fake_parallel_apply)
    monkeypatch.setattr(dp, "gather", fake_gather)

    class FakeEngine:
        def ddim_step(self, x, noise=None, delta_h=None, t=0, t_next=0, **kw):
            nz = noise if noise is not None else torch.zeros_like(x)
            dh = delta_h if delta_h is not None else (x[:, :1, :2, :2] * 3 if kw.get("apply_edit") else None)
            return x * 2 + nz + t, x - nz, dh, x[:, :1, :1, :1] + t_next

        def run_edit(self, x, seq_inv, seq_gen, noise=None, hs_coeff=(1.0, 1.0), want_latent=False, **kw):
            per = hs_coeff if isinstance(hs_coeff[0], (tuple, list)) else [hs_coeff] * x.shape[0]
            assert len(per) == x.shape[0]
            c = torch.tensor([float(p_[1]) for p_ in per]).view(-1, 1, 1, 1)
            y = x * c + (noise.sum(0) if noise is not None else 0) + len(seq_inv) + 10 * len(seq_gen)
            return (y, x + 1) if want_latent else y

    class FakeModel:
        def _ready_engine(self, x):
            return FakeEngine()

    class Wrapper(torch.nn.DataParallel):
        def __init__(self):          # no CUDA here: only the attributes the sharding code reads
            torch.nn.Module.__init__(self)
            self.device_ids, self.output_device = [0, 1, 2], 0

    w, m, eng = Wrapper(), FakeModel(), FakeEngine()
    assert dp.wrapper_devices(w) == [0, 1, 2]
    B = 7
    x = torch.arange(B * 3 * 4 * 4, dtype=torch.float32).reshape(B, 3, 4, 4)
    nz = torch.randn(B, 3, 4, 4)
    dh = torch.randn(B, 1, 2, 2)
    got = dp.sharded_step(w, m, x, noise=nz, delta_h=None, t=5, t_next=4, apply_edit=True)
    want = eng.ddim_step(x, noise=nz, t=5, t_next=4, apply_edit=True)
    assert all(torch.equal(g, v) for g, v in zip(got, want))
    assert sorted(d for d, _ in seen) == [0, 1, 2]                                                  # one chunk per wrapper device
    got = dp.sharded_step(w, m, x, noise=None, delta_h=dh, t=5, t_next=4, apply_edit=True)
    assert got[2] is dh and torch.equal(got[0], x * 2 + 5)                                          # the caller's own delta_h object
    got = dp.sharded_step(w, m, x, t=5, t_next=4, apply_edit=False)
    assert got[2] is None
    # both loops: noise stack [n_eta, B, ...] split along its batch dimension, one coefficient tuple per image split with the batch
    stack = torch.randn(2, B, 3, 4, 4)
    tuples = [(1.0, 0.5 * i) for i in range(B)]
    got, got_T = dp.sharded_edit(w, m, x, [0, 1], [0, 1, 2], noise=stack, hs_coeff=tuples, want_latent=True, t_edit=1)
    want, want_T = eng.run_edit(x, [0, 1], [0, 1, 2], noise=stack, hs_coeff=tuples, want_latent=True)
    assert torch.equal(got, want) and torch.equal(got_T, want_T)
    assert torch.equal(dp.sharded_edit(w, m, x[:2], [], [0], hs_coeff=(1.0, 2.0)), eng.run_edit(x[:2], [], [0], hs_coeff=(1.0, 2.0)))
    with pytest.raises(ValueError):
        dp.sharded_edit(w, m, x, [], [0], hs_coeff=tuples[:3])
