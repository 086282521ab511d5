"""RCCL path: only meaningful with >= 2 GPUs in the box; skipped on the 1-GPU test boxes.  The same sharding and
gather logic is covered on CPU with gloo (tests/test_host_cpu.py) and on one GPU bitwise (tests/test_gpu_edit.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


need2 = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL over xGMI)")


@need2
def test_run_edit_sharded_over_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "tests", "multi_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MULTI_GPU_OK world=2 backend=nccl" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@need2
def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` (no launcher) re-executes itself under torch.distributed.run and reports n_gpus = 2."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["collective_backend"] == "nccl" and res["parity_check"]["batch_invariance_bitwise"]


def test_two_engine_processes_under_the_real_launcher_share_one_gpu():
    """Two REAL engine processes + the real self-launch + the gather on a 1-GPU box: `bench.py --gpus 2` with the
    dry-run backend (gloo: the ranks share device 0 and the gather is staged through host memory; RCCL needs one GPU per rank).
    Asserts the world size the collective library saw, per-rank seeds, two processes, that every rank received every other rank's
    result (sha256 per slice), and that the gathered batch equals ONE unsharded call on rank 0's engine bit for bit."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ASYRP_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "4",
                        "--no-kernel-events", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["config"]["world_size"] == 2 and res["config"]["collective_backend"] == "gloo"
    assert res["config"]["launcher"].startswith("self") and res["config"]["batch_per_gpu"] == 4
    gc = res["gather_check"]
    assert gc["world_size"] == 2 and gc["rank_seeds"] == [1234, 1235] and gc["distinct_processes"] and gc["shards_differ"]
    assert gc["gathered_slices_equal_rank_results_sha256"] and gc["gathered_equals_unsharded_bitwise"] is True
    assert res["parity_check"]["batch_invariance_bitwise"]
    assert res["value"] > 0 and "other_configs" not in res


def test_bench_refuses_more_gpus_than_visible():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
