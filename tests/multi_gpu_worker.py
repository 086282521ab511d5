"""Worker of tests/test_gpu_multi.py: launched by torch.distributed.run, one process per GPU, RCCL (backend "nccl").
Every rank edits its batch shard with no data-path collective; one all-gather of x_edit; every rank compares the
gathered batch with the unsharded run on its own GPU (bitwise); rank 0 prints OK with the RCCL version."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    from asyrp_official_amd import run_edit, run_edit_sharded
    from oracle import sampler as osamp
    from oracle.weights import SMALL, hash_normal
    from util_models import hip_model, synthetic
    sd = synthetic(SMALL, 1, seed=7)
    m = hip_model(SMALL, sd, 1, device=f"cuda:{local}", max_batch=8)
    b = osamp.beta_schedule()
    x = hash_normal("multi.x", (2 * world + 1, 3, 32, 32), seed=9).cuda()       # uneven shards on purpose
    kw = dict(n_inv=4, n_gen=4, t_edit=500)
    full = run_edit_sharded(m, x, b, **kw)
    assert full.shape == x.shape
    # EVERY rank holds the full batch after the one all-gather, and it equals the unsharded edit on that rank's own GPU bit for
    # bit (batch-invariant kernels: sharding cannot change an image)
    alone = run_edit(m, x, b, **kw)
    same = torch.tensor([int(torch.equal(full, alone))], device="cuda")
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    assert int(same.item()) == 1, "sharded + all-gathered result differs from the unsharded one on some rank"
    if rank == 0:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        print(f"MULTI_GPU_OK world={world} backend={dist.get_backend()} rccl={ver}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
