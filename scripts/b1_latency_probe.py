"""Where does a single-image edit (B = 1, BASELINE config 1's shape) spend its 0.76 s?  Host enqueue time vs device time.

  enqueue_s   wall time until asyrp_run_edit RETURNS (all launches queued; the call is asynchronous)
  total_s     wall time until the stream has drained
If enqueue_s ~ total_s the host (one thread issuing ~68 000 launches) is the bound and a captured graph would help; if
enqueue_s << total_s the device is: tiny kernels and the gaps between dependent launches.  Run it under
`rocprofv3 --kernel-trace --stats` for the sum of kernel durations of the same edit.

  python scripts/b1_latency_probe.py [nominal_batch=0|1] [edits=3]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from asyrp_official_amd import DDPM  # noqa: E402
from asyrp_official_amd.diffusion_utils import get_beta_schedule  # noqa: E402
from asyrp_official_amd.sampler import timestep_seq  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 0
EDITS = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def main():
    torch.manual_seed(1234)
    m = DDPM(bench.celeba_namespace(), max_batch=1, nominal_batch=NB)
    m.setattr_layers(1)
    m = m.cuda().eval()
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    m.set_schedule(betas)
    x0 = (2 * torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(1)) - 1).cuda()
    eng = m._ready_engine(x0)
    seq = timestep_seq(40, 999)[0]
    kw = dict(t_edit=500, index=0, hs_coeff=(1.0, 1.0))
    eng.run_edit(x0, seq, seq, **kw)
    torch.cuda.synchronize()
    rows = []
    for _ in range(EDITS):
        t0 = time.perf_counter()
        eng.run_edit(x0, seq, seq, **kw)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append({"enqueue_s": t1 - t0, "total_s": t2 - t0})
    print(json.dumps({"nominal_batch": NB or 32, "edits": rows,
                      "enqueue_over_total": sum(r["enqueue_s"] for r in rows) / sum(r["total_s"] for r in rows)}))


if __name__ == "__main__":
    main()
