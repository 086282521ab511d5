#!/usr/bin/env python
"""Calibration of bench.py's CPU baseline: on the GPU box `/root/reference` does not exist, so the driver's
line times the oracle PORT (`"kind": "port"`).  This script times BOTH kinds on one host (the build container, where the reference
is importable) on exactly bench.py's sample -- CelebA-HQ DDPM, B=1, 4 inversion + 4 dual-decoder Asyrp steps, the same seeded
weights and image, the same thread count -- and records the ratio and the difference of their outputs in
profiles/cpu_port_vs_reference_same_host.json.  bench.py copies the ratio into `cpu_baseline` when it reports the port.
usage: python scripts/cpu_port_vs_reference.py [threads]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else len(os.sched_getaffinity(0))
    from asyrp_official_amd import DDPM
    from asyrp_official_amd.diffusion_utils import get_beta_schedule
    torch.manual_seed(1234)
    model = DDPM(bench.celeba_namespace(), max_batch=1)
    model.setattr_layers(1)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    betas = torch.from_numpy(get_beta_schedule(beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float()
    inv_pairs = ((0, 25), (25, 51), (51, 76), (76, 102))
    gen_pairs = ((999, 973), (973, 947), (947, 922), (922, 896))
    one = torch.ones(1)
    res, outs = {}, {}
    ref_root = os.environ.get("ASYRP_REFERENCE", "/root/reference")
    for kind_wanted, root in (("reference", ref_root), ("port", "/nonexistent")):
        os.environ["ASYRP_REFERENCE"] = root
        kind, step = bench._cpu_step_fn(sd, betas, "ddpm", False)
        assert kind == kind_wanted, (kind, kind_wanted)
        torch.set_num_threads(threads)
        g = torch.Generator().manual_seed(1234)
        x = 2 * torch.rand((1, 3, 256, 256), generator=g) - 1
        step(x, one * 0.0, one * 25.0, eta=0)           # warm
        best = None
        for rep in range(2):
            xx = x.clone()
            t0 = time.perf_counter()
            for (i, j) in inv_pairs:
                xx, _, _, _ = step(xx, one * float(i), one * float(j), eta=0)
            t_inv = (time.perf_counter() - t0) / len(inv_pairs)
            x_inv = xx.clone()
            t0 = time.perf_counter()
            for (i, j) in gen_pairs:
                xx, _, dh, _ = step(xx, one * float(i), one * float(j), eta=0.0, index=0, t_edit=bench.T_EDIT, hs_coeff=(1.0, 1.0))
            t_gen = (time.perf_counter() - t0) / len(gen_pairs)
            if best is None or t_inv + t_gen < best[0] + best[1]:
                best = (t_inv, t_gen)
        res[kind] = {"s_per_inversion_step": best[0], "s_per_dual_decoder_step": best[1],
                     "images_per_s_extrapolated": 1.0 / ((bench.N_INV - 1) * best[0] + bench.N_GEN * best[1])}
        outs[kind] = (x_inv, xx, dh)
        print(kind, res[kind], flush=True)
    d = {n: float((a - b).abs().max()) for n, a, b in zip(("x_after_4_inversion_steps", "x_after_4_dual_steps", "delta_h_last_step"),
                                                           outs["reference"], outs["port"])}
    out = {"what": "bench.py's cpu_baseline sample (CelebA-HQ DDPM, B=1, 4 inversion + 4 dual-decoder Asyrp steps, best of 2) timed with the "
                   "reference's own modules and with the oracle port on the SAME host and thread count",
           "host": {"visible_cores": len(os.sched_getaffinity(0)), "threads": threads, "torch": torch.__version__},
           "reference": res["reference"], "port": res["port"],
           "port_over_reference_rate": res["port"]["images_per_s_extrapolated"] / res["reference"]["images_per_s_extrapolated"],
           "max_abs_diff_of_outputs": d,
           "outputs_bit_identical": all(v == 0.0 for v in d.values())}
    path = os.path.join(ROOT, "profiles", "cpu_port_vs_reference_same_host.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
