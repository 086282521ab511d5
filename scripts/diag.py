"""Quick on-GPU diagnostic: setup time, per-forward time, conv kernel TFLOP/s (not part of the product)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import celeba_namespace
from asyrp_official_amd import DDPM
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t0 = time.perf_counter()
m = DDPM(celeba_namespace(), max_batch=B); m.setattr_layers(1)
print("construct", time.perf_counter() - t0, flush=True)
t0 = time.perf_counter(); m = m.cuda().eval(); eng = m.engine(); torch.cuda.synchronize()
print("engine setup", time.perf_counter() - t0, flush=True)
x = torch.randn(B, 3, 256, 256, device="cuda"); t = torch.ones(B, device="cuda") * 700
for name, kw in (("single", {}), ("dual", dict(index=0, t_edit=500, hs_coeff=(1.0, 1.0)))):
    m(x, t, **kw); torch.cuda.synchronize()
    eng.profile_read(); eng.profile_enable(True)
    t0 = time.perf_counter(); m(x, t, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    eng.profile_enable(False); p = eng.profile_read()
    gf = 497.03 if name == "single" else 858.97
    print(f"{name}: {dt*1e3:.1f} ms/forward B={B} -> {gf*B/dt/1e3:.1f} TFLOP/s overall; dominant {p['kernel']} "
          f"{p['flops']/p['ms']/1e9:.1f} TFLOP/s over {p['launches']} launches ({p['ms']:.1f} ms); all gemm {p['all_flops']/p['all_ms']/1e9:.1f} TFLOP/s ({p['all_ms']:.1f} ms)", flush=True)
print("device bytes", eng.device_bytes() / 2**30, "GiB")
