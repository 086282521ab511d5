"""CPU experiment (not a test): how far does a split-precision MFMA conv drift from the fp32 oracle?

Emulates  conv(x, w) ~= conv(x_hi, w_hi) + conv(x_hi, w_lo) + conv(x_lo, w_hi)  with x_hi/x_lo, w_hi/w_lo
the two-term f16 (or bf16) split of the fp32 operands.  Products of two f16 values are exact in fp32,
and the CPU conv accumulates in fp32 like the matrix core does, so this is a faithful model of a
v_mfma_f32_32x32x16_f16 "x3" implicit GEMM.  Run:  python tests/experiments/split_precision_numerics.py [mode]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ddpm as O  # noqa: E402
from oracle.weights import CELEBA, ddpm_param_shapes, hash_normal, synthetic_state_dict  # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 else "f16x3"


def split(v, dt, rtz=False):
    hi = v.to(dt).float()
    lo = (v - hi).to(dt).float()
    return hi, lo


def conv_split(x, sd, p, stride=1, padding=0):
    w, b = sd[p + ".weight"], sd[p + ".bias"]
    if MODE == "f32":
        return F.conv2d(x, w, b, stride=stride, padding=padding)
    dt = torch.float16 if MODE.startswith("f16") else torch.bfloat16
    # per-layer power-of-two weight scale so the f16 lo term stays out of the subnormal range
    s = 2.0 ** torch.floor(torch.log2(1.0 / w.abs().max())).item() if MODE.startswith("f16") else 1.0
    xh, xl = split(x, dt)
    wh, wl = split(w * s, dt)
    kw = dict(stride=stride, padding=padding)
    if MODE.endswith("x1"):
        y = F.conv2d(xh, wh, None, **kw)
    elif MODE.endswith("x3"):
        y = F.conv2d(xh, wh, None, **kw) + (F.conv2d(xh, wl, None, **kw) + F.conv2d(xl, wh, None, **kw))
    else:
        raise SystemExit("mode")
    return y / s + b[None, :, None, None]


def main():
    torch.manual_seed(0)
    sd = synthetic_state_dict(ddpm_param_shapes(CELEBA, n_delta=1), seed=1234)
    x = hash_normal("celeba.x", (1, 3, 256, 256), seed=1234)
    t = torch.ones(1) * 768.0
    with torch.no_grad():
        t0 = time.time()
        ref = O.ddpm_forward(sd, CELEBA, x, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        print("fp32 oracle", time.time() - t0, "s")
        O._conv = conv_split
        t0 = time.time()
        got = O.ddpm_forward(sd, CELEBA, x, t, index=0, t_edit=500, hs_coeff=(1.0, 1.0))
        print(MODE, time.time() - t0, "s")
    for name, g, r in zip(("et", "et_mod", "delta_h", "middle_h"), got, ref):
        err = (g - r).abs()
        bad = err > 1e-4 + 1e-3 * r.abs()
        print(f"{MODE} {name}: max abs {err.max():.3e} mean {err.mean():.3e} frac outside {bad.float().mean():.2e} "
              f"(ref absmax {r.abs().max():.3f}, std {r.std():.3f})")


if __name__ == "__main__":
    main()
