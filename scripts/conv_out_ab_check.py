"""sha256 of conv_out (tile 13) outputs on fixed inputs: run once per value of ASYRP_CONV_OUT_PATCH with ASYRP_LIBRARY=bench and compare
(the 14 x 14 and the 8 x 16 patch forms do the same products in the same order per output: the hashes must be equal)."""
import hashlib, os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_ops import hip_conv
from oracle.weights import hash_normal, hash_uniform
out = []
for (B, Cout, Cin, H, W) in ((2, 3, 128, 64, 64), (1, 3, 128, 256, 256), (2, 3, 32, 40, 24), (1, 1, 256, 16, 48), (3, 3, 128, 30, 17)):
    x = hash_normal(f"coab.x.{Cin}.{H}", (B, Cin, H, W)); w = hash_uniform(f"coab.w.{Cout}.{Cin}", (Cout, Cin, 3, 3), -1, 1) / (Cin * 9) ** 0.5
    b = 0.1 * hash_uniform(f"coab.b.{Cout}", (Cout,)); gn = (1 + 0.1 * hash_uniform("coab.g", (Cin,)), 0.1 * hash_uniform("coab.be", (Cin,)))
    y = hip_conv(x, w, b, gn=gn, silu=True, tile=13)
    out.append(hashlib.sha256(y.numpy().tobytes()).hexdigest()[:16])
print(os.environ.get("ASYRP_CONV_OUT_PATCH", "default(14x14)"), " ".join(out))
