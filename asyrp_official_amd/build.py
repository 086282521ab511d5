"""Build libasyrp_hip.so (gfx950) in-tree with hipcc.  `python -m asyrp_official_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = PRODUCT_LIB = os.path.join(HERE, "libasyrp_hip.so")
BENCH_LIB = os.path.join(HERE, "libasyrp_hip_bench.so")
SOURCES = ["kernels.hip", "conv_f16x3.hip", "conv_out.hip", "conv_in.hip", "gemm1x1.hip", "attention.hip", "backward.hip", "engine.hip"]
BENCH_SOURCES = SOURCES + ["bench_hooks.hip"]   # the profiling library only (-DASYRP_BENCH_HOOKS): micro-benchmark / phase-stamp entry points
DEPS = BENCH_SOURCES + ["kernels.h", os.path.join("..", "..", "include", "asyrp.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-comment",
         "-fvisibility=hidden"]   # exports = the extern "C" entry points of include/asyrp.h (visibility pragma there)


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def needs_build(lib=PRODUCT_LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = DEPS if lib == BENCH_LIB else [d for d in DEPS if d != "bench_hooks.hip"]
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in deps)


def build_library(force=False, verbose=True, bench_hooks=False):
    """Compile the HIP sources into asyrp_official_amd/libasyrp_hip.so; returns the path.
    One hipcc -c per source, run concurrently (objects under csrc/build/, git-ignored), then one link.
    bench_hooks: the profiling library libasyrp_hip_bench.so instead (same sources with -DASYRP_BENCH_HOOKS: adds
    asyrp_op_conv_bench and the ablation instantiations of the main tile; used by scripts/conv_bench.py only)."""
    LIB = BENCH_LIB if bench_hooks else PRODUCT_LIB
    if not force and not needs_build(LIB):
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build", "bench" if bench_hooks else "")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + (["-DASYRP_BENCH_HOOKS"] if bench_hooks else [])
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, d)) for d in DEPS if not d.endswith(".hip"))

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        srcp = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), hdr_t):
            return obj
        cmd = [hipcc] + cflags + ["-c", srcp, "-o", obj]
        if verbose:
            print("[asyrp build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    sources = BENCH_SOURCES if bench_hooks else SOURCES
    with ThreadPoolExecutor(max_workers=len(sources)) as ex:
        objs = list(ex.map(compile_one, sources))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB + ".tmp"]
    if verbose:
        print("[asyrp build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, bench_hooks="--bench" in sys.argv))
