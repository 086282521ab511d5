"""ctypes binding of libasyrp_hip.so (the C ABI in include/asyrp.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the compute
entry points raises.  torch is imported first so the process shares torch's HIP runtime
(libamdhip64.so.7 is resolved by SONAME to the already-loaded copy).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: one HIP runtime per process)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libasyrp_hip.so")

MAX_LEVELS = 8
FAMILY_DDPM, FAMILY_IDDPM = 0, 1
MATH_F16X3, MATH_F32, MATH_F16 = 0, 1, 2
CONV_MATH = {"f16x3": MATH_F16X3, "f32": MATH_F32, "f16": MATH_F16}   # f16 = the single-product fast mode (not fp32-equivalent)


class AsyrpConfig(C.Structure):
    _fields_ = [("family", C.c_int32), ("resolution", C.c_int32), ("in_channels", C.c_int32),
                ("out_channels", C.c_int32), ("ch", C.c_int32), ("n_levels", C.c_int32),
                ("ch_mult", C.c_int32 * MAX_LEVELS), ("num_res_blocks", C.c_int32), ("n_attn", C.c_int32),
                ("attn_resolutions", C.c_int32 * MAX_LEVELS), ("num_head_channels", C.c_int32),
                ("n_delta", C.c_int32), ("conv_math", C.c_int32), ("num_classes", C.c_int32), ("nominal_batch", C.c_int32), ("reserved", C.c_int32 * 5)]


_P, _F, _I = C.c_void_p, C.c_float, C.c_int
ABI_VERSION = 8   # include/asyrp.h ASYRP_ABI_VERSION

_SIGS = {
    "asyrp_abi_version": (C.c_int, []),
    "asyrp_last_error": (C.c_char_p, []),
    "asyrp_create": (C.c_int, [C.POINTER(_P), C.POINTER(AsyrpConfig), _I, _I]),
    "asyrp_destroy": (None, [_P]),
    "asyrp_load_param": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "asyrp_set_schedule": (C.c_int, [_P, _P, _I]),
    "asyrp_set_temb_freqs": (C.c_int, [_P, _P, _I]),
    "asyrp_finalize_params": (C.c_int, [_P]),
    "asyrp_num_params": (C.c_int, [_P]),
    "asyrp_param_info": (C.c_int, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(_I)]),
    "asyrp_unet_forward": (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P, _P]),
    "asyrp_ddim_step": (C.c_int, [_P, _P, _I, _I, _I, _F, _P, _I, _I, _I, _P, _I, _I, _P, _I, _F, _I, _P, _P, _P, _P, _P]),
    "asyrp_run_edit": (C.c_int, [_P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P, _P, _P]),
    "asyrp_run_inversion": (C.c_int, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "asyrp_get_temb": (C.c_int, [_P, _P, _I, _P, _P]),
    "asyrp_train_forward": (C.c_int, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, C.POINTER(C.c_int64), _P]),
    "asyrp_train_backward": (C.c_int, [_P, C.c_int64, _P, _I, C.POINTER(C.c_char_p), C.POINTER(_P), _P]),
    "asyrp_train_discard": (None, [_P, C.c_int64]),
    "asyrp_sampler_update": (C.c_int, [_I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "asyrp_device_bytes": (C.c_int64, [_P]),
    "asyrp_profile_table": (C.c_int, [_P, _I, _P, _P, _P, _P, _P]),
    "asyrp_profile_enable": (C.c_int, [_P, _I]),
    "asyrp_profile_read": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]),
    "asyrp_op_conv2d": (C.c_int, [_I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _F, _I, _P, _P,
                                  _P, _I, _I, _P]),
    "asyrp_op_conv2d_stats": (C.c_int, [_I, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P]),
    "asyrp_op_resblock_tail": (C.c_int, [_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _F, _P, _P]),
    "asyrp_op_attention": (C.c_int, [_I, _P, _I, _I, _I, _I, _I, _P, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)

# the profiling library (same sources + -DASYRP_BENCH_HOOKS): scripts/conv_bench.py only, never loaded by the package
BENCH_LIB_PATH = os.environ.get("ASYRP_BENCH_LIB") or os.path.join(_HERE, "libasyrp_hip_bench.so")   # (override: A/B builds of the profiling library)
BENCH_SIGS = {"asyrp_op_conv_bench": (C.c_int, [_I] * 16 + [C.POINTER(C.c_float), _P]),
              "asyrp_op_conv_stamps": (C.c_int, [_I] * 16 + [C.POINTER(C.c_float), _P, _I]),
              "asyrp_op_attention_phases": (C.c_int, [_I] * 7 + [C.POINTER(C.c_float), _P, _P]),
              "asyrp_op_gemm1x1_phases": (C.c_int, [_I] * 9 + [C.POINTER(C.c_float), _P, _P])}

_lib = None


def load_bench():
    """The profiling build (python -m asyrp_official_amd.build --bench): every product symbol + asyrp_op_conv_bench."""
    if not os.path.exists(BENCH_LIB_PATH):
        raise RuntimeError(f"{BENCH_LIB_PATH} not found: build it with `python -m asyrp_official_amd.build --bench`")
    lib = C.CDLL(BENCH_LIB_PATH)
    for name, (res, args) in {**_SIGS, **BENCH_SIGS}.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Load the shared library (once) and declare every signature.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    if os.environ.get("ASYRP_LIBRARY") == "bench":
        # explicit opt-in for A/B runs: the profiling build honours the ASYRP_* kernel switches, the product library reads none.
        # Said out loud, so that a variable left over in a shell cannot silently change which kernels a product run executes.
        import warnings
        path = BENCH_LIB_PATH
        warnings.warn(f"ASYRP_LIBRARY=bench: loading the PROFILING build {path} in place of the product library; it honours the "
                      "ASYRP_* kernel A/B switches of the environment", RuntimeWarning, stacklevel=2)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the Asyrp HIP engine has no CPU/PyTorch fallback. "
            "Build it with `python -m asyrp_official_amd.build` (needs hipcc, targets gfx950).")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)   # AttributeError here == ABI mismatch; let it propagate
        fn.restype = res
        fn.argtypes = args
    if lib.asyrp_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path} implements ABI v{lib.asyrp_abi_version()}, this package binds v{ABI_VERSION}: "
                           "rebuild with `python -m asyrp_official_amd.build`")
    _lib = lib
    lib._asyrp_path = path
    return lib


class AsyrpError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = load().asyrp_last_error().decode(errors="replace")
        raise AsyrpError(f"asyrp engine error {rc}: {msg}")
