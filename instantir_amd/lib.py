"""ctypes binding of libinstantir_hip.so (the C ABI declared in include/instantir_hip.h).

The library is the product: there is no PyTorch / CPU fallback.  `load()` raises if the shared
object is missing or does not export a declared symbol, and every wrapper raises on a non-zero
return code.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libinstantir_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "instantir_hip.h")

EPI_PLAIN, EPI_GEGLU, EPI_SFT, EPI_XATTN = 0, 1, 2, 3
ACT_NONE, ACT_SILU, ACT_GELU, ACT_QUICKGELU = 0, 1, 2, 3


class AttnKV(C.Structure):
    _fields_ = [
        ("K", C.c_void_p), ("ldk", C.c_int64), ("k_batch_stride", C.c_int64),
        ("Vt", C.c_void_p), ("ldvt", C.c_int64), ("vt_batch_stride", C.c_int64),
        ("Tkv", C.c_int32),
    ]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("W", C.c_void_p),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("bias", C.c_void_p),
        ("rowbias", C.c_void_p), ("ldrb", C.c_int64), ("rows_per_rb", C.c_int32),
        ("res", C.c_void_p), ("ldr", C.c_int64),
        ("epi", C.c_int32), ("act", C.c_int32),
        ("out_scale", C.c_float),
        ("tile", C.c_int32),
        ("prefetch", C.c_void_p), ("prefetch_bytes", C.c_int64),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64),
        ("Ct", C.c_void_p), ("ldct", C.c_int64), ("tr_from", C.c_int32),
        ("dtype", C.c_int32), ("c_f32", C.c_int32),
        ("wscale", C.c_void_p),
        ("ln_stats_out", C.c_void_p), ("ln_stats_in", C.c_void_p), ("ln_parts", C.c_int32), ("ln_part_cols", C.c_int32),
        ("ln_eps", C.c_float), ("ln_colsum", C.c_void_p),
        ("gn_stats_out", C.c_void_p),
        ("xattn_kv", C.POINTER(AttnKV)), ("xattn_tq", C.c_int32),
        ("a_fp8", C.c_int32), ("a_scale", C.c_float), ("c_fp8", C.c_int32),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("ldx", C.c_int64),
        ("R", C.c_int32), ("H", C.c_int32), ("Wd", C.c_int32), ("Cin", C.c_int32),
        ("Wt", C.c_void_p),
        ("Y", C.c_void_p), ("ldy", C.c_int64),
        ("Cout", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("upsample", C.c_int32),
        ("bias", C.c_void_p),
        ("rowbias", C.c_void_p), ("ldrb", C.c_int64), ("rows_per_rb", C.c_int32),
        ("res", C.c_void_p), ("ldr", C.c_int64),
        ("epi", C.c_int32), ("act", C.c_int32),
        ("out_scale", C.c_float),
        ("tile", C.c_int32),
        ("zero_page", C.c_void_p),
        ("x_img_stride", C.c_int64), ("y_img_rows", C.c_int32), ("res_img_rows", C.c_int32), ("pad_mode", C.c_int32),
        ("prefetch", C.c_void_p), ("prefetch_bytes", C.c_int64),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64),
        ("dtype", C.c_int32),
        ("gn_stats_out", C.c_void_p),
    ]


class AdaLNJob(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("Y", C.c_void_p), ("shift", C.c_void_p), ("scale", C.c_void_p),
        ("ldx", C.c_int64), ("ldy", C.c_int64), ("C", C.c_int32), ("transposed", C.c_int32),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("ldq", C.c_int64), ("q_batch_stride", C.c_int64),
        ("O", C.c_void_p), ("ldo", C.c_int64), ("o_batch_stride", C.c_int64),
        ("batch", C.c_int32), ("heads", C.c_int32), ("Tq", C.c_int32), ("nseg", C.c_int32),
        ("scale", C.c_float),
        ("kv", AttnKV * 2),
        ("causal", C.c_int32),
        ("q_prescaled", C.c_int32),
        ("o_fp8", C.c_int32),
    ]


_P, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# symbol -> (restype, argtypes); must list every function declared in include/instantir_hip.h
SIGNATURES = {
    "iir_gemm_f16": (C.c_int, [C.POINTER(GemmDesc), _P]),
    "iir_gemm_splitk_workspace_bytes": (C.c_int64, [_I32, _I32]),
    "iir_gemm_uses_splitk": (C.c_int, [_I32, _I32, _I32, _I64]),
    "iir_gemm_pick_tile": (C.c_int, [_I32, _I32, _I32, _I32]),
    "iir_gemm_resolve_tile": (C.c_int, [_P]),
    "iir_gemm_gn_supported": (C.c_int, [_I32, _I32, _I32, _I32]),
    "iir_gemm_fp8_out_supported": (C.c_int, [_I32, _I32, _I32, _I32]),
    "iir_gemm_tile_bn": (C.c_int, [_I32]),
    "iir_gemm_ln_parts": (C.c_int, [_I32, _I32, _I32]),
    "iir_conv2d_nhwc_f16": (C.c_int, [C.POINTER(ConvDesc), _P]),
    "iir_attention_d64_f16": (C.c_int, [C.POINTER(AttnDesc), _P]),
    "iir_groupnorm_nhwc_f16": (C.c_int, [_P, _I64, _P, _I64, _I32, _I32, _I32, _I32, _P, _P, _F, _I32, _P, _I64, _P]),
    "iir_groupnorm_nhwc": (C.c_int, [_P, _I64, _P, _I64, _I32, _I32, _I32, _I32, _P, _P, _F, _I32, _P, _I64, _I32, _P]),
    "iir_groupnorm_from_partials": (C.c_int, [_P, _I64, _P, _I64, _P, _I64, _I32, _I32, _I32, _I32, _P, _P, _F, _I32, _P, _I64, _I32, _P]),
    "iir_groupnorm_workspace_bytes": (C.c_int64, [_I32, _I32]),
    "iir_layernorm_f16": (C.c_int, [_P, _I64, _P, _I64, _I32, _I32, _P, _P, _F, _P, _P, _I64, _I32, _I32, _I32, _I64, _P]),
    "iir_adaln_batch_f16": (C.c_int, [_P, _I32, _I32, _I32, _F, _I64, _I32, _I32, _I64, _P]),
    "iir_softmax_rows_f16": (C.c_int, [_P, _I64, _I32, _I32, _P]),
    "iir_softmax_rows_f32": (C.c_int, [_P, _I64, _P, _I64, _I32, _I32, _I32, _P]),
    "iir_sinusoid_f16": (C.c_int, [_P, _I32, _I32, _I32, _P, _I64, _I32, _P]),
    "iir_silu_f16": (C.c_int, [_P, _P, _I64, _P]),
    "iir_copy_add_f16": (C.c_int, [_P, _I64, _P, _I64, _I64, _I64, _I32, _P, _I64, _P, _I32, _P]),
    "iir_pack_latent": (C.c_int, [_P, _I32, _I32, _I32, _P, _I64, _I32, _F, _P]),
    "iir_unpack_latent": (C.c_int, [_P, _I64, _I32, _I32, _I32, _P, _P]),
    "iir_pack_latent_t": (C.c_int, [_P, _I32, _I32, _I32, _P, _I64, _I32, _F, _I32, _P]),
    "iir_unpack_latent_t": (C.c_int, [_P, _I64, _I32, _I32, _I32, _P, _I32, _P]),
    "iir_sched_step": (C.c_int, [_P, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "iir_cfg_rescale_factor": (C.c_int, [_P, _I64, _I32, _I32, _I32, _P, _F, _P, _P]),
    "iir_lcm_step": (C.c_int, [_P, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _I64, _P, _P]),
    "iir_sched_step_f32": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _P]),
    "iir_axpby_f32": (C.c_int, [_P, _P, _P, _I64, _P, _P]),
    "iir_prefetch": (C.c_int, [_P, _I64, _I32, _P]),
    "iir_blend_tiles_f32": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "iir_transpose_f16": (C.c_int, [_P, _I64, _I32, _I32, _P, _I64, _I32, _P]),
    "iir_timing_event_create": (C.c_void_p, []),
    "iir_timing_event_destroy": (None, [_P]),
    "iir_timing_arm": (C.c_int, [_P, _P]),
    "iir_timing_elapsed_us": (C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    "iir_abi_version": (C.c_int, []),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def declared_symbols(header_path: str = HEADER_PATH):
    """Function names declared in include/instantir_hip.h (used by the load check and the CPU tests)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(iir_[a-z0-9_]+)\s*\(", text)))


def load():
    """dlopen the in-tree library and bind every declared symbol; raises HipLibraryError otherwise."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; importing it first makes our library bind to that same
    # runtime instance (one HIP runtime per process), otherwise torch streams/pointers are foreign to us.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with instantir_amd/csrc/build.sh (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the denoising path.")
    lib = C.CDLL(LIB_PATH)
    for name in declared_symbols():
        if name not in SIGNATURES:
            raise HipLibraryError(f"header declares {name} but lib.py has no signature for it")
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = SIGNATURES[name]
    if lib.iir_abi_version() != 1:
        raise HipLibraryError("ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        raise HipLibraryError(f"{what} failed with code {rc} ({'invalid argument' if rc == -1 else 'launch failure'})")
