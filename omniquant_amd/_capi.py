"""ctypes binding of the gfx950 C-ABI library (include/oq_hip.h -> omniquant_amd/lib/liboq_hip.so).

There is NO fallback: if the library is missing or a call fails, an exception is raised.  PyTorch is used
only as the owner of device memory and of the HIP stream the kernels are enqueued on.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liboq_hip.so")

OQ_F32, OQ_F16, OQ_BF16 = 0, 1, 2
_DT = {torch.float32: OQ_F32, torch.float16: OQ_F16, torch.bfloat16: OQ_BF16}

_i64, _i32, _f32, _vp = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# name -> argtypes, exactly as declared in include/oq_hip.h (tests/test_capi_symbols.py cross-checks the header)
SIGNATURES = {
    "oq_fakequant_fwd": [_vp, _i32, _i64, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                         _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "oq_fakequant_bwd": [_vp, _i32, _i64, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                         _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "oq_gemm": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32,
                _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp],
    "oq_gemm_ws": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32,
                _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp, _i64, _vp],
    "oq_gemm_i8": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _vp],
    "oq_wgrad_group": [_vp, _i32, _vp, _i64, _vp],
    "oq_colsum": [_vp, _i32, _i64, _i64, _vp, _vp, _i64, _vp],
    "oq_norm_fwd": [_vp, _i32, _i64, _i64, _vp, _vp, _f32, _i32, _vp, _vp, _vp, _vp],
    "oq_norm_bwd": [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "oq_rope": [_vp, _vp, _i32, _i64, _i64, _i64, _vp, _vp, _i32, _vp],
    "oq_silu_mul_fwd": [_vp, _vp, _vp, _i32, _i64, _vp],
    "oq_silu_mul_bwd": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp],
    "oq_silu_mul_fwd_2d": [_vp, _vp, _vp, _i32, _i64, _i64, _i64, _vp],
    "oq_silu_mul_bwd_2d": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _vp],
    "oq_norm_quant_fwd": [_vp, _i32, _i64, _i64, _vp, _vp, _f32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "oq_norm_quant_bwd": [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "oq_rope_quant_fwd": [_vp, _i32, _i64, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "oq_rope_quant_bwd": [_vp, _i32, _i64, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp],
    "oq_qkv_rope_quant_fwd": [_vp, _i32, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp,
                              _vp],
    "oq_qkv_rope_quant_bwd": [_vp, _i32, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp],
    "oq_silu_mul_quant_fwd": [_vp, _vp, _i32, _i64, _i64, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "oq_silu_mul_quant_bwd": [_vp, _vp, _vp, _i32, _i32, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp],
    "oq_group_sum": [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _i64, _vp],
    "oq_relu_fwd": [_vp, _vp, _i32, _i64, _vp],
    "oq_relu_bwd": [_vp, _vp, _vp, _i32, _i64, _vp],
    "oq_softmax_fwd": [_vp, _vp, _i32, _i64, _i64, _f32, _vp, _i64, _i32, _vp],
    "oq_softmax_bwd": [_vp, _vp, _vp, _i32, _i64, _i64, _f32, _i32, _vp],
    "oq_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _f32, _i32, _vp],
    "oq_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _f32, _i32, _vp],
    "oq_attn_fwd_grid": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _f32, _i32, _vp],
    "oq_attn_bwd_grid": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _f32,
                         _i32, _vp],
    "oq_mse_fwd_bwd": [_vp, _i32, _vp, _vp, _i32, _i64, _f32, _vp, _vp, _vp],
    "oq_add": [_vp, _vp, _vp, _i32, _i64, _vp],
    "oq_scale": [_vp, _f32, _vp, _i32, _i64, _vp],
    "oq_copy_samples": [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "oq_gradnorm": [_vp, _i64, _vp, _vp, _vp],
    "oq_adamw": [_vp, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _vp],
    "oq_adamw_step": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32, _i32, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _vp,
                      _vp, _vp, _i64, _vp],
    "oq_truncate": [_vp, _i64, _f32, _vp],
    "oq_act_stats": [_vp, _i32, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _i64, _vp],
    "oq_pack_weights": [_vp, _i32, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp],
    "oq_sum_vectors": [_i32, _vp, _vp, _vp, _vp, _vp],
    "oq_cast": [_vp, _i32, _vp, _i32, _i64, _vp],
    "oq_let_vectors_fwd": [_i64] + [_vp] * 28,
    "oq_let_vectors_bwd": [_i64] + [_vp] * 39,
}



class FakeQuantFwdArgs(ctypes.Structure):
    """oq_fakequant_fwd_args of include/oq_hip.h (same fields, same order as oq_fakequant_fwd's parameters)."""
    _fields_ = [("w", _vp), ("w_dtype", _i32), ("rows", _i64), ("cols", _i64), ("seg", _i64), ("nbits", _i32),
                ("symmetric", _i32), ("col_mul", _vp), ("row_div", _vp), ("row_mul", _vp), ("shift", _vp), ("up", _vp),
                ("low", _vp), ("y", _vp), ("y_dtype", _i32), ("scale", _vp), ("zp", _vp), ("xmin", _vp), ("xmax", _vp),
                ("wshift", _vp), ("codes", _vp), ("csum", _vp)]


class WgradItem(ctypes.Structure):
    """item of oq_wgrad_group (include/oq_hip.h): gw[N][K] = gy[T][N]^T x[T][K]."""
    _fields_ = [("gy", _vp), ("x", _vp), ("gw", _vp), ("N", _i64), ("K", _i64), ("T", _i64), ("ld_gy", _i64), ("ld_x", _i64),
                ("ld_gw", _i64)]


class FakeQuantBwdArgs(ctypes.Structure):
    """oq_fakequant_bwd_args of include/oq_hip.h."""
    _fields_ = [("w", _vp), ("w_dtype", _i32), ("rows", _i64), ("cols", _i64), ("seg", _i64), ("nbits", _i32),
                ("symmetric", _i32), ("col_mul", _vp), ("row_div", _vp), ("row_mul", _vp), ("shift", _vp), ("up", _vp),
                ("low", _vp), ("xmin", _vp), ("xmax", _vp), ("g", _vp), ("g_dtype", _i32), ("g_wshift", _vp),
                ("g_up", _vp), ("g_low", _vp), ("gx", _vp), ("gx_dtype", _i32), ("g_col_mul", _vp), ("g_shift", _vp),
                ("g_row_div", _vp), ("g_row_mul", _vp), ("workspace", _vp), ("workspace_floats", _i64)]


SIGNATURES["oq_fakequant_fwd_multi"] = [_vp, _i32, _vp]
SIGNATURES["oq_fakequant_bwd_multi"] = [_vp, _i32, _vp]

# functions returning a size instead of an error code
SIZE_FUNCS = {"oq_fakequant_bwd_workspace": [_i64, _i64], "oq_norm_bwd_workspace": [_i64, _i64],
              "oq_attn_supported": [_i32, _i64, _i32, _i32], "oq_wgrad_group_workspace": [_vp, _i32], "oq_act_stats_workspace": [_i64, _i64],
              "oq_colsum_workspace": [_i64, _i64], "oq_gemm_workspace": [_i64, _i64, _i64, _i32, _i64, _i32], "oq_rope_quant_supported": [_i32, _i32],
              "oq_norm_quant_supported": [_i32, _i64], "oq_norm_quant_bwd_workspace": [_i64, _i64],
              "oq_fakequant_codes_supported": [_i64, _i64, _i32, _i32]}

_lib = None


class OQError(RuntimeError):
    pass


def load():
    """Load liboq_hip.so (built by __graft_entry__.build() / `make -C omniquant_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OQError(f"{LIB_PATH} is missing: the HIP extension is not built "
                      "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.oq_version.restype = ctypes.c_int
    lib.oq_last_error.restype = ctypes.c_char_p
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    for name, argtypes in SIZE_FUNCS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int64
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise OQError(f"{name} failed (rc={rc}): {lib.oq_last_error().decode()}")


def size_call(name, *args):
    return int(getattr(load(), name)(*args))


def dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise OQError(f"dtype {t.dtype} is not supported by the HIP path")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must live on the GPU and be contiguous."""
    if t is None:
        return None
    if not t.is_cuda:
        raise OQError("the OmniQuant HIP path needs GPU tensors (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise OQError("non-contiguous tensor passed to the HIP path")
    return t.data_ptr()


def fptr(t):
    """float32 device pointer."""
    if t is not None and t.dtype != torch.float32:
        raise OQError(f"expected a float32 tensor, got {t.dtype}")
    return ptr(t)


def stream():
    return torch.cuda.current_stream().cuda_stream
