"""ctypes binding of ``include/npf_hip.h`` (the C ABI of the HIP path).

The library is loaded lazily.  There is no fallback: if ``libnpf_hip.so`` is missing or
does not load, every entry point raises ``RuntimeError`` -- the product path never runs
on anything but the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _build

NPF_MAX_OPS = 40
NPF_MAX_FEATURES = 512        # widest layer side of a chain, forward and backward
NPF_WGRAD_BLOCK = 256         # widest block of dW one wgrad job covers (wider layers are split into block jobs)
NPF_MAX_FUSED_ROW = 256       # features of one fused attention score row / LayerNorm row / bf16 layer (16-block instances)

# opcodes (enum npf_opcode)
OP_END, OP_LOAD_PT, OP_STORE_PT, OP_LOAD_ROWS, OP_STORE_ROWS, OP_LINEAR, OP_SOFTMAX, OP_ADD_PT, OP_MASK_POS, \
    OP_ADD_TASKVEC, OP_ROWDOT_PT, OP_SOFTMAX_BWD, OP_RELU, OP_SCALE, OP_STORE_TR, OP_LAYERNORM, OP_LAYERNORM_BWD, OP_LOAD_RM, OP_STORE_WB, OP_STORE_TRB, \
    OP_STORE_MASK, OP_MASK_BITS = range(22)
# weight modes (enum npf_wmode)
W_ROWMAJOR, W_PT_ROWS, W_PT_COLS = range(3)
F_RELU, F_ADD_PT, F_MASK_PT, F_ADD_RM, F_P16, F_MASK_BITS, F_STORE_IN, F_STORE_P16, F_STORE_BITS = 1, 2, 4, 8, 16, 32, 64, 128, 256


class NpfOp(C.Structure):
    _fields_ = [
        ("op", C.c_int32), ("i0", C.c_int32), ("i1", C.c_int32), ("i2", C.c_int32), ("i3", C.c_int32),
        ("flags", C.c_uint32), ("f0", C.c_float), ("i4", C.c_int32),
        ("p0", C.c_void_p), ("p1", C.c_void_p), ("p2", C.c_void_p),
        ("s0", C.c_int64), ("s1", C.c_int64), ("p3", C.c_void_p),
    ]


class NpfProgram(C.Structure):
    _fields_ = [
        ("n_ops", C.c_int32), ("n_tasks", C.c_int32), ("pts_per_task", C.c_int32), ("tiles_per_task", C.c_int32),
        ("wg_per_task", C.c_int32), ("reserved", C.c_int32 * 3),
        ("ops", NpfOp * NPF_MAX_OPS),
    ]


class NpfWgradJob(C.Structure):
    _fields_ = [
        ("dZ", C.c_void_p), ("A", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p), ("ldw", C.c_int64),
        ("N", C.c_int32), ("K", C.c_int32), ("per_task", C.c_int32), ("accumulate", C.c_int32),
        ("ldz", C.c_int32), ("lda", C.c_int32), ("ldo", C.c_int32), ("reserved", C.c_int32),
    ]


NPF_MAX_WGRAD_JOBS = 16
WGRAD_F32X6 = 16              # npf_wgrad_job_t.accumulate bit: fp32 contraction as six bf16 products per term (NPF_WGRAD_F32X6)
WGRAD_NO_H16 = 32             # ... and keep wgrad_x6_kernel for 256 x 256 jobs too (NPF_WGRAD_NO_H16, an A/B switch)


class NpfWprepJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n_rows", C.c_int32), ("n_cols", C.c_int32), ("ld", C.c_int32),
                ("kind", C.c_int32)]


NPF_MAX_WPREP_JOBS = 32
NPF_X6_MAX_LAYERS = 8


class NpfX6Layer(C.Structure):
    _fields_ = [("w_img", C.c_void_p), ("bias", C.c_void_p), ("mask", C.c_void_p), ("store_in", C.c_void_p),
                ("store_out", C.c_void_p), ("addend", C.c_void_p), ("store_bits", C.c_void_p), ("mask_bits", C.c_void_p),
                ("relu", C.c_int32), ("reserved", C.c_int32)]


NPF_X6_MAX_OPS = 12
X6_IN_RM, X6_ADD_RM = 1, 2  # npf_x6_op_t.reserved[0]: in_pt / addend are row-major [n_tasks][pts][F] tensors
X6_STORE_IN_F32, X6_STORE_OUT_F32 = 4, 8  # npf_b16_run: that store takes the fp32 value (PT32) instead of bf16(value) (PT16)


class NpfX6Op(C.Structure):
    _fields_ = [("in_pt", C.c_void_p), ("in_rows", C.c_void_p), ("in_w", C.c_void_p), ("in_b", C.c_void_p),
                ("pre_add", C.c_void_p), ("mask", C.c_void_p), ("mask_bits", C.c_void_p), ("sbwd_p", C.c_void_p),
                ("store_in", C.c_void_p), ("store_in_bits", C.c_void_p), ("w_img", C.c_void_p), ("w_task_stride", C.c_int64),
                ("bias", C.c_void_p), ("bias_task_stride", C.c_int64), ("addend", C.c_void_p), ("store_out", C.c_void_p),
                ("store_bits", C.c_void_p), ("in_n", C.c_int32), ("in_relu", C.c_int32), ("relu", C.c_int32),
                ("softmax_n", C.c_int32), ("softmax_scale", C.c_float), ("sbwd_scale", C.c_float), ("reserved", C.c_int32 * 2)]


assert C.sizeof(NpfX6Op) == 17 * 8 + 8 * 4
assert C.sizeof(NpfOp) == 80 and C.sizeof(NpfProgram) == 32 + 80 * NPF_MAX_OPS and C.sizeof(NpfWgradJob) == 72 and C.sizeof(NpfWprepJob) == 32

# name -> (restype, argtypes); must list every symbol declared in include/npf_hip.h
_i32, _i64, _p = C.c_int32, C.c_int64, C.c_void_p
SIGNATURES = {
    "npf_chain_run": (C.c_int, [C.POINTER(NpfProgram), _p]),
    "npf_wgrad_run": (C.c_int, [C.POINTER(NpfWgradJob), _i32, _i32, _i32, _p, _i64, _p]),
    "npf_wgrad_partials_bytes": (_i64, [C.POINTER(NpfWgradJob), _i32, _i32, _i32]),
    "npf_gauss_head_fwd": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _i32, _p, _p, _p, _p]),
    "npf_gauss_head_bwd": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _p, _i32, _p, _p, _p, _p, _p]),
    "npf_mc_objective_fwd": (C.c_int, [_p, _i32, _i32, _i32, _p, _i32, _p, _p]),
    "npf_mc_objective_bwd": (C.c_int, [_p, _i32, _i32, _i32, _p, _i32, _p, _p, _p, _p]),
    "npf_mean_agg_fwd": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "npf_mean_agg_bwd": (C.c_int, [_p, _i32, _i32, _i32, _p, _i32, _p]),
    "npf_pack_pt": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "npf_unpack_pt": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "npf_transpose": (C.c_int, [_p, _i32, _i32, _p, _p]),
    "npf_cast_bf16_weights": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "npf_prepare_weights": (C.c_int, [C.POINTER(NpfWprepJob), _i32, _p]),
    "npf_gather_points": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "npf_split_heads": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "npf_merge_heads": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "npf_mlp_x6_run": (C.c_int, [C.POINTER(NpfX6Layer), _i32, _p, _p, _i32, _i32, _p]),
    "npf_mlp_x6_run_rows": (C.c_int, [C.POINTER(NpfX6Layer), _i32, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _p]),
    "npf_x6_run": (C.c_int, [C.POINTER(NpfX6Op), _i32, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "npf_x6_run_ex": (C.c_int, [C.POINTER(NpfX6Op), _i32, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "npf_x6_task_images": (C.c_int, [_p, _i32, _i32, _i32, _p, _p, _p]),
    "npf_b16_run": (C.c_int, [C.POINTER(NpfX6Op), _i32, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "npf_b16_task_images": (C.c_int, [_p, _i32, _i32, _i32, _p, _p, _p]),
    "npf_mha_fwd": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "npf_mha_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p]),
    "npf_add_layernorm_fwd": (C.c_int, [_p, _p, _p, _p, C.c_float, _i32, _i32, _i32, _p, _p, _p]),
    "npf_add_layernorm_bwd": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p]),
    "npf_version": (C.c_int, []),
}

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    # NPF_HIP_LIB: development override to A/B two builds of the same ABI on one box
    return os.environ.get("NPF_HIP_LIB") or _build.LIB_PATH


def load() -> C.CDLL:
    """Load ``libnpf_hip.so`` (built by ``_build.build()``); raise loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"HIP extension {path} is missing. Build it with `python -m npf_gwwaveform_amd._build` "
            "(needs hipcc); there is no CPU fallback for the neural-process path."
        )
    try:
        lib = C.CDLL(path)
    except OSError as e:  # pragma: no cover - depends on the box
        raise RuntimeError(f"HIP extension {path} failed to load: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc} "
                           f"({'invalid argument/unsupported size' if rc == -1 else 'kernel launch error'})")


def stream_ptr(device: Optional[torch.device] = None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t: Optional[torch.Tensor], strided: bool = False) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("the HIP path takes device tensors only (got a CPU tensor); there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"the HIP path computes in fp32 (got {t.dtype})")
    if not strided and not t.is_contiguous():
        raise RuntimeError("tensor handed to the HIP path must be contiguous")
    return t.data_ptr()
