"""Host runtime of the HIP chain kernel: describe a per-point computation once, get the
forward program, the backward (dgrad) program and the weight-gradient jobs from it.

A :class:`Chain` is a straight-line computation on the register-resident activation of
every point (``cur`` in ``csrc/chain_kernel.hip``): inputs, Linear layers (shared weights
or the task's keys / values as weights), softmax, adds.  ``Chain.run()`` executes it through
``npf_chain_run`` inside one ``torch.autograd.Function``; the backward pass is the chain
walked in reverse (relu masks, transposed weights) plus one batched ``npf_wgrad_run``.

Tensors between chains are "PT32" tensors: ``[n_tasks, tiles, F/4, 32, 4]`` fp32 (see
``include/npf_hip.h``).  Nothing here computes on the CPU: every step is a kernel launch.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L


# bench.py sets this to a list to time every launch with HIP events on the launch stream:
# entries are (kernel, algorithmic flops, start event, end event, algorithmic HBM bytes)
PROFILE = None
# tests set this to a list: every chain execution then leaves a record of the tensors it read and wrote -- ("fwd", chain,
# inputs, saved tensors by (step, role), outputs, bf16) and ("bwd", chain, incoming gradients, buffers by (step, role), wgrad
# jobs, bf16) -- so that a test can check each step against the kernel's OWN stored inputs (tests/teacher.py)
TRACE = None
DEBUG_ABLATE = 0  # development only: chain_kernel ablation bits (tools/microbench.py)
# tests / tools: 1 = 64-point workgroups, 2 = 128-point (paired) workgroups, 0 = the library's choice
FORCE_WG = int(os.environ.get("NPF_FORCE_WG", "0"))
PT16_INTERNAL = os.environ.get("NPF_NO_PT16", "0") != "1"  # bf16 mode: backward-only tensors as bf16 tiles (debug switch)
MASK_BITS = os.environ.get("NPF_NO_MASK_BITS", "0") != "1"  # bf16 mode: ReLU masks of the backward pass as bits (debug switch)
# bf16 mode: a layer with <= 32 inputs and 256 outputs right behind LOAD_ROWS (x-encoder first layers, the dgrad of a narrow
# output layer) costs twice a 256 -> 256 layer on the generic slab loop (one DMA round trip per slab); with its image
# rows zero-padded to 256 inputs -- the other input registers ARE zero behind LOAD_ROWS -- it is a pipelined layer
PAD_SMALL_K = os.environ.get("NPF_NO_PAD_SMALL_K", "0") != "1"
DUMP_PROGRAMS = os.environ.get("NPF_DUMP_PROGRAMS", "0") == "1"
FUSE_STORES = os.environ.get("NPF_NO_FUSED_STORE", "0") != "1"  # bf16 mode: STORE_PT + LINEAR -> LINEAR | F_STORE_IN (debug switch)
# fp32 mode: weight / key / value gradients on the bf16 matrix pipe -- every fp32 operand split exactly into three bf16
# terms, six cross products accumulated in fp32 (NPF_WGRAD_F32X6: the result agrees with the v_mfma_f32_16x16x4_f32
# kernel to fp32 summation-order noise, at 6/16 of its matrix-pipe time).  NPF_NO_WGRAD_X6=1: the native fp32 kernel.
WGRAD_X6 = os.environ.get("NPF_NO_WGRAD_X6", "0") != "1"
X6_NARROW_NATIVE = os.environ.get("NPF_X6_ALL", "0") != "1"  # narrow jobs of an fp32 launch keep the fp32-MFMA kernel
# launches whose jobs are all 256 x 256 run on wgrad_h16_kernel (every operand value split once per workgroup);
# NPF_NO_WGRAD_H16=1: wgrad_x6_kernel for those too
WGRAD_H16 = os.environ.get("NPF_NO_WGRAD_H16", "0") != "1"


# Compute mode of the MLP chains ("fp32" | "bf16"), see set_compute_dtype.  In "bf16" every chain made only
# of shared-weight Linear layers and elementwise steps multiplies in bf16 (weights and layer inputs rounded
# to bf16 at the MFMA, fp32 accumulation / bias / activations in HBM) and every weight / key / value
# gradient is a bf16-product contraction; attention chains, LayerNorm and row-major inference inputs stay
# on the fp32 instances.
COMPUTE_DTYPE = "fp32"
_BF16_STEPS = {"input_pt", "input_rows", "linear", "add_pt", "add_taskvec", "tap", "output_pt", "output_rows", "store_tr",
               "store_wb", "store_trb", "softmax"}
_BF16_ATTN_STEPS = {"attn_scores", "attn_values"}  # eligible when the chain was given bf16 images of keys / values


def set_compute_dtype(dtype: str) -> None:
    """"fp32" (default, the parity-gated path) or "bf16" (BASELINE config 3)."""
    global COMPUTE_DTYPE
    if dtype not in ("fp32", "bf16"):
        raise ValueError(f"unknown compute dtype {dtype!r}")
    COMPUTE_DTYPE = dtype


def _bf16_image(W: torch.Tensor, transposed: bool) -> torch.Tensor:
    """``cast_bf16_weights`` image of a weight tensor.  Not cached: the tensors that reach this point are
    per-call aliases and temporaries (slices, zero-padded fan-ins) whose addresses get reused, so neither
    the address nor the version counter identifies their content; a cast is a few microseconds."""
    return cast_bf16_weights(W, transposed=transposed)


def pad32(n: int) -> int:
    return (n + 31) // 32 * 32


def tiles_of(pts: int) -> int:
    return (pts + 31) // 32


def pt_shape(n_tasks: int, pts: int, F: int):
    return (n_tasks, tiles_of(pts), pad32(F) // 4, 32, 4)


def _pad_vec(v: torch.Tensor) -> torch.Tensor:
    """A per-feature vector zero-padded to a multiple of 32 floats (LayerNorm gamma / beta)."""
    n = v.shape[0]
    if n == pad32(n) and v.is_contiguous() and v.data_ptr() % 16 == 0:
        return v
    out = torch.zeros(pad32(n), dtype=torch.float32, device=v.device)
    out[:n] = v
    return out


def pt16_shape(n_tasks: int, pts: int, F: int):
    """PT16: the bf16 tile layout of the bf16 compute mode -- [n_tasks, tiles, F/8 rows, 32 points, 8 features], the 8
    features of row 4 s + g being {32 s + 4 g + i} and {32 s + 16 + 4 g + i}, i < 4 (a lane's values of two adjacent
    16-feature blocks: one 16-byte access per lane and 32-feature group)."""
    return (n_tasks, tiles_of(pts), pad32(F) // 8, 32, 8)


def pt16_empty(n_tasks: int, pts: int, F: int, device) -> torch.Tensor:
    return torch.empty(pt16_shape(n_tasks, pts, F), dtype=torch.bfloat16, device=device)


def ptm_empty(n_tasks: int, pts: int, F: int, device) -> torch.Tensor:
    """PTM: a ReLU mask as bits (bf16 compute mode, ``NPF_OP_STORE_MASK``) -- [n_tasks, tiles, ceil(F / 128) words, 4 lane
    groups, 32 points] int32, 32 bytes per point and 256 features."""
    return torch.empty((n_tasks, tiles_of(pts), (pad32(F) + 127) // 128, 4, 32), dtype=torch.int32, device=device)


def pt_empty(n_tasks: int, pts: int, F: int, device) -> torch.Tensor:
    return torch.empty(pt_shape(n_tasks, pts, F), dtype=torch.float32, device=device)


@dataclass
class PTensor:
    """A per-point activation between chains: the PT32 tensor together with everything a later stage has to
    know about it.  The geometry travels with the data (a PT32 tensor's own shape only gives the padded
    sizes), and so do the companion copies an attention chain streams as per-task weights."""
    t: torch.Tensor                       # PT32 [n_tasks, tiles, pad32(F)/4, 32, 4] fp32 (carries the autograd graph)
    pts: int                              # valid points per task
    F: int                                # valid features
    tr: Optional[torch.Tensor] = None     # feature-major copy [n_tasks, F, 32*tiles] (``Chain.store_tr``)
    img: Optional[Tuple[torch.Tensor, torch.Tensor]] = None  # bf16 row / transposed images (bf16 compute mode)
    proj: Optional[torch.Tensor] = None   # PT32: the attender's query projection of these points, made by their producer (x6.xenc_proj)

    def __post_init__(self):
        want = pt_shape(self.t.shape[0], self.pts, self.F)
        if tuple(self.t.shape) != want:
            raise ValueError(f"PT32 tensor of shape {tuple(self.t.shape)} does not hold {self.pts} points x {self.F} "
                             f"features per task (expected {want})")

    @property
    def n_tasks(self) -> int:
        return self.t.shape[0]


# ---------------------------------------------------------------------------------------
# low level: program assembly + launch
# ---------------------------------------------------------------------------------------
class Program:
    """A list of ``npf_op_t`` + geometry; ``launch()`` calls ``npf_chain_run``."""

    def __init__(self, n_tasks: int, pts_per_task: int, wg_per_task: bool):
        self.n_tasks, self.pts, self.wg_per_task = n_tasks, pts_per_task, wg_per_task
        self.ops: List[L.NpfOp] = []
        self.keep: list = []  # tensors referenced by raw pointer must outlive the launch call
        self.bf16 = False     # every LINEAR takes a bf16 weight image (linear_bf16)
        self._pad_k = 0       # sum over zero-padded layers (PAD_SMALL_K) of (padded K - K) * N: not algorithmic work

    def _op(self, **kw) -> None:
        if len(self.ops) >= L.NPF_MAX_OPS:
            raise RuntimeError(f"chain program longer than NPF_MAX_OPS={L.NPF_MAX_OPS}")
        o = L.NpfOp()
        for k, v in kw.items():
            setattr(o, k, v)
        self.ops.append(o)

    def _p(self, t: Optional[torch.Tensor]):
        if t is None:
            return None
        self.keep.append(t)
        return L.ptr(t)

    def _pt(self, t: torch.Tensor):
        """(pointer, flags) of a PT operand: fp32 PT32, or a bf16 PT16 tensor (bf16 compute mode)."""
        if t.dtype == torch.bfloat16:
            if not t.is_cuda or not t.is_contiguous():
                raise RuntimeError("PT16 tensors must be contiguous device tensors")
            self.keep.append(t)
            self.bf16 = True
            return t.data_ptr(), L.F_P16
        return self._p(t), 0

    def load_pt(self, t, F, modulus=0):
        p, fl = self._pt(t)
        self._op(op=L.OP_LOAD_PT, i0=pad32(F), i4=modulus, p0=p, flags=fl)

    def load_rm(self, t, F, modulus=0):
        """cur <- row-major [n_tasks (or modulus), pts, F] tensor, F a multiple of 32."""
        self._op(op=L.OP_LOAD_RM, i0=F, i4=modulus, p0=self._p(t))

    def store_pt(self, t, F):
        p, fl = self._pt(t)
        self._op(op=L.OP_STORE_PT, i0=pad32(F), p0=p, flags=fl)

    def add_pt(self, t, F, relu=False, modulus=0):
        p, fl = self._pt(t)
        self._op(op=L.OP_ADD_PT, i0=pad32(F), i1=int(relu), i4=modulus, p0=p, flags=fl)

    def mask_pos(self, t, F):
        last = self.ops[-1] if self.ops else None
        if (last is not None and last.op == L.OP_LINEAR and not (last.flags & (L.F_ADD_PT | L.F_MASK_PT | L.F_RELU))
                and pad32(last.i1) == pad32(F)):
            # fuse the relu-backward mask into the producing layer's epilogue: the mask tile is
            # then prefetched under that layer's MFMAs instead of being waited for afterwards
            p, fl = self._pt(t)
            last.flags |= L.F_MASK_PT | fl
            last.p2 = p
            last.i4 = 0
            return
        p, fl = self._pt(t)
        self._op(op=L.OP_MASK_POS, i0=pad32(F), p0=p, flags=fl)

    def store_mask(self, m, F):
        """PTM tensor ``m`` <- (cur > 0) as bits (bf16 programs)."""
        self.bf16 = True
        self.keep.append(m)
        last = self.ops[-1] if self.ops else None
        if (FUSE_STORES and last is not None and last.op == L.OP_LINEAR and pad32(last.i1) == pad32(F) and F <= 256
                and (last.flags & L.F_RELU) and last.p2 is None
                and not (last.flags & (L.F_ADD_PT | L.F_MASK_PT | L.F_ADD_RM | L.F_MASK_BITS | L.F_STORE_BITS))):
            last.flags |= L.F_STORE_BITS  # the layer's epilogue shifts the bits in as it goes (NPF_F_STORE_BITS)
            last.p2 = m.data_ptr()
            return
        self._op(op=L.OP_STORE_MASK, i0=pad32(F), p0=m.data_ptr())

    def mask_bits(self, m, F):
        """cur <- bit ? cur : 0 with the PTM tensor ``m``; fused into the producing layer's epilogue when there is one."""
        self.bf16 = True
        self.keep.append(m)
        last = self.ops[-1] if self.ops else None
        if (last is not None and last.op == L.OP_LINEAR and pad32(last.i1) == pad32(F)
                and not (last.flags & (L.F_ADD_PT | L.F_MASK_PT | L.F_RELU | L.F_MASK_BITS))):
            last.flags |= L.F_MASK_BITS
            last.p2 = m.data_ptr()
            last.i4 = 0
            return
        self._op(op=L.OP_MASK_BITS, i0=pad32(F), p0=m.data_ptr())

    def scale(self, f: float):
        self._op(op=L.OP_SCALE, f0=float(f))

    def mask_sign(self, t, F):
        """cur <- t > 0 ? cur : 0 as an op of its own (dropout: ``t`` holds +1 / -1)."""
        p, fl = self._pt(t)
        self._op(op=L.OP_MASK_POS, i0=pad32(F), p0=p, flags=fl)

    def rowdot_pt(self, t, F):
        p, fl = self._pt(t)
        self._op(op=L.OP_ROWDOT_PT, i0=pad32(F), p0=p, flags=fl)

    def softmax_bwd(self, t, F, scale):
        p, fl = self._pt(t)
        self._op(op=L.OP_SOFTMAX_BWD, i0=pad32(F), f0=scale, p0=p, flags=fl)

    def store_tr(self, t, F, ld):
        self._op(op=L.OP_STORE_TR, i0=F, i1=ld, p0=self._p(t))

    def load_rows(self, t, kd, modulus=0):
        self._op(op=L.OP_LOAD_ROWS, i0=kd, i4=modulus, p0=self._p(t))

    def store_rows(self, t, nd):
        self._op(op=L.OP_STORE_ROWS, i0=nd, p0=self._p(t))

    def softmax(self, n_valid, scale, mode=0, stats=None):
        """mode 1: also store (row max, row sum) to ``stats`` [n_tasks, pts, 2]; mode 2: take them from it."""
        self._op(op=L.OP_SOFTMAX, i0=n_valid, i1=mode, f0=scale, p0=self._p(stats))

    def layernorm(self, gamma, beta, F, eps):
        self._op(op=L.OP_LAYERNORM, i0=F, f0=eps, p0=self._p(gamma), p1=self._p(beta))

    def layernorm_bwd(self, x_saved, gamma, F, eps, dy_xhat=None):
        self._op(op=L.OP_LAYERNORM_BWD, i0=F, f0=eps, p0=self._p(x_saved), p1=self._p(gamma), p2=self._p(dy_xhat))

    def add_taskvec(self, t, F, relu=False, modulus=0):
        self._op(op=L.OP_ADD_TASKVEC, i0=pad32(F), i1=int(relu), i4=modulus, p0=self._p(t))

    def linear(self, W, K, N, bias=None, relu=False, addend=None, addend_modulus=0, mode=L.W_ROWMAJOR, ldw=None,
               w_tiles=0, w_task_stride=0, b_task_stride=0, addend_rm=False):
        if self.bf16:
            raise ValueError("a program is either all-fp32 or all-bf16")
        flags = (L.F_RELU if relu else 0) | ((L.F_ADD_RM if addend_rm else L.F_ADD_PT) if addend is not None else 0)
        i3 = (ldw if ldw is not None else K) if mode == L.W_ROWMAJOR else w_tiles
        self.keep.append(W)
        self._op(op=L.OP_LINEAR, i0=K, i1=N, i2=mode, i3=i3, flags=flags, i4=addend_modulus, p0=L.ptr(W, strided=True),
                 p1=self._p(bias), p2=self._p(addend), s0=w_task_stride, s1=b_task_stride)

    def linear_bf16(self, W_img, K, N, bias=None, relu=False, addend=None, addend_modulus=0, b_task_stride=0,
                    per_task=False, true_K=None):
        """LINEAR in the bf16 compute mode: ``W_img`` = ``cast_bf16_weights`` image [N, pad32(K)] (bf16), or with
        ``per_task`` a ``store_wb`` / ``store_trb`` image [n_tasks, rows >= N, pad32(K)] of activations."""
        if self.ops and not self.bf16 and any(o.op == L.OP_LINEAR for o in self.ops):
            raise ValueError("a program is either all-fp32 or all-bf16")
        ok = W_img.dtype == torch.bfloat16 and W_img.is_contiguous() and W_img.shape[-1] == pad32(K)
        ok = ok and ((W_img.dim() == 3 and W_img.shape[0] == self.n_tasks and W_img.shape[1] >= N) if per_task
                     else tuple(W_img.shape) == (N, pad32(K)))
        if not ok:
            raise ValueError(f"bad bf16 weight image {tuple(W_img.shape)} {W_img.dtype} for a {K}->{N} layer")
        self.bf16 = True
        if true_K is not None:  # (image rows zero-padded from true_K to K inputs)
            self._pad_k += (K - true_K) * N
        p_add, fl_add = self._pt(addend) if addend is not None else (None, 0)  # (fp32 PT32 or PT16)
        flags = (L.F_RELU if relu else 0) | ((L.F_ADD_PT | fl_add) if addend is not None else 0)
        # a store of the layer's input right in front of it rides inside the layer (NPF_F_STORE_IN): the pipelined
        # layers spread it over their stages instead of bursting 8 store instructions per wave between two layers
        p3 = None
        last = self.ops[-1] if self.ops else None
        if FUSE_STORES and last is not None and last.op == L.OP_STORE_PT and last.i0 == pad32(K):
            self.ops.pop()
            p3 = last.p0
            flags |= L.F_STORE_IN | (L.F_STORE_P16 if last.flags & L.F_P16 else 0)
        self.keep.append(W_img)
        self._op(op=L.OP_LINEAR, i0=K, i1=N, i2=L.W_ROWMAJOR, i3=pad32(K) // 2, flags=flags, i4=addend_modulus,
                 p0=W_img.data_ptr(), p1=self._p(bias), p2=p_add, p3=p3,
                 s0=(W_img.shape[1] * pad32(K) // 2 if per_task else 0), s1=b_task_stride)

    def store_wb(self, img, F):
        """bf16 row image [n_tasks, 32*tiles, pad32(F)] <- cur (bf16 mode)."""
        self.bf16 = True
        self.keep.append(img)
        self._op(op=L.OP_STORE_WB, i0=F, i1=img.shape[1], p0=img.data_ptr())

    def store_trb(self, img, F):
        """bf16 transposed image [n_tasks, F, 32*tiles] <- cur (bf16 mode)."""
        self.bf16 = True
        self.keep.append(img)
        self._op(op=L.OP_STORE_TRB, i0=F, i1=img.shape[2], p0=img.data_ptr())

    def flops(self) -> int:
        """Algorithmic GEMM FLOPs of one launch: 2*K*N per LINEAR per valid point."""
        return (sum(2 * o.i0 * o.i1 for o in self.ops if o.op == L.OP_LINEAR) - 2 * self._pad_k) * self.n_tasks * self.pts

    def hbm_bytes(self) -> int:
        """Algorithmic HBM bytes of one launch: every per-point tensor the program loads or stores
        once (padded tiles included), every weight matrix once (per task for per-task weights).
        Weight re-reads by the other workgroups are L2 traffic, not counted."""
        pts = self.n_tasks * tiles_of(self.pts) * 32
        per_pt = 0
        fixed = 0
        for o in self.ops:
            if o.op in (L.OP_LOAD_PT, L.OP_STORE_PT, L.OP_ADD_PT, L.OP_MASK_POS, L.OP_ROWDOT_PT, L.OP_SOFTMAX_BWD,
                        L.OP_LOAD_RM):
                per_pt += (2 if o.flags & L.F_P16 else 4) * o.i0
            elif o.op in (L.OP_LOAD_ROWS, L.OP_STORE_ROWS, L.OP_STORE_TR):
                per_pt += 4 * o.i0
            elif o.op in (L.OP_STORE_WB, L.OP_STORE_TRB):
                per_pt += 2 * pad32(o.i0)
            elif o.op in (L.OP_STORE_MASK, L.OP_MASK_BITS):
                per_pt += 16 * ((o.i0 + 127) // 128)
            elif o.op == L.OP_SOFTMAX and o.i1:
                per_pt += 8
            elif o.op == L.OP_LAYERNORM_BWD:
                per_pt += 4 * pad32(o.i0) * (2 if o.p2 else 1)
            elif o.op == L.OP_LINEAR:
                if o.flags & (L.F_ADD_PT | L.F_MASK_PT | L.F_ADD_RM):
                    per_pt += (2 if o.flags & L.F_P16 else 4) * pad32(o.i1)
                if o.flags & (L.F_MASK_BITS | L.F_STORE_BITS):
                    per_pt += 16 * ((pad32(o.i1) + 127) // 128)
                if o.flags & L.F_STORE_IN:
                    per_pt += (2 if o.flags & L.F_STORE_P16 else 4) * pad32(o.i0)
                per_task = o.i2 != L.W_ROWMAJOR or o.s0 != 0
                fixed += 4 * o.i0 * o.i1 * (self.n_tasks if per_task else 1)
        return per_pt * pts + fixed - 4 * self._pad_k

    def launch(self) -> None:
        if not self.ops:
            return
        if PROFILE is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            self._launch()
            ev1.record()
            PROFILE.append(("chain_kernel", self.flops(), ev0, ev1, self.hbm_bytes()))
        else:
            self._launch()

    def only_rows_loaded(self) -> bool:
        """cur = a LOAD_ROWS result (<= 32 features, every other register zero), stored at most since."""
        return bool(self.ops) and self.ops[0].op == L.OP_LOAD_ROWS and all(o.op == L.OP_STORE_PT for o in self.ops[1:])

    def describe(self) -> str:
        """One line per op (debugging aid: NPF_DUMP_PROGRAMS=1 prints it at every launch)."""
        names = {v: k[3:] for k, v in vars(L).items() if k.startswith("OP_")}
        flag_names = [(L.F_RELU, "relu"), (L.F_ADD_PT, "add_pt"), (L.F_MASK_PT, "mask_pt"), (L.F_ADD_RM, "add_rm"),
                      (L.F_P16, "p16"), (L.F_MASK_BITS, "mask_bits"), (L.F_STORE_IN, "store_in"),
                      (L.F_STORE_P16, "store_p16"), (L.F_STORE_BITS, "store_bits")]
        lines = [f"program: {self.n_tasks} tasks x {self.pts} points, wg_per_task={int(self.wg_per_task)}, bf16={int(self.bf16)}"]
        for o in self.ops:
            fl = "|".join(n for b, n in flag_names if o.flags & b)
            lines.append(f"  {names.get(o.op, o.op):12s} i0={o.i0:4d} i1={o.i1:4d} i2={o.i2} i4={o.i4} {fl}")
        return "\n".join(lines)

    def _launch(self) -> None:
        if DUMP_PROGRAMS:
            print(self.describe(), file=sys.stderr)
        prog = L.NpfProgram()
        prog.n_ops = len(self.ops)
        prog.n_tasks, prog.pts_per_task, prog.tiles_per_task = self.n_tasks, self.pts, tiles_of(self.pts)
        prog.wg_per_task = int(self.wg_per_task)
        prog.reserved[0] = DEBUG_ABLATE
        prog.reserved[1] = FORCE_WG
        prog.reserved[2] = int(self.bf16)
        for i, o in enumerate(self.ops):
            prog.ops[i] = o
        L.check(L.load().npf_chain_run(C.byref(prog), L.stream_ptr()), "npf_chain_run")


def run_wgrad(jobs: Sequence[dict], n_tasks: int, pts: int, device, tag: str = "") -> None:
    """jobs: dicts(dZ, A, N, K, dW, db=None, ldw=None, per_task=False, accumulate=False)."""
    if not jobs:
        return
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        _run_wgrad(jobs, n_tasks, pts, device)
        ev1.record()
        padded = n_tasks * tiles_of(pts) * 32
        nbytes = sum((j["dZ"].element_size() * pad32(j["N"]) + j["A"].element_size() * pad32(j["K"])) * padded
                     + 4 * j["N"] * j["K"] * (n_tasks if j.get("per_task") else 1) for j in jobs)
        PROFILE.append(("wgrad_kernel", sum(2 * j["N"] * j["K"] for j in jobs) * n_tasks * pts, ev0, ev1, nbytes, tag))
    else:
        _run_wgrad(jobs, n_tasks, pts, device)


def _block_jobs(jobs: Sequence[dict]) -> List[dict]:
    """One wgrad job covers at most 256 x 256 of dW (``NPF_WGRAD_BLOCK``): a wider layer becomes one job per block,
    each pointing at its block of features inside the (wider) operand tensors (``ldz`` / ``lda`` / ``ldo`` = features
    per tile of those tensors).  Blocks are multiples of 256 features, i.e. whole rows of a PT tile."""
    Bk = L.NPF_WGRAD_BLOCK
    out = []
    for jb in jobs:
        N, K = jb["N"], jb["K"]
        if N <= Bk and K <= Bk:
            out.append(jb)
            continue
        if jb.get("per_task") and N > Bk:
            raise NotImplementedError("per-task weight gradients (attention keys / values) over more than 256 points")
        dZ, A = jb["dZ"], jb["A"]
        Fz, Fa = pad32(N), pad32(K)
        ldw = jb.get("ldw") or K
        for n0 in range(0, N, Bk):
            for k0 in range(0, K, Bk):
                blk = dict(jb, N=min(Bk, N - n0), K=min(Bk, K - k0), ldz=Fz, lda=Fa,
                           dZ_off=n0 * 32, A_off=k0 * 32)  # element offsets of the block inside every tile
                if jb.get("per_task"):
                    blk.update(dW_off=k0 * 32, ldo=Fa)  # PT32 output: the block of features k0.. of every row tile
                else:
                    blk.update(dW_off=n0 * ldw + k0, ldw=ldw, db=jb.get("db") if k0 == 0 else None, db_off=n0)
                out.append(blk)
    return out


def _run_wgrad(jobs: Sequence[dict], n_tasks: int, pts: int, device) -> None:
    jobs = _block_jobs(jobs)
    x6 = WGRAD_X6 and COMPUTE_DTYPE != "bf16"
    if x6 and X6_NARROW_NATIVE:
        # narrow jobs (first / last layers: 256 x 4, 4 x 256, 32 x 2 ...) are bandwidth-bound and leave most of a split-kernel
        # workgroup idle (a 256 x 4 job over 262 144 points: 0.19 ms there, 0.11 ms on the fp32 kernel): their own launch
        narrow = [jb for jb in jobs if min(jb["N"], jb["K"]) <= 32]
        wide = [jb for jb in jobs if min(jb["N"], jb["K"]) > 32]
        if narrow and wide:
            _launch_wgrad(wide, n_tasks, pts, device, True)
            _launch_wgrad(narrow, n_tasks, pts, device, False)
            return
        x6 = bool(wide)
    _launch_wgrad(jobs, n_tasks, pts, device, x6)


def _launch_wgrad(jobs: Sequence[dict], n_tasks: int, pts: int, device, x6: bool) -> None:
    lib = L.load()
    for i0 in range(0, len(jobs), L.NPF_MAX_WGRAD_JOBS):
        chunk = jobs[i0:i0 + L.NPF_MAX_WGRAD_JOBS]
        arr = (L.NpfWgradJob * len(chunk))()
        for j, jb in enumerate(chunk):
            z16, a16 = jb["dZ"].dtype == torch.bfloat16, jb["A"].dtype == torch.bfloat16  # PT16 operands (bf16 mode)
            if (z16 or a16) and COMPUTE_DTYPE != "bf16":
                raise RuntimeError("PT16 operands exist in the bf16 compute mode only")
            arr[j].dZ = (jb["dZ"].data_ptr() if z16 else L.ptr(jb["dZ"])) + jb.get("dZ_off", 0) * jb["dZ"].element_size()
            arr[j].A = (jb["A"].data_ptr() if a16 else L.ptr(jb["A"])) + jb.get("A_off", 0) * jb["A"].element_size()
            arr[j].dW = L.ptr(jb["dW"]) + 4 * jb.get("dW_off", 0)
            arr[j].db = (L.ptr(jb["db"]) + 4 * jb.get("db_off", 0)) if jb.get("db") is not None else None
            arr[j].ldz, arr[j].lda, arr[j].ldo = jb.get("ldz", 0), jb.get("lda", 0), jb.get("ldo", 0)
            arr[j].ldw = jb.get("ldw") or jb["K"]
            arr[j].N, arr[j].K = jb["N"], jb["K"]
            arr[j].per_task = int(jb.get("per_task", False))
            # bit 1: bf16 products (bf16 compute mode: every weight / key / value gradient of the step)
            arr[j].accumulate = (int(jb.get("accumulate", False)) | (2 if COMPUTE_DTYPE == "bf16" else 0)
                                 | (4 if z16 else 0) | (8 if a16 else 0)
                                 | (L.WGRAD_F32X6 if x6 else 0) | (L.WGRAD_NO_H16 if (x6 and not WGRAD_H16) else 0))
        nbytes = lib.npf_wgrad_partials_bytes(arr, len(chunk), n_tasks, tiles_of(pts))
        if nbytes < 0:
            raise RuntimeError("npf_wgrad_partials_bytes: invalid wgrad jobs")
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        L.check(lib.npf_wgrad_run(arr, len(chunk), n_tasks, tiles_of(pts), L.ptr(ws), nbytes, L.stream_ptr()),
                "npf_wgrad_run")


def cast_bf16_weights(W: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """The bf16 image of a row-major fp32 matrix ``W`` [N, K] (or of ``W^T``) that the bf16 chain instance
    streams: [rows, pad32(cols)] bfloat16, k-permuted inside groups of 32 (``npf_cast_bf16_weights``)."""
    N, K = W.shape
    if W.stride(1) != 1:
        raise ValueError("weight rows must be contiguous")
    rows, cols = (K, N) if transposed else (N, K)
    out = torch.empty((rows, pad32(cols)), dtype=torch.bfloat16, device=W.device)
    L.check(L.load().npf_cast_bf16_weights(L.ptr(W, strided=True), N, K, W.stride(0), int(transposed), out.data_ptr(),
                                           L.stream_ptr()), "npf_cast_bf16_weights")
    return out


def prepare_weights(specs: Sequence[Tuple[torch.Tensor, int]], dsts: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
    """``npf_prepare_weights``: for every (W [N, K] row-major with unit column stride, kind) one output, all in one
    launch per 32 matrices -- kind 0: W^T (fp32 [K, N]); 1: the bf16 image of W; 2: the bf16 image of W^T; 5 / 6: images 1 / 2
    with their rows zero-padded to 256 inputs (PAD_SMALL_K)."""
    outs: List[torch.Tensor] = []
    lib = L.load()
    if dsts is not None:  # (``dsts``: write the images into these contiguous tensors -- e.g. slices of one buffer -- instead)
        assert len(dsts) == len(specs) and os.environ.get("NPF_NO_BATCH_PREP") != "1"
    if os.environ.get("NPF_NO_BATCH_PREP") == "1":  # debug switch: one launch per matrix
        if any(kind & 4 for _, kind in specs):
            raise RuntimeError("NPF_NO_BATCH_PREP needs NPF_NO_PAD_SMALL_K=1 as well")
        return [transpose(W.contiguous()) if kind == 0 else cast_bf16_weights(W, transposed=kind == 2) for W, kind in specs]
    for i0 in range(0, len(specs), L.NPF_MAX_WPREP_JOBS):
        chunk = specs[i0:i0 + L.NPF_MAX_WPREP_JOBS]
        arr = (L.NpfWprepJob * len(chunk))()
        for j, (W, kind) in enumerate(chunk):
            N, K = W.shape
            if W.stride(1) != 1 or W.dtype != torch.float32:
                raise ValueError("weight rows must be contiguous fp32")
            if kind == 0:
                shape, dt = (K, N), torch.float32
            else:
                rows, cols = (K, N) if kind & 3 == 2 else (N, K)   # (kind bits 4-5: a term of the three-term split)
                shape, dt = (rows, 256 if kind & 4 else pad32(cols)), torch.bfloat16
            if dsts is not None:
                out = dsts[i0 + j]
                if tuple(out.shape) != shape or out.dtype != dt or not out.is_contiguous():
                    raise ValueError(f"prepare_weights: destination {tuple(out.shape)} {out.dtype} for an image {shape} {dt}")
            else:
                out = torch.empty(shape, dtype=dt, device=W.device)
            arr[j].src, arr[j].dst = L.ptr(W, strided=True), out.data_ptr()
            arr[j].n_rows, arr[j].n_cols, arr[j].ld, arr[j].kind = N, K, W.stride(0), kind
            outs.append(out)
        L.check(lib.npf_prepare_weights(arr, len(chunk), L.stream_ptr()), "npf_prepare_weights")
    return outs


def transpose(W: torch.Tensor) -> torch.Tensor:
    """W^T of a row-major [rows, cols] device matrix (npf_transpose)."""
    rows, cols = W.shape
    out = torch.empty((cols, rows), dtype=torch.float32, device=W.device)
    L.check(L.load().npf_transpose(L.ptr(W.contiguous()), rows, cols, L.ptr(out), L.stream_ptr()), "npf_transpose")
    return out


# ---------------------------------------------------------------------------------------
# chain description
# ---------------------------------------------------------------------------------------
@dataclass
class _Step:
    kind: str
    t: dict = field(default_factory=dict)   # role -> index into Chain.tensors
    a: dict = field(default_factory=dict)   # static attributes


class Chain:
    """Straight-line per-point computation (see module docstring).

    ``n_tasks`` x ``pts_per_task`` points; ``wg_per_task`` must be True when the chain uses
    the task's keys / values as weights (attention)."""

    def __init__(self, n_tasks: int, pts_per_task: int, device, wg_per_task: bool = False):
        self.n_tasks, self.pts, self.device, self.wg_per_task = n_tasks, pts_per_task, device, wg_per_task
        self.steps: List[_Step] = []
        self.tensors: List[Optional[torch.Tensor]] = []
        self.F = 0  # valid features of cur

    def _t(self, t: Optional[torch.Tensor]) -> int:
        if t is None:
            return -1
        self.tensors.append(t)
        return len(self.tensors) - 1

    # ---- inputs
    def input_pt(self, t: torch.Tensor, F: int, modulus: int = 0) -> "Chain":
        self.steps.append(_Step("input_pt", {"x": self._t(t)}, {"F": F, "mod": modulus}))
        self.F = F
        return self

    def input_rm(self, t: torch.Tensor, F: int, modulus: int = 0) -> "Chain":
        """Row-major [n_tasks (or modulus), pts, F] input without a layout pass (F % 32 == 0, contiguous,
        inference only: no gradient flows into it)."""
        if F % 32 or not t.is_contiguous() or t.shape[-1] != F:
            raise ValueError("input_rm needs a contiguous tensor with a multiple of 32 features")
        self.steps.append(_Step("input_rm", {"x": self._t(t)}, {"F": F, "mod": modulus}))
        self.F = F
        return self

    def input_rows(self, t: torch.Tensor, kd: int, modulus: int = 0) -> "Chain":
        if kd > 32:
            raise NotImplementedError("row-major chain inputs are limited to 32 features")
        self.steps.append(_Step("input_rows", {"x": self._t(t)}, {"kd": kd, "mod": modulus}))
        self.F = kd
        return self

    # ---- layers
    def linear(self, W: torch.Tensor, b: Optional[torch.Tensor], relu: bool = False,
               addend: Optional[torch.Tensor] = None, addend_modulus: int = 0,
               bias_per_task: bool = False, addend_rm: bool = False, residual: bool = False) -> "Chain":
        """cur <- act(W cur + b [+ addend]) [+ cur].  ``W`` [N, K] may be a column slice of a wider
        matrix (row stride = ``W.stride(0)``); ``bias_per_task``: ``b`` is [n_tasks, pad32(N)];
        ``addend_rm``: the addend is a row-major [n_tasks (or modulus), pts, N] tensor (N % 32 == 0,
        inference only); ``residual``: the layer's input is added to its activated output (the hidden
        layers of an ``MLP(is_res=True)``, mlp.py:100-104)."""
        N, K = W.shape
        if residual and N != K:
            raise ValueError(f"a residual layer keeps its width, got {K} -> {N}")
        if addend_rm and (addend is None or N % 32 or not addend.is_contiguous() or addend.shape[-1] != N):
            raise ValueError("a row-major addend must be contiguous with N % 32 == 0 features")
        if K != self.F:
            raise ValueError(f"Linear expects {K} inputs, chain carries {self.F}")
        if max(N, K) > L.NPF_MAX_FEATURES:
            raise NotImplementedError(
                f"layer {K}->{N}: the HIP chain keeps at most {L.NPF_MAX_FEATURES} features in registers")
        if W.stride(1) != 1:
            raise ValueError("weight rows must be contiguous")
        if bias_per_task and not self.wg_per_task:
            raise ValueError("per-task biases need wg_per_task=True")
        self.steps.append(_Step("linear", {"W": self._t(W), "b": self._t(b), "add": self._t(addend)},
                                {"N": N, "K": K, "relu": relu, "mod": addend_modulus, "bpt": bias_per_task,
                                 "add_rm": addend_rm, "res": residual}))
        self.F = N
        return self

    def add_pt(self, t: torch.Tensor, relu: bool = False, modulus: int = 0) -> "Chain":
        self.steps.append(_Step("add_pt", {"x": self._t(t)}, {"F": self.F, "relu": relu, "mod": modulus}))
        return self

    def dropout(self, sign_mask: torch.Tensor, p: float) -> "Chain":
        """cur <- cur * keep / (1 - p) (``nn.Dropout`` in training mode, mlp.py:81,98,105): ``sign_mask`` is a PT32 tensor
        holding +1 where the unit is kept and -1 where it is dropped (no gradient)."""
        if not 0.0 <= p < 1.0:
            raise ValueError(f"dropout probability {p}")
        self.steps.append(_Step("dropout", {"m": self._t(sign_mask)}, {"F": self.F, "p": float(p)}))
        return self

    def add_taskvec(self, v: torch.Tensor, relu: bool = False, modulus: int = 0) -> "Chain":
        """cur += v[task] with v row-major [n_tasks (or modulus), pad32(F)]."""
        self.steps.append(_Step("add_taskvec", {"v": self._t(v)}, {"F": self.F, "relu": relu, "mod": modulus}))
        return self

    def layernorm(self, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5) -> "Chain":
        """cur <- LayerNorm(cur) over the features (nn.LayerNorm with affine parameters)."""
        if self.F > L.NPF_MAX_FUSED_ROW or weight.shape != (self.F,) or bias.shape != (self.F,):
            raise ValueError(f"LayerNorm over {self.F} features with parameters {tuple(weight.shape)}")
        self.steps.append(_Step("layernorm", {"g": self._t(weight), "b": self._t(bias)}, {"F": self.F, "eps": float(eps)}))
        return self

    def store_bf16_images(self) -> "Chain":
        """Two extra (non-differentiable) outputs, bf16 compute mode only: cur as a bf16 row image
        [n_tasks, 32*tiles, pad32(F)] and as a transposed image [n_tasks, F, 32*tiles] -- what an attention
        chain takes as ``keys_img`` / ``values_img``."""
        self.steps.append(_Step("store_wb", {}, {"F": self.F}))
        self.steps.append(_Step("store_trb", {}, {"F": self.F}))
        return self

    def attn_scores(self, keys_pt: torch.Tensor, n_keys: int, keys_tr: Optional[torch.Tensor] = None,
                    keys_img=None) -> "Chain":
        """cur[c] <- sum_d keys[c][d] cur[d] (DotAttender.score, attention.py:204-220, unscaled).
        ``keys_tr``: optional feature-major copy [n_tasks, d, 32*tiles] of the keys (``store_tr``):
        lets the backward pass stream K^T by LDS-DMA."""
        if not self.wg_per_task:
            raise ValueError("attention needs wg_per_task=True")
        if n_keys > L.NPF_MAX_FUSED_ROW:
            raise NotImplementedError(
                f"a fused score row holds at most {L.NPF_MAX_FUSED_ROW} context points; longer contexts go "
                "through DotAttender.attend_pt (attention_long.py)")
        self.steps.append(_Step("attn_scores", {"k": self._t(keys_pt)}, {"C": n_keys, "r": self.F, "tr": keys_tr,
                                                                          "img": keys_img}))
        self.F = n_keys
        return self

    def softmax(self, scale: float) -> "Chain":
        self.steps.append(_Step("softmax", {}, {"n": self.F, "scale": float(scale)}))
        return self

    def attn_values(self, values_pt: torch.Tensor, r: int, values_tr: Optional[torch.Tensor] = None,
                    values_img=None) -> "Chain":
        """cur[n] <- sum_c values[c][n] cur[c] (torch.bmm(attn, values), attention.py:151).
        ``values_tr``: optional feature-major copy [n_tasks, r, 32*tiles] of the values."""
        self.steps.append(_Step("attn_values", {"v": self._t(values_pt)}, {"C": self.F, "r": r, "tr": values_tr,
                                                                            "img": values_img}))
        self.F = r
        return self

    # ---- outputs
    def tap(self, alias_input: bool = False) -> "Chain":
        """Extra output: cur at this point.  ``alias_input``: directly behind the chain's ``input_pt`` (no modulus) the output IS
        the input tensor (no copy) -- a way to hand the same tensor on to a later consumer such that its gradient comes back into
        THIS chain's dgrad launch and is added there, instead of autograd summing two point-sized tensors in a pass of its own."""
        if alias_input and not (len(self.steps) == 1 and self.steps[0].kind == "input_pt" and self.steps[0].a["mod"] == 0):
            raise ValueError("tap(alias_input=True) goes directly behind input_pt (modulus 0)")
        self.steps.append(_Step("tap", {}, {"F": self.F, "alias": bool(alias_input)}))
        return self

    def store_tr(self) -> "Chain":
        """Extra (non-differentiable) output: feature-major copy [n_tasks, F, 32*tiles] of cur."""
        self.steps.append(_Step("store_tr", {}, {"F": self.F}))
        return self

    def output_pt(self) -> "Chain":
        self.steps.append(_Step("output_pt", {}, {"F": self.F}))
        return self

    def output_rows(self) -> "Chain":
        if self.F > 32:
            raise NotImplementedError("row-major chain outputs are limited to 32 features")
        self.steps.append(_Step("output_rows", {}, {"nd": self.F}))
        return self

    def run(self):
        """Execute; returns the tuple of outputs in declaration order (taps and final)."""
        self.grad_enabled = torch.is_grad_enabled()
        outs = _ChainFn.apply(self, *[t for t in self.tensors])
        return outs

    def run_pt(self, as_weights: bool = False) -> PTensor:
        """Declare cur as the PT32 output, execute, and return it as a :class:`PTensor`.  ``as_weights``: a
        later attention chain streams these activations as per-task weights (keys / values), so the
        feature-major copy -- and in the bf16 compute mode the two bf16 images -- are produced by the same
        launch and travel with the result."""
        F = self.F
        self.output_pt()
        widest = max([max(st.a["N"], st.a["K"]) for st in self.steps if st.kind == "linear"], default=0)
        imgs = as_weights and COMPUTE_DTYPE == "bf16" and max(widest, F) <= L.NPF_MAX_FUSED_ROW  # (a bf16 chain: <= 256 wide)
        if as_weights:
            self.store_tr()
        if imgs:
            self.store_bf16_images()
        outs = self.run()
        return PTensor(outs[0], self.pts, F, tr=outs[1] if as_weights else None,
                       img=(outs[2], outs[3]) if imgs else None)


class _ChainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, chain: Chain, *tensors):
        needs_grad = [t is not None and t.requires_grad for t in chain.tensors]
        if not chain.grad_enabled:
            needs_grad = [False] * len(needs_grad)
        train = any(needs_grad)  # (grad mode is always off inside Function.forward)
        prog = Program(chain.n_tasks, chain.pts, chain.wg_per_task)
        dev = chain.device
        T = [t.detach() if t is not None else None for t in chain.tensors]
        ctx_tiles = lambda k: T[k].shape[1]  # noqa: E731
        bf16 = (COMPUTE_DTYPE == "bf16"
                and all(st.kind in _BF16_STEPS or (st.kind in _BF16_ATTN_STEPS and st.a.get("img") is not None)
                        for st in chain.steps)
                and not any(st.kind == "linear" and st.a.get("add_rm") for st in chain.steps)
                and max([max(st.a["N"], st.a["K"]) for st in chain.steps if st.kind == "linear"], default=0) <= 256)

        saved = {}      # (step index, role) -> PT tensor
        outputs = []
        non_diff = []
        backed = None   # PT tensor currently holding cur (None if cur only lives in registers)
        upstream = False  # does cur depend on something that needs a gradient
        upstream_before = []

        backed16 = None  # bf16 mode: PT16 copy of cur kept for the backward pass only (mask / wgrad operand)

        def ensure_saved(F, internal=False):
            """A tensor holding cur: an fp32 PT32 tensor (required for outputs and for operands of fp32-only
            consumers), or -- bf16 mode, ``internal`` -- a PT16 tensor at half the HBM traffic."""
            nonlocal backed, backed16
            if internal and bf16 and PT16_INTERNAL and backed is None:
                if backed16 is None:
                    backed16 = pt16_empty(chain.n_tasks, chain.pts, F, dev)
                    prog.store_pt(backed16, F)
                return backed16
            if backed is None:
                backed = pt_empty(chain.n_tasks, chain.pts, F, dev)
                prog.store_pt(backed, F)
            return backed

        def save_relu_mask(i, F):
            """What the backward pass needs of a ReLU output: where it is positive.  bf16 mode: 32 bytes of bits per point
            (the activation itself is only stored if a wgrad job reads it -- the next layer asks for that itself);
            fp32 mode: the activation tensor doubles as the mask."""
            if bf16 and MASK_BITS:
                m = ptm_empty(chain.n_tasks, chain.pts, F, dev)
                prog.store_mask(m, F)
                saved[(i, "mask")] = m
            else:
                saved[(i, "out")] = ensure_saved(F, internal=True)

        images = {}  # bf16 mode: the weight images of all LINEAR steps, one launch
        pad_k = set()  # steps run as zero-padded 256-input layers (PAD_SMALL_K)
        if bf16:
            for i, st in enumerate(chain.steps):
                if (PAD_SMALL_K and st.kind == "linear" and i > 0 and chain.steps[i - 1].kind == "input_rows" and st.a["K"] <= 32
                        and st.a["N"] == 256 and st.t["add"] < 0 and not st.a.get("add_rm")):
                    pad_k.add(i)
            lin = [(i, st.t["W"]) for i, st in enumerate(chain.steps) if st.kind == "linear"]
            images = dict(zip(lin, prepare_weights([(T[w], 5 if i in pad_k else 1) for i, w in lin])))
        for i, st in enumerate(chain.steps):
            upstream_before.append(upstream)
            k, a = st.kind, st.a
            if k == "input_pt":
                prog.load_pt(T[st.t["x"]], a["F"], a["mod"])
                backed = T[st.t["x"]] if a["mod"] == 0 else None
                backed16 = None
                upstream = needs_grad[st.t["x"]]
            elif k == "input_rm":
                if train and needs_grad[st.t["x"]]:
                    raise NotImplementedError("row-major chain inputs carry no gradient (inference path)")
                prog.load_rm(T[st.t["x"]], a["F"], a["mod"])
                backed = backed16 = None
                upstream = False
            elif k == "input_rows":
                prog.load_rows(T[st.t["x"]], a["kd"], a["mod"])
                backed = backed16 = None
                upstream = False
            elif k == "linear":
                W, b, add = st.t["W"], st.t["b"], st.t["add"]
                # residual layer: its input has to be in HBM (fp32) to come back after the activation
                res_in = ensure_saved(a["K"]) if a.get("res") else None
                if train and (needs_grad[W] or (b >= 0 and needs_grad[b])):
                    saved[(i, "in")] = ensure_saved(a["K"], internal=True)
                if a.get("add_rm") and train and (upstream or needs_grad[W] or needs_grad[add]):
                    raise NotImplementedError("row-major addends carry no gradient (inference path)")
                if bf16:
                    padded = i in pad_k and prog.only_rows_loaded()
                    if i in pad_k and not padded:
                        raise AssertionError("zero-padded layer not behind LOAD_ROWS")
                    prog.linear_bf16(images[(i, W)], 256 if padded else a["K"], a["N"], bias=T[b] if b >= 0 else None,
                                     relu=a["relu"], addend=T[add] if add >= 0 else None, addend_modulus=a["mod"],
                                     b_task_stride=(T[b].stride(0) if a["bpt"] else 0), true_K=a["K"] if padded else None)
                else:
                    prog.linear(T[W], a["K"], a["N"], bias=T[b] if b >= 0 else None, relu=a["relu"],
                                addend=T[add] if add >= 0 else None, addend_modulus=a["mod"], ldw=T[W].stride(0),
                                b_task_stride=(T[b].stride(0) if a["bpt"] else 0), addend_rm=bool(a.get("add_rm")))
                backed = backed16 = None
                upstream = upstream or needs_grad[W] or (b >= 0 and needs_grad[b]) or (add >= 0 and needs_grad[add])
                if train and a["relu"] and upstream:
                    save_relu_mask(i, a["N"])
                if res_in is not None:
                    prog.add_pt(res_in, a["N"])
                    backed = backed16 = None
            elif k == "add_pt":
                prog.add_pt(T[st.t["x"]], a["F"], a["relu"], a["mod"])
                backed = backed16 = None
                upstream = upstream or needs_grad[st.t["x"]]
                if train and a["relu"] and upstream:
                    save_relu_mask(i, a["F"])
            elif k == "dropout":
                prog.mask_sign(T[st.t["m"]], a["F"])
                prog.scale(1.0 / (1.0 - a["p"]))
                backed = backed16 = None
            elif k == "add_taskvec":
                prog.add_taskvec(T[st.t["v"]], a["F"], a["relu"], a["mod"])
                backed = backed16 = None
                upstream = upstream or needs_grad[st.t["v"]]
                if train and a["relu"] and upstream:
                    save_relu_mask(i, a["F"])
            elif k == "layernorm":
                gi, bi = st.t["g"], st.t["b"]
                upstream = upstream or needs_grad[gi] or needs_grad[bi]
                if train and upstream:
                    saved[(i, "in")] = ensure_saved(a["F"])
                prog.layernorm(_pad_vec(T[gi]), _pad_vec(T[bi]), a["F"], a["eps"])
                backed = backed16 = None
            elif k == "attn_scores":
                kk = st.t["k"]
                if train and needs_grad[kk]:
                    saved[(i, "in")] = ensure_saved(a["r"], internal=True)
                if bf16:
                    prog.linear_bf16(a["img"][0], a["r"], a["C"], per_task=True)
                else:
                    prog.linear(T[kk], a["r"], a["C"], mode=L.W_PT_ROWS, w_tiles=ctx_tiles(kk))
                backed = backed16 = None
                upstream = upstream or needs_grad[kk]
            elif k == "softmax":
                prog.softmax(a["n"], a["scale"])
                backed = backed16 = None
                if train and upstream:
                    saved[(i, "out")] = ensure_saved(a["n"], internal=True)
            elif k == "attn_values":
                vv = st.t["v"]
                if train and needs_grad[vv]:
                    saved[(i, "in")] = ensure_saved(a["C"], internal=True)
                if bf16:
                    prog.linear_bf16(a["img"][1], a["C"], a["r"], per_task=True)
                elif a["tr"] is not None:
                    ld = a["tr"].shape[2]
                    prog.linear(a["tr"], a["C"], a["r"], mode=L.W_ROWMAJOR, ldw=ld, w_task_stride=a["r"] * ld)
                else:
                    prog.linear(T[vv], a["C"], a["r"], mode=L.W_PT_COLS, w_tiles=ctx_tiles(vv))
                backed = backed16 = None
                upstream = upstream or needs_grad[vv]
            elif k == "store_tr":
                o = torch.empty((chain.n_tasks, a["F"], 32 * tiles_of(chain.pts)), dtype=torch.float32, device=dev)
                prog.store_tr(o, a["F"], o.shape[2])
                outputs.append(o)
                non_diff.append(o)
            elif k in ("store_wb", "store_trb"):
                if not bf16:
                    raise RuntimeError("bf16 images exist in the bf16 compute mode only")
                cols = 32 * tiles_of(chain.pts)
                shape = (chain.n_tasks, cols, pad32(a["F"])) if k == "store_wb" else (chain.n_tasks, a["F"], cols)
                o = torch.empty(shape, dtype=torch.bfloat16, device=dev)
                (prog.store_wb if k == "store_wb" else prog.store_trb)(o, a["F"])
                outputs.append(o)
                non_diff.append(o)
            elif k == "tap" and a.get("alias"):
                outputs.append(tensors[chain.steps[0].t["x"]])  # the input itself (autograd makes the output a view of it)
            elif k in ("tap", "output_pt"):
                # an output must own its storage: never alias an input tensor
                if backed is not None and any(backed is t for t in T):
                    backed = backed16 = None
                outputs.append(ensure_saved(a["F"]))
            elif k == "output_rows":
                o = torch.empty((chain.n_tasks, chain.pts, a["nd"]), dtype=torch.float32, device=dev)
                prog.store_rows(o, a["nd"])
                outputs.append(o)
            else:  # pragma: no cover
                raise AssertionError(k)
        prog.launch()
        if TRACE is not None:
            TRACE.append(("fwd", chain, T, dict(saved), tuple(outputs), bf16))
        # A saved tensor that is also an output (a chain ending in a ReLU layer: its output doubles as the mask) goes
        # through save_for_backward: kept in the dict it would close a reference cycle (output -> grad_fn -> ctx -> output)
        # that only the cyclic collector frees -- a point-sized tensor leaked per step until then
        out_pos = {id(o): j for j, o in enumerate(outputs)}
        ctx.alias_keys = [key for key, t in saved.items() if id(t) in out_pos]
        ctx.save_for_backward(*[saved[key] for key in ctx.alias_keys])
        for key in ctx.alias_keys:
            del saved[key]
        ctx.chain, ctx.saved, ctx.T, ctx.needs_grad, ctx.upstream_before = chain, saved, T, needs_grad, upstream_before
        ctx.bf16 = bf16
        ctx.train = train
        if non_diff:
            ctx.mark_non_differentiable(*non_diff)
        ctx.set_materialize_grads(False)  # outputs without a gradient arrive as None, not as zero-filled tensors
        return tuple(outputs)

    @staticmethod
    def backward(ctx, *gouts):
        chain: Chain = ctx.chain
        saved, T, needs_grad, upstream_before = dict(ctx.saved), ctx.T, ctx.needs_grad, ctx.upstream_before
        saved.update(zip(ctx.alias_keys, ctx.saved_tensors))
        dev = chain.device
        prog = Program(chain.n_tasks, chain.pts, chain.wg_per_task)
        grads: List[Optional[torch.Tensor]] = [None] * len(T)
        pending_taskvec = []  # (tensor index, PT buffer, F, modulus): reduced over the points after the launch
        pending_vec = []      # (tensor index, PT buffer, F): reduced over points and tasks (LayerNorm gamma / beta)
        jobs = []
        gouts = list(gouts)
        gouts_in = tuple(gouts)
        bufs = {}  # (step, role) -> gradient buffer this launch writes (TRACE)
        started = False  # has cur been initialised with a gradient yet

        def new_pt(F, internal=False):
            """Gradient buffer: fp32 PT32 when it is returned to autograd or reduced by an fp32 kernel; in the bf16
            mode a PT16 tensor when it only feeds the wgrad kernel (``internal``)."""
            if internal and ctx.bf16 and PT16_INTERNAL:
                return pt16_empty(chain.n_tasks, chain.pts, F, dev)
            return pt_empty(chain.n_tasks, chain.pts, F, dev)

        def relu_backward(i, F):
            if (i, "mask") in saved:
                prog.mask_bits(saved[(i, "mask")], F)
            else:
                prog.mask_pos(saved[(i, "out")], F)

        def acc_grad(idx, g):
            grads[idx] = g if grads[idx] is None else grads[idx] + g

        def reduce_modulus(buf, mod):
            if mod and mod != chain.n_tasks:
                return buf.view(chain.n_tasks // mod, mod, *buf.shape[1:]).sum(0)
            return buf

        n_out = sum(1 for s in chain.steps if s.kind in ("tap", "output_pt", "output_rows", "store_tr", "store_wb", "store_trb"))
        assert len(gouts) == n_out
        # W^T (bf16 mode: the transposed image) of every layer the dgrad passes through, one launch
        lin = [(i, st.t["W"]) for i, st in enumerate(chain.steps) if st.kind == "linear" and upstream_before[i]]
        # (PAD_SMALL_K: the dgrad of a narrow last layer -- dOut rows behind LOAD_ROWS, <= 32 of them, 256 outputs -- as a
        # zero-padded 256-input layer)
        pad_t = set()
        if ctx.bf16 and PAD_SMALL_K and lin:
            i_last = max(i for i, _ in lin)
            st_l = chain.steps[i_last]
            tail = [s.kind for s in chain.steps[i_last + 1:]]
            if st_l.a["N"] <= 32 and st_l.a["K"] == 256 and tail == ["output_rows"] and not st_l.a["relu"]:
                pad_t.add(i_last)
        w_t = dict(zip(lin, prepare_weights([(T[w], (6 if i in pad_t else 2) if ctx.bf16 else 0) for i, w in lin]))) if lin else {}
        for i in range(len(chain.steps) - 1, -1, -1):
            st = chain.steps[i]
            k, a = st.kind, st.a
            if k in ("store_tr", "store_wb", "store_trb"):
                gouts.pop()
                continue
            if k in ("output_pt", "output_rows", "tap"):
                g = gouts.pop()
                if g is None:
                    continue
                g = g.contiguous()
                if k == "output_rows":
                    assert not started
                    prog.load_rows(g, a["nd"])
                elif not started:
                    prog.load_pt(g, a["F"])
                else:
                    prog.add_pt(g, a["F"])
                started = True
                continue
            if not started:
                continue  # no gradient reaches this step
            if k == "linear":
                W, b, add = st.t["W"], st.t["b"], st.t["add"]
                g_res = None
                if a.get("res") and upstream_before[i]:
                    # y = act(W x + b) + x: the incoming gradient also reaches x directly; park it while cur goes
                    # through the mask and W^T (the same wave reads back the addresses it wrote)
                    g_res = new_pt(a["N"])
                    prog.store_pt(g_res, a["N"])
                    bufs[(i, "g_res")] = g_res
                if a["relu"]:
                    relu_backward(i, a["N"])
                need_dz = needs_grad[W] or (b >= 0 and needs_grad[b]) or (add >= 0 and needs_grad[add])
                if need_dz:
                    dz = new_pt(a["N"], internal=not ((add >= 0 and needs_grad[add]) or (b >= 0 and a["bpt"] and needs_grad[b])))
                    prog.store_pt(dz, a["N"])
                    bufs[(i, "dz")] = dz
                    if needs_grad[W] or (b >= 0 and needs_grad[b]):
                        dW = torch.empty((a["N"], a["K"]), dtype=torch.float32, device=dev)
                        db = torch.empty((a["N"],), dtype=torch.float32, device=dev) if (b >= 0 and not a["bpt"]) else None
                        jobs.append(dict(dZ=dz, A=saved[(i, "in")], N=a["N"], K=a["K"], dW=dW, db=db))
                        grads[W] = dW
                        if b >= 0 and not a["bpt"]:
                            grads[b] = db
                    if b >= 0 and a["bpt"] and needs_grad[b]:
                        pending_taskvec.append((b, dz, a["N"], 0))
                    if add >= 0 and needs_grad[add]:
                        grads[add] = (dz, a["mod"])  # resolved after the launch
                if upstream_before[i]:
                    if ctx.bf16:
                        padded = i in pad_t and prog.only_rows_loaded()
                        if i in pad_t and not padded:
                            raise AssertionError("zero-padded dgrad layer not behind LOAD_ROWS")
                        prog.linear_bf16(w_t[(i, W)], 256 if padded else a["N"], a["K"], true_K=a["N"] if padded else None)
                    else:
                        prog.linear(w_t[(i, W)], a["N"], a["K"])  # W^T [K, N]
                    if g_res is not None:
                        prog.add_pt(g_res, a["K"])
                else:
                    started = False  # nothing upstream needs this gradient
                    break
            elif k in ("add_pt", "add_taskvec"):
                if a["relu"]:
                    relu_backward(i, a["F"])
                idx = st.t["x"] if k == "add_pt" else st.t["v"]
                if needs_grad[idx]:
                    buf = new_pt(a["F"])
                    prog.store_pt(buf, a["F"])
                    bufs[(i, "dz")] = buf
                    if k == "add_pt":
                        grads[idx] = (buf, a["mod"])
                    else:
                        pending_taskvec.append((idx, buf, a["F"], a["mod"]))
            elif k == "dropout":
                prog.mask_sign(T[st.t["m"]], a["F"])
                prog.scale(1.0 / (1.0 - a["p"]))
            elif k == "layernorm":
                gi, bi = st.t["g"], st.t["b"]
                dyx = None
                if needs_grad[bi]:
                    dyb = new_pt(a["F"])
                    prog.store_pt(dyb, a["F"])
                    pending_vec.append((bi, dyb, a["F"]))
                if needs_grad[gi]:
                    dyx = new_pt(a["F"])
                    pending_vec.append((gi, dyx, a["F"]))
                prog.layernorm_bwd(saved[(i, "in")], _pad_vec(T[gi]), a["F"], a["eps"], dy_xhat=dyx)
            elif k == "attn_values":
                vv = st.t["v"]
                if needs_grad[vv]:
                    dO = new_pt(a["r"], internal=True)
                    prog.store_pt(dO, a["r"])
                    bufs[(i, "dz")] = dO
                    dV = torch.empty_like(T[vv])
                    jobs.append(dict(dZ=saved[(i, "in")], A=dO, N=a["C"], K=a["r"], dW=dV, per_task=True))
                    grads[vv] = dV
                if ctx.bf16:
                    prog.linear_bf16(a["img"][0], a["r"], a["C"], per_task=True)   # dP = dO V^T (row image of the values)
                else:
                    prog.linear(T[vv], a["r"], a["C"], mode=L.W_PT_ROWS, w_tiles=T[vv].shape[1])
            elif k == "softmax":
                P = saved[(i, "out")]
                prog.rowdot_pt(P, a["n"])
                prog.softmax_bwd(P, a["n"], a["scale"])
            elif k == "attn_scores":
                kk = st.t["k"]
                if needs_grad[kk]:
                    dS = new_pt(a["C"], internal=True)
                    prog.store_pt(dS, a["C"])
                    bufs[(i, "dz")] = dS
                    dK = torch.empty_like(T[kk])
                    jobs.append(dict(dZ=dS, A=saved[(i, "in")], N=a["C"], K=a["r"], dW=dK, per_task=True))
                    grads[kk] = dK
                if upstream_before[i]:
                    if ctx.bf16:
                        prog.linear_bf16(a["img"][1], a["C"], a["r"], per_task=True)  # dQ = dS K (transposed image of the keys)
                    elif a["tr"] is not None:
                        ld = a["tr"].shape[2]
                        prog.linear(a["tr"], a["C"], a["r"], mode=L.W_ROWMAJOR, ldw=ld, w_task_stride=a["r"] * ld)
                    else:
                        prog.linear(T[kk], a["C"], a["r"], mode=L.W_PT_COLS, w_tiles=T[kk].shape[1])
                else:
                    break
            elif k == "input_pt":
                idx = st.t["x"]
                if needs_grad[idx]:
                    # Gradient fan-in: the same tensor enters this chain again further down as a layer's addend
                    # (targets: attention query here, decoder input there).  Its other gradient is that layer's dZ,
                    # stored by this very launch (same wave, same addresses): add it here and return one buffer,
                    # instead of leaving autograd a separate pass over two point-sized tensors.
                    for j2, gj in enumerate(grads):
                        if (j2 != idx and isinstance(gj, tuple) and gj[1] == a["mod"] and gj[0].dtype == torch.float32
                                and T[j2].data_ptr() == T[idx].data_ptr() and T[j2].shape == T[idx].shape
                                and tuple(gj[0].shape) == pt_shape(chain.n_tasks, chain.pts, a["F"])
                                and not any(g2 is gj for j3, g2 in enumerate(grads) if j3 != j2)):
                            prog.add_pt(gj[0], a["F"])
                            bufs.setdefault((i, "fan_in"), []).append(gj[0])
                            grads[j2] = None
                    buf = new_pt(a["F"])
                    prog.store_pt(buf, a["F"])
                    bufs[(i, "dx")] = buf
                    grads[idx] = (buf, a["mod"])
            elif k == "input_rows":
                pass
        prog.launch()
        run_wgrad(jobs, chain.n_tasks, chain.pts, dev)
        if TRACE is not None:
            TRACE.append(("bwd", chain, gouts_in, bufs, list(jobs), ctx.bf16, saved, T))
        from .functional import sum_points_pt  # late import (cycle)

        for idx, buf, F, mod in pending_taskvec:
            g = sum_points_pt(buf, chain.pts, F)  # [n_tasks, pad32(F)]
            acc_grad(idx, reduce_modulus(g, mod))
        for idx, buf, F in pending_vec:
            acc_grad(idx, sum_points_pt(buf, chain.pts, F).sum(0)[:F])
        out = []
        for g in grads:
            if isinstance(g, tuple):
                g = reduce_modulus(g[0], g[1])
            out.append(g)
        return (None, *out)
