"""x6 programs (``npf_x6_run``, ``csrc/x6_kernel.hip``): whole sides of the model as one launch with every multiply an fp32
product on the bf16 matrix pipe (three exact bf16 terms per operand, six cross products, fp32 accumulation; in the bf16 compute mode one bf16 term per
operand: b16 programs, ``npf_b16_run``, csrc/b16_kernel.hip).

``target_side`` is the fused target side of an attentive deterministic model (AttnCNP with scaled-dot attention,
npf/neuralproc/attnnp.py:118-131 + base.py:327-367): x-encoder from the raw frequencies, cross attention over the task's context
points, decoder and its output layer -- ONE launch forward, ONE launch for the dgrad of all of it, then the weight / key / value
gradient jobs.  ``context_side`` is the x-encoder + XY-encoder of the context points (attnnp.py:105-116) the same way.
Nothing here computes on the CPU; there is no fallback (callers test ``*_usable`` and otherwise take the chain path).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional, Sequence

import torch

from . import _lib as L
from . import chain as CH
from . import functional as FN

# NPF_NO_X6_FUSED=1: the fused sides are off (the chain / mlp_x6 launches of round 2 run instead)
ENABLED = os.environ.get("NPF_NO_X6_FUSED", "0") != "1"
WIDTH = 256
# which instance of the program kernel runs (npf_x6_run_ex ``variant``): 0 = the library's choice; 256-wide programs: 1 = 16
# points per wave, two workgroups of four waves per CU, 2 = 32 points per wave (one wave per SIMD), 3 = 16 points per wave, eight
# waves sharing one slab ring (the default); 512-wide programs: 1 = one wave per 16 points
# and all 512 features (one wave per SIMD), 0 / 2 = the contraction split over pairs of waves (plain layers only, two waves per
# SIMD, the default).  NPF_X6_VARIANT is a development / A-B switch
VARIANT = int(os.environ.get("NPF_X6_VARIANT", "0"))
# the bf16 compute mode on the fused sides (b16 programs); NPF_NO_B16_FUSED=1: the bf16 chain launches of round 2 instead
B16_ENABLED = os.environ.get("NPF_NO_B16_FUSED", "0") != "1"
B16_VARIANT = int(os.environ.get("NPF_B16_VARIANT", "0"))  # npf_b16_run ``variant`` (A-B switch)


class Program:
    """A list of ``npf_x6_op_t`` + geometry; ``launch()`` calls ``npf_x6_run``."""

    def __init__(self, n_tasks: int, tiles: int, per_task: bool, width: int = WIDTH, pts: Optional[int] = None,
                 bf16: bool = False):
        self.n_tasks, self.tiles, self.per_task, self.width = n_tasks, tiles, per_task, width
        # bf16: a b16 program (``npf_b16_run``, csrc/b16_kernel.hip) -- the bf16 compute mode: one-term images, store_in /
        # store_out / sbwd_p are PT16 tensors unless the op says ``store_in_f32`` / ``store_out_f32``
        self.bf16 = bf16
        self.pts = tiles * 32 if pts is None else pts  # valid points per task (row-major operands are indexed with it)
        self.ops: List[dict] = []
        self.tail = None  # (W [4, F], b [4] or None, rows [points, 4])
        self.tag = ""     # which side / direction this launch is (bench.py lists the launches of a step by it)

    def op(self, **kw) -> "Program":
        self.ops.append(kw)
        return self

    @staticmethod
    def _ptr(t, what):
        if t is None:
            return None
        if not t.is_cuda or not t.is_contiguous():
            raise RuntimeError(f"x6 program operand {what} must be a contiguous device tensor")
        return t.data_ptr()

    def _pt_ok(self, t, what, p16=False):
        want = (self.n_tasks, self.tiles, self.width // 8, 32, 8) if p16 else (self.n_tasks, self.tiles, self.width // 4, 32, 4)
        dt = torch.bfloat16 if p16 else torch.float32
        if t is not None and (tuple(t.shape) != want or t.dtype != dt):
            raise ValueError(f"x6 program operand {what}: {'PT16' if p16 else 'PT32'} tensor {tuple(t.shape)} {t.dtype}, "
                             f"expected {want} {dt}")

    def launch(self) -> None:
        if len(self.ops) > L.NPF_X6_MAX_OPS:
            raise RuntimeError(f"x6 program longer than NPF_X6_MAX_OPS={L.NPF_X6_MAX_OPS}")
        arr = (L.NpfX6Op * len(self.ops))()
        F, pts = self.width, self.n_tasks * self.tiles * 32
        flops = 0
        nbytes = 0
        for j, o in enumerate(self.ops):
            flags = 0
            for k in ("in_pt", "pre_add", "mask", "sbwd_p", "store_in", "addend", "store_out"):
                t = o.get(k)
                if t is not None and o.get(k + "_rm"):  # a row-major [n_tasks, pts, F] operand (in_pt / addend only)
                    if k not in ("in_pt", "addend") or tuple(t.shape) != (self.n_tasks, self.pts, F) or t.dtype != torch.float32:
                        raise ValueError(f"x6 program operand {k}: row-major tensor {tuple(t.shape)} {t.dtype}")
                    if self.bf16:
                        raise ValueError("b16 programs take no row-major operands")
                    flags |= L.X6_IN_RM if k == "in_pt" else L.X6_ADD_RM
                else:
                    p16 = self.bf16 and (k == "sbwd_p" or (k in ("store_in", "store_out") and not o.get(k + "_f32")))
                    if self.bf16 and k in ("store_in", "store_out") and o.get(k + "_f32") and t is not None:
                        flags |= L.X6_STORE_IN_F32 if k == "store_in" else L.X6_STORE_OUT_F32
                    if self.bf16 and k == "mask" and t is not None:
                        raise ValueError("b16 programs take ReLU masks as bits")
                    self._pt_ok(t, k, p16)
                setattr(arr[j], k, self._ptr(t, k))
                nbytes += pts * F * (t.element_size() if t is not None else 0)
            arr[j].reserved[0] = flags
            for k in ("mask_bits", "store_in_bits", "store_bits"):
                t = o.get(k)
                if t is not None and (tuple(t.shape) != (self.n_tasks, self.tiles, 2, 64) or t.dtype != torch.int64):
                    raise ValueError(f"x6 program operand {k}: bits tensor {tuple(t.shape)} {t.dtype}")
                setattr(arr[j], k, self._ptr(t, k))
                nbytes += pts * F // 8 if t is not None else 0
            if o.get("in_rows") is not None:
                rows, w, b = o["in_rows"], o["in_w"], o.get("in_b")
                n = w.shape[1]
                if tuple(rows.shape) != (self.n_tasks, self.tiles * 32, 4) or w.shape[0] != 4 or n % 16 or n > F:
                    raise ValueError(f"x6 rows prologue: rows {tuple(rows.shape)}, matrix {tuple(w.shape)}")
                if b is not None and tuple(b.shape) != (n,):
                    raise ValueError("x6 rows prologue: bias shape")
                arr[j].in_rows, arr[j].in_w, arr[j].in_b = L.ptr(rows), L.ptr(w), L.ptr(b)
                arr[j].in_n, arr[j].in_relu = n, int(bool(o.get("in_relu", False)))
                flops += 2 * 4 * n * pts
                nbytes += pts * 16
            img = o.get("img")
            if img is not None:
                per_task = bool(o.get("img_per_task", False))
                terms = (1,) if self.bf16 else (3,)
                want = ((self.n_tasks,) if per_task else ()) + (() if self.bf16 else (3,)) + (F, F)
                if tuple(img.shape) != want or img.dtype != torch.bfloat16 or not img.is_contiguous():
                    raise ValueError(f"x6 program weight image {tuple(img.shape)} {img.dtype}, expected {want} bfloat16")
                arr[j].w_img = img.data_ptr()
                arr[j].w_task_stride = terms[0] * F * F * 2 if per_task else 0
                bias = o.get("bias")
                if bias is not None:
                    bpt = bool(o.get("bias_per_task", False))
                    if tuple(bias.shape) != ((self.n_tasks, F) if bpt else (F,)):
                        raise ValueError(f"x6 program bias {tuple(bias.shape)}")
                    arr[j].bias = L.ptr(bias)
                    arr[j].bias_task_stride = F if bpt else 0
                arr[j].relu = int(bool(o.get("relu", False)))
                arr[j].softmax_n = int(o.get("softmax_n", 0))
                arr[j].softmax_scale = float(o.get("softmax_scale", 1.0))
                flops += 2 * o.get("true_k", F) * o.get("true_n", F) * pts
                nbytes += terms[0] * 2 * F * F * (self.n_tasks if per_task else 1)
            elif any(o.get(k) is not None for k in ("bias", "addend", "store_out", "store_bits")) or o.get("relu") or o.get("softmax_n"):
                raise ValueError("x6 program: an op without a multiply has no output side")
            arr[j].sbwd_scale = float(o.get("sbwd_scale", 1.0))
        tail = self.tail
        if tail is not None:
            flops += 2 * 4 * F * pts
            nbytes += pts * 16
        if CH.PROFILE is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        tw, tb, tr = (L.ptr(tail[0]) if tail else None, L.ptr(tail[1]) if (tail and tail[1] is not None) else None,
                      L.ptr(tail[2]) if tail else None)
        if self.bf16:
            if self.pts != self.tiles * 32:
                raise ValueError("b16 programs run on whole tiles")
            L.check(L.load().npf_b16_run(arr, len(self.ops), tw, tb, tr, self.n_tasks, self.tiles, int(self.per_task), F,
                                         B16_VARIANT, L.stream_ptr()), "npf_b16_run")
        else:
            L.check(L.load().npf_x6_run_ex(arr, len(self.ops), tw, tb, tr, self.n_tasks, self.tiles, self.pts, int(self.per_task), F,
                                           VARIANT if F in (256, 512) else 0, L.stream_ptr()), "npf_x6_run_ex")
        if CH.PROFILE is not None:
            ev1.record()
            CH.PROFILE.append(("b16_program_kernel" if self.bf16 else "x6_program_kernel", flops, ev0, ev1, nbytes, self.tag))
        if CH.TRACE is not None:
            CH.TRACE.append(("prog", self))


def task_images(pt: torch.Tensor, pts: int, row: bool = True, tr: bool = True, width: int = WIDTH, bf16: bool = False):
    """Three-term images of a PT32 tensor [n_tasks, tiles, F/4, 32, 4] taken as per-task weights (``npf_x6_task_images``):
    (row image W[n = point][k = feature], transposed image W[n = feature][k = point]), each [n_tasks, 3, F, F] bf16 or None.
    ``bf16``: the rounded value alone, [n_tasks, F, F] (``npf_b16_task_images``, the bf16 compute mode)."""
    n_tasks = pt.shape[0]
    if tuple(pt.shape) != (n_tasks, CH.tiles_of(pts), width // 4, 32, 4) or pts > width:
        raise ValueError(f"task_images: PT32 tensor {tuple(pt.shape)} for {pts} points x {width} features")
    shape = (n_tasks, width, width) if bf16 else (n_tasks, 3, width, width)
    mk = lambda: torch.empty(shape, dtype=torch.bfloat16, device=pt.device)  # noqa: E731
    ri, ti = (mk() if row else None), (mk() if tr else None)
    if CH.PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    fn = L.load().npf_b16_task_images if bf16 else L.load().npf_x6_task_images
    L.check(fn(L.ptr(pt.detach().contiguous()), n_tasks, pts, width, ri.data_ptr() if row else None,
               ti.data_ptr() if tr else None, L.stream_ptr()), "npf_x6_task_images")
    if CH.PROFILE is not None:
        ev1.record()
        CH.PROFILE.append(("x6_task_images_kernel", 0, ev0, ev1,
                           pt.numel() * 4 + (int(row) + int(tr)) * n_tasks * (1 if bf16 else 3) * width * width * 2))
    return ri, ti


def _weight_images(Ws: Sequence[torch.Tensor], kinds: Sequence[int], width: int = WIDTH, bf16: bool = False) -> List[List[torch.Tensor]]:
    """Per kind (1: of W, 2: of W^T) and weight matrix [F, F] its image: [3, F, F] (three exact bf16 terms), or -- ``bf16`` -- the
    one-term image [F, F] = bf16(W) (both k-permuted, ``npf_prepare_weights``)."""
    if bf16:
        n = len(Ws)
        buf = torch.empty((len(kinds), n, width, width), dtype=torch.bfloat16, device=Ws[0].device)
        CH.prepare_weights([(Ws[i].detach(), kind) for kind in kinds for i in range(n)],
                           dsts=[buf[k, i] for k in range(len(kinds)) for i in range(n)])
        return [[buf[k, i] for i in range(n)] for k in range(len(kinds))]
    from .mlp_x6 import _three_term_images

    return _three_term_images(Ws, kinds, width)


def _r16(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """A tensor as the bf16 mode's multiplies see it: rounded to bfloat16 (nearest even), kept as fp32."""
    return None if t is None else t.to(torch.bfloat16).to(torch.float32)


def _bits(n_tasks, tiles, dev):
    return torch.empty((n_tasks, tiles, 2, 64), dtype=torch.int64, device=dev)


_ZEROS = {}


def _zeros(shape, device) -> torch.Tensor:
    """A cached block of zeros (read-only: it only ever enters ``torch.cat``) -- padding by one copy launch instead of a fill and a
    copy per call."""
    key = (tuple(shape), str(device))
    z = _ZEROS.get(key)
    if z is None:
        z = torch.zeros(shape, dtype=torch.float32, device=device)
        if not (z.is_cuda and torch.cuda.is_current_stream_capturing()):  # (a block born inside a graph capture belongs to that graph)
            _ZEROS[key] = z
    return z


def _pad_rows4(X: torch.Tensor) -> torch.Tensor:
    """[B, P, d] (d <= 4) -> contiguous [B, P, 4], zero padded."""
    d = X.shape[-1]
    if d == 4:
        return X.contiguous()
    return torch.cat((X, _zeros((*X.shape[:-1], 4 - d), X.device)), dim=-1)


def _first_layer_matrix(W: torch.Tensor, n_out: int) -> torch.Tensor:
    """nn.Linear weight [n, d] (d <= 4) -> the rows-prologue matrix [4, n_out] = W^T, zero padded."""
    n, d = W.shape
    Wt = W.detach().t()
    if n < n_out:
        Wt = torch.cat((Wt, _zeros((d, n_out - n), W.device)), dim=1)
    if d < 4:
        Wt = torch.cat((Wt, _zeros((4 - d, n_out), W.device)), dim=0)
    return Wt.contiguous()


def decode_rows_usable(mod, x1: torch.Tensor, x2: torch.Tensor) -> bool:
    """Does ``decode_rows`` cover this sum-merge module and these row-major inference inputs: fp32 mode, x1 / x2 / every hidden
    layer F wide with F in (128, 256, 512), an output layer of <= 4 features, no residual, one x1 row per x2 row."""
    from .architectures import MLP

    if not (ENABLED and CH.COMPUTE_DTYPE == "fp32" and mod.is_sum_merge and isinstance(mod.flat_module, MLP)):
        return False
    F = x1.shape[-1]
    fm, rs = mod.flat_module, mod.resizer
    if F not in (128, 256, 512) or x2.shape[-1] != F or x1.shape[:-1] != x2.shape[:-1] or fm.is_res or rs.is_res:
        return False
    if any(m.dropout_p > 0 and m.training for m in (fm, rs)):  # (active dropout: the chain path draws and applies the masks)
        return False
    if not (_square(rs.layers(), F) and _square([fm.to_hidden, *fm.linears], F) and fm.out.in_features == F
            and fm.out.out_features <= 4):
        return False
    return len(rs.layers()) + len(fm.linears) + 1 <= L.NPF_X6_MAX_OPS and x1.shape[-2] > 0


def decode_rows(mod, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """``MergeFlatInputs.forward`` at inference (encoders.py:175-183: flat(relu(x1 + resizer(x2)))) as ONE x6 program straight
    from the row-major module-boundary tensors x1, x2 [n, T, F] -- what the reference's ``decode(X_trgt_enc, R_trgt)``
    (base.py:327-367) hands its decoder: no layout pass, no per-layer HBM traffic, the F -> 2 dy output layer on the registers
    the last hidden layer leaves.  Returns [n, T, n_out]."""
    n, T, F = x1.shape
    fm, rs = mod.flat_module, mod.resizer
    lins = [*rs.layers(), fm.to_hidden, *fm.linears]
    imgs = _weight_images([l.weight for l in lins], (1,), F)[0]
    tiles = CH.tiles_of(T)
    prog = Program(n, tiles, per_task=False, width=F, pts=T)
    n_res = len(rs.layers())
    for i, lin in enumerate(lins):
        o = dict(img=imgs[i], bias=lin.bias.detach() if lin.bias is not None else None, relu=True)
        if i == 0:
            o.update(in_pt=x2.detach().contiguous(), in_pt_rm=True)
        if i == n_res - 1:
            o.update(addend=x1.detach().contiguous(), addend_rm=True)
        prog.op(**o)
    rows = torch.empty((n, tiles * 32, 4), dtype=torch.float32, device=x1.device)
    Wo, bo = _pad_out(fm.out.weight, fm.out.bias)
    prog.tail = (Wo, bo, rows)
    prog.tag = f"decode (merge + {F}-wide decoder + output layer) from row-major inputs"
    prog.launch()
    return rows[:, :T, :fm.out.out_features]


def _pad_out(W: torch.Tensor, b: Optional[torch.Tensor]):
    """The F -> n_out (<= 4) output layer as a [4, F] matrix and a [4] bias."""
    n = W.shape[0]
    if n == 4:
        return W.detach().contiguous(), (b.detach().contiguous() if b is not None else None)
    Wp = torch.zeros((4, W.shape[1]), dtype=torch.float32, device=W.device)
    Wp[:n] = W.detach()
    bp = None
    if b is not None:
        bp = torch.zeros((4,), dtype=torch.float32, device=W.device)
        bp[:n] = b.detach()
    return Wp, bp


class _TargetSideFn(torch.autograd.Function):
    """rows [B, T, n_out] = decoder(x_encoder(X_trgt), attention(x_encoder(X_trgt), K, V)).

    Arguments: X [B, T, dx] raw target features; K_pt, V_pt PT32 [B, tilesC, 64, 32, 4] (encoded context points /
    their representations), C context points; ``spec`` = (n_xenc, n_mrz, n_res, n_flat): numbers of 256 -> 256 layers of the
    x-encoder behind its first layer, of the latent merge (0 or 1: AttnLNP's merge_r_z, base.py:554-575 / attnnp.py:183-202, as
    relu(W_R R_trgt + zb[task]) -- the z half of the concatenated input is constant per task and enters as the per-task bias
    ``zb`` [B, 256]), of the resizer, of the flat MLP in front of its output layer; params = W, b pairs in the order x-encoder
    (first layer, then the 256 -> 256 ones), merge (W_R, None), resizer, flat, output layer."""

    @staticmethod
    def forward(ctx, X, K_pt, V_pt, C, scale, spec, zb, *params):
        n_x, n_mrz, n_res, n_flat = spec
        B, T, dx = X.shape
        dev = X.device
        tiles = T // 32
        Ws, bs = list(params[0::2]), list(params[1::2])
        W1, b1 = Ws[0], bs[0]
        W_out, b_out = Ws[-1], bs[-1]
        n_out = W_out.shape[0]
        mid_W, mid_b = Ws[1:-1], bs[1:-1]  # the F -> F layers (F = 128 or 256): x-encoder rest, latent merge, resizer, flat
        F = mid_W[0].shape[0]
        if n_mrz:
            mid_b[n_x] = zb
        train = any(ctx.needs_input_grad)
        # bf16 compute mode (config 3): a b16 program -- one-term images, bf16-rounded rows / small matrices, and the tensors only
        # the backward pass reads as PT16 (``pb``); what an fp32 consumer reads (the merge's addend, the outputs) stays PT32
        bf16 = CH.COMPUTE_DTYPE == "bf16"
        imgs = _weight_images(mid_W, (1, 2) if train else (1,), F, bf16)
        fw = imgs[0]
        K_row, K_tr = task_images(K_pt, C, row=True, tr=train, width=F, bf16=bf16)
        V_row, V_tr = task_images(V_pt, C, row=train, tr=True, width=F, bf16=bf16)
        X4 = _pad_rows4(X.detach())
        W1p = _first_layer_matrix(W1, F)
        if bf16:
            X4, W1p = _r16(X4), _r16(W1p)
        pt = lambda: CH.pt_empty(B, T, F, dev)  # noqa: E731
        pb = (lambda: CH.pt16_empty(B, T, F, dev)) if bf16 else pt  # noqa: E731
        prog = Program(B, tiles, per_task=True, width=F, bf16=bf16)
        saved_acts, saved_bits = [], []
        # x-encoder: first layer in the prologue, then its 256 -> 256 layers (ReLU on all but the last, mlp.py:95-109)
        h1 = pb() if train else None
        bits_h1 = _bits(B, tiles, dev) if train else None
        Xt_enc = pt()
        for i in range(n_x):
            last = i == n_x - 1
            o = dict(img=fw[i], w_ref=("shared", mid_W[i]), bias=mid_b[i].detach() if mid_b[i] is not None else None, relu=not last)
            if i == 0:
                o.update(in_rows=X4, in_w=W1p, in_b=b1.detach() if b1 is not None else None, in_relu=True, store_in=h1,
                         store_in_bits=bits_h1)
            if last:
                o["store_out"], o["store_out_f32"] = Xt_enc, True
            elif train:
                o["store_out"], o["store_bits"] = pb(), _bits(B, tiles, dev)
                saved_acts.append(o["store_out"])
                saved_bits.append(o["store_bits"])
            prog.op(**o)
        # attention (attention.py:129-164, 204-220): scores = K q, softmax(scale .), attn . V
        P = pb() if train else None
        prog.op(img=K_row, w_ref=("task_row", K_pt, C), img_per_task=True, softmax_n=C, softmax_scale=scale, store_out=P, true_n=C)
        R_trgt = pb() if train else None
        prog.op(img=V_tr, w_ref=("task_tr", V_pt, C), img_per_task=True, store_out=R_trgt, true_k=C)
        # [latent merge relu(W_R R_trgt + zb[task])], decoder: resizer, merge relu(x1 + .) (encoders.py:178-179), flat MLP
        for i in range(n_mrz + n_res + n_flat):
            j = n_x + i
            o = dict(img=fw[j], w_ref=("shared", mid_W[j]), bias=mid_b[j].detach().contiguous() if mid_b[j] is not None else None,
                     relu=True, bias_per_task=bool(n_mrz and i == 0))
            if i == n_mrz + n_res - 1:
                o["addend"] = Xt_enc
            if train:
                o["store_out"], o["store_bits"] = pb(), _bits(B, tiles, dev)
                saved_acts.append(o["store_out"])
                saved_bits.append(o["store_bits"])
            prog.op(**o)
        rows = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
        Wo, bo = _pad_out(W_out, b_out)
        if bf16:
            Wo = _r16(Wo)
        prog.tail = (Wo, bo, rows)
        prog.tag = "target side forward (x-encoder, attention, decoder, output layer)"
        prog.launch()
        ctx.geom = (B, T, tiles, dx, C, float(scale), n_out, F)
        ctx.spec, ctx.bf16 = spec, bf16
        # (a traced run -- tests/teacher.py -- names the matrices behind the images of the backward launch too)
        ctx.trace_ref = ([W.detach() for W in mid_W], K_pt, V_pt) if CH.TRACE is not None else None
        ctx.has_b = [b is not None for b in bs]
        ctx.set_materialize_grads(False)
        if train:
            ctx.n_acts, ctx.n_bits = len(saved_acts), len(saved_bits)
            ctx.save_for_backward(X4, h1, bits_h1, Xt_enc, P, R_trgt, *saved_acts, *saved_bits, *imgs[1], V_row, K_tr, Wo)
        return rows[..., :n_out] if n_out != 4 else rows

    @staticmethod
    def backward(ctx, g):
        n_x, n_mrz, n_res, n_flat = ctx.spec
        n_dec = n_mrz + n_res + n_flat
        n_mid = n_x + n_dec
        n_par = 2 * (n_mid + 2)
        if g is None:
            return (None,) * (7 + n_par)
        B, T, tiles, dx, C, scale, n_out, F = ctx.geom
        sv = list(ctx.saved_tensors)
        X4, h1, bits_h1, Xt_enc, P, R_trgt = sv[:6]
        acts = sv[6:6 + ctx.n_acts]
        bits = sv[6 + ctx.n_acts:6 + ctx.n_acts + ctx.n_bits]
        rest = sv[6 + ctx.n_acts + ctx.n_bits:]
        bw, (V_row, K_tr, Wo) = rest[:n_mid], rest[n_mid:]
        dev = g.device
        bf16 = ctx.bf16
        g4 = g.contiguous() if n_out == 4 else torch.nn.functional.pad(g, (0, 4 - n_out)).contiguous()
        if bf16:
            g4 = _r16(g4)  # (the output layer's dZ as its dgrad and weight gradient see it; Wo was saved rounded)
        pt = lambda: CH.pt_empty(B, T, F, dev)  # noqa: E731
        pb = (lambda: CH.pt16_empty(B, T, F, dev)) if bf16 else pt  # noqa: E731
        # the saved outputs by layer: x-encoder hidden layers (n_x - 1 of them), then resizer + flat (all ReLU layers)
        x_acts, x_bits = acts[:n_x - 1], bits[:n_x - 1]
        d_acts, d_bits = acts[n_x - 1:], bits[n_x - 1:]
        prog = Program(B, tiles, per_task=True, width=F, bf16=bf16)
        tr = ctx.trace_ref
        wref = lambda j: ("shared", tr[0][j].t()) if tr is not None else None  # noqa: E731
        dz = [None] * n_mid  # dZ of every 256 -> 256 layer (index as in the forward: x-encoder, resizer, flat)
        # decoder layers (and the latent merge), last to first.  (bf16 mode: a dZ is a PT16 tensor -- the rounded values its
        # weight gradient multiplies and its bias gradient sums -- except the merge layer's, which is also the fp32 gradient of x1)
        for i in range(n_dec - 1, -1, -1):
            j = n_x + i
            # (... and the latent merge's: the per-task bias gradient is the fp32 sum of it over the task's points)
            is_merge = i == n_mrz + n_res - 1 or (n_mrz and i == 0)
            dz[j] = pt() if is_merge else pb()
            o = dict(mask_bits=d_bits[i], store_in=dz[j], store_in_f32=is_merge, img=bw[j], w_ref=wref(j))
            if i == n_dec - 1:
                o.update(in_rows=g4, in_w=Wo)  # the dgrad of the output layer in the prologue
            prog.op(**o)
        # attention backward: dO -> dP = V dO ; dS = softmax'(dP) ; dq = K^T dS, + the merge's gradient wrt x1 (fan-in)
        dO, dS = pb(), pb()
        prog.op(store_in=dO, img=V_row, w_ref=("task_row", tr[2], C) if tr is not None else None, img_per_task=True, true_n=C)
        prog.op(sbwd_p=P, sbwd_scale=scale, store_in=dS, img=K_tr, w_ref=("task_tr", tr[1], C) if tr is not None else None,
                img_per_task=True, addend=dz[n_x + n_mrz + n_res - 1], true_k=C)
        # x-encoder, last to first; the first layer's dZ behind its ReLU mask closes the program
        for i in range(n_x - 1, -1, -1):
            dz[i] = pb()
            o = dict(store_in=dz[i], img=bw[i], w_ref=wref(i))
            if i < n_x - 1:
                o["mask_bits"] = x_bits[i]
            prog.op(**o)
        dz1 = pb()
        prog.op(mask_bits=bits_h1, store_in=dz1)
        prog.tag = "target side dgrad (decoder, attention backward, x-encoder)"
        prog.launch()
        # weight / key / value gradients
        jobs, grads = [], []
        # (the first layer's fan-in zero-padded to 4, like the chain path pads skinny first layers)
        dW1p = torch.empty((F, 4), dtype=torch.float32, device=dev)
        db1 = torch.empty((F,), dtype=torch.float32, device=dev) if ctx.has_b[0] else None
        jobs.append(dict(dZ=dz1, A=FN._pack(X4), N=F, K=4, dW=dW1p, db=db1))
        grads += [None, db1]  # (dW1 sliced out of dW1p after the launch)
        ins = [h1, *x_acts, R_trgt, *d_acts[:-1]]  # input of every 256 -> 256 layer
        for j in range(n_mid):
            dW = torch.empty((F, F), dtype=torch.float32, device=dev)
            db = torch.empty((F,), dtype=torch.float32, device=dev) if (ctx.has_b[1 + j] and not (n_mrz and j == n_x)) else None
            jobs.append(dict(dZ=dz[j], A=ins[j], N=F, K=F, dW=dW, db=db))
            grads += [dW, db]
        dz_out = torch.zeros(CH.pt_shape(B, T, 4), dtype=torch.float32, device=dev)
        dz_out[:, :, 0] = g4.view(B, tiles, 32, 4)
        dWo = torch.empty((n_out, F), dtype=torch.float32, device=dev)
        dbo = torch.empty((n_out,), dtype=torch.float32, device=dev) if ctx.has_b[-1] else None
        jobs.append(dict(dZ=dz_out, A=d_acts[-1], N=n_out, K=F, dW=dWo, db=dbo))
        grads += [dWo, dbo]
        # (context points that do not fill their last tile: the padding rows of dK / dV are zero -- their keys have no weight)
        mk = torch.zeros if C % 32 else torch.empty
        dK = mk(CH.pt_shape(B, C, F), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        dV = mk(CH.pt_shape(B, C, F), dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        if dV is not None:
            jobs.append(dict(dZ=P, A=dO, N=C, K=F, dW=dV, per_task=True, ldz=F))  # (P has F features per tile)
        if dK is not None:
            jobs.append(dict(dZ=dS, A=Xt_enc, N=C, K=F, dW=dK, per_task=True, ldz=F))
        CH.run_wgrad(jobs, B, T, dev, tag="target side weight / key / value gradients")
        if CH.TRACE is not None:
            CH.TRACE.append(("wgrad", jobs, B, T, bf16))
        grads[0] = dW1p[:, :dx].contiguous() if dx != 4 else dW1p
        # the per-task bias of the latent merge: the sum of its dZ over the task's points (padding points carry zeros: their
        # incoming gradient is zero)
        d_zb = FN.sum_points_pt(dz[n_x], T, F)[:, :F].contiguous() if (n_mrz and ctx.needs_input_grad[6]) else None
        return (None, dK, dV, None, None, None, d_zb, *grads)


class _DecoderSideFn(torch.autograd.Function):
    """rows [B, T, n_out] = decoder(x1, R) = flat(relu(x1 + resizer(R))) + output layer (MergeFlatInputs.forward, encoders.py:175-183,
    as ``NeuralProcessFamily.decode`` calls it, base.py:327-367) from PT32 tensors: ONE launch forward, ONE for its dgrad, then the
    weight-gradient jobs -- the decoder half of ``_TargetSideFn`` for models whose attention is not the fused scaled-dot one
    (multihead / transformer attention, more than 256 keys).  ``spec`` = (n_mrz, n_res, n_flat): AttnLNP's latent merge in front
    (0 or 1: relu(W_R R + zb[task]), base.py:554-575, ``zb`` [B, F] the latent half as a per-task bias), F -> F layers of the
    resizer, of the flat MLP in front of its output layer; params = W, b pairs: [merge (W_R, None)], resizer, flat, output layer."""

    @staticmethod
    def forward(ctx, R_pt, X1_pt, T, spec, zb, *params):
        n_mrz, n_res, n_flat = spec
        B, tiles = R_pt.shape[0], R_pt.shape[1]
        dev = R_pt.device
        Ws, bs = list(params[0::2]), list(params[1::2])
        W_out, b_out = Ws[-1], bs[-1]
        n_out = W_out.shape[0]
        mid_W, mid_b = Ws[:-1], bs[:-1]
        F = mid_W[0].shape[0]
        n_mid = n_mrz + n_res + n_flat
        if n_mrz:
            mid_b[0] = zb
        train = any(ctx.needs_input_grad)
        imgs = _weight_images(mid_W, (1, 2) if train else (1,), F)
        fw = imgs[0]
        R_pt, X1_pt = R_pt.contiguous(), X1_pt.contiguous()
        pt = lambda: CH.pt_empty(B, tiles * 32, F, dev)  # noqa: E731
        prog = Program(B, tiles, per_task=bool(n_mrz), width=F)
        acts, bits = [], []
        for i in range(n_mid):
            o = dict(img=fw[i], w_ref=("shared", mid_W[i]), bias=mid_b[i].detach().contiguous() if mid_b[i] is not None else None,
                     relu=True, bias_per_task=bool(n_mrz and i == 0))
            if i == 0:
                o["in_pt"] = R_pt.detach()
            if i == n_mrz + n_res - 1:
                o["addend"] = X1_pt.detach()
            if train:
                o["store_out"], o["store_bits"] = pt(), _bits(B, tiles, dev)
                acts.append(o["store_out"])
                bits.append(o["store_bits"])
            prog.op(**o)
        rows = torch.empty((B, tiles * 32, 4), dtype=torch.float32, device=dev)
        Wo, bo = _pad_out(W_out, b_out)
        prog.tail = (Wo, bo, rows)
        prog.tag = "decoder forward (resizer, merge, flat MLP, output layer)"
        prog.launch()
        ctx.geom = (B, T, tiles, n_out, F)
        ctx.spec = spec
        ctx.has_b = [b is not None for b in bs]
        ctx.set_materialize_grads(False)
        ctx.trace_ref = [W.detach() for W in mid_W] if CH.TRACE is not None else None
        if train:
            ctx.save_for_backward(R_pt.detach(), *acts, *bits, *imgs[1], Wo)
        return rows[:, :T, :n_out]

    @staticmethod
    def backward(ctx, g):
        n_mrz, n_res, n_flat = ctx.spec
        n_mid = n_mrz + n_res + n_flat
        if g is None:
            return (None,) * (5 + 2 * (n_mid + 1))
        B, T, tiles, n_out, F = ctx.geom
        sv = list(ctx.saved_tensors)
        R_pt, acts, bits, bw, Wo = sv[0], sv[1:1 + n_mid], sv[1 + n_mid:1 + 2 * n_mid], sv[1 + 2 * n_mid:1 + 3 * n_mid], sv[-1]
        dev = g.device
        if T == tiles * 32 and n_out == 4:
            g4 = g.contiguous()
        else:
            g4 = torch.zeros((B, tiles * 32, 4), dtype=torch.float32, device=dev)  # (padding points and outputs: zero gradient)
            g4[:, :T, :n_out] = g
        pt = lambda: CH.pt_empty(B, tiles * 32, F, dev)  # noqa: E731
        tr = ctx.trace_ref
        prog = Program(B, tiles, per_task=bool(n_mrz), width=F)
        dz = [None] * n_mid
        dR = pt() if ctx.needs_input_grad[0] else None
        for i in range(n_mid - 1, -1, -1):
            dz[i] = pt()
            o = dict(mask_bits=bits[i], store_in=dz[i], img=bw[i], w_ref=("shared", tr[i].t()) if tr is not None else None)
            if i == n_mid - 1:
                o.update(in_rows=g4, in_w=Wo)  # the dgrad of the output layer in the prologue
            if i == 0 and dR is not None:
                o["store_out"] = dR
            prog.op(**o)
        prog.tag = "decoder dgrad"
        prog.launch()
        jobs, grads = [], []
        ins = [R_pt, *acts[:-1]]
        for j in range(n_mid):
            dW = torch.empty((F, F), dtype=torch.float32, device=dev)
            db = torch.empty((F,), dtype=torch.float32, device=dev) if (ctx.has_b[j] and not (n_mrz and j == 0)) else None
            jobs.append(dict(dZ=dz[j], A=ins[j], N=F, K=F, dW=dW, db=db))
            grads += [dW, db]
        dz_out = torch.zeros(CH.pt_shape(B, tiles * 32, 4), dtype=torch.float32, device=dev)
        dz_out[:, :, 0] = g4.view(B, tiles, 32, 4)
        dWo = torch.empty((n_out, F), dtype=torch.float32, device=dev)
        dbo = torch.empty((n_out,), dtype=torch.float32, device=dev) if ctx.has_b[-1] else None
        jobs.append(dict(dZ=dz_out, A=acts[-1], N=n_out, K=F, dW=dWo, db=dbo))
        grads += [dWo, dbo]
        CH.run_wgrad(jobs, B, tiles * 32, dev, tag="decoder weight gradients")
        if CH.TRACE is not None:
            CH.TRACE.append(("wgrad", jobs, B, tiles * 32, False))
        dX1 = dz[n_mrz + n_res - 1] if ctx.needs_input_grad[1] else None  # (the merge adds x1 in front of its ReLU: its dZ is x1's gradient)
        # the per-task bias of the latent merge: the sum of its dZ over the task's points (padding points carry zeros)
        d_zb = FN.sum_points_pt(dz[0], tiles * 32, F)[:, :F].contiguous() if (n_mrz and ctx.needs_input_grad[4]) else None
        return (dR, dX1, None, None, d_zb, *grads)


def decoder_side_usable(model, T: int) -> bool:
    """Does ``decoder_side`` cover this model: fp32 mode, a sum-merge decoder whose layers are all F x F with F = 128 (the
    256-wide ones already run as one launch on ``mlp_x6``), an output layer of <= 4 features, no residual / dropout."""
    from .architectures import MLP, MergeFlatInputs

    if not (ENABLED and CH.COMPUTE_DTYPE == "fp32") or T <= 0:
        return False
    dec = getattr(model, "decoder", None)
    F = getattr(model, "r_dim", 0)
    if F != 128 or getattr(model, "x_transf_dim", F) != F or not isinstance(dec, MergeFlatInputs):
        return False
    if not (dec.is_sum_merge and isinstance(dec.flat_module, MLP)):
        return False
    fm, rs = dec.flat_module, dec.resizer
    for m in (fm, rs):
        if m.is_res or (m.dropout_p > 0 and m.training):
            return False
    if not (_square(rs.layers(), F) and _square([fm.to_hidden, *fm.linears], F) and fm.out.in_features == F
            and fm.out.out_features <= 4):
        return False
    return len(rs.layers()) + len(fm.linears) + 2 <= L.NPF_X6_MAX_OPS


def decoder_side(model, R_pt: torch.Tensor, X1_pt: torch.Tensor, T: int, zb: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The decoder's sufficient statistics [B, T, 2 dy] from the target representations and the encoded targets (PT32 tensors
    [B, tiles, F/4, 32, 4]; ``decoder_side_usable``).  ``zb`` [B, F]: AttnLNP's merge_r_z in front, as in ``target_side``."""
    dec = model.decoder
    fm, rs = dec.flat_module, dec.resizer
    lins = [*rs.layers(), fm.to_hidden, *fm.linears, fm.out]
    params = []
    for lin in lins:
        params += [lin.weight, lin.bias]
    if zb is not None:
        params[0:0] = [model.r_z_merger.weight[:, :model.r_dim], None]
    return _DecoderSideFn.apply(R_pt, X1_pt, T, (int(zb is not None), len(rs.layers()), len(fm.linears) + 1), zb, *params)


class _PairLinearFn(torch.autograd.Function):
    """(W_a a, W_b b) for two PT32 tensors of the same geometry and two bias-free F x F matrices as ONE program launch (two ops, each
    with its own input), their dgrad as one launch, both weight gradients as one launch: the key and value projections of
    MultiheadAttender (key_transform / value_transform, attention.py:397-404: Linear(bias=False) on the encoded context points
    and on their representations)."""

    @staticmethod
    def forward(ctx, a_pt, b_pt, pts, Wa, Wb):
        B, tiles = a_pt.shape[0], a_pt.shape[1]
        F = Wa.shape[0]
        dev = a_pt.device
        train = any(ctx.needs_input_grad)
        imgs = _weight_images([Wa, Wb], (1, 2) if train else (1,), F)
        a_pt, b_pt = a_pt.contiguous(), b_pt.contiguous()
        oa, ob = CH.pt_empty(B, tiles * 32, F, dev), CH.pt_empty(B, tiles * 32, F, dev)
        prog = Program(B, tiles, per_task=False, width=F)
        prog.op(in_pt=a_pt.detach(), img=imgs[0][0], w_ref=("shared", Wa), store_out=oa)
        prog.op(in_pt=b_pt.detach(), img=imgs[0][1], w_ref=("shared", Wb), store_out=ob)
        prog.tag = "key / value projections"
        prog.launch()
        ctx.geom = (B, tiles, F, pts)
        ctx.set_materialize_grads(False)
        if train:
            ctx.save_for_backward(a_pt.detach(), b_pt.detach(), *imgs[1])
        return oa, ob

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return (None,) * 5
        B, tiles, F, pts = ctx.geom
        a_pt, b_pt, ia, ib = ctx.saved_tensors
        dev = a_pt.device
        zeros = lambda: torch.zeros(CH.pt_shape(B, tiles * 32, F), dtype=torch.float32, device=dev)  # noqa: E731
        ga = ga.contiguous() if ga is not None else zeros()
        gb = gb.contiguous() if gb is not None else zeros()
        da, db = CH.pt_empty(B, tiles * 32, F, dev), CH.pt_empty(B, tiles * 32, F, dev)
        prog = Program(B, tiles, per_task=False, width=F)
        prog.op(in_pt=ga, img=ia, store_out=da)
        prog.op(in_pt=gb, img=ib, store_out=db)
        prog.tag = "key / value projections dgrad"
        prog.launch()
        dWa, dWb = (torch.empty((F, F), dtype=torch.float32, device=dev) for _ in range(2))
        # (``pts`` valid points: the padding points of the incoming gradients are zero or never written -- count only the valid ones)
        CH.run_wgrad([dict(dZ=ga, A=a_pt, N=F, K=F, dW=dWa, db=None), dict(dZ=gb, A=b_pt, N=F, K=F, dW=dWb, db=None)], B, pts, dev,
                     tag="key / value projection weight gradients")
        return da, db, None, dWa, dWb


def pair_linear_usable(lin_a, lin_b) -> bool:
    F = lin_a.in_features
    return (ENABLED and CH.COMPUTE_DTYPE == "fp32" and F in (128, 256) and lin_a.bias is None and lin_b.bias is None
            and _square([lin_a, lin_b], F))


def pair_linear(a_pt: torch.Tensor, b_pt: torch.Tensor, pts: int, lin_a, lin_b):
    """(lin_a(a), lin_b(b)) on PT32 tensors [B, tiles, F/4, 32, 4] with ``pts`` valid points per task (``pair_linear_usable``)."""
    return _PairLinearFn.apply(a_pt, b_pt, pts, lin_a.weight, lin_b.weight)


class _XEncProjFn(torch.autograd.Function):
    """(X_enc, Q) PT32 = x_encoder(X), query_transform(X_enc) from the raw features X [B, P, dx] (P whole tiles) in one launch, the
    dgrad of both in one launch, their weight gradients in one: the target side in front of a multihead / transformer attention
    (MLP.forward mlp.py:95-109; MultiheadAttender.query_transform, attention.py:397-404, a Linear with bias).  ``n_x`` = F -> F
    layers of the x-encoder behind its first layer; params = W, b pairs: x-encoder (first layer, the F -> F ones), projection."""

    @staticmethod
    def forward(ctx, X, n_x, *params):
        B, P, dx = X.shape
        dev = X.device
        tiles = P // 32
        Ws, bs = list(params[0::2]), list(params[1::2])
        W1, b1 = Ws[0], bs[0]
        x_W, x_b = Ws[1:1 + n_x], bs[1:1 + n_x]
        Wq, bq = Ws[-1], bs[-1]
        F = Wq.shape[0]
        train = any(ctx.needs_input_grad)
        imgs = _weight_images([*x_W, Wq], (1, 2) if train else (1,), F)
        fw = imgs[0]
        X4 = _pad_rows4(X.detach())
        pt = lambda: CH.pt_empty(B, P, F, dev)  # noqa: E731
        prog = Program(B, tiles, per_task=False, width=F)
        acts, bits = [], []
        h1 = pt() if train else None
        bits_h1 = _bits(B, tiles, dev) if train else None
        X_enc, Q = pt(), pt()
        for i in range(n_x):
            last = i == n_x - 1
            o = dict(img=fw[i], w_ref=("shared", x_W[i]), bias=x_b[i].detach() if x_b[i] is not None else None, relu=not last)
            if i == 0:
                o.update(in_rows=X4, in_w=_first_layer_matrix(W1, F), in_b=b1.detach() if b1 is not None else None, in_relu=True,
                         store_in=h1, store_in_bits=bits_h1)
            if last:
                o["store_out"] = X_enc
            elif train:
                o["store_out"], o["store_bits"] = pt(), _bits(B, tiles, dev)
                acts.append(o["store_out"])
                bits.append(o["store_bits"])
            prog.op(**o)
        prog.op(img=fw[n_x], w_ref=("shared", Wq), bias=bq.detach() if bq is not None else None, store_out=Q)
        prog.tag = "x-encoder + query projection"
        prog.launch()
        ctx.geom = (B, P, tiles, dx, F)
        ctx.n_x = n_x
        ctx.has_b = [b is not None for b in bs]
        ctx.set_materialize_grads(False)
        if train:
            ctx.save_for_backward(X4, h1, bits_h1, X_enc, *acts, *bits, *imgs[1])
        return X_enc, Q

    @staticmethod
    def backward(ctx, gX, gQ):
        n_x = ctx.n_x
        n_par = 2 * (n_x + 2)
        if gX is None and gQ is None:
            return (None,) * (2 + n_par)
        B, P, tiles, dx, F = ctx.geom
        sv = list(ctx.saved_tensors)
        X4, h1, bits_h1, X_enc = sv[:4]
        acts, bits, bw = sv[4:4 + n_x - 1], sv[4 + n_x - 1:4 + 2 * (n_x - 1)], sv[4 + 2 * (n_x - 1):]
        dev = X4.device
        pt = lambda: CH.pt_empty(B, P, F, dev)  # noqa: E731
        prog = Program(B, tiles, per_task=False, width=F)
        jobs, grads = [], [None] * n_par

        def wjob(pos, dZ, A, N, K):
            dW = torch.empty((N, K), dtype=torch.float32, device=dev)
            db = torch.empty((N,), dtype=torch.float32, device=dev) if ctx.has_b[pos] else None
            jobs.append(dict(dZ=dZ, A=A, N=N, K=K, dW=dW, db=db))
            grads[2 * pos], grads[2 * pos + 1] = dW, db
            return dW

        if gQ is not None:
            gQ = gQ.contiguous()
            prog.op(in_pt=gQ, img=bw[n_x])          # the projection's dgrad; the x-encoder's own gradient joins in front of the next op
            wjob(n_x + 1, gQ, X_enc, F, F)
            first = dict(pre_add=gX.contiguous()) if gX is not None else {}
        else:
            first = dict(in_pt=gX.contiguous())
        x_in = [h1, *acts]
        for i in range(n_x - 1, -1, -1):
            dz = pt()
            o = dict(store_in=dz, img=bw[i])
            if i == n_x - 1:
                o.update(first)
            else:
                o["mask_bits"] = bits[i]
            prog.op(**o)
            wjob(1 + i, dz, x_in[i], F, F)
        dz1 = pt()
        prog.op(mask_bits=bits_h1, store_in=dz1)
        dW1p = wjob(0, dz1, FN._pack(X4), F, 4)
        prog.tag = "x-encoder + query projection dgrad"
        prog.launch()
        CH.run_wgrad(jobs, B, P, dev, tag="x-encoder + query projection weight gradients")
        if dx != 4:
            grads[0] = dW1p[:, :dx].contiguous()
        return (None, None, *grads)


def xenc_proj_usable(model, lin_q, T: int) -> bool:
    """Does ``xenc_proj`` cover this model: fp32, stock MLP x-encoder (<= 4 inputs) with F-wide layers, F in (128, 256), a F -> F
    projection, no residual / dropout."""
    from .architectures import MLP

    if not (ENABLED and CH.COMPUTE_DTYPE == "fp32") or T <= 0:
        return False
    F = _width_of(model)
    xe = model.x_encoder
    if not F or not (isinstance(xe, MLP) and xe.input_size <= 4 and xe.hidden_size == F and xe.output_size == F
                     and not xe.is_res and not (xe.dropout_p > 0 and xe.training)):
        return False
    return lin_q.in_features == F and lin_q.out_features == F and len(xe.linears) + 3 <= L.NPF_X6_MAX_OPS


def xenc_proj(model, X: torch.Tensor, lin_q):
    """(encoded points as a :class:`~chain.PTensor`, their projection as a PT32 tensor) -- ``xenc_proj_usable``."""
    xe = model.x_encoder
    lins = [xe.to_hidden, *xe.linears, xe.out, lin_q]
    params = []
    for lin in lins:
        params += [lin.weight, lin.bias]
    T = X.shape[1]
    if T % 32:
        X = torch.nn.functional.pad(X, (0, 0, 0, CH.pad32(T) - T))
    Xe, Q = _XEncProjFn.apply(X, len(xe.linears) + 1, *params)
    return CH.PTensor(Xe, T, model.r_dim), Q


class _MlpPtFn(torch.autograd.Function):
    """``MLP.forward`` (mlp.py:95-109: to_hidden -> relu -> [linears -> relu]* -> out) on a PT32 tensor whose layers are all F x F,
    as one program launch each way + one weight-gradient launch: the MLP block of TransformerAttender (attention.py:576-588) at
    F = 128 / 256.  params = W, b pairs in layer order; ReLU behind all but the last."""

    @staticmethod
    def forward(ctx, x_pt, pts, *params):
        B, tiles = x_pt.shape[0], x_pt.shape[1]
        dev = x_pt.device
        Ws, bs = list(params[0::2]), list(params[1::2])
        n, F = len(Ws), Ws[0].shape[0]
        train = any(ctx.needs_input_grad)
        imgs = _weight_images(Ws, (1, 2) if train else (1,), F)
        x_pt = x_pt.contiguous()
        pt = lambda: CH.pt_empty(B, tiles * 32, F, dev)  # noqa: E731
        prog = Program(B, tiles, per_task=False, width=F)
        acts, bits = [], []
        y = pt()
        for i in range(n):
            last = i == n - 1
            o = dict(img=imgs[0][i], w_ref=("shared", Ws[i]), bias=bs[i].detach() if bs[i] is not None else None, relu=not last)
            if i == 0:
                o["in_pt"] = x_pt.detach()
            if last:
                o["store_out"] = y
            elif train:
                o["store_out"], o["store_bits"] = pt(), _bits(B, tiles, dev)
                acts.append(o["store_out"])
                bits.append(o["store_bits"])
            prog.op(**o)
        prog.tag = "MLP block forward"
        prog.launch()
        ctx.geom = (B, tiles, F, pts, n)
        ctx.has_b = [b is not None for b in bs]
        ctx.set_materialize_grads(False)
        if train:
            ctx.save_for_backward(x_pt.detach(), *acts, *bits, *imgs[1])
        return y

    @staticmethod
    def backward(ctx, g):
        B, tiles, F, pts, n = ctx.geom
        if g is None:
            return (None,) * (2 + 2 * n)
        sv = list(ctx.saved_tensors)
        x_pt, acts, bits, bw = sv[0], sv[1:n], sv[n:2 * n - 1], sv[2 * n - 1:]
        dev = g.device
        g = g.contiguous()
        pt = lambda: CH.pt_empty(B, tiles * 32, F, dev)  # noqa: E731
        prog = Program(B, tiles, per_task=False, width=F)
        dz = [None] * n
        dx = pt()
        for i in range(n - 1, -1, -1):
            o = dict(img=bw[i])
            if i == n - 1:
                o["in_pt"], dz[i] = g, g       # (no activation behind the last layer: its dZ is the incoming gradient)
            else:
                dz[i] = pt()
                o.update(mask_bits=bits[i], store_in=dz[i])
            if i == 0:
                o["store_out"] = dx
            prog.op(**o)
        prog.tag = "MLP block dgrad"
        prog.launch()
        ins = [x_pt, *acts]
        jobs, grads = [], []
        for i in range(n):
            dW = torch.empty((F, F), dtype=torch.float32, device=dev)
            db = torch.empty((F,), dtype=torch.float32, device=dev) if ctx.has_b[i] else None
            jobs.append(dict(dZ=dz[i], A=ins[i], N=F, K=F, dW=dW, db=db))
            grads += [dW, db]
        CH.run_wgrad(jobs, B, pts, dev, tag="MLP block weight gradients")
        return (dx, None, *grads)


def mlp_pt_usable(mlp) -> bool:
    """Does ``mlp_pt`` cover this MLP: fp32 mode, every layer F x F with F in (128, 256), no residual / dropout."""
    from .architectures import MLP

    if not (ENABLED and CH.COMPUTE_DTYPE == "fp32" and isinstance(mlp, MLP)):
        return False
    F = mlp.input_size
    if F not in (128, 256) or mlp.is_res or (mlp.dropout_p > 0 and mlp.training):
        return False
    return _square(mlp.layers(), F) and len(mlp.layers()) <= L.NPF_X6_MAX_OPS


def mlp_pt(mlp, x_pt: torch.Tensor, pts: int) -> torch.Tensor:
    """``mlp(x)`` on a PT32 tensor [B, tiles, F/4, 32, 4] (``mlp_pt_usable``)."""
    params = []
    for lin in mlp.layers():
        params += [lin.weight, lin.bias]
    return _MlpPtFn.apply(x_pt, pts, *params)


def _width_of(model) -> int:
    """The feature width F of the model's wide layers if the x6 programs have an instance for it (128, 256), else 0."""
    F = getattr(model, "r_dim", 0)
    return F if (F in (128, 256) and getattr(model, "x_transf_dim", F) == F) else 0


def _mode_ok() -> bool:
    """The fused sides exist for the fp32 mode (x6 programs) and the bf16 compute mode (b16 programs)."""
    return ENABLED and (CH.COMPUTE_DTYPE == "fp32" or (CH.COMPUTE_DTYPE == "bf16" and B16_ENABLED))


def _square(lins, width=WIDTH) -> bool:
    return all(l.in_features == width and l.out_features == width for l in lins)


def target_side_usable(model, C: int, T: int, latent_merge: bool = False) -> bool:
    """Does the fused target side cover this model and batch: scaled-dot attention over 128 < C <= 256 context points, every
    wide layer 256 -> 256, no residual / dropout; ``latent_merge``: with AttnLNP's merge_r_z between attention and decoder."""
    from .architectures import MLP, DotAttender

    if not _mode_ok():
        return False
    WIDTH = _width_of(model)
    if not WIDTH:
        return False
    xe, dec, att = model.x_encoder, model.decoder, getattr(model, "attender", None)
    if not isinstance(att, DotAttender) or att.kq_size != WIDTH or att.value_size != WIDTH:
        return False
    # (fewer keys than half a score row: at 256 features the chain path's 32-key granularity wins; the 128-wide programs take any
    # 1 <= C <= 128 -- the reference's 1-D experiments draw 0..50 context points)
    if not ((WIDTH // 2 < C or (WIDTH == 128 and C >= 1)) and C <= WIDTH and T > 0):
        return False
    if not (isinstance(xe, MLP) and xe.input_size <= 4 and xe.hidden_size == WIDTH and xe.output_size == WIDTH
            and not xe.is_res and not (xe.dropout_p > 0 and xe.training)):
        return False
    if not (dec.is_sum_merge and isinstance(dec.flat_module, MLP)):
        return False
    fm, rs = dec.flat_module, dec.resizer
    for m in (fm, rs):
        if m.is_res or (m.dropout_p > 0 and m.training):
            return False
    if not (_square(rs.layers(), WIDTH) and _square([fm.to_hidden, *fm.linears], WIDTH) and fm.out.in_features == WIDTH
            and fm.out.out_features <= 4):
        return False
    n_mrz = int(latent_merge)
    if n_mrz:
        mz = getattr(model, "r_z_merger", None)
        if mz is None or mz.out_features != WIDTH or mz.in_features <= WIDTH or model.r_dim != WIDTH:
            return False
    n_x, n_res, n_flat = len(xe.linears) + 1, len(rs.layers()), len(fm.linears) + 1
    # forward: n_x + 2 + n_mrz + n_res + n_flat ops; dgrad: the same + 1
    return n_x + 2 + n_mrz + n_res + n_flat + 1 <= L.NPF_X6_MAX_OPS


def target_side(model, X_trgt: torch.Tensor, K: CH.PTensor, V: CH.PTensor, zb: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The decoder's sufficient statistics [B, T, 2 dy] of an attentive model from the raw target features and the context
    side's PT32 outputs (``target_side_usable`` must hold).  ``zb`` [B, 256]: the latent half of AttnLNP's merge_r_z,
    W_z z + b per task (one latent sample) -- the merge then runs between attention and decoder as relu(W_R R_trgt + zb)."""
    xe, dec, att = model.x_encoder, model.decoder, model.attender
    fm, rs = dec.flat_module, dec.resizer
    lins = [xe.to_hidden, *xe.linears, xe.out, *rs.layers(), fm.to_hidden, *fm.linears, fm.out]
    params = []
    for lin in lins:
        params += [lin.weight, lin.bias]
    n_x = len(xe.linears) + 1
    if zb is not None:
        params[2 * (n_x + 1):2 * (n_x + 1)] = [model.r_z_merger.weight[:, :model.r_dim], None]
    spec = (n_x, int(zb is not None), len(rs.layers()), len(fm.linears) + 1)
    scale = 1.0 / math.sqrt(att.kq_size) if att.is_scale else 1.0
    T = X_trgt.shape[1]
    Tp = CH.pad32(T)
    if Tp != T:  # whole tiles of targets: the padding points (x = 0) are computed and dropped, their gradients are zero
        X_trgt = torch.nn.functional.pad(X_trgt, (0, 0, 0, Tp - T))
    rows = _TargetSideFn.apply(X_trgt, K.t, V.t, K.pts, scale, spec, zb, *params)
    return rows if Tp == T else rows[:, :T].contiguous()


class _ContextSideFn(torch.autograd.Function):
    """(Xc_enc, R) PT32 [B, tilesC, 64, 32, 4] each = x_encoder(X_cntxt), xy_encoder(Xc_enc, Y_cntxt) (attnnp.py:105-116 /
    np.py:86-95 up to the pooling; encoders.py:175-183: flat(relu(x1 + resizer(y)))) in one launch; the dgrad of both in one
    launch.  ``spec`` = (n_xenc, n_flat): 256 -> 256 layers of the x-encoder behind its first layer, of the flat MLP (its output
    layer included); params = W, b pairs: x-encoder, resizer (dy -> h, h -> 256), flat."""

    @staticmethod
    def forward(ctx, X, Y, spec, *params):
        n_x, n_flat = spec
        B, Cn, dx = X.shape
        dy = Y.shape[-1]
        dev = X.device
        tiles = Cn // 32
        Ws, bs = list(params[0::2]), list(params[1::2])
        W1, b1 = Ws[0], bs[0]
        x_W, x_b = Ws[1:1 + n_x], bs[1:1 + n_x]
        Wr1, br1, Wr2, br2 = Ws[1 + n_x], bs[1 + n_x], Ws[2 + n_x], bs[2 + n_x]
        f_W, f_b = Ws[3 + n_x:], bs[3 + n_x:]
        h = Wr1.shape[0]
        F = Wr2.shape[0]  # the width of every other layer: 128 or 256
        train = any(ctx.needs_input_grad)
        bf16 = CH.COMPUTE_DTYPE == "bf16"  # (see _TargetSideFn)
        Wr2p = torch.zeros((F, F), dtype=torch.float32, device=dev)  # (the h -> 256 layer as a 256-input layer:
        Wr2p[:, :h] = Wr2.detach()                                           #  the input's other registers are zero)
        imgs = _weight_images([*x_W, Wr2p, *f_W], (1, 2) if train else (1,), F, bf16)
        fw = imgs[0]
        X4, Y4 = _pad_rows4(X.detach()), _pad_rows4(Y.detach())
        W1p, Wr1p = _first_layer_matrix(W1, F), _first_layer_matrix(Wr1, h)
        if bf16:
            X4, Y4, W1p, Wr1p = _r16(X4), _r16(Y4), _r16(W1p), _r16(Wr1p)
        pt = lambda: CH.pt_empty(B, Cn, F, dev)  # noqa: E731
        pb = (lambda: CH.pt16_empty(B, Cn, F, dev)) if bf16 else pt  # noqa: E731
        prog = Program(B, tiles, per_task=False, width=F, bf16=bf16)
        acts, bits = [], []
        h1 = pb() if train else None
        bits_h1 = _bits(B, tiles, dev) if train else None
        Xc_enc = pt()
        for i in range(n_x):
            last = i == n_x - 1
            o = dict(img=fw[i], w_ref=("shared", x_W[i]), bias=x_b[i].detach() if x_b[i] is not None else None, relu=not last)
            if i == 0:
                o.update(in_rows=X4, in_w=W1p, in_b=b1.detach() if b1 is not None else None,
                         in_relu=True, store_in=h1, store_in_bits=bits_h1)
            if last:
                o["store_out"], o["store_out_f32"] = Xc_enc, True
            elif train:
                o["store_out"], o["store_bits"] = pb(), _bits(B, tiles, dev)
                acts.append(o["store_out"])
                bits.append(o["store_bits"])
            prog.op(**o)
        # resizer(y): first layer in the prologue (h features), second layer + x1 + ReLU = the merge
        hy = pb() if train else None
        bits_hy = _bits(B, tiles, dev) if train else None
        o = dict(in_rows=Y4, in_w=Wr1p, in_b=br1.detach() if br1 is not None else None, in_relu=True,
                 store_in=hy, store_in_bits=bits_hy, img=fw[n_x], w_ref=("shared", Wr2p),
                 bias=br2.detach() if br2 is not None else None, relu=True, addend=Xc_enc, true_k=h)
        if train:
            o["store_out"], o["store_bits"] = pb(), _bits(B, tiles, dev)
            acts.append(o["store_out"])
            bits.append(o["store_bits"])
        prog.op(**o)
        R = pt()
        for i in range(n_flat):
            last = i == n_flat - 1
            o = dict(img=fw[n_x + 1 + i], w_ref=("shared", f_W[i]), bias=f_b[i].detach() if f_b[i] is not None else None,
                     relu=not last)
            if last:
                o["store_out"], o["store_out_f32"] = R, True
            elif train:
                o["store_out"], o["store_bits"] = pb(), _bits(B, tiles, dev)
                acts.append(o["store_out"])
                bits.append(o["store_bits"])
            prog.op(**o)
        prog.tag = "context side forward (x-encoder, XY-encoder)"
        prog.launch()
        ctx.geom = (B, Cn, tiles, dx, dy, h, F)
        ctx.spec, ctx.bf16 = spec, bf16
        ctx.trace_ref = [W.detach() for W in (*x_W, Wr2p, *f_W)] if CH.TRACE is not None else None
        ctx.has_b = [b is not None for b in bs]
        ctx.set_materialize_grads(False)
        if train:
            ctx.n_acts = len(acts)
            ctx.save_for_backward(X4, Y4, h1, bits_h1, hy, bits_hy, *acts, *bits, *imgs[1])
        return Xc_enc, R

    @staticmethod
    def backward(ctx, gK, gR):
        n_x, n_flat = ctx.spec
        n_par = 2 * (1 + n_x + 2 + n_flat)
        if gK is None and gR is None:
            return (None,) * (3 + n_par)
        B, Cn, tiles, dx, dy, h, F = ctx.geom
        sv = list(ctx.saved_tensors)
        X4, Y4, h1, bits_h1, hy, bits_hy = sv[:6]
        acts, bits = sv[6:6 + ctx.n_acts], sv[6 + ctx.n_acts:6 + 2 * ctx.n_acts]
        bw = sv[6 + 2 * ctx.n_acts:]
        dev = X4.device
        bf16 = ctx.bf16
        pt = lambda: CH.pt_empty(B, Cn, F, dev)  # noqa: E731
        pb = (lambda: CH.pt16_empty(B, Cn, F, dev)) if bf16 else pt  # noqa: E731
        x_acts, x_bits = acts[:n_x - 1], bits[:n_x - 1]
        m_act, m_bits = acts[n_x - 1], bits[n_x - 1]          # the merge's output
        f_acts, f_bits = acts[n_x:], bits[n_x:]                 # flat hidden outputs (n_flat - 1 of them)
        prog = Program(B, tiles, per_task=False, width=F, bf16=bf16)
        wref = lambda j: ("shared", ctx.trace_ref[j].t()) if ctx.trace_ref is not None else None  # noqa: E731
        jobs, grads = [], [None] * n_par

        def wjob(pos, dZ, A, N, K, **kw):
            dW = torch.empty((N, K), dtype=torch.float32, device=dev)
            db = torch.empty((N,), dtype=torch.float32, device=dev) if ctx.has_b[pos] else None
            jobs.append(dict(dZ=dZ, A=A, N=N, K=K, dW=dW, db=db, **kw))
            grads[2 * pos], grads[2 * pos + 1] = dW, db
            return dW

        dz_m = None
        if gR is not None:
            gR = gR.contiguous()
            # flat MLP, last to first; the output layer's dZ is the incoming gradient itself
            f_in = [m_act, *f_acts]
            for i in range(n_flat - 1, -1, -1):
                o = dict(img=bw[n_x + 1 + i], w_ref=wref(n_x + 1 + i))
                if i == n_flat - 1:
                    o["in_pt"] = gR
                    dz = gR
                    if bf16:  # (the rounded incoming gradient: what the weight gradient multiplies and the bias gradient sums)
                        dz = pb()
                        o["store_in"] = dz
                else:
                    dz = pb()
                    o.update(mask_bits=f_bits[i], store_in=dz)
                prog.op(**o)
                wjob(1 + n_x + 2 + i, dz, f_in[i], F, F)
            # the merge relu(x1 + W_r2 hy + b): its dZ is also the gradient wrt x1 (fp32 in either mode); then back through the
            # h-wide first layer
            dz_m, dz_hy = pt(), pb()
            prog.op(mask_bits=m_bits, store_in=dz_m, store_in_f32=True, img=bw[n_x], w_ref=wref(n_x))
            prog.op(mask_bits=bits_hy, store_in=dz_hy)
            wjob(1 + n_x + 1, dz_m, hy, F, h, lda=F)
            dWr1 = wjob(1 + n_x, dz_hy, FN._pack(Y4), h, 4, ldz=F)
        # x-encoder: gradient of Xc_enc = what the attention sends back for the keys + the merge's x1 gradient
        if gK is not None or dz_m is not None:
            first = dict(in_pt=gK.contiguous(), pre_add=dz_m) if gK is not None else dict(in_pt=dz_m)
            x_in = [h1, *x_acts]
            for i in range(n_x - 1, -1, -1):
                dz = pb()
                o = dict(store_in=dz, img=bw[i], w_ref=wref(i))
                if i == n_x - 1:
                    o.update(first)
                else:
                    o["mask_bits"] = x_bits[i]
                prog.op(**o)
                wjob(1 + i, dz, x_in[i], F, F)
            dz1 = pb()
            prog.op(mask_bits=bits_h1, store_in=dz1)
            dW1p = wjob(0, dz1, FN._pack(X4), F, 4)
        prog.tag = "context side dgrad (XY-encoder, x-encoder)"
        prog.launch()
        CH.run_wgrad(jobs, B, Cn, dev, tag="context side weight gradients")
        if CH.TRACE is not None:
            CH.TRACE.append(("wgrad", jobs, B, Cn, bf16))
        if gR is not None and dy != 4:
            grads[2 * (1 + n_x)] = dWr1[:, :dy].contiguous()
        if (gK is not None or dz_m is not None) and dx != 4:
            grads[0] = dW1p[:, :dx].contiguous()
        return (None, None, None, *grads)


def context_side_usable(model, C: int) -> bool:
    """Does the fused context side cover this model and batch: stock MLP x-encoder (<= 4 inputs) and sum-merge MLP XY-encoder
    with 256-wide layers, no residual / dropout, a two-layer resizer (dy -> h -> 256, h a multiple of 16), whole tiles."""
    from .architectures import MLP, MergeFlatInputs

    if not _mode_ok() or C <= 0:
        return False
    WIDTH = _width_of(model)
    if not WIDTH:
        return False
    xe, xy = model.x_encoder, getattr(model, "xy_encoder", None)
    if not (isinstance(xe, MLP) and xe.input_size <= 4 and xe.hidden_size == WIDTH and xe.output_size == WIDTH
            and not xe.is_res and not (xe.dropout_p > 0 and xe.training)):
        return False
    if not (isinstance(xy, MergeFlatInputs) and xy.is_sum_merge and isinstance(xy.flat_module, MLP)):
        return False
    fm, rs = xy.flat_module, xy.resizer
    for m in (fm, rs):
        if m.is_res or (m.dropout_p > 0 and m.training):
            return False
    if not (_square([fm.to_hidden, *fm.linears, fm.out], WIDTH) and len(rs.linears) == 0 and rs.input_size <= 4
            and rs.output_size == WIDTH and rs.hidden_size % 16 == 0 and rs.hidden_size <= WIDTH):
        return False
    n_x, n_flat = len(xe.linears) + 1, len(fm.linears) + 2
    return n_flat + 2 + n_x + 1 <= L.NPF_X6_MAX_OPS


def context_side(model, X_cntxt: torch.Tensor, Y_cntxt: torch.Tensor):
    """(encoded context points, their per-point representations) as :class:`~chain.PTensor` s (``context_side_usable``)."""
    xe, xy = model.x_encoder, model.xy_encoder
    fm, rs = xy.flat_module, xy.resizer
    lins = [xe.to_hidden, *xe.linears, xe.out, rs.to_hidden, rs.out, fm.to_hidden, *fm.linears, fm.out]
    params = []
    for lin in lins:
        params += [lin.weight, lin.bias]
    Cn = X_cntxt.shape[1]
    if Cn % 32:  # whole tiles: the padding points are computed (finite values) and never read -- every consumer knows ``pts``
        pad = CH.pad32(Cn) - Cn
        X_cntxt = torch.nn.functional.pad(X_cntxt, (0, 0, 0, pad))
        Y_cntxt = torch.nn.functional.pad(Y_cntxt, (0, 0, 0, pad))
    Xc, R = _ContextSideFn.apply(X_cntxt, Y_cntxt, (len(xe.linears) + 1, len(fm.linears) + 2), *params)
    return CH.PTensor(Xc, Cn, model.r_dim), CH.PTensor(R, Cn, model.r_dim)
