"""Stacks of 256 -> 256 Linear layers with their fp32 products on the bf16 matrix pipe (``npf_mlp_x6_run``,
``csrc/mlp_x6_kernel.hip``): the hidden layers of the reference's flat MLPs (npf/architectures/mlp.py:95-109) as one
launch forward and one launch of their dgrad, weight gradients through the usual ``run_wgrad`` jobs.

The arithmetic is fp32: every operand is split exactly into three bf16 terms and six of the nine cross products are
accumulated in fp32.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import _lib as L
from . import chain as CH

# On by default in the fp32 compute mode (NPF_NO_MLP_X6=1: the layers stay inside the fp32 chains).  Config 2: the decoder's
# resizer + merge + hidden layers (6 of its 9 layer-equivalents) and the XY-encoder's flat module run here at 175 - 217 TF/s
# fp32-equivalent where the fp32 chain kernel does 105 - 116; step 9.23 -> 8.7 - 8.9 ms.
ENABLED = os.environ.get("NPF_NO_MLP_X6", "0") != "1"
WIDTH = 256


def _three_term_images(Ws: Sequence[torch.Tensor], kinds: Sequence[int], width: int = 256) -> List[List[torch.Tensor]]:
    """Per weight matrix [F, F] (F = ``width``) and per kind (1: of W, 2: of W^T) its three-term image [3, F, F] bf16 (k-permuted like
    ``npf_cast_bf16_weights``): W0 = bf16(W), W1 = bf16(W - W0), W2 = bf16(W - W0 - W1), split and permuted by
    ``npf_prepare_weights`` (kinds 1 / 2 + 16 (s + 1)), one launch per 32 images, written in place."""
    n, dev = len(Ws), Ws[0].device
    buf = torch.empty((len(kinds), n, 3, width, width), dtype=torch.bfloat16, device=dev)  # (the images land in place)
    specs = [(Ws[i].detach(), kind | ((s + 1) << 4)) for kind in kinds for i in range(n) for s in range(3)]
    CH.prepare_weights(specs, dsts=[buf[k, i, s] for k in range(len(kinds)) for i in range(n) for s in range(3)])
    return [[buf[k, i] for i in range(n)] for k in range(len(kinds))]


def _launch(layers: Sequence[dict], x: Optional[torch.Tensor], y: Optional[torch.Tensor], n_tasks: int, tiles: int,
            rows: Optional[torch.Tensor] = None, rows_w: Optional[torch.Tensor] = None,
            out: Optional[tuple] = None) -> None:
    """``rows`` / ``rows_w`` in place of ``x``: the stack's input is rows [points, 4] through rows_w [4, 256], computed in the
    kernel's prologue; ``out`` = (W [4, 256], b [4] or None, rows [points, 4]): a 256 -> 4 layer on the registers the last
    layer leaves (``npf_mlp_x6_run_rows``)."""
    for i0 in range(0, len(layers), L.NPF_X6_MAX_LAYERS):
        chunk = layers[i0:i0 + L.NPF_X6_MAX_LAYERS]
        last = i0 + L.NPF_X6_MAX_LAYERS >= len(layers)
        arr = (L.NpfX6Layer * len(chunk))()
        for j, ly in enumerate(chunk):
            arr[j].w_img = ly["img"].data_ptr()
            arr[j].bias = L.ptr(ly.get("bias")) if ly.get("bias") is not None else None
            arr[j].mask = L.ptr(ly.get("mask")) if ly.get("mask") is not None else None
            arr[j].store_in = L.ptr(ly.get("store_in")) if ly.get("store_in") is not None else None
            arr[j].store_out = L.ptr(ly.get("store_out")) if ly.get("store_out") is not None else None
            arr[j].addend = L.ptr(ly.get("addend")) if ly.get("addend") is not None else None
            arr[j].store_bits = ly["store_bits"].data_ptr() if ly.get("store_bits") is not None else None
            arr[j].mask_bits = ly["mask_bits"].data_ptr() if ly.get("mask_bits") is not None else None
            arr[j].relu = int(bool(ly.get("relu", False)))
        tail = out if (last and out is not None) else None
        out_t = y if last else torch.empty_like(y)
        if CH.PROFILE is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if x is None or tail is not None:
            L.check(L.load().npf_mlp_x6_run_rows(
                arr, len(chunk), L.ptr(x) if x is not None else None, L.ptr(rows) if x is None else None,
                L.ptr(rows_w) if x is None else None, L.ptr(out_t), L.ptr(tail[0]) if tail else None,
                L.ptr(tail[1]) if (tail and tail[1] is not None) else None, L.ptr(tail[2]) if tail else None,
                n_tasks, tiles, L.stream_ptr()), "npf_mlp_x6_run_rows")
        else:
            L.check(L.load().npf_mlp_x6_run(arr, len(chunk), L.ptr(x), L.ptr(out_t), n_tasks, tiles, L.stream_ptr()),
                    "npf_mlp_x6_run")
        if CH.PROFILE is not None:
            ev1.record()
            padded = n_tasks * tiles * 32
            nbytes = padded * 32 * sum((ly.get("mask_bits") is not None) + (ly.get("store_bits") is not None) for ly in chunk)
            nbytes += padded * 1024 * ((2 if x is not None else 1) + sum((ly.get("mask") is not None) + (ly.get("store_in") is not None)
                                              + (ly.get("store_out") is not None) + (ly.get("addend") is not None)
                                              for ly in chunk)) + len(chunk) * 3 * 2 * WIDTH * WIDTH
            CH.PROFILE.append(("mlp_x6_kernel", 2 * WIDTH * WIDTH * len(chunk) * n_tasks * tiles * 32, ev0, ev1, nbytes))
        x = out_t


class _MlpX6Fn(torch.autograd.Function):
    """y = stack(x): x, y PT32 [n_tasks, tiles, 64, 32, 4]; ``addend`` (PT32 or None) enters layer ``add_at`` before its
    ReLU; params = W_0, b_0, W_1, b_1, ... (b may be None).  ``tail``: the last (W, b) pair is a 256 -> 4 layer behind the
    stack (the decoder's output layer) and the result is its row-major [n_tasks, pts, 4] output: forward on the registers the
    last layer leaves, backward in the prologue of the stack's dgrad launch (``npf_mlp_x6_run_rows``)."""

    @staticmethod
    def forward(ctx, x, pts, relus, addend, add_at, tail, *params):
        n_tasks, tiles = x.shape[0], x.shape[1]
        Ws, bs = list(params[0::2]), list(params[1::2])
        W_out = b_out = None
        if tail:
            W_out, b_out = Ws.pop(), bs.pop()
        train = any(ctx.needs_input_grad)  # (grad mode is always off inside Function.forward)
        x = x.detach().contiguous()
        add = addend.detach().contiguous() if addend is not None else None
        both = _three_term_images(Ws, (1, 2) if train else (1,))  # (the images of W^T for the dgrad in the same batch)
        imgs = both[0]
        y = torch.empty_like(x)
        outs, bits = [], []
        layers = []
        for i, (img, b, r) in enumerate(zip(imgs, bs, relus)):
            ly = dict(img=img, bias=b.detach() if b is not None else None, relu=r, addend=add if i == add_at else None)
            if train and i + 1 < len(imgs):  # the layer's output = the next layer's input (the weight gradient's operand)
                ly["store_out"] = torch.empty_like(x)
                outs.append(ly["store_out"])
            if train and r:  # where it is positive, as bits: what the dgrad needs of it (8 bytes per lane instead of 256)
                ly["store_bits"] = torch.empty((n_tasks, tiles, 2, 64), dtype=torch.int64, device=x.device)
                bits.append(ly["store_bits"])
            layers.append(ly)
        rows = None
        if tail:  # the output layer on the registers the last layer leaves: [n_tasks, pts, 4] (pts a multiple of 32)
            rows = torch.empty((n_tasks, pts, 4), dtype=torch.float32, device=x.device)
            _launch(layers, x, y, n_tasks, tiles,
                    out=(W_out.detach().contiguous(), b_out.detach() if b_out is not None else None, rows))
        else:
            _launch(layers, x, y, n_tasks, tiles)
        ctx.pts, ctx.relus, ctx.geom, ctx.add_at = pts, tuple(relus), (n_tasks, tiles), (add_at if addend is not None else -1)
        ctx.n = len(Ws)
        # acts[i] = input of layer i, acts[i + 1] = its output; through save_for_backward: the output y among them would
        # otherwise close a reference cycle (y -> grad_fn -> ctx -> y) that only the cyclic collector frees -- GBs per step
        ctx.has_b = [b is not None for b in bs]
        ctx.set_materialize_grads(False)
        ctx.tail = bool(tail)
        if tail:
            ctx.tail_b = b_out is not None
            ctx.save_for_backward(x, *outs, y, *(both[1] if train else []), *bits, W_out.detach())
            return rows
        ctx.save_for_backward(x, *outs, y, *(both[1] if train else []), *bits)
        return y

    @staticmethod
    def backward(ctx, g):
        n = ctx.n
        if g is None:
            return (None,) * (6 + 2 * n + 2 * ctx.tail)
        saved = list(ctx.saved_tensors)
        W_out = saved.pop() if ctx.tail else None
        acts, imgs_t, bits = saved[:n + 1], saved[n + 1:2 * n + 1], saved[2 * n + 1:]
        bit_of = dict(zip([i for i in range(n) if ctx.relus[i]], bits))
        n_tasks, tiles = ctx.geom
        g = g.contiguous()
        dzs = [torch.empty_like(acts[0]) for _ in range(n)]
        layers = []
        for i in range(n - 1, -1, -1):  # dZ_i = g_i masked by the layer's own output; g_{i-1} = W_i^T dZ_i
            layers.append(dict(img=imgs_t[i], mask_bits=bit_of.get(i), store_in=dzs[i]))
        dx = torch.empty_like(acts[0])
        if ctx.tail:  # g = dOut rows [n_tasks, pts, 4] (pts a multiple of 32): W_out^T dOut is formed inside the launch
            _launch(layers, None, dx, n_tasks, tiles, rows=g, rows_w=W_out)
        else:
            _launch(layers, g, dx, n_tasks, tiles)
        jobs, grads = [], []
        for i in range(n):
            dW = torch.empty((WIDTH, WIDTH), dtype=torch.float32, device=g.device)
            db = torch.empty((WIDTH,), dtype=torch.float32, device=g.device) if ctx.has_b[i] else None
            jobs.append(dict(dZ=dzs[i], A=acts[i], N=WIDTH, K=WIDTH, dW=dW, db=db))
            grads += [dW, db]
        if ctx.tail:
            # the output layer's weight gradient: its dZ = dOut as a PT32 tensor (4 of a tile's 32 feature slots; the rows of
            # a tile are exactly the tile's first feature group)
            dz_out = torch.zeros(CH.pt_shape(n_tasks, ctx.pts, 4), dtype=torch.float32, device=g.device)
            dz_out[:, :, 0] = g.view(n_tasks, tiles, 32, 4)
            dW = torch.empty((4, WIDTH), dtype=torch.float32, device=g.device)
            db = torch.empty((4,), dtype=torch.float32, device=g.device) if ctx.tail_b else None
            jobs.append(dict(dZ=dz_out, A=acts[n], N=4, K=WIDTH, dW=dW, db=db))
            grads += [dW, db]
        CH.run_wgrad(jobs, n_tasks, ctx.pts, g.device)
        # the addend's gradient is the dZ of its layer (it enters in front of the ReLU, with unit weight)
        d_add = dzs[ctx.add_at] if ctx.add_at >= 0 else None
        return (dx, None, None, d_add, None, None, *grads)


def usable(linears: Sequence[torch.nn.Linear]) -> bool:
    """Can this run of Linear layers go through the split kernel: fp32 compute mode, every layer 256 -> 256."""
    return (ENABLED and CH.COMPUTE_DTYPE == "fp32" and len(linears) > 0
            and all(l.in_features == WIDTH and l.out_features == WIDTH for l in linears))


# The decoder's 256 -> 4 output layer rides on the stack (NPF_NO_X6_TAIL=1: a chain of its own, forward and backward).
TAIL = os.environ.get("NPF_NO_X6_TAIL", "0") != "1"


def tail_usable(out: torch.nn.Linear, pts: int) -> bool:
    return TAIL and out.in_features == WIDTH and out.out_features == 4 and pts % 32 == 0


def run_stack(x_pt: torch.Tensor, pts: int, linears: Sequence[torch.nn.Linear], relus: Sequence[bool],
              addend: Optional[torch.Tensor] = None, add_at: int = 0, tail: Optional[torch.nn.Linear] = None) -> torch.Tensor:
    """PT32 [n_tasks, tiles, 64, 32, 4] -> the same shape through ``linears`` (256 -> 256 each), ReLU behind layer i when
    ``relus[i]``; ``addend`` (PT32, same shape) is added in front of the ReLU of layer ``add_at``.  With ``tail`` (a
    256 -> 4 Linear, ``tail_usable``): the row-major [n_tasks, pts, 4] output of that layer behind the stack."""
    params = []
    for lin in list(linears) + ([tail] if tail is not None else []):
        params += [lin.weight, lin.bias]
    return _MlpX6Fn.apply(x_pt, pts, tuple(bool(r) for r in relus), addend, int(add_at), tail is not None, *params)
