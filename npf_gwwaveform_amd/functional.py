"""Autograd wrappers of the non-chain kernels: layout changes, mean aggregation over the
points of a task, and the Gaussian head.  Every function launches HIP kernels through the
C ABI (``_lib``); none has a CPU path."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib as L
from .chain import pad32, pt_empty, tiles_of


# ---- layout ---------------------------------------------------------------------------
def _pack(rows: torch.Tensor) -> torch.Tensor:
    n_tasks, pts, F = rows.shape
    out = pt_empty(n_tasks, pts, F, rows.device)
    L.check(L.load().npf_pack_pt(L.ptr(rows.contiguous()), n_tasks, pts, F, L.ptr(out), L.stream_ptr()), "npf_pack_pt")
    return out


def _unpack(pt: torch.Tensor, pts: int, F: int) -> torch.Tensor:
    n_tasks = pt.shape[0]
    out = torch.empty((n_tasks, pts, F), dtype=torch.float32, device=pt.device)
    L.check(L.load().npf_unpack_pt(L.ptr(pt.contiguous()), n_tasks, pts, F, L.ptr(out), L.stream_ptr()), "npf_unpack_pt")
    return out


class _PackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows):
        ctx.pts, ctx.F = rows.shape[1], rows.shape[2]
        return _pack(rows)

    @staticmethod
    def backward(ctx, g):
        return _unpack(g.contiguous(), ctx.pts, ctx.F)


class _UnpackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pt, pts, F):
        return _unpack(pt, pts, F)

    @staticmethod
    def backward(ctx, g):
        return _pack(g.contiguous()), None, None


def pack_pt(rows: torch.Tensor) -> torch.Tensor:
    """row-major [n_tasks, pts, F] -> PT32 (padding points / features are zero)."""
    return _PackFn.apply(rows)


def unpack_pt(pt: torch.Tensor, pts: int, F: int) -> torch.Tensor:
    """PT32 -> row-major [n_tasks, pts, F]."""
    return _UnpackFn.apply(pt, pts, F)


# ---- mean over the points of a task ---------------------------------------------------
def sum_points_pt(pt: torch.Tensor, pts: int, F: int) -> torch.Tensor:
    """[n_tasks, pad32(F)] sum over the valid points (no autograd)."""
    n_tasks = pt.shape[0]
    Fp = pad32(F)
    out = torch.empty((n_tasks, Fp), dtype=torch.float32, device=pt.device)
    L.check(L.load().npf_mean_agg_fwd(L.ptr(pt), n_tasks, pts, Fp, L.ptr(out), L.stream_ptr()), "npf_mean_agg_fwd")
    return out * float(pts)


class _MeanAggFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pt, pts, F):
        n_tasks, Fp = pt.shape[0], pad32(F)
        ctx.geo = (n_tasks, pts, Fp)
        out = torch.empty((n_tasks, Fp), dtype=torch.float32, device=pt.device)
        L.check(L.load().npf_mean_agg_fwd(L.ptr(pt.contiguous()), n_tasks, pts, Fp, L.ptr(out), L.stream_ptr()),
                "npf_mean_agg_fwd")
        return out

    @staticmethod
    def backward(ctx, g):
        n_tasks, pts, Fp = ctx.geo
        d = pt_empty(n_tasks, pts, Fp, g.device)
        L.check(L.load().npf_mean_agg_bwd(L.ptr(g.contiguous()), n_tasks, pts, Fp, L.ptr(d), 0, L.stream_ptr()),
                "npf_mean_agg_bwd")
        return d, None, None


def mean_agg(pt: torch.Tensor, pts: int, F: int) -> torch.Tensor:
    """torch.mean(R, dim=1) of a PT32 tensor -> row-major [n_tasks, pad32(F)]
    (npf/neuralproc/np.py:95, attnnp.py:181)."""
    return _MeanAggFn.apply(pt, pts, F)


# ---- Gaussian head --------------------------------------------------------------------
class _GaussHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, suff, Y, dy, homosk, want_dist):
        n_rows, pts, two_dy = suff.shape
        assert two_dy == 2 * dy
        suff = suff.contiguous()
        loc = scale = None
        if want_dist:
            loc = torch.empty((n_rows, pts, dy), dtype=torch.float32, device=suff.device)
            scale = torch.empty_like(loc)
        slp = None
        n_y = 0
        if Y is not None:
            Y = Y.contiguous()
            n_y = Y.shape[0]
            assert Y.shape[1:] == (pts, dy) and n_rows % n_y == 0
            slp = torch.empty((n_rows,), dtype=torch.float32, device=suff.device)
        elif not want_dist:
            raise ValueError("a loss-only head launch needs the targets")
        L.check(L.load().npf_gauss_head_fwd(L.ptr(suff), n_rows, pts, dy, int(homosk), L.ptr(Y), n_y, L.ptr(loc),
                                            L.ptr(scale), L.ptr(slp), L.stream_ptr()), "npf_gauss_head_fwd")
        ctx.save_for_backward(suff, loc, scale, Y)
        ctx.cfg = (dy, homosk)
        empty = suff.new_zeros((0,))
        outs = [loc if want_dist else empty, scale if want_dist else empty, slp if slp is not None else suff.new_zeros((n_rows,))]
        nd = ([] if want_dist else [outs[0], outs[1]]) + ([] if slp is not None else [outs[2]])
        if nd:
            ctx.mark_non_differentiable(*nd)
        return tuple(outs)

    @staticmethod
    def backward(ctx, d_loc, d_scale, d_slp):
        suff, loc, scale, Y = ctx.saved_tensors
        dy, homosk = ctx.cfg
        n_rows, pts, _ = suff.shape
        d_suff = torch.empty_like(suff)
        c = lambda t: t.contiguous() if t is not None else None  # noqa: E731
        if loc is None:
            d_loc = d_scale = None
        L.check(L.load().npf_gauss_head_bwd(L.ptr(suff), L.ptr(loc), L.ptr(scale), n_rows, pts, dy, int(homosk),
                                            L.ptr(Y), Y.shape[0] if Y is not None else 0, L.ptr(c(d_loc)),
                                            L.ptr(c(d_scale)), L.ptr(c(d_slp)) if Y is not None else None,
                                            L.ptr(d_suff), L.stream_ptr()), "npf_gauss_head_bwd")
        return d_suff, None, None, None, None


def gauss_head(suff: torch.Tensor, Y: Optional[torch.Tensor], dy: int, homoskedastic: bool, want_dist: bool = True):
    """(loc, scale, sum_log_prob) from the raw decoder output ``suff`` [rows, pts, 2*dy]
    (npf/neuralproc/base.py:350-365; losses.py:18-24).  ``sum_log_prob`` [rows] is the
    log-likelihood of ``Y`` [rows or B, pts, dy] summed over targets and y-dims.  ``want_dist=False``: a
    loss-only launch -- loc and scale come back empty and nothing of size [rows, pts, dy] is written."""
    return _GaussHeadFn.apply(suff, Y, dy, homoskedastic, want_dist)


# ---- Monte-Carlo objectives over the latent samples -------------------------------------
MC_MEAN, MC_LOGMEANEXP, MC_SUMO = 0, 1, 2


class _McObjectiveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_w, mode, inv_w, m):
        n_z, B = log_w.shape
        log_w = log_w.contiguous()
        out = torch.empty((B,), dtype=torch.float32, device=log_w.device)
        L.check(L.load().npf_mc_objective_fwd(L.ptr(log_w), n_z, B, mode, L.ptr(inv_w), m, L.ptr(out), L.stream_ptr()),
                "npf_mc_objective_fwd")
        ctx.save_for_backward(log_w, inv_w)
        ctx.cfg = (mode, m)
        return out

    @staticmethod
    def backward(ctx, d_out):
        log_w, inv_w = ctx.saved_tensors
        mode, m = ctx.cfg
        n_z, B = log_w.shape
        d = torch.empty_like(log_w)
        ws = torch.empty_like(log_w) if mode == MC_SUMO else None
        L.check(L.load().npf_mc_objective_bwd(L.ptr(log_w), n_z, B, mode, L.ptr(inv_w), m, L.ptr(d_out.contiguous()), L.ptr(d),
                                              L.ptr(ws), L.stream_ptr()), "npf_mc_objective_bwd")
        return d, None, None, None


def mc_objective(log_w: torch.Tensor, mode: int, inv_weights: Optional[torch.Tensor] = None, m: int = 0) -> torch.Tensor:
    """Per-task estimate [B] from the log weights ``log_w`` [n_z, B] of the latent samples
    (``npf_mc_objective_fwd``): their mean, log-mean-exp, or the SUMO estimate."""
    return _McObjectiveFn.apply(log_w, mode, inv_weights, m)


class _HeadsFn(torch.autograd.Function):
    """Heads as tasks and back on PT32 tensors (attention.py:505-527); the two directions are each
    other's adjoint."""

    @staticmethod
    def forward(ctx, x_pt, n_tasks, pts, F, n_heads, split):
        ctx.geom = (n_tasks, pts, F, n_heads, split)
        return _heads(x_pt, n_tasks, pts, F, n_heads, split)

    @staticmethod
    def backward(ctx, g):
        n_tasks, pts, F, n_heads, split = ctx.geom
        return _heads(g.contiguous(), n_tasks, pts, F, n_heads, not split), None, None, None, None, None


def _heads(x_pt, n_tasks, pts, F, n_heads, split):
    from .chain import pt_empty

    lib = L.load()
    x_pt = x_pt.contiguous()
    if split:
        out = pt_empty(n_heads * n_tasks, pts, F // n_heads, x_pt.device)
        L.check(lib.npf_split_heads(L.ptr(x_pt), n_tasks, pts, F, n_heads, L.ptr(out), L.stream_ptr()), "npf_split_heads")
    else:
        out = pt_empty(n_tasks, pts, F, x_pt.device)
        L.check(lib.npf_merge_heads(L.ptr(x_pt), n_tasks, pts, F, n_heads, L.ptr(out), L.stream_ptr()), "npf_merge_heads")
    return out


def split_heads(x_pt: torch.Tensor, n_tasks: int, pts: int, F: int, n_heads: int) -> torch.Tensor:
    """PT32 [n_tasks, pts, F] -> PT32 [n_heads * n_tasks, pts, F / n_heads] (task index h * n_tasks + b)."""
    return _HeadsFn.apply(x_pt, n_tasks, pts, F, n_heads, True)


def merge_heads(x_pt: torch.Tensor, n_tasks: int, pts: int, F: int, n_heads: int) -> torch.Tensor:
    """Inverse of :func:`split_heads`."""
    return _HeadsFn.apply(x_pt, n_tasks, pts, F, n_heads, False)


MHA_MAX_KEYS = {16: 256, 32: 128}  # head size -> keys the fused multihead attention kernel takes (csrc/mha_kernel.hip)
MHA_ENABLED = os.environ.get("NPF_NO_MHA", "0") != "1"  # NPF_NO_MHA=1: heads as extra tasks on the chain kernel (round 2)


class _MhaFn(torch.autograd.Function):
    """out[b, q, D h + :] = softmax_k(Q_h K_h^T / sqrt(D)) V_h on the PT32 tensors of the K / Q / V projections (``npf_mha_fwd`` /
    ``npf_mha_bwd``; MultiheadAttender.forward, npf/architectures/attention.py:505-527 with DotAttender :204-220 per head)."""

    @staticmethod
    def forward(ctx, q_pt, k_pt, v_pt, n_tasks, n_keys, n_queries, n_heads, head):
        from . import chain as CH

        F = n_heads * head
        q_pt, k_pt, v_pt = q_pt.contiguous(), k_pt.contiguous(), v_pt.contiguous()
        train = any(ctx.needs_input_grad[:3])
        # (zeros where the kernel leaves something unwritten: the features of a padded tile beyond F, the points beyond n_queries)
        whole = F % 32 == 0 and n_queries % 32 == 0
        out = (torch.empty if whole else torch.zeros)(CH.pt_shape(n_tasks, n_queries, F), dtype=torch.float32, device=q_pt.device)
        lse = torch.empty((n_tasks, n_heads, n_queries), dtype=torch.float32, device=q_pt.device) if train else None
        if CH.PROFILE is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        L.check(L.load().npf_mha_fwd(L.ptr(q_pt), L.ptr(k_pt), L.ptr(v_pt), n_tasks, n_heads, n_keys, n_queries, F, L.ptr(out),
                                     L.ptr(lse) if lse is not None else None, L.stream_ptr()), "npf_mha_fwd")
        if CH.PROFILE is not None:
            ev1.record()
            CH.PROFILE.append(("mha_fwd_kernel", 4 * n_tasks * n_queries * n_keys * F, ev0, ev1,
                               4 * F * n_tasks * (2 * n_queries + 2 * n_keys), "multihead attention"))
        ctx.geom = (n_tasks, n_keys, n_queries, n_heads, F)
        if train:
            ctx.save_for_backward(q_pt, k_pt, v_pt, out, lse)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import chain as CH

        n_tasks, n_keys, n_queries, n_heads, F = ctx.geom
        q_pt, k_pt, v_pt, out, lse = ctx.saved_tensors
        g = g.contiguous()
        # (zeros where the kernel leaves something unwritten, as in the forward pass)
        mk = lambda t, n: (torch.empty_like if (F % 32 == 0 and n % 32 == 0) else torch.zeros_like)(t)  # noqa: E731
        dq, dk, dv = mk(q_pt, n_queries), mk(k_pt, n_keys), mk(v_pt, n_keys)
        if CH.PROFILE is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        L.check(L.load().npf_mha_bwd(L.ptr(q_pt), L.ptr(k_pt), L.ptr(v_pt), L.ptr(out), L.ptr(g), L.ptr(lse), n_tasks, n_heads,
                                     n_keys, n_queries, F, L.ptr(dq), L.ptr(dk), L.ptr(dv), L.stream_ptr()), "npf_mha_bwd")
        if CH.PROFILE is not None:
            ev1.record()
            CH.PROFILE.append(("mha_bwd_kernel", 14 * n_tasks * n_queries * n_keys * F, ev0, ev1,
                               4 * F * n_tasks * (4 * n_queries + 4 * n_keys), "multihead attention backward"))
        return dq, dk, dv, None, None, None, None, None


def mha_usable(kq_head: int, v_head: int, n_keys: int) -> bool:
    """Does the fused multihead attention kernel take this: fp32 mode, 16-feature heads and at most 256 keys, or 32-feature heads
    and at most 128."""
    from . import chain as CH

    return (MHA_ENABLED and CH.COMPUTE_DTYPE == "fp32" and kq_head == v_head and kq_head in MHA_MAX_KEYS
            and 0 < n_keys <= MHA_MAX_KEYS[kq_head])


def mha(q_pt: torch.Tensor, k_pt: torch.Tensor, v_pt: torch.Tensor, n_tasks: int, n_keys: int, n_queries: int,
        n_heads: int, head: int = 16) -> torch.Tensor:
    """PT32 [n_tasks, n_queries, head * n_heads]: per-head scaled-dot attention of the projected queries over the projected keys /
    values (``mha_usable``), no split / merge of heads in memory."""
    return _MhaFn.apply(q_pt, k_pt, v_pt, n_tasks, n_keys, n_queries, n_heads, head)


class _AddLayerNormFn(torch.autograd.Function):
    """LayerNorm(a + b) over the features on PT32 tensors (``npf_add_layernorm_fwd`` / ``_bwd``; the first LayerNorm of
    TransformerAttender.forward, npf/architectures/attention.py:566-575)."""

    @staticmethod
    def forward(ctx, a_pt, b_pt, gamma, beta, eps, n_tasks, pts, F):
        from . import chain as CH

        a_pt, b_pt = a_pt.contiguous(), b_pt.contiguous()
        tiles = tiles_of(pts)
        train = any(ctx.needs_input_grad[:4])
        y = (torch.zeros if pad32(F) != F else torch.empty)(CH.pt_shape(n_tasks, pts, F), dtype=torch.float32, device=a_pt.device)
        stats = torch.empty((n_tasks * tiles * 32, 2), dtype=torch.float32, device=a_pt.device) if train else None
        g, bt = gamma.detach().contiguous(), beta.detach().contiguous()
        if CH.PROFILE is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        L.check(L.load().npf_add_layernorm_fwd(L.ptr(a_pt), L.ptr(b_pt), L.ptr(g), L.ptr(bt), float(eps), n_tasks, pts, F, L.ptr(y),
                                               L.ptr(stats) if stats is not None else None, L.stream_ptr()), "npf_add_layernorm_fwd")
        if CH.PROFILE is not None:
            ev1.record()
            CH.PROFILE.append(("add_layernorm_fwd_kernel", 0, ev0, ev1, 12 * n_tasks * tiles * 32 * F, "LayerNorm(a + b)"))
        ctx.geom = (n_tasks, pts, F, tiles)
        if train:
            ctx.save_for_backward(a_pt, b_pt, g, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import chain as CH

        n_tasks, pts, F, tiles = ctx.geom
        a_pt, b_pt, g, stats = ctx.saved_tensors
        dy = dy.contiguous()
        dx = (torch.zeros if pad32(F) != F else torch.empty)(CH.pt_shape(n_tasks, pts, F), dtype=torch.float32, device=dy.device)
        partials = torch.empty((n_tasks * tiles, 2, F), dtype=torch.float32, device=dy.device)
        if CH.PROFILE is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        L.check(L.load().npf_add_layernorm_bwd(L.ptr(a_pt), L.ptr(b_pt), L.ptr(g), L.ptr(stats), L.ptr(dy), n_tasks, pts, F, L.ptr(dx),
                                               L.ptr(partials), L.stream_ptr()), "npf_add_layernorm_bwd")
        sums = partials.sum(0)
        if CH.PROFILE is not None:
            ev1.record()
            CH.PROFILE.append(("add_layernorm_bwd_kernel", 0, ev0, ev1, 16 * n_tasks * tiles * 32 * F, "LayerNorm(a + b) backward"))
        return dx, dx, sums[0], sums[1], None, None, None, None


def add_layernorm_usable(F: int) -> bool:
    from . import chain as CH

    return MHA_ENABLED and CH.COMPUTE_DTYPE == "fp32" and F % 4 == 0 and F <= 256


def add_layernorm(a_pt: torch.Tensor, b_pt: torch.Tensor, ln: torch.nn.LayerNorm, n_tasks: int, pts: int) -> torch.Tensor:
    """PT32 LayerNorm(a + b) with the module's gamma / beta / eps (``add_layernorm_usable``)."""
    F = ln.normalized_shape[0]
    return _AddLayerNormFn.apply(a_pt, b_pt, ln.weight, ln.bias, ln.eps, n_tasks, pts, F)
