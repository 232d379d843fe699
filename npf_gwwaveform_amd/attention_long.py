"""Scaled-dot cross attention over more context points than one register-resident row holds.

The chain kernel keeps a point's whole score row in registers, which caps the fused attention of
``Chain.attn_scores / softmax / attn_values`` at 256 keys.  The reference has no such limit
(``DotAttender``, npf/architectures/attention.py:129-164,204-220: ``softmax(Q K^T / sqrt(d)) V``
over any number of keys), so longer contexts run here in blocks of <= 256 keys:

  pass A  per block j: s_j = q K_j^T, row statistics (m_j, l_j) = (max s_j, sum exp(scale (s_j - m_j)))
  combine m = max_j m_j,  l = sum_j l_j exp(scale (m_j - m))          (two [B, T] tensors)
  pass B  per block j: P_j = exp(scale (s_j - m)) / l,  O += P_j V_j   (s_j recomputed; P_j kept for backward)

and the backward pass is the usual blocked form with D = rowsum(dO * O):
  dP_j = dO V_j^T,  dS_j = scale P_j (dP_j - D),  dQ += dS_j K_j,  dK_j = dS_j^T Q,  dV_j = P_j^T dO.

Every contraction is a LINEAR op of the chain kernel or a per-task job of the wgrad kernel; only
the statistics merge is elementwise torch glue on [B, T] tensors.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L
from .chain import Program, pad32, pt_empty, run_wgrad, tiles_of

BLOCK = 256  # keys per block = the chain kernel's widest register-resident row in training
_OPS_PER_PROGRAM = L.NPF_MAX_OPS - 4


def _blocks(n_keys: int):
    return [(c0, min(BLOCK, n_keys - c0)) for c0 in range(0, n_keys, BLOCK)]


class _Launcher:
    """Packs ops into as few chain programs as NPF_MAX_OPS allows (same geometry for all)."""

    def __init__(self, n_tasks: int, pts: int):
        self.n_tasks, self.pts = n_tasks, pts
        self.prog: Optional[Program] = None

    def room(self, n_ops: int) -> Program:
        if self.prog is not None and len(self.prog.ops) + n_ops > _OPS_PER_PROGRAM:
            self.flush()
        if self.prog is None:
            self.prog = Program(self.n_tasks, self.pts, True)
        return self.prog

    def flush(self) -> None:
        if self.prog is not None:
            self.prog.launch()
            self.prog = None


def _score_op(prog: Program, k_pt: torch.Tensor, c0: int, cj: int, r: int) -> None:
    # s[c] = sum_d K[c0 + c][d] q[d]: the block's keys are the layer's weights (points = rows)
    prog.linear(k_pt[:, c0 // 32:], r, cj, mode=L.W_PT_ROWS, w_tiles=k_pt.shape[1])


def _contract_keys_op(prog: Program, x_pt: torch.Tensor, x_tr: Optional[torch.Tensor], c0: int, cj: int, r: int) -> None:
    # out[f] = sum_c X[c0 + c][f] cur[c]: contraction over the block's points
    if x_tr is not None:
        ld = x_tr.shape[2]
        prog.linear(x_tr[:, :, c0:], cj, r, mode=L.W_ROWMAJOR, ldw=ld, w_task_stride=r * ld)
    else:
        prog.linear(x_pt[:, c0 // 32:], cj, r, mode=L.W_PT_COLS, w_tiles=x_pt.shape[1])


class _LongScaledDot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_pt, k_pt, v_pt, n_keys, n_queries, r, scale, k_tr, v_tr, d):
        B, dev = q_pt.shape[0], q_pt.device
        T = n_queries
        blocks = _blocks(n_keys)
        train = any(ctx.needs_input_grad[:3])
        q_pt, k_pt, v_pt = q_pt.contiguous(), k_pt.contiguous(), v_pt.contiguous()
        # pass A: block statistics
        stats = torch.empty((len(blocks), B, T, 2), dtype=torch.float32, device=dev)
        run = _Launcher(B, T)
        for j, (c0, cj) in enumerate(blocks):
            prog = run.room(3)
            prog.load_pt(q_pt, d)
            _score_op(prog, k_pt, c0, cj, d)
            prog.softmax(cj, scale, mode=1, stats=stats[j])
        run.flush()
        m = stats[..., 0].amax(dim=0)                                            # [B, T]
        lsum = (stats[..., 1] * torch.exp((stats[..., 0] - m) * scale)).sum(dim=0)
        row = torch.stack([m, lsum], dim=-1).contiguous()                        # [B, T, 2]
        # pass B: normalised probabilities and the weighted sum of the values
        out = pt_empty(B, T, r, dev)
        probs = []
        for j, (c0, cj) in enumerate(blocks):
            prog = run.room(7)
            prog.load_pt(q_pt, d)
            _score_op(prog, k_pt, c0, cj, d)
            prog.softmax(cj, scale, mode=2, stats=row)
            if train:
                p_j = pt_empty(B, T, cj, dev)
                prog.store_pt(p_j, cj)
                probs.append(p_j)
            _contract_keys_op(prog, v_pt, v_tr, c0, cj, r)
            if j > 0:
                prog.add_pt(out, r)
            prog.store_pt(out, r)
            run.flush()  # one launch per block: the running sum is re-read by the next block
        ctx.geom = (B, T, n_keys, r, scale, d)
        ctx.tr = (k_tr, v_tr)
        ctx.save_for_backward(q_pt, k_pt, v_pt, out, *probs)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, d_out):
        if d_out is None:
            return (None,) * 10
        B, T, n_keys, r, scale, d = ctx.geom
        k_tr, v_tr = ctx.tr
        q_pt, k_pt, v_pt, out, *probs = ctx.saved_tensors
        dev = q_pt.device
        d_out = d_out.contiguous()
        need_q, need_k, need_v = ctx.needs_input_grad[:3]
        blocks = _blocks(n_keys)
        dq = pt_empty(B, T, d, dev) if need_q else None
        run = _Launcher(B, T)
        jobs, dk_parts, dv_parts = [], [], []
        for j, (c0, cj) in enumerate(blocks):
            prog = run.room(8)
            prog.load_pt(d_out, r)
            prog.rowdot_pt(out, r)                                               # D = rowsum(dO * O)
            prog.linear(v_pt[:, c0 // 32:], r, cj, mode=L.W_PT_ROWS, w_tiles=v_pt.shape[1])   # dP_j = dO V_j^T
            prog.softmax_bwd(probs[j], cj, scale)                                # dS_j = scale P_j (dP_j - D)
            ds_j = pt_empty(B, T, cj, dev)
            prog.store_pt(ds_j, cj)
            if need_q:
                _contract_keys_op(prog, k_pt, k_tr, c0, cj, d)                   # dQ_j = dS_j K_j
                if j > 0:
                    prog.add_pt(dq, d)
                prog.store_pt(dq, d)
            run.flush()  # (as in the forward pass: dq is re-read by the next block's launch)
            if need_k:
                dk_j = pt_empty(B, cj, d, dev)
                jobs.append(dict(dZ=ds_j, A=q_pt, N=cj, K=d, dW=dk_j, per_task=True))
                dk_parts.append(dk_j)
            if need_v:
                dv_j = pt_empty(B, cj, r, dev)
                jobs.append(dict(dZ=probs[j], A=d_out, N=cj, K=r, dW=dv_j, per_task=True))
                dv_parts.append(dv_j)
        run_wgrad(jobs, B, T, dev)

        def assemble(parts, like):
            if not parts:
                return None
            g = torch.cat(parts, dim=1)
            assert g.shape == like.shape, (g.shape, like.shape)
            return g

        return dq, assemble(dk_parts, k_pt), assemble(dv_parts, v_pt), None, None, None, None, None, None, None


def long_scaledot_attention(q_pt: torch.Tensor, k_pt: torch.Tensor, v_pt: torch.Tensor, n_keys: int, n_queries: int,
                            r: int, scale: float, k_tr: Optional[torch.Tensor] = None,
                            v_tr: Optional[torch.Tensor] = None, d: Optional[int] = None) -> torch.Tensor:
    """``softmax(scale * q K^T) V`` on PT32 tensors (queries [B, T, d], keys [B, C, d], values [B, C, r]) for any
    number of keys; returns the PT32 context vectors [B, T, r].  ``d``: width of keys and queries (default r);
    ``k_tr`` / ``v_tr``: optional feature-major copies of keys / values (``Chain.store_tr``)."""
    d = r if d is None else d
    if max(r, d) > L.NPF_MAX_FUSED_ROW:
        raise NotImplementedError(f"attention over {max(r, d)}-wide keys / values: the HIP path keeps at most "
                                  f"{L.NPF_MAX_FUSED_ROW} features per point in training")
    if k_pt.shape[1] != tiles_of(n_keys) or pad32(d) // 4 != k_pt.shape[2]:
        raise ValueError("keys tensor does not match (n_keys, d)")
    if v_pt.shape[1] != tiles_of(n_keys) or pad32(r) // 4 != v_pt.shape[2]:
        raise ValueError("values tensor does not match (n_keys, r)")
    return _LongScaledDot.apply(q_pt, k_pt, v_pt, n_keys, n_queries, r, float(scale), k_tr, v_tr, d)
