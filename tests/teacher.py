"""Teacher-forced check of the HIP chain launches (test infrastructure; used by tests/test_hip_bf16.py).

``npf_gwwaveform_amd.chain.TRACE`` makes every chain execution leave a record of the tensors it read and wrote.  This module
walks such a record step by step and checks each step against the kernel's OWN stored inputs:

* forward: every LINEAR (npf/architectures/mlp.py:95-109) / attention contraction (attention.py:129-164,204-220) is recomputed
  in float64 from the input the launch stored for the weight gradient (the PT16 / PT32 tensor itself) and the weights rounded
  where the kernel rounds them (``oracle._r16``: what ``_LinearBf16`` / ``_ScaledotBf16`` emulate), and compared with what the
  launch stored next -- the next layer's stored input, the softmax probabilities, the ReLU bits, the fp32 outputs;
* backward: every stored dZ / dO / dS against the mask and W^T applied to the previous stored gradient, every weight / key /
  value gradient of the wgrad launch against dZ_stored^T A_stored, bias gradients against the sum of the stored dZ.

Because every step starts from what the HIP path itself stored, a bf16 rounding that falls the other way than in the CPU
emulation (the "flip" of DESIGN.md section 4) cannot propagate into the next check: the tolerances are fp32-accumulation sized
(2e-6 of max|ref| per step, 1e-5 for sums over all points), with no exception list.  A value stored as bfloat16 is accepted
when it is a correct rounding of SOME value within that tolerance of the reference (|stored - ref| <= tol + half a bf16 ulp);
elements whose stored value differs from the rounding of the float64 reference are counted and listed as flips -- these are
the elements an end-to-end comparison against the emulation can disagree on.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch

TOL_STEP = 2e-6   # one layer / one softmax / one dgrad step: fp32 accumulation over <= 256 (512) terms
TOL_SUM = 1e-5    # sums over every point of the batch (weight, key / value and bias gradients)


def r16(x: torch.Tensor) -> torch.Tensor:
    """Round fp32-representable float64 values to bfloat16 (nearest even: v_cvt_pk_bf16_f32) and back to float64."""
    return x.float().to(torch.bfloat16).double()


def unpack32(t: torch.Tensor, pts: int, F: int) -> torch.Tensor:
    """PT32 [n, tiles, F/4, 32, 4] -> [n, pts, F] float64 (include/npf_hip.h: feature 4 f4 + e of point 32 tile + p)."""
    n, tiles, f4 = t.shape[:3]
    return t.detach().cpu().double().permute(0, 1, 3, 2, 4).reshape(n, tiles * 32, f4 * 4)[:, :pts, :F]


def unpack16(t: torch.Tensor, pts: int, F: int) -> torch.Tensor:
    """PT16 [n, tiles, F/8, 32, 8] bf16 -> [n, pts, F] float64 (chain.pt16_shape: row 4 s + g holds features
    {32 s + 4 g + i} and {32 s + 16 + 4 g + i})."""
    n, tiles, rows = t.shape[:3]
    Fp = rows * 8
    v = t.detach().cpu().double().view(n, tiles, Fp // 32, 4, 32, 2, 4).permute(0, 1, 4, 2, 5, 3, 6)
    return v.reshape(n, tiles * 32, Fp)[:, :pts, :F]


def unpack_any(t: torch.Tensor, pts: int, F: int) -> torch.Tensor:
    return unpack16(t, pts, F) if t.dtype == torch.bfloat16 else unpack32(t, pts, F)


def unpack_bits(m: torch.Tensor, pts: int, F: int) -> torch.Tensor:
    """PTM [n, tiles, words, 4 lane groups, 32 points] int32 -> bool [n, pts, F]: word q of lane group g holds feature
    128 q + 16 bb + 4 g + e at bit 31 - (4 bb + e) (csrc/chain_kernel.hip, "ReLU masks as bits")."""
    n, tiles, words = m.shape[:3]
    w = m.detach().cpu().to(torch.int64) & 0xFFFFFFFF
    out = torch.zeros(n, tiles, 32, words * 128, dtype=torch.bool)
    for q in range(words):
        for bb in range(8):
            for e in range(4):
                bit = (w[:, :, q] >> (31 - (4 * bb + e))) & 1                     # [n, tiles, 4 g, 32 p]
                for g in range(4):
                    out[:, :, :, 128 * q + 16 * bb + 4 * g + e] = bit[:, :, g].bool()
    return out.reshape(n, tiles * 32, words * 128)[:, :pts, :F]


def _half_ulp16(x: torch.Tensor) -> torch.Tensor:
    """Half a bfloat16 ulp (8 significant bits) at magnitude |x|."""
    _, e = torch.frexp(x.abs().clamp_min(1e-38))  # |x| = m 2^e, m in [0.5, 1)
    return torch.exp2((e - 9).double())


@dataclass
class Report:
    rows: List[tuple] = field(default_factory=list)    # (what, error / max|ref|, tolerance)
    flips: List[str] = field(default_factory=list)     # human-readable flipped elements
    n_flips: int = 0
    n_values: int = 0
    unforced: int = 0                                   # checks skipped because no stored input preceded them

    def add(self, what, err, tol):
        self.rows.append((what, float(err), tol))

    def worst(self):
        return max(self.rows, key=lambda r: r[1] / r[2]) if self.rows else ("", 0.0, 1.0)

    def failures(self):
        return [r for r in self.rows if not r[1] <= r[2]]

    def summary(self) -> str:
        w = self.worst()
        return (f"{len(self.rows)} teacher-forced checks over {self.n_values} stored values, worst {w[1]:.2e} of max|ref| "
                f"(tolerance {w[2]:.0e}; {w[0]}); {self.n_flips} bf16 roundings differ from the float64 reference's")


def _check(rep: Report, what: str, stored: torch.Tensor, ref: torch.Tensor, is16: bool, tol: float):
    assert stored.shape == ref.shape, (what, stored.shape, ref.shape)
    assert torch.isfinite(stored).all(), f"{what}: non-finite values"
    m = max(float(ref.abs().max()), 1e-30)
    d = (stored - ref).abs()
    if is16:
        d = (d - _half_ulp16(torch.maximum(stored.abs(), ref.abs()))).clamp_min(0.0)
        flipped = (stored != r16(ref)).nonzero()
        rep.n_flips += flipped.shape[0]
        for idx in flipped[:2].tolist():
            if len(rep.flips) < 12:
                i = tuple(idx)
                rep.flips.append(f"{what}{list(i)}: float64 reference {float(ref[i]):.9g} rounds to {float(r16(ref)[i]):.9g}, "
                                 f"the launch stored {float(stored[i]):.9g}")
    rep.n_values += stored.numel()
    rep.add(what, float(d.max()) / m, tol)


def _bcast(x: torch.Tensor, n_tasks: int) -> torch.Tensor:
    """A tensor with ``modulus`` tasks seen by a chain of n_tasks (task t reads t % modulus)."""
    if x.shape[0] == n_tasks:
        return x
    assert n_tasks % x.shape[0] == 0, (x.shape, n_tasks)
    return x.repeat(n_tasks // x.shape[0], *([1] * (x.dim() - 1)))


def walk_forward(rec, rep: Report, tag: str = "") -> None:
    _, chain, T, saved, outputs, bf16 = rec
    R = r16 if bf16 else (lambda x: x)
    n, pts = chain.n_tasks, chain.pts
    cur: Optional[torch.Tensor] = None
    forced = True   # every rounded operand since the chain's input came from a tensor the launch stored
    outs = list(outputs)
    oi = 0

    def operand(i, F, what):
        """The operand of a contraction at step i: the launch's own stored copy when there is one (checked against cur)."""
        nonlocal cur, forced
        if cur is not None and cur.shape[-1] < F:  # (a skinny input zero-padded to the layer's fan-in, e.g. dx = 1 -> 4)
            cur = torch.nn.functional.pad(cur, (0, F - cur.shape[-1]))
        t = saved.get((i, "in"))
        if t is None:
            if bf16:
                forced = False
            return R(cur)
        stored = unpack_any(t, pts, F)
        if cur is not None and forced:
            _check(rep, f"{tag}step {i} {what}: stored input", stored, cur, t.dtype == torch.bfloat16, TOL_STEP)
        forced = True
        if t.dtype != torch.bfloat16:
            cur = stored
        return R(stored)

    def relu_saved(i, F, pre):
        """A ReLU output's mask as the launch kept it: bits, or the stored activation."""
        nonlocal cur
        if (i, "mask") in saved and forced:
            bits = unpack_bits(saved[(i, "mask")], pts, F)
            tol = TOL_STEP * max(float(pre.abs().max()), 1e-30)
            bad = (bits != (pre > 0)) & (pre.abs() > tol)
            rep.n_values += bits.numel()
            rep.add(f"{tag}step {i}: ReLU bits that disagree with the recomputed pre-activation beyond the tolerance",
                    float(bad.sum()), 0.5)  # (a count: any such element fails)
        if (i, "out") in saved and forced:
            t = saved[(i, "out")]
            _check(rep, f"{tag}step {i}: stored ReLU output", unpack_any(t, pts, F), cur, t.dtype == torch.bfloat16, TOL_STEP)

    for i, st in enumerate(chain.steps):
        k, a = st.kind, st.a
        if k == "input_pt":
            cur, forced = _bcast(unpack_any(T[st.t["x"]], pts, a["F"]), n), True
        elif k in ("input_rm", "input_rows"):
            cur, forced = _bcast(T[st.t["x"]].detach().cpu().double(), n)[:, :pts], True
        elif k == "linear":
            N, K = a["N"], a["K"]
            W = T[st.t["W"]].detach().cpu().double()
            x_in = cur
            xop = operand(i, K, f"linear {K}->{N}")
            if a.get("res"):
                x_in = cur  # (the residual adds the fp32 input, which is the stored tensor here)
            y = xop @ R(W).t()
            if st.t["b"] >= 0:
                b = T[st.t["b"]].detach().cpu().double()
                y = y + (_bcast(b[:, :N], n)[:, None, :] if a["bpt"] else b[:N])
            if st.t["add"] >= 0:
                ad = T[st.t["add"]]
                y = y + _bcast(ad.detach().cpu().double() if a.get("add_rm") else unpack_any(ad, pts, N), n)
            pre = y
            cur = torch.relu(y) if a["relu"] else y
            if a["relu"]:
                relu_saved(i, N, pre)
            if a.get("res"):
                cur = cur + x_in
        elif k == "add_pt":
            pre = cur + _bcast(unpack_any(T[st.t["x"]], pts, a["F"]), n)
            cur = torch.relu(pre) if a["relu"] else pre
            if a["relu"]:
                relu_saved(i, a["F"], pre)
        elif k == "add_taskvec":
            v = _bcast(T[st.t["v"]].detach().cpu().double()[:, :a["F"]], n)
            pre = cur + v[:, None, :]
            cur = torch.relu(pre) if a["relu"] else pre
            if a["relu"]:
                relu_saved(i, a["F"], pre)
        elif k == "dropout":
            m = unpack32(T[st.t["m"]], pts, a["F"])
            cur = torch.where(m > 0, cur, torch.zeros_like(cur)) / (1.0 - a["p"])
        elif k == "layernorm":
            g, b = T[st.t["g"]].detach().cpu().double(), T[st.t["b"]].detach().cpu().double()
            if (i, "in") in saved:
                stored = unpack32(saved[(i, "in")], pts, a["F"])
                if forced:
                    _check(rep, f"{tag}step {i} layernorm: stored input", stored, cur, False, TOL_STEP)
                cur, forced = stored, True
            mu = cur.mean(-1, keepdim=True)
            var = ((cur - mu) ** 2).mean(-1, keepdim=True)
            cur = (cur - mu) / torch.sqrt(var + a["eps"]) * g + b
        elif k == "attn_scores":
            Kk = unpack32(T[st.t["k"]], a["C"], a["r"])
            cur = operand(i, a["r"], "attention scores") @ R(Kk).transpose(1, 2)
        elif k == "softmax":
            cur = torch.softmax(a["scale"] * cur, dim=-1)
            if (i, "out") in saved and forced:
                t = saved[(i, "out")]
                _check(rep, f"{tag}step {i}: stored softmax probabilities", unpack_any(t, pts, a["n"]), cur,
                       t.dtype == torch.bfloat16, TOL_STEP)
        elif k == "attn_values":
            V = unpack32(T[st.t["v"]], a["C"], a["r"])
            cur = operand(i, a["C"], "attention values") @ R(V)
        elif k == "store_tr":
            o = outs[oi]
            oi += 1
            if forced:
                _check(rep, f"{tag}step {i}: feature-major copy", o.detach().cpu().double()[:, :, :pts], cur.transpose(1, 2),
                       False, TOL_STEP)
        elif k in ("store_wb", "store_trb"):
            oi += 1
        elif k == "tap" and a.get("alias"):
            oi += 1
        elif k in ("tap", "output_pt"):
            o = outs[oi]
            oi += 1
            if forced:
                _check(rep, f"{tag}step {i}: fp32 output", unpack32(o, pts, a["F"]), cur, False, TOL_STEP)
            else:
                rep.unforced += 1
        elif k == "output_rows":
            o = outs[oi]
            oi += 1
            if forced:
                _check(rep, f"{tag}step {i}: output rows", o.detach().cpu().double(), cur, False, TOL_STEP)
            else:
                rep.unforced += 1
        else:  # pragma: no cover
            raise AssertionError(k)


def walk_backward(rec, rep: Report, tag: str = "") -> None:
    _, chain, gouts, bufs, jobs, bf16, saved, T = rec
    R = r16 if bf16 else (lambda x: x)
    n, pts = chain.n_tasks, chain.pts
    gouts = list(gouts)
    cur: Optional[torch.Tensor] = None
    started = False

    def job_of(dz_t, a_is_dz=False):
        for jb in jobs:
            if (jb["A"] if a_is_dz else jb["dZ"]) is dz_t:
                return jb
        return None

    def mask_of(i, F):
        if (i, "mask") in saved:
            return unpack_bits(saved[(i, "mask")], pts, F)
        return unpack_any(saved[(i, "out")], pts, F) > 0

    def stored_grad(i, F, what):
        """The gradient buffer the launch stored at step i (checked against cur); returns the dgrad operand."""
        nonlocal cur
        t = bufs.get((i, "dz"))
        if t is None:
            return R(cur), None
        stored = unpack_any(t, pts, F)
        _check(rep, f"{tag}step {i} {what}", stored, cur, t.dtype == torch.bfloat16, TOL_STEP)
        if t.dtype != torch.bfloat16:
            cur = stored
        return R(stored), t

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
            gv = g.detach().cpu().double() if k == "output_rows" else unpack32(g.contiguous(), pts, a["F"])
            cur = gv if not started else cur + gv
            started = True
            continue
        if not started:
            continue
        if k == "linear":
            N, K = a["N"], a["K"]
            W = T[st.t["W"]].detach().cpu().double()
            g_res = None
            if (i, "g_res") in bufs:
                _check(rep, f"{tag}step {i}: parked residual gradient", unpack32(bufs[(i, "g_res")], pts, N), cur, False, TOL_STEP)
                g_res = cur
            if a["relu"]:
                cur = cur * mask_of(i, N)
            dop, dz_t = stored_grad(i, N, f"linear {K}->{N}: stored dZ")
            if dz_t is not None:
                jb = job_of(dz_t)
                if jb is not None:
                    A = R(unpack_any(jb["A"], pts, K))
                    dW = torch.einsum("bpn,bpk->nk", dop, A)
                    rep.n_values += dW.numel()
                    m = max(float(dW.abs().max()), 1e-30)
                    rep.add(f"{tag}step {i} linear {K}->{N}: weight gradient vs dZ_stored^T A_stored",
                            float((jb["dW"].detach().cpu().double()[:N, :K] - dW).abs().max()) / m, TOL_SUM)
                    if jb.get("db") is not None:
                        db = unpack_any(dz_t, pts, N).sum((0, 1))
                        rep.add(f"{tag}step {i} linear {K}->{N}: bias gradient vs the sum of the stored dZ",
                                float((jb["db"].detach().cpu().double()[:N] - db).abs().max()) / max(float(db.abs().max()), 1e-30),
                                TOL_SUM)
            if chain_upstream(rec, i):
                cur = dop @ R(W)
                if g_res is not None:
                    cur = cur + g_res
            else:
                break
        elif k in ("add_pt", "add_taskvec"):
            if a["relu"]:
                cur = cur * mask_of(i, a["F"])
            stored_grad(i, a["F"], "addend gradient")
        elif k == "dropout":
            m = unpack32(T[st.t["m"]], pts, a["F"])
            cur = torch.where(m > 0, cur, torch.zeros_like(cur)) / (1.0 - a["p"])
        elif k == "layernorm":
            g = T[st.t["g"]].detach().cpu().double()
            x = unpack32(saved[(i, "in")], pts, a["F"])
            mu = x.mean(-1, keepdim=True)
            rstd = 1.0 / torch.sqrt(((x - mu) ** 2).mean(-1, keepdim=True) + a["eps"])
            xh = (x - mu) * rstd
            dxh = cur * g
            cur = rstd * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))
        elif k == "attn_values":
            V = unpack32(T[st.t["v"]], a["C"], a["r"])
            dop, dO_t = stored_grad(i, a["r"], "attention: stored dO")
            if dO_t is not None:
                jb = job_of(dO_t, a_is_dz=True)
                P = R(unpack_any(jb["dZ"], pts, a["C"]))
                dV = torch.einsum("bpc,bpr->bcr", P, dop)
                rep.n_values += dV.numel()
                rep.add(f"{tag}step {i} attention: value gradient vs P_stored^T dO_stored",
                        float((unpack32(jb["dW"], a["C"], a["r"]) - dV).abs().max()) / max(float(dV.abs().max()), 1e-30), TOL_SUM)
            cur = dop @ R(V).transpose(1, 2)
        elif k == "softmax":
            P = unpack_any(saved[(i, "out")], pts, a["n"])
            cur = a["scale"] * P * (cur - (cur * P).sum(-1, keepdim=True))
        elif k == "attn_scores":
            Kk = unpack32(T[st.t["k"]], a["C"], a["r"])
            dop, dS_t = stored_grad(i, a["C"], "attention: stored dS")
            if dS_t is not None:
                jb = job_of(dS_t)
                q = R(unpack_any(jb["A"], pts, a["r"]))
                dK = torch.einsum("bpc,bpr->bcr", dop, q)
                rep.n_values += dK.numel()
                rep.add(f"{tag}step {i} attention: key gradient vs dS_stored^T q_stored",
                        float((unpack32(jb["dW"], a["C"], a["r"]) - dK).abs().max()) / max(float(dK.abs().max()), 1e-30), TOL_SUM)
            if chain_upstream(rec, i):
                cur = dop @ R(Kk)
            else:
                break
        elif k == "input_pt":
            for t in bufs.get((i, "fan_in"), []):
                cur = cur + unpack32(t, pts, a["F"])
            if (i, "dx") in bufs:
                _check(rep, f"{tag}step {i}: gradient of the chain input", unpack32(bufs[(i, "dx")], pts, a["F"]), cur, False,
                       TOL_STEP)


# ---------------------------------------------------------------------------------------------------------------------------
# x6 / b16 programs (npf_gwwaveform_amd/x6.py ``Program``: one launch = a list of ops on the register-resident activation)
# ---------------------------------------------------------------------------------------------------------------------------
def unpack_xbits(b: torch.Tensor, F: int) -> torch.Tensor:
    """ReLU bits of the program kernels [n, tiles, 2 halves, 64 lanes] int64 -> bool [n, tiles * 32, F]: bit 4 b + e of lane
    (p = lane % 16, g = lane // 16) = feature 16 b + 4 g + e of point 16 half + p (include/npf_hip.h)."""
    n, tiles = b.shape[:2]
    w = b.detach().cpu().view(n, tiles, 2, 4, 16)           # [.., half, g, p]
    out = torch.zeros(n, tiles, 2, 16, F, dtype=torch.bool)  # [.., half, p, feature]
    for blk in range(F // 16):
        for e in range(4):
            bit = ((w >> (4 * blk + e)) & 1).bool()          # [n, tiles, half, g, p]
            for g in range(4):
                out[..., 16 * blk + 4 * g + e] = bit[:, :, :, g, :]
    return out.reshape(n, tiles * 32, F)


def _op_weights(o, n: int, F: int) -> torch.Tensor:
    """The fp32 matrix W [n or 1, F out, F in] an op multiplies by (``w_ref`` of the op: what its image was made from)."""
    kind = o["w_ref"][0]
    if kind == "shared":
        W = o["w_ref"][1].detach().cpu().double()
        Wp = torch.zeros(1, F, F, dtype=torch.float64)
        Wp[0, :W.shape[0], :W.shape[1]] = W
        return Wp
    _, src, C = o["w_ref"]
    M = unpack32(src, C, F)                                   # [n, C points, F features]
    Wp = torch.zeros(n, F, F, dtype=torch.float64)
    if kind == "task_row":                                    # W[point][feature]
        Wp[:, :C, :] = M
    else:                                                     # "task_tr": W[feature][point]
        Wp[:, :, :C] = M.transpose(1, 2)
    return Wp


def walk_program(prog, rep: Report, tag: str = "") -> None:
    """One program launch, op by op (include/npf_hip.h, npf_x6_op_t): every stored tensor against the op applied to the previous
    STORED tensor.  b16 programs round the input of every multiply and the PT16 stores; x6 programs are fp32 throughout."""
    n, pts, F = prog.n_tasks, prog.tiles * 32, prog.width
    R = r16 if prog.bf16 else (lambda x: x)
    cur = None
    forced = True   # cur is a tensor the launch read or stored (not a value this walker computed)
    for l, o in enumerate(prog.ops):
        name = f"{tag}op {l}"
        if o.get("in_pt") is not None:
            cur, forced = unpack32(o["in_pt"], pts, F), True
        if o.get("in_rows") is not None:
            rows, w = o["in_rows"].detach().cpu().double(), o["in_w"].detach().cpu().double()
            v = rows @ w
            if o.get("in_b") is not None:
                v = v + o["in_b"].detach().cpu().double()
            if o.get("in_relu"):
                v = v.clamp_min(0.0)
            cur = torch.zeros(n, pts, F, dtype=torch.float64)
            cur[..., :v.shape[-1]] = v
            forced = False
        if o.get("pre_add") is not None:
            cur, forced = cur + unpack32(o["pre_add"], pts, F), False
        if o.get("mask_bits") is not None:
            cur = torch.where(unpack_xbits(o["mask_bits"], F), cur, torch.zeros_like(cur))
        if o.get("sbwd_p") is not None:
            P = unpack_any(o["sbwd_p"], pts, F)
            cur, forced = o["sbwd_scale"] * P * (cur - (cur * P).sum(-1, keepdim=True)), False
        if o.get("store_in") is not None:
            t = o["store_in"]
            stored = unpack_any(t, pts, F)
            _check(rep, f"{name}: stored input", stored, cur, t.dtype == torch.bfloat16, TOL_STEP)
            cur, forced = stored, True
        if o.get("store_in_bits") is not None:
            bits = unpack_xbits(o["store_in_bits"], F)
            bad = (bits != (cur > 0)) & (cur.abs() > TOL_STEP * float(cur.abs().max()))
            rep.add(f"{name}: ReLU bits of the input", float(bad.sum()), 0.5)
        if o.get("img") is None:
            continue
        rep.unforced += 0 if forced else 1  # (a multiply whose rounded input no stored tensor pins)
        W = R(_op_weights(o, n, F))
        y = torch.einsum("bpk,bnk->bpn", R(cur), W.expand(n, F, F))
        if o.get("bias") is not None:
            b = o["bias"].detach().cpu().double()
            y = y + (b[:, None, :] if b.dim() == 2 else b)
        if o.get("addend") is not None:
            y = y + unpack32(o["addend"], pts, F)
        if o.get("relu"):
            y = y.clamp_min(0.0)
        if o.get("softmax_n"):
            c = o["softmax_n"]
            P = torch.softmax(o.get("softmax_scale", 1.0) * y[..., :c], dim=-1)
            y = torch.zeros_like(y)
            y[..., :c] = P
        if o.get("store_bits") is not None:
            bits = unpack_xbits(o["store_bits"], F)
            bad = (bits != (y > 0)) & (y.abs() > TOL_STEP * float(y.abs().max()))
            rep.add(f"{name}: ReLU bits of the output", float(bad.sum()), 0.5)
        if o.get("store_out") is not None:
            t = o["store_out"]
            stored = unpack_any(t, pts, F)
            _check(rep, f"{name}: stored output", stored, y, t.dtype == torch.bfloat16, TOL_STEP)
            cur, forced = stored, True
        else:
            cur, forced = y, False
    if prog.tail is not None:
        Wo, bo, rows = prog.tail
        y = R(cur) @ R(Wo.detach().cpu().double()).t()
        if bo is not None:
            y = y + bo.detach().cpu().double()
        _check(rep, f"{tag}output layer", rows.detach().cpu().double(), y, False, TOL_STEP)


def walk_wgrad(rec, rep: Report, tag: str = "") -> None:
    """The weight / key / value gradient jobs of a fused side: dW = dZ_stored^T A_stored (operands rounded where the launch rounds
    them), db = the sum of the dZ buffer as stored."""
    _, jobs, n, pts, bf16 = rec
    R = r16 if bf16 else (lambda x: x)
    for j, jb in enumerate(jobs):
        N, K = jb["N"], jb["K"]
        Fz, Fa = jb["dZ"].shape[2] * jb["dZ"].shape[4], jb["A"].shape[2] * jb["A"].shape[4]
        dZ, A = unpack_any(jb["dZ"], pts, Fz)[..., :N], unpack_any(jb["A"], pts, Fa)[..., :K]
        if jb.get("per_task"):
            ref = torch.einsum("bpn,bpk->bnk", R(dZ), R(A))
            got = unpack32(jb["dW"], N, jb["dW"].shape[2] * 4)[..., :K]
        else:
            ref = torch.einsum("bpn,bpk->nk", R(dZ), R(A))
            got = jb["dW"].detach().cpu().double()[:N, :K]
        rep.n_values += ref.numel()
        rep.add(f"{tag}job {j} ({N} x {K}): weight gradient vs dZ_stored^T A_stored",
                float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30), TOL_SUM)
        if jb.get("db") is not None:
            db = dZ.sum((0, 1))
            rep.add(f"{tag}job {j}: bias gradient vs the sum of the stored dZ",
                    float((jb["db"].detach().cpu().double()[:N] - db).abs().max()) / max(float(db.abs().max()), 1e-30), TOL_SUM)


def chain_upstream(rec, i: int) -> bool:
    """Does a gradient continue below step i (the ``upstream_before`` of chain._ChainFn, re-derived from the record)."""
    chain, T = rec[1], rec[7]
    up = False
    for j, st in enumerate(chain.steps[:i]):
        if st.kind == "input_pt":
            up = T[st.t["x"]] is not None and chain_needs(rec, st.t["x"])
        elif st.kind in ("input_rm", "input_rows"):
            up = False
        else:
            up = up or any(chain_needs(rec, idx) for idx in st.t.values() if idx >= 0)
    return up


def chain_needs(rec, idx: int) -> bool:
    t = rec[1].tensors[idx]
    return t is not None and t.requires_grad and getattr(rec[1], "grad_enabled", True)


def check_trace(trace, tag: str = "") -> Report:
    """Walk every chain execution of a ``chain.TRACE`` list (forward and backward records)."""
    rep = Report()
    for j, rec in enumerate(trace):
        if rec[0] == "prog":
            walk_program(rec[1], rep, f"{tag}launch {j} ({'b16' if rec[1].bf16 else 'x6'} program: {rec[1].tag}) ")
            continue
        if rec[0] == "wgrad":
            walk_wgrad(rec, rep, f"{tag}launch {j} (fused side's gradient jobs) ")
            continue
        name = f"{tag}chain {j} ({'bf16' if rec[5] else 'fp32'} instance) "
        if rec[0] == "fwd":
            walk_forward(rec, rep, name)
        else:
            walk_backward(rec, rep, name)
    return rep
