"""CPU restatement (pure PyTorch, fp32) of the reference's neural-process hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  This file is the checker the
HIP path is compared against and the ``cpu_baseline`` timed by ``bench.py``; the product
never imports it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
(``/root/reference/npf``) in the build container, runs it on seeded inputs and stores the
results under ``tests/golden/*.npz``; ``tests/test_oracle.py`` requires this restatement
to reproduce them bit-for-bit on the small cases (G1/G2/G6/G7) and within the stated
fp32 tolerance on the large ones (G3-G5; the last-ulp level there depends on the BLAS
blocking).

bf16 emulation (``matmul_mode("bf16")``, BASELINE config 3): the same restatement with every
contraction's operands rounded to bfloat16 (round-to-nearest-even) exactly where the HIP kernels of
the bf16 compute mode round them, fp32 accumulation, everything else in fp32 -- forward AND backward
(hand-written autograd functions, because the backward contractions round *their* operands too).
The reference has no bf16 path, so this mode is not pinned by reference vectors: it is the pinned fp32
restatement plus the rounding points DESIGN.md section 8 states, and it exists so that the bf16
kernels face a full-tensor gate instead of a cosine.  It models the scaled-dot / mean-aggregation
family with at most 256 features and 256 context points (what configs 3 and 4 run).

Style: functional and state_dict driven -- every function takes the flat parameter
dict (reference key names, e.g. ``decoder.flat_module.linears.0.weight``) so that the
same dict drives the reference, this oracle and the HIP path.  Every function cites the
reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

MODEL_KINDS = ("CNP", "LNP", "AttnCNP", "AttnLNP")


@dataclass
class OracleConfig:
    """What the reference's constructors fix (npf/neuralproc/base.py:104-152,423-460;
    np.py:54,141; attnnp.py:66-101,161-162).  Layer sizes are NOT listed here: they are
    read off the parameter dict."""

    kind: str
    x_dim: int = 1
    y_dim: int = 2
    r_dim: int = 128
    encoded_path: Optional[str] = None  # default per kind, as the reference classes force it
    is_heteroskedastic: bool = True
    is_q_zCct: bool = False
    z_dim: Optional[int] = None
    attention: str = "scaledot"  # attentive kinds: "scaledot" | "multihead" | "transformer" (attention.py:16-86)
    n_heads: int = 8
    x_transf_dim: Optional[int] = None   # width of the encoded x (base.py:126-131); default r_dim
    is_sum_merge: bool = True            # XY-encoder merge flavour (encoders.py:163-183; False = cat(x, y) -> MLP.  The
                                         # reference's concatenating merge cannot serve as a decoder: its torch.cat of a
                                         # 3-d X_trgt and a 4-d R_trgt raises, encoders.py:181)
    is_res: bool = False                 # residual hidden layers in the XY-encoder / decoder flat MLPs (mlp.py:100-104)
    dropout: float = 0.0                 # dropout of those two MLPs (mlp.py:81,98,105); masks via ``DROPOUT_MASKS``

    def __post_init__(self):
        if self.kind not in MODEL_KINDS:
            raise ValueError(f"unknown kind {self.kind}")
        if self.attention not in ("scaledot", "multihead", "transformer"):
            raise ValueError(f"unknown attention {self.attention}")
        if self.encoded_path is None:
            self.encoded_path = {
                "CNP": "deterministic",
                "AttnCNP": "deterministic",
                "LNP": "latent",
                "AttnLNP": "both",
            }[self.kind]
        if self.z_dim is None:
            self.z_dim = self.r_dim
        if self.x_transf_dim is None:
            self.x_transf_dim = self.r_dim

    @property
    def is_latent(self) -> bool:
        return self.kind in ("LNP", "AttnLNP")

    @property
    def is_attentive(self) -> bool:
        return self.kind in ("AttnCNP", "AttnLNP")


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
# Tests may set this to a list: every ReLU then appends the smallest |pre-activation| it saw.  A value
# within fp32 rounding of zero means the ReLU derivative there is decided by rounding noise, and two
# correct fp32 implementations can legitimately disagree on the gradient (tests/test_hip_stress.py).
RELU_MARGINS = None


# --------------------------------------------------------------------------------------
# bf16 emulation of the HIP bf16 compute mode
# --------------------------------------------------------------------------------------
MATMUL_MODE = "fp32"


class matmul_mode:
    """``with matmul_mode("bf16"): ...`` -- emulate the HIP path's bf16 compute mode."""

    def __init__(self, mode: str):
        if mode not in ("fp32", "bf16"):
            raise ValueError(mode)
        self.mode = mode

    def __enter__(self):
        global MATMUL_MODE
        self.prev, MATMUL_MODE = MATMUL_MODE, self.mode
        return self

    def __exit__(self, *exc):
        global MATMUL_MODE
        MATMUL_MODE = self.prev


def _r16(t: torch.Tensor) -> torch.Tensor:
    """Round to bfloat16 (nearest even, what v_cvt_pk_bf16_f32 does) and back to fp32."""
    return t.to(torch.bfloat16).to(torch.float32)


class _LinearBf16(torch.autograd.Function):
    """y = bf16(x) bf16(W)^T + b with fp32 accumulation (chain_kernel BF16 instance: the weight image is
    rounded once per step, the layer input at the MFMA, the bias initialises the fp32 accumulator).
    Backward as the dgrad chain + wgrad launch do it: dZ is rounded at the dgrad MFMA and in the PT16
    buffer the wgrad kernel reads, so dx = bf16(dy) bf16(W), dW = bf16(dy)^T bf16(x); db sums the dZ buffer
    as stored -- PT16 (rounded) unless the same buffer also carries an addend's gradient back to autograd
    (then it is an fp32 tensor: ``db_rounded=False``)."""

    @staticmethod
    def forward(ctx, x, W, b, db_rounded):
        xr, Wr = _r16(x), _r16(W)
        ctx.save_for_backward(xr, Wr)
        ctx.has_b, ctx.db_rounded = b is not None, db_rounded
        y = xr @ Wr.t()
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        xr, Wr = ctx.saved_tensors
        N, K = Wr.shape
        dyr = _r16(dy)
        dx = dyr @ Wr
        dW = dyr.reshape(-1, N).t() @ xr.reshape(-1, K)
        db = (dyr if ctx.db_rounded else dy).reshape(-1, N).sum(0) if ctx.has_b else None
        return dx, dW, db, None


def linear(x, W, b=None, db_rounded: bool = True):
    """``F.linear`` of the reference, or its bf16-mode emulation."""
    if MATMUL_MODE == "bf16":
        return _LinearBf16.apply(x, W, b, db_rounded)
    return F.linear(x, W, b)


class _ScaledotBf16(torch.autograd.Function):
    """softmax(bf16(q) bf16(K)^T / sqrt(d)) in fp32, then bf16(P) bf16(V) (two per-task-weight LINEARs of the
    bf16 chain with the in-register softmax between them).  Backward: dP = bf16(dO) bf16(V)^T; the softmax
    backward uses the SAVED probabilities, which are a PT16 tensor, i.e. bf16(P): dS = scale P16 (dP - <dP, P16>);
    dQ = bf16(dS) bf16(K), dK = bf16(dS)^T bf16(q), dV = bf16(P)^T bf16(dO)."""

    @staticmethod
    def forward(ctx, keys, queries, values):
        kr, qr, vr = _r16(keys), _r16(queries), _r16(values)
        scale = 1.0 / math.sqrt(queries.size(-1))
        s = torch.einsum("bkd,bqd->bqk", kr, qr)
        m = s.max(dim=-1, keepdim=True).values
        e = torch.exp((s - m) * scale)
        P = e / e.sum(-1, keepdim=True)
        Pr = _r16(P)
        ctx.save_for_backward(kr, qr, vr, Pr)
        ctx.scale = scale
        return torch.bmm(Pr, vr)

    @staticmethod
    def backward(ctx, dO):
        kr, qr, vr, Pr = ctx.saved_tensors
        dOr = _r16(dO)
        dP = torch.bmm(dOr, vr.transpose(1, 2))
        dS = ctx.scale * Pr * (dP - (dP * Pr).sum(-1, keepdim=True))
        dSr = _r16(dS)
        dQ = torch.bmm(dSr, kr)
        dK = torch.bmm(dSr.transpose(1, 2), qr)
        dV = torch.bmm(Pr.transpose(1, 2), dOr)
        return dK, dQ, dV


def _relu(v: torch.Tensor) -> torch.Tensor:
    if RELU_MARGINS is not None and v.numel():
        RELU_MARGINS.append(float(v.detach().abs().min()))
    return torch.relu(v)


# Training-mode dropout: an iterator of keep masks (0 / 1, shaped like the activation) consumed in call order.  The
# reference draws them from torch's global CPU generator inside nn.Dropout; tests capture those draws (forward hooks in
# tests/golden/make_golden.py) and hand them to the oracle and to the HIP path alike.
DROPOUT_MASKS = None
TRAINING = True


def _dropout(h: torch.Tensor, p: float) -> torch.Tensor:
    """``nn.Dropout(p)`` in training mode (mlp.py:81) with the mask taken from ``DROPOUT_MASKS``: torch multiplies by
    mask / (1 - p)."""
    if p <= 0.0 or DROPOUT_MASKS is None:
        return h
    keep = next(DROPOUT_MASKS).reshape(h.shape).to(h.dtype)
    return h * keep.div(1.0 - p)


def mlp(params: Params, prefix: str, x: torch.Tensor, out_db_rounded: bool = True, is_res: bool = False,
        dropout: float = 0.0) -> torch.Tensor:
    """``MLP.forward`` (npf/architectures/mlp.py:95-109) with ReLU:
    to_hidden -> relu -> dropout -> [linears.i -> relu (+ its input when ``is_res``, :103-104) -> dropout]* -> out.
    (``out_db_rounded``: bf16 emulation only, see ``_LinearBf16``.)"""
    h = _dropout(_relu(linear(x, params[f"{prefix}.to_hidden.weight"], params[f"{prefix}.to_hidden.bias"])), dropout)
    i = 0
    while f"{prefix}.linears.{i}.weight" in params:
        o = _relu(linear(h, params[f"{prefix}.linears.{i}.weight"], params[f"{prefix}.linears.{i}.bias"]))
        h = _dropout(o + h if is_res else o, dropout)
        i += 1
    return linear(h, params[f"{prefix}.out.weight"], params[f"{prefix}.out.bias"], out_db_rounded)


def merge_flat_sum(params: Params, prefix: str, x1: torch.Tensor, x2: torch.Tensor, fused_addend: bool = True,
                   is_res: bool = False, dropout: float = 0.0) -> torch.Tensor:
    """``MergeFlatInputs.forward`` with ``is_sum_merge=True``
    (npf/architectures/encoders.py:175-183): flat(relu(x1 + resizer(x2))).
    (``fused_addend``: bf16 emulation only -- x1 enters the resizer's last layer as its addend inside one
    chain, whose dZ buffer is then an fp32 tensor.)"""
    x2 = mlp(params, f"{prefix}.resizer", x2, out_db_rounded=not fused_addend)
    return mlp(params, f"{prefix}.flat_module", _relu(x1 + x2), is_res=is_res, dropout=dropout)


def merge_flat(cfg: "OracleConfig", params: Params, prefix: str, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """``MergeFlatInputs.forward`` (encoders.py:175-183) in the configured flavour: the sum merge above, or
    ``flat(cat(x1, x2))`` (:180-181; x1 broadcast over the leading dims of x2 like the sum would)."""
    if cfg.is_sum_merge:
        return merge_flat_sum(params, prefix, x1, x2, is_res=cfg.is_res, dropout=cfg.dropout if TRAINING else 0.0)
    if MATMUL_MODE == "bf16":
        raise NotImplementedError("the bf16 emulation models the sum-merge path only")
    x1 = x1.expand(*x2.shape[:-1], x1.shape[-1])
    return mlp(params, f"{prefix}.flat_module", torch.cat((x1, x2), dim=-1), is_res=cfg.is_res,
               dropout=cfg.dropout if TRAINING else 0.0)


def scaledot_attend(keys: torch.Tensor, queries: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
    """``get_attender("scaledot")`` = ``DotAttender`` on ``BaseAttender.forward``
    (npf/architectures/attention.py:129-164,204-220): softmax(Q K^T / sqrt(d)) V, single
    head, no learned projections, no resizer (value_size == out_size on this path)."""
    if MATMUL_MODE == "bf16":
        return _ScaledotBf16.apply(keys, queries, values)
    logits = torch.einsum("bkd,bqd->bqk", keys, queries) / math.sqrt(queries.size(-1))
    attn = logits.softmax(dim=-1)
    return torch.bmm(attn, values)


def multihead_attend(params: Params, prefix: str, keys: torch.Tensor, queries: torch.Tensor, values: torch.Tensor,
                     n_heads: int, post_process: bool = True) -> torch.Tensor:
    """``MultiheadAttender.forward`` (npf/architectures/attention.py:456-527): K (no bias), Q
    (bias), V (no bias) projections, heads stacked as extra batches (``_make_multiheaded``
    :507-516: index h * B + b), scaled-dot attention per head with the *head* size in the
    scale (:216-218), heads concatenated back (:518-527), optional post Linear (:500-503)."""
    K = F.linear(keys, params[f"{prefix}.key_transform.weight"])
    Q = F.linear(queries, params[f"{prefix}.query_transform.weight"], params[f"{prefix}.query_transform.bias"])
    V = F.linear(values, params[f"{prefix}.value_transform.weight"])
    B = keys.shape[0]
    hs = K.shape[-1] // n_heads

    def heads(x):
        return x.view(B, -1, n_heads, hs).permute(2, 0, 1, 3).contiguous().view(B * n_heads, -1, hs)

    ctx = scaledot_attend(heads(K), heads(Q), heads(V))
    ctx = ctx.view(n_heads, B, -1, hs).permute(1, 2, 0, 3).contiguous().view(B, -1, n_heads * hs)
    if post_process:
        ctx = F.linear(ctx, params[f"{prefix}.post_processor.weight"], params[f"{prefix}.post_processor.bias"])
    return ctx


def transformer_attend(params: Params, prefix: str, keys: torch.Tensor, queries: torch.Tensor, values: torch.Tensor,
                       n_heads: int) -> torch.Tensor:
    """``TransformerAttender.forward`` (attention.py:566-588): multihead attention without the post
    Linear, residual with the (untransformed) queries + LayerNorm, then residual MLP + LayerNorm."""
    d = queries.shape[-1]
    ctx = multihead_attend(params, prefix, keys, queries, values, n_heads, post_process=False)
    ctx = F.layer_norm(ctx + queries, (d,), params[f"{prefix}.layer_norm1.weight"], params[f"{prefix}.layer_norm1.bias"])
    ctx = F.layer_norm(ctx + mlp(params, f"{prefix}.mlp", ctx), (d,), params[f"{prefix}.layer_norm2.weight"],
                       params[f"{prefix}.layer_norm2.bias"])
    return ctx


def attend(cfg: "OracleConfig", params: Params, keys: torch.Tensor, queries: torch.Tensor, values: torch.Tensor):
    """``self.attender(keys, queries, values)`` of attnnp.py:128 for the configured flavour."""
    if cfg.attention == "scaledot":
        return scaledot_attend(keys, queries, values)
    if cfg.attention == "multihead":
        return multihead_attend(params, "attender", keys, queries, values, cfg.n_heads)
    return transformer_attend(params, "attender", keys, queries, values, cfg.n_heads)


def p_y_scale_transform(raw: torch.Tensor) -> torch.Tensor:
    """npf/neuralproc/base.py:116."""
    return 0.01 + 0.99 * F.softplus(raw)


def q_z_scale_transform(raw: torch.Tensor) -> torch.Tensor:
    """npf/neuralproc/base.py:432."""
    return 0.1 + 0.9 * torch.sigmoid(raw)


# --------------------------------------------------------------------------------------
# model stages (same split as the reference's public methods)
# --------------------------------------------------------------------------------------
def encode_globally(cfg: OracleConfig, params: Params, X_enc: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    """``CNP.encode_globally`` (npf/neuralproc/np.py:86-101): per-point XY encoding then
    mean over the context points; ``AttnCNP.encode_globally``
    (npf/neuralproc/attnnp.py:105-116): per-point XY encoding, no pooling."""
    B, C, _ = X_enc.shape
    if cfg.is_attentive:
        if C == 0:
            return torch.zeros(B, 0, cfg.r_dim)
        return merge_flat(cfg, params, "xy_encoder", X_enc, Y)
    R_cntxt = merge_flat(cfg, params, "xy_encoder", X_enc, Y)
    R = torch.mean(R_cntxt, dim=1, keepdim=True)
    if C == 0:
        R = torch.zeros(B, 1, cfg.r_dim)
    return R


def rep_to_lat_input(cfg: OracleConfig, R: torch.Tensor) -> torch.Tensor:
    """Identity for LNP (npf/neuralproc/base.py:549-552); mean over context for AttnLNP
    (npf/neuralproc/attnnp.py:172-181)."""
    if not cfg.is_attentive:
        return R
    B, C, _ = R.shape
    if C == 0:
        R = torch.zeros(B, 1, cfg.r_dim)
    return torch.mean(R, dim=1, keepdim=True)


def infer_latent_dist(cfg: OracleConfig, params: Params, R: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """``infer_latent_dist`` (npf/neuralproc/base.py:516-547) -> (loc, scale) of the
    diagonal Gaussian."""
    suff = mlp(params, "latent_encoder", rep_to_lat_input(cfg, R))
    loc, raw = suff.split(cfg.z_dim, dim=-1)
    return loc, q_z_scale_transform(raw)


def merge_r_z(cfg: OracleConfig, params: Params, R: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """``merge_r_z`` (npf/neuralproc/base.py:554-575)."""
    if MATMUL_MODE == "bf16":
        # the HIP path never materialises the concatenation: the latent half is a per-(sample, task) bias
        # computed once per row of z (``z`` must come in un-expanded, [n_z, B, 1, z]) and added in fp32
        W, r = params["r_z_merger.weight"], cfg.r_dim
        zb = linear(z, W[:, r:], params["r_z_merger.bias"])
        Re = R if R.dim() == z.dim() else R.unsqueeze(0)
        Re = Re.expand(z.shape[0], *Re.shape[1:])  # the deterministic half is contracted per (sample, task, point) row
        return torch.relu(linear(Re, W[:, :r], None) + zb)
    if R.shape != z.shape:
        R = R.unsqueeze(0).expand(*z.shape[:-1], cfg.r_dim)
    return torch.relu(F.linear(torch.cat((R, z), dim=-1), params["r_z_merger.weight"], params["r_z_merger.bias"]))


def trgt_dependent_representation(cfg, params, Xc_enc, z_samples, R, Xt_enc) -> torch.Tensor:
    """np.py:103-110 (CNP), np.py:144-163 (LNP), attnnp.py:118-131 (AttnCNP),
    attnnp.py:183-202 (AttnLNP).  Returns [n_z, B, T, r]."""
    B, T, _ = Xt_enc.shape
    if cfg.kind == "CNP":
        return R.expand(B, T, cfg.r_dim).unsqueeze(0)
    if cfg.kind == "LNP":
        n_z = z_samples.size(0)
        if cfg.encoded_path == "both":
            R_trgt = merge_r_z(cfg, params, R, z_samples)
        else:
            R_trgt = z_samples
            if cfg.z_dim != cfg.r_dim:
                R_trgt = linear(R_trgt, params["reshaper_z.weight"], params["reshaper_z.bias"])
        return R_trgt.expand(n_z, B, T, cfg.r_dim)
    # attentive
    if Xc_enc.shape[1] == 0:
        R_det = torch.zeros(B, T, cfg.r_dim)
    else:
        R_det = attend(cfg, params, Xc_enc, Xt_enc, R)
    if cfg.kind == "AttnCNP":
        return R_det.unsqueeze(0)
    n_z = z_samples.size(0)
    if MATMUL_MODE == "bf16":
        return merge_r_z(cfg, params, R_det.unsqueeze(0), z_samples)
    z = z_samples.expand(n_z, B, T, cfg.z_dim)
    return merge_r_z(cfg, params, R_det, z)


def decode(cfg: OracleConfig, params: Params, Xt_enc: torch.Tensor, R_trgt: torch.Tensor):
    """``NeuralProcessFamily.decode`` (npf/neuralproc/base.py:327-367) -> (loc, scale),
    each [n_z, B, T, y_dim]; homoskedastic pooling per
    npf/neuralproc/helpers.py:21-32."""
    if MATMUL_MODE == "bf16" and not cfg.is_attentive:
        # mean-aggregation models: R_trgt is one vector per (sample, task) expanded over the targets; the HIP
        # path resizes it once per task (its own small chain) and adds the result to every target in fp32
        x2 = mlp(params, "decoder.resizer", R_trgt[..., :1, :])
        suff = mlp(params, "decoder.flat_module", _relu(Xt_enc + x2), is_res=cfg.is_res, dropout=cfg.dropout if TRAINING else 0.0)
    else:
        suff = merge_flat_sum(params, "decoder", Xt_enc, R_trgt, is_res=cfg.is_res, dropout=cfg.dropout if TRAINING else 0.0)
    loc, raw = suff.split(cfg.y_dim, dim=-1)
    scale = p_y_scale_transform(raw)
    if not cfg.is_heteroskedastic:
        n_z, B, T, dy = scale.shape
        s = scale.view(n_z * B, T, dy).mean(1, keepdim=True)
        scale = s.expand(n_z * B, T, dy).view(n_z, B, T, dy)
    return loc, scale


def forward(
    cfg: OracleConfig,
    params: Params,
    X_cntxt: torch.Tensor,
    Y_cntxt: torch.Tensor,
    X_trgt: torch.Tensor,
    Y_trgt: Optional[torch.Tensor] = None,
    eps: Optional[torch.Tensor] = None,
    n_z: int = 1,
    training: bool = True,
) -> dict:
    """``NeuralProcessFamily.forward`` (npf/neuralproc/base.py:177-239) and the latent
    path (base.py:475-514).  ``eps`` [n_z, B, 1, z] replaces the reference's global-RNG
    draw inside ``rsample`` (base.py:512): z = loc + eps * scale.

    Returns a dict with ``loc``/``scale`` [n_z,B,T,dy] and, for latent models,
    ``z_samples`` and the (loc, scale) pairs ``q_zCc`` / ``q_zCct``.
    """
    global TRAINING
    TRAINING = training  # (nn.Dropout is the identity in evaluation mode)
    if training:
        # base.py:241-247 / npf/utils/helpers.py:55-57
        ok = ((X_cntxt >= -1) & (X_cntxt <= 1)).all() and ((X_trgt >= -1) & (X_trgt <= 1)).all()
        if not ok:
            raise ValueError("Features during training should be in [-1,1].")
    Xc_enc = mlp(params, "x_encoder", X_cntxt)
    Xt_enc = mlp(params, "x_encoder", X_trgt)
    R = encode_globally(cfg, params, Xc_enc, Y_cntxt)

    out = {"z_samples": None, "q_zCc": None, "q_zCct": None}
    z_samples = None
    if cfg.is_latent:
        q_zCc = infer_latent_dist(cfg, params, R)
        if cfg.is_q_zCct and Y_trgt is not None:
            R_from_trgt = encode_globally(cfg, params, Xt_enc, Y_trgt)
            q_zCct = infer_latent_dist(cfg, params, R_from_trgt)
            samp = q_zCct
        else:
            q_zCct = None
            samp = q_zCc
        if eps is None:
            raise ValueError("latent models need an explicit eps (the reference draws it from torch's global RNG)")
        assert eps.shape[0] == n_z
        z_samples = samp[0] + eps * samp[1]
        out.update(z_samples=z_samples, q_zCc=q_zCc, q_zCct=q_zCct)
    if cfg.encoded_path == "latent":
        R = None
    R_trgt = trgt_dependent_representation(cfg, params, Xc_enc, z_samples, R, Xt_enc)
    loc, scale = decode(cfg, params, Xt_enc, R_trgt)
    out.update(loc=loc, scale=scale, Xc_enc=Xc_enc, Xt_enc=Xt_enc, R=R, R_trgt=R_trgt)
    return out


# --------------------------------------------------------------------------------------
# losses (npf/losses.py)
# --------------------------------------------------------------------------------------
def _normal_log_prob(loc, scale, value):
    """``torch.distributions.Normal.log_prob`` restated op-for-op (var, log, same
    association) so that results are bit-identical to the reference's
    ``Independent(Normal)`` (npf/utils/helpers.py:125-129)."""
    var = scale ** 2
    log_scale = scale.log()
    return -((value - loc) ** 2) / (2 * var) - log_scale - math.log(math.sqrt(2 * math.pi))


def sum_log_prob(loc, scale, value, batch_ndim: int = 2):
    """``sum_log_prob`` (npf/losses.py:18-24) for ``Independent(Normal(loc, scale), 1)``:
    sum over the event dim then over everything past the first two dims."""
    lp = _normal_log_prob(loc, scale, value).sum(-1)
    return lp.view(*lp.shape[:batch_ndim], -1).sum(-1)


def _kl_normal(p_loc, p_scale, q_loc, q_scale):
    """``kl_divergence(Normal p, Normal q)`` as torch/distributions/kl.py states it."""
    var_ratio = (p_scale / q_scale).pow(2)
    t1 = ((p_loc - q_loc) / q_scale).pow(2)
    return 0.5 * (var_ratio + t1 - 1 - var_ratio.log())


def cnpf_loss(out: dict, Y_trgt: torch.Tensor, reduction: Optional[str] = "mean"):
    """``CNPFLoss`` (npf/losses.py:112-123) + batch reduction (losses.py:71-81)."""
    assert out["q_zCc"] is None
    nll = -sum_log_prob(out["loc"], out["scale"], Y_trgt).squeeze(0)
    return _reduce(nll, reduction)


def elbo_loss(out: dict, Y_trgt: torch.Tensor, reduction: Optional[str] = "mean"):
    """``ELBOLossLNPF`` (npf/losses.py:126-150): needs q_zCct (is_q_zCct=True)."""
    s = sum_log_prob(out["loc"], out["scale"], Y_trgt).mean(0)
    kl = _kl_normal(*out["q_zCct"], *out["q_zCc"]).sum(-1)  # Independent: sum event dim
    kl = kl.view(kl.shape[0], -1).sum(-1)
    return _reduce(-(s - kl), reduction)


def nll_loss(out: dict, Y_trgt: torch.Tensor, reduction: Optional[str] = "mean"):
    """``NLLLossLNPF`` (npf/losses.py:153-203) incl. the importance weights when q_zCct
    is present."""
    s = sum_log_prob(out["loc"], out["scale"], Y_trgt)
    n_z = s.shape[0]
    if out["q_zCct"] is not None:
        z = out["z_samples"]
        s = s + sum_log_prob(*out["q_zCc"], z) - sum_log_prob(*out["q_zCct"], z)
    return _reduce(-(torch.logsumexp(s, 0) - math.log(n_z)), reduction)


def sumo_loss(out: dict, Y_trgt: torch.Tensor, reduction: Optional[str] = "mean", a: int = 5, alpha: int = 85):
    """``SUMOLossLNPF`` (npf/losses.py:207-276) with the default number-of-samples distribution
    ``LightTailPareto(a=5).freeze(85)`` (npf/utils/helpers.py:36-53: P(K >= k) = 1/(k - a) below
    ``alpha - a``, geometric beyond).  ``logcumsumexp`` is the reference's running logsumexp
    (helpers.py:20-33)."""
    s = sum_log_prob(out["loc"], out["scale"], Y_trgt)
    n_z = s.shape[0]
    if out["q_zCct"] is not None:
        z = out["z_samples"]
        s = s + sum_log_prob(*out["q_zCc"], z) - sum_log_prob(*out["q_zCct"], z)
    ks = torch.arange(1, n_z + 1).unsqueeze(-1)
    cum_iwae = torch.cat([torch.logsumexp(s[:i], dim=0, keepdim=True) for i in range(1, n_z + 1)], dim=0) - ks.float().log()
    kk = (ks - 1 + 1 - a).clamp(min=1).double()            # cdf(ks - 1): k -> k + 1 - m, clipped at 1
    al = float(alpha - a)
    tail = torch.where(kk < al, 1.0 / kk, (1.0 / al) * 0.9 ** (kk - al))
    inv_weights = tail                                      # 1 - cdf = P(K >= k)
    sumo = cum_iwae[a - 1] + (inv_weights[a:] * (cum_iwae[a:] - cum_iwae[a - 1:-1])).sum(0)
    return _reduce(-sumo, reduction)


def _reduce(loss, reduction):
    if reduction is None:
        return loss
    if reduction == "mean":
        return loss.mean(0)
    if reduction == "sum":
        return loss.sum(0)
    raise ValueError(f"Unknown {reduction}")


# --------------------------------------------------------------------------------------
# parameter construction (reference initialisation)
# --------------------------------------------------------------------------------------
def mlp_shapes(prefix: str, n_in: int, n_out: int, hidden: int, n_hidden_layers: int, force_smaller=False):
    """Layer shapes of ``MLP.__init__`` incl. the hidden-size clamp
    (npf/architectures/mlp.py:64-79)."""
    if force_smaller and hidden > max(n_out, n_in):
        hidden = max(n_out, n_in)
    elif hidden < min(n_out, n_in):
        hidden = min(n_out, n_in)
    shapes = [(f"{prefix}.to_hidden", hidden, n_in)]
    for i in range(n_hidden_layers - 1):
        shapes.append((f"{prefix}.linears.{i}", hidden, hidden))
    shapes.append((f"{prefix}.out", n_out, hidden))
    return shapes


def model_shapes(cfg: OracleConfig, n_layers_xy: int = 2, n_layers_dec: int = 4):
    """(name, out_features, in_features) for every Linear of the stock model, in the
    reference's state_dict order (base.py:143-146,157-175; np.py:62-82; base.py:447-458)."""
    r, dx, dy, xt = cfg.r_dim, cfg.x_dim, cfg.y_dim, cfg.x_transf_dim
    shapes = mlp_shapes("x_encoder", dx, xt, r, 1)
    shapes += mlp_shapes("decoder.resizer", r, xt, 32, 1)
    shapes += mlp_shapes("decoder.flat_module", xt, 2 * dy, r, n_layers_dec)
    if cfg.is_sum_merge:
        shapes += mlp_shapes("xy_encoder.resizer", dy, xt, 32, 1)
        shapes += mlp_shapes("xy_encoder.flat_module", xt, r, r, n_layers_xy, force_smaller=True)
    else:  # encoders.py:163-173: no resizer, the flat module takes cat(x, y)
        shapes += mlp_shapes("xy_encoder.flat_module", xt + dy, r, r, n_layers_xy, force_smaller=True)
    if cfg.is_attentive and cfg.attention == "transformer":
        shapes += mlp_shapes("attender.mlp", r, r, r, 1)  # attention.py:556-561
    if cfg.is_latent:
        shapes += mlp_shapes("latent_encoder", r, 2 * cfg.z_dim, r, 1)
        if cfg.encoded_path == "both":
            shapes.append(("r_z_merger", r, r + cfg.z_dim))
        if cfg.z_dim != r and cfg.encoded_path == "latent":
            shapes.append(("reshaper_z", r, cfg.z_dim))
    return shapes


def init_params(cfg: OracleConfig, seed: int = 0, n_layers_xy: int = 2, n_layers_dec: int = 4) -> Params:
    """Deterministic parameters with the reference's *effective* init statistics
    (npf/utils/initialization.py:7-94, behaviour documented in SURVEY.md 8a row 12):
    hidden Linear weights U(+-1/sqrt(fan_in)), ``.out`` weights U(+-sqrt(6/fan_in)), MLP
    biases 0, ``r_z_merger``/``reshaper_z`` torch-default incl. non-zero bias.  The draw
    itself is this project's own (numpy Philox stream), so the same dict can be
    regenerated on the GPU box without the reference."""
    import numpy as np

    rng = np.random.Generator(np.random.Philox(seed))
    params: Params = {}
    for name, n_out, n_in in model_shapes(cfg, n_layers_xy, n_layers_dec):
        if name.endswith(".out"):
            bound = math.sqrt(6.0 / n_in)
        else:
            bound = 1.0 / math.sqrt(n_in)
        w = rng.uniform(-bound, bound, size=(n_out, n_in)).astype("float32")
        if name in ("r_z_merger", "reshaper_z"):
            b = rng.uniform(-bound, bound, size=(n_out,)).astype("float32")
        else:
            b = np.zeros((n_out,), dtype="float32")
        params[f"{name}.weight"] = torch.from_numpy(w)
        params[f"{name}.bias"] = torch.from_numpy(b)
    if cfg.is_attentive and cfg.attention != "scaledot":
        # attention.py:444-455: normal(0, sqrt(2 / (size + head_size))) for the three projections;
        # LayerNorm weights are drawn away from 1 here so that parity tests exercise them
        r = cfg.r_dim
        std = math.sqrt(2.0 / (r + r // cfg.n_heads))
        f32 = lambda a: torch.from_numpy(a.astype("float32"))  # noqa: E731
        params["attender.key_transform.weight"] = f32(rng.normal(0.0, std, size=(r, r)))
        params["attender.query_transform.weight"] = f32(rng.normal(0.0, std, size=(r, r)))
        params["attender.query_transform.bias"] = f32(np.zeros((r,)))
        params["attender.value_transform.weight"] = f32(rng.normal(0.0, std, size=(r, r)))
        if cfg.attention == "multihead":
            bound = 1.0 / math.sqrt(r)
            params["attender.post_processor.weight"] = f32(rng.uniform(-bound, bound, size=(r, r)))
            params["attender.post_processor.bias"] = f32(rng.uniform(-bound, bound, size=(r,)))
        else:
            for ln in ("layer_norm1", "layer_norm2"):
                params[f"attender.{ln}.weight"] = f32(rng.uniform(0.5, 1.5, size=(r,)))
                params[f"attender.{ln}.bias"] = f32(np.zeros((r,)))
    return params


def perturb_biases(params: Params, seed: int = 1, scale: float = 0.05) -> Params:
    """Non-zero biases for parity tests (the reference zero-inits them, which would hide
    a bias-indexing bug)."""
    import numpy as np

    rng = np.random.Generator(np.random.Philox(seed))
    out = dict(params)
    for k, v in params.items():
        if k.endswith(".bias"):
            out[k] = torch.from_numpy(rng.uniform(-scale, scale, size=tuple(v.shape)).astype("float32"))
    return out
