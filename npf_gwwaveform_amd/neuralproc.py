"""Members of the neural-process family with the reference's class interface and HIP
execution: ``CNP``, ``LNP``, ``AttnCNP``, ``AttnLNP``.

Mirrors (constructor kwargs, method names, return values, ``state_dict`` keys):
  * ``NeuralProcessFamily`` / ``LatentNeuralProcessFamily``   npf/neuralproc/base.py:23-575
  * ``CNP`` / ``LNP``                                         npf/neuralproc/np.py:19-163
  * ``AttnCNP`` / ``AttnLNP``                                 npf/neuralproc/attnnp.py:27-202

``forward`` runs the whole path as a handful of fused chain-kernel launches on PT32 tensors
(x-encoder, xy-encoder, [attention + merge +] decoder, Gaussian head); the stage methods
(``encode_globally``, ``trgt_dependent_representation``, ``latent_path``, ``decode``) keep
the reference's row-major signatures and run the same kernels stage by stage.
The reference's building blocks are supported (MLP x-encoder incl. ``is_res``, sum- or concatenating-merge
MLP xy-encoder / decoder, ``x_transf_dim`` != ``r_dim``, ``attention`` = scaledot / multihead / transformer);
anything else raises ``NotImplementedError`` -- there is no fallback path.
"""
from __future__ import annotations

import abc
import math
from functools import partial
from typing import Optional

import torch
import torch.nn as nn
from torch.distributions import Independent, Normal

from . import functional as FN
from .architectures import (MLP, DotAttender, MergeFlatInputs, MultiheadAttender, SelfAttention, get_attender,
                            merge_flat_input)
from .chain import Chain, PTensor, pad32, pt_shape

__all__ = ["NeuralProcessFamily", "LatentNeuralProcessFamily", "CNP", "LNP", "AttnCNP", "AttnLNP",
           "MultivariateNormalDiag", "HeadDistribution"]


def MultivariateNormalDiag(loc, scale_diag):
    """npf/utils/helpers.py:125-129 (argument validation off: it would sync the stream)."""
    if loc.dim() < 1:
        raise ValueError("loc must be at least one-dimensional.")
    return Independent(Normal(loc, scale_diag, validate_args=False), 1)


class HeadDistribution(Independent):
    """``MultivariateNormalDiag(loc, scale)`` of the decoder's raw output (base.py:350-367) -- an
    ``Independent(Normal(loc, scale), 1)`` with batch shape [n_z, B, T] and event shape [y_dim] -- evaluated lazily:
    ``loc`` / ``scale`` are only materialised ([n_z, B, T, y_dim] each, one ``npf_gauss_head_fwd`` launch) when
    ``base_dist`` (or anything that needs it: ``mean``, ``log_prob``, ``sample`` ...) is first touched.  The
    training and evaluation objectives never do: :meth:`sum_log_prob` is a loss-only launch that writes one float
    per (z-sample, task)."""

    def __init__(self, suff, y_dim, homoskedastic, n_z, B, T):
        torch.distributions.Distribution.__init__(self, torch.Size((n_z, B, T)), torch.Size((y_dim,)), validate_args=False)
        self.reinterpreted_batch_ndims = 1
        self._suff, self._y_dim, self._homosk = suff, y_dim, homoskedastic
        self._base = None
        self._slp = None  # (targets, [n_z, B] sum of log-probabilities) of the last sum_log_prob call

    @property
    def base_dist(self):
        if self._base is None:
            n_z, B, T = self.batch_shape
            loc, scale, _ = FN.gauss_head(self._suff, None, self._y_dim, self._homosk)
            self._base = Normal(loc.view(n_z, B, T, self._y_dim), scale.view(n_z, B, T, self._y_dim), validate_args=False)
        return self._base

    def sum_log_prob(self, Y_trgt):
        """sum_t log p(y_t) -> [n_z, B] (npf/losses.py:18-24), fused with the head in one loss-only launch."""
        if self._slp is None or self._slp[0] is not Y_trgt:
            n_z, B, _ = self.batch_shape
            _, _, slp = FN.gauss_head(self._suff, Y_trgt.contiguous(), self._y_dim, self._homosk, want_dist=False)
            self._slp = (Y_trgt, slp.view(n_z, B))
        return self._slp[1]


def _q_z_scale(z_scale):
    return 0.1 + 0.9 * torch.sigmoid(z_scale)


class NeuralProcessFamily(nn.Module, abc.ABC):
    """Base class (npf/neuralproc/base.py:23-371)."""

    _valid_paths = ["deterministic", "latent", "both"]

    def __init__(self, x_dim, y_dim, encoded_path, r_dim=128, x_transf_dim=-1, is_heteroskedastic=True, XEncoder=None,
                 Decoder=None, PredictiveDistribution=MultivariateNormalDiag, p_y_loc_transformer=None,
                 p_y_scale_transformer=None):
        super().__init__()
        self.x_dim, self.y_dim, self.r_dim = x_dim, y_dim, r_dim
        self.is_heteroskedastic = is_heteroskedastic
        if x_transf_dim is None:
            self.x_transf_dim = self.x_dim
        elif x_transf_dim == -1:
            self.x_transf_dim = self.r_dim
        else:
            self.x_transf_dim = x_transf_dim
        self.encoded_path = encoded_path.lower()
        if self.encoded_path not in self._valid_paths:
            raise ValueError(f"Unknown encoded_path={self.encoded_path}.")
        if PredictiveDistribution is not MultivariateNormalDiag or p_y_loc_transformer is not None \
                or p_y_scale_transformer is not None:
            raise NotImplementedError("the HIP Gaussian head implements the reference defaults only: diagonal "
                                      "Gaussian, identity loc, scale = 0.01 + 0.99 softplus (base.py:114-116)")
        if XEncoder is None:
            XEncoder = self.dflt_Modules["XEncoder"]
        if Decoder is None:
            Decoder = self.dflt_Modules["Decoder"]
        self.x_encoder = XEncoder(self.x_dim, self.x_transf_dim)
        self.decoder = Decoder(self.x_transf_dim, self.r_dim, self.y_dim * 2)
        if not isinstance(self.x_encoder, MLP) or not isinstance(self.decoder, MergeFlatInputs):
            raise NotImplementedError("the HIP path needs the stock MLP XEncoder and merge_flat_input(MLP) Decoder")
        if max(self.x_dim, 2 * self.y_dim) > 32:
            raise NotImplementedError("x_dim and 2*y_dim are limited to 32 on the hot path")
        self.PredictiveDistribution = PredictiveDistribution
        self.p_y_loc_transformer = nn.Identity()
        self.validate_inputs = True

    def reset_parameters(self):  # no-op in the reference as well
        pass

    @property
    def dflt_Modules(self):
        d = dict()
        d["XEncoder"] = partial(MLP, n_hidden_layers=1, hidden_size=self.r_dim)
        d["SubDecoder"] = partial(MLP, n_hidden_layers=4, hidden_size=self.r_dim)
        d["Decoder"] = merge_flat_input(d["SubDecoder"], is_sum_merge=True)
        return d

    # ------------------------------------------------------------------ forward
    def forward(self, X_cntxt, Y_cntxt, X_trgt, Y_trgt=None):
        """Same contract as base.py:177-239: returns ``(p_yCc, z_samples, q_zCc, q_zCct)``."""
        self._validate_inputs(X_cntxt, Y_cntxt, X_trgt, Y_trgt)
        B, C, _ = X_cntxt.shape
        T = X_trgt.shape[1]
        if T == 0:
            raise ValueError("no target points")
        from . import chain as _chain

        fused_t = self._fused_target_side(C, T)
        fused_c = self._fused_context_side(C)
        if fused_c and self._attentive and not fused_t and _chain.COMPUTE_DTYPE == "bf16":
            fused_c = False  # (the bf16 attention chain streams the bf16 images of keys / values its producer chains store)
        if fused_c:
            # x-encoder + XY-encoder of the context points as one x6 program (x6.py); the pooling stays the subclass's
            from . import x6

            Xc_pt, R_pts = x6.context_side(self, X_cntxt, Y_cntxt)
            R = self._pool_pt(R_pts, B)
        else:
            Xc_pt = self._xenc_pt(X_cntxt, with_tr=self._attentive and not fused_t) if C > 0 else None
            R = self._encode_globally_pt(Xc_pt, Y_cntxt, B, C)
        if fused_t:
            Xt_pt, self._X_trgt_raw = None, X_trgt  # (the target side runs as one x6 program from the raw features)
        elif self._xenc_with_query_projection(C, T):
            # multihead / transformer attention with 16-wide heads: the target x-encoder and the attender's query projection as
            # one x6 program (x6.xenc_proj)
            from . import x6

            Xt_pt, q_proj = x6.xenc_proj(self, X_trgt, self.attender.query_transform)
            Xt_pt.proj = q_proj
        else:
            Xt_pt = self._xenc_pt(X_trgt)
        if self.encoded_path in ["latent", "both"]:
            z_samples, q_zCc, q_zCct = self._latent_path_pt(R, C, Xt_pt, Y_trgt, B, T)
        else:
            z_samples, q_zCc, q_zCct = None, None, None
        if self.encoded_path == "latent":
            R = None
        suff = self._target_suffstat(Xc_pt, z_samples, R, Xt_pt, B, C, T)  # [n_z * B, T, 2 dy]
        p_yCc = self._head(suff, Y_trgt, B, T)
        return p_yCc, z_samples, q_zCc, q_zCct

    def _validate_inputs(self, X_cntxt, Y_cntxt, X_trgt, Y_trgt):
        """base.py:241-247: features must be in [-1, 1] during training."""
        for t in (X_cntxt, Y_cntxt, X_trgt, Y_trgt):
            if t is not None and (not t.is_cuda or t.dtype != torch.float32):
                raise RuntimeError("the HIP path takes fp32 device tensors only; there is no CPU fallback")
        if self.training and self.validate_inputs:
            # one device reduction and ONE host sync per step (every sync drains the stream and
            # costs a launch bubble); NaNs fail the test like the reference's (x>=-1)&(x<=1)
            m = X_trgt.abs().amax() if X_trgt.numel() else X_trgt.new_zeros(())
            if X_cntxt.numel():
                m = torch.maximum(m, X_cntxt.abs().amax())
            if self.validate_inputs == "deferred":
                # (a step replayed from a HIP graph cannot stop for the host: the reduction stays in the step, the
                # running maximum -- NaN-propagating -- is read by ``check_deferred_inputs`` at the next sync point)
                if getattr(self, "_range_seen", None) is None or self._range_seen.device != m.device:
                    self._range_seen = torch.zeros((), device=m.device)
                self._range_seen.copy_(torch.maximum(self._range_seen, m))
                return
            if not (m.item() <= 1.0):
                lo = min(X_cntxt.min().item() if X_cntxt.numel() else 0.0, X_trgt.min().item())
                hi = max(X_cntxt.max().item() if X_cntxt.numel() else 0.0, X_trgt.max().item())
                raise ValueError(f"Features during training should be in [-1,1]. Got [{lo}, {hi}].")

    def check_deferred_inputs(self):
        """``validate_inputs == "deferred"``: raise the reference's ValueError (base.py:244-247) if any training batch
        since the last call had features outside [-1, 1] (one host sync, resets the record)."""
        seen = getattr(self, "_range_seen", None)
        if seen is None:
            return
        worst = seen.item()
        seen.zero_()
        if not (worst <= 1.0):
            raise ValueError(f"Features during training should be in [-1,1]. Got max |x| = {worst}.")

    # ------------------------------------------------------------------ PT-level stages
    _attentive = False  # attentive subclasses also keep feature-major copies of keys / values

    def _fused_target_side(self, C, T) -> bool:
        """Does ``forward`` hand the whole target side (x-encoder, attention, decoder) to one x6 program (x6.py)."""
        return False

    def _xenc_with_query_projection(self, C, T) -> bool:
        """Does ``forward`` hand the target x-encoder + the attender's query projection to one x6 program (x6.xenc_proj)."""
        from . import x6

        att = getattr(self, "attender", None)
        if not (self._attentive and isinstance(att, MultiheadAttender) and C > 0):
            return False
        return FN.mha_usable(att.kq_head_size, att.value_head_size, C) and x6.xenc_proj_usable(self, att.query_transform, T)

    def _fused_context_side(self, C) -> bool:
        """Does ``forward`` hand x-encoder + XY-encoder of the context points to one x6 program (x6.py)."""
        from . import x6

        return hasattr(self, "xy_encoder") and hasattr(self, "_pool_pt") and x6.context_side_usable(self, C)

    def _xenc_pt(self, X, with_tr=False) -> PTensor:
        B, P, dx = X.shape
        ch = Chain(B, P, X.device)
        ch.input_rows(X.contiguous(), dx)
        return self.x_encoder.append_to(ch).run_pt(as_weights=with_tr)

    def _xyenc_pt(self, X_enc: PTensor, Y) -> PTensor:
        """Per-point XY encoding (the per-point part of encode_globally) -> [B, P, r]."""
        ch = Chain(X_enc.n_tasks, X_enc.pts, Y.device)
        ch.input_rows(Y.contiguous(), self.y_dim)
        return self.xy_encoder.run_pt(ch, X_enc.t, X_enc.n_tasks, X_enc.pts, with_tr=self._attentive)

    def _head(self, suff, Y_trgt, B, T):
        return HeadDistribution(suff, self.y_dim, not self.is_heteroskedastic, suff.shape[0] // B, B, T)

    def _decode_taskvec(self, Xt_pt, vec, B, T, n_rows):
        """decoder(X_trgt_enc, R_trgt) when R_trgt is one vector per (z-sample, task)
        (np.py:107-110,161: the reference expands it over the targets and recomputes the
        resizer per target; here the resizer runs once per task)."""
        ch = Chain(n_rows, T, Xt_pt.t.device)
        ch.input_pt(Xt_pt.t, self.x_transf_dim, modulus=(B if n_rows != B else 0))
        return self.decoder.finish_rows(ch, taskvec=vec)

    # ------------------------------------------------------------------ reference stage API (row-major)
    @abc.abstractmethod
    def encode_globally(self, X_cntxt, Y_cntxt):
        pass

    @abc.abstractmethod
    def trgt_dependent_representation(self, X_cntxt, z_samples, R, X_trgt):
        pass

    def latent_path(self, X_cntxt, R, X_trgt, Y_trgt):
        raise NotImplementedError(
            f"`latent_path` not implemented. Cannot use encoded_path={self.encoded_path} in such case.")

    def decode(self, X_trgt, R_trgt):
        """base.py:327-367 with row-major ``X_trgt`` [B,T,x_transf] and ``R_trgt`` [n_z,B,T,r]."""
        n_z, B, T, _ = R_trgt.shape
        suff = self.decoder(X_trgt, R_trgt).reshape(n_z * B, T, 2 * self.y_dim)
        return self._head(suff, None, B, T)

    def set_extrapolation(self, min_max):
        pass


class LatentNeuralProcessFamily(NeuralProcessFamily):
    """Latent (sub-)family (npf/neuralproc/base.py:374-575)."""

    _valid_paths = ["latent", "both"]

    def __init__(self, *args, is_q_zCct=False, n_z_samples_train=32, n_z_samples_test=32, LatentEncoder=None,
                 LatentDistribution=MultivariateNormalDiag, q_z_loc_transformer=None, q_z_scale_transformer=None,
                 z_dim=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.is_q_zCct = is_q_zCct
        self.n_z_samples_train, self.n_z_samples_test = n_z_samples_train, n_z_samples_test
        self.z_dim = self.r_dim if z_dim is None else z_dim
        if LatentEncoder is None:
            LatentEncoder = self.dflt_Modules["LatentEncoder"]
        self.latent_encoder = LatentEncoder(self.r_dim, self.z_dim * 2)
        if self.encoded_path == "both":
            self.r_z_merger = nn.Linear(self.r_dim + self.z_dim, self.r_dim)
        self.LatentDistribution = LatentDistribution
        self.q_z_loc_transformer = nn.Identity() if q_z_loc_transformer is None else q_z_loc_transformer
        self.q_z_scale_transformer = _q_z_scale if q_z_scale_transformer is None else q_z_scale_transformer
        if self.z_dim != self.r_dim and self.encoded_path == "latent":
            self.reshaper_z = nn.Linear(self.z_dim, self.r_dim)

    @property
    def dflt_Modules(self):
        d = NeuralProcessFamily.dflt_Modules.__get__(self)
        d["LatentEncoder"] = partial(MLP, n_hidden_layers=1, hidden_size=self.r_dim)
        return d

    def forward(self, *args, **kwargs):
        try:  # scipy random variable = random number of samples (base.py:478-486)
            self.n_z_samples = self.n_z_samples_train.rvs() if self.training else self.n_z_samples_test.rvs()
        except AttributeError:
            self.n_z_samples = self.n_z_samples_train if self.training else self.n_z_samples_test
        return super().forward(*args, **kwargs)

    def infer_latent_dist(self, X, R):
        """base.py:516-547 on row-major R."""
        R_lat_inp = self.rep_to_lat_input(R)
        return self._latent_dist_from(R_lat_inp)

    def _latent_dist_from(self, R_lat_inp):
        suff = self.latent_encoder(R_lat_inp)
        q_z_loc, q_z_scale = suff.split(self.z_dim, dim=-1)
        return self.LatentDistribution(self.q_z_loc_transformer(q_z_loc), self.q_z_scale_transformer(q_z_scale))

    def rep_to_lat_input(self, R):
        return R

    def latent_path(self, X_cntxt, R, X_trgt, Y_trgt):
        """base.py:495-514 (row-major stage API)."""
        q_zCc = self.infer_latent_dist(X_cntxt, R)
        if self.is_q_zCct and Y_trgt is not None:
            R_from_trgt = self.encode_globally(X_trgt, Y_trgt)
            q_zCct = self.infer_latent_dist(X_trgt, R_from_trgt)
            sampling_dist = q_zCct
        else:
            q_zCct, sampling_dist = None, q_zCc
        return sampling_dist.rsample([self.n_z_samples]), q_zCc, q_zCct

    def _latent_path_pt(self, R, C, Xt_pt, Y_trgt, B, T):
        q_zCc = self._latent_dist_from(self._lat_input(R, B))
        if self.is_q_zCct and Y_trgt is not None:
            if Xt_pt is None:
                # (forward left the target side to one x6 program, which encodes the targets itself: the target-side latent
                # encode of base.py:501-506 is the context-side program over the target points, or the chain launches)
                if self._fused_context_side(T):
                    from . import x6

                    R_t = self._pool_pt(x6.context_side(self, self._X_trgt_raw, Y_trgt)[1], B)
                else:
                    R_t = self._encode_globally_pt(self._xenc_pt(self._X_trgt_raw), Y_trgt, B, T)
            else:
                R_t = self._encode_globally_pt(Xt_pt, Y_trgt, B, T)
            q_zCct = self._latent_dist_from(self._lat_input(R_t, B))
            sampling_dist = q_zCct
        else:
            q_zCct, sampling_dist = None, q_zCc
        return sampling_dist.rsample([self.n_z_samples]), q_zCc, q_zCct

    def merge_r_z(self, R, z_samples):
        """base.py:554-575 on row-major tensors: relu(Linear(cat(R, z)))."""
        if R.shape != z_samples.shape:
            R = R.unsqueeze(0).expand(*z_samples.shape[:-1], self.r_dim)
        lead = z_samples.shape[:-1]
        rows = int(math.prod(lead))
        out = self._merge_rows(R.reshape(rows, self.r_dim), z_samples.reshape(rows, self.z_dim))
        return out.reshape(*lead, self.r_dim)

    def _merge_rows(self, R_rows, z_rows):
        """relu(W_R R + (W_z z + b)) on [rows, .] tensors; the concatenation of the reference
        is never materialised (two accumulating layers)."""
        rows = R_rows.shape[0]
        W, b, r = self.r_z_merger.weight, self.r_z_merger.bias, self.r_dim
        ch = Chain(1, rows, R_rows.device)
        ch.input_pt(FN.pack_pt(z_rows.reshape(1, rows, self.z_dim)), self.z_dim).linear(W[:, r:], b).output_pt()
        (zb,) = ch.run()
        ch = Chain(1, rows, R_rows.device)
        ch.input_pt(FN.pack_pt(R_rows.reshape(1, rows, r)), r).linear(W[:, :r], None, relu=True, addend=zb).output_pt()
        return FN.unpack_pt(ch.run()[0], rows, r).reshape(rows, r)


class CNP(NeuralProcessFamily):
    """Conditional neural process: mean aggregation (npf/neuralproc/np.py:19-110)."""

    _valid_paths = ["deterministic"]

    def __init__(self, x_dim, y_dim, XYEncoder=None, **kwargs):
        kwargs["encoded_path"] = kwargs.get("encoded_path", "deterministic")
        super().__init__(x_dim, y_dim, **kwargs)
        if XYEncoder is None:
            XYEncoder = self.dflt_Modules["XYEncoder"]
        self.xy_encoder = XYEncoder(self.x_transf_dim, self.y_dim, self.r_dim)
        if not isinstance(self.xy_encoder, MergeFlatInputs):
            raise NotImplementedError("the HIP path needs the stock merge_flat_input(MLP) XYEncoder")

    @property
    def dflt_Modules(self):
        d = NeuralProcessFamily.dflt_Modules.__get__(self)
        sub = partial(MLP, n_hidden_layers=2, is_force_hid_smaller=True, hidden_size=self.r_dim)
        d["XYEncoder"] = merge_flat_input(sub, is_sum_merge=True)
        return d

    # reference stage API
    def encode_globally(self, X_cntxt, Y_cntxt):
        B, C, _ = X_cntxt.shape
        if C == 0:
            return torch.zeros(B, 1, self.r_dim, device=X_cntxt.device)
        return self._encode_globally_pt(PTensor(FN.pack_pt(X_cntxt), C, self.x_transf_dim), Y_cntxt, B, C)

    def trgt_dependent_representation(self, _, __, R, X_trgt):
        B, T, _ = X_trgt.shape
        return R.expand(B, T, self.r_dim).unsqueeze(0)

    # fused path
    def _encode_globally_pt(self, X_enc, Y, B, P):
        """-> row-major R [B, 1, r] (np.py:86-101)."""
        if P == 0:
            return torch.zeros(B, 1, self.r_dim, device=Y.device)
        return self._pool_pt(self._xyenc_pt(X_enc, Y), B)

    def _pool_pt(self, R_pts: PTensor, B):
        """np.py:95: the mean over the context points of the per-point representations -> row-major [B, 1, r]."""
        return FN.mean_agg(R_pts.t, R_pts.pts, self.r_dim)[:, : self.r_dim].reshape(B, 1, self.r_dim)

    def _target_suffstat(self, Xc_pt, z_samples, R, Xt_pt, B, C, T):
        return self._decode_taskvec(Xt_pt, R.reshape(B, self.r_dim), B, T, B)


class LNP(LatentNeuralProcessFamily, CNP):
    """(Latent) neural process (npf/neuralproc/np.py:113-163)."""

    _valid_paths = ["latent", "both"]

    def __init__(self, x_dim, y_dim, encoded_path="latent", **kwargs):
        super().__init__(x_dim, y_dim, encoded_path=encoded_path, **kwargs)

    def _lat_input(self, R, B):
        return R

    def _rep_rows(self, z_samples, R, B):
        n_z = z_samples.size(0)
        if self.encoded_path == "both":
            R_trgt = self.merge_r_z(R, z_samples)
        else:
            R_trgt = z_samples
            if self.z_dim != self.r_dim:
                R_trgt = _rows_mlp_linear(self.reshaper_z, R_trgt)
        return R_trgt.reshape(n_z * B, self.r_dim)

    def trgt_dependent_representation(self, _, z_samples, R, X_trgt):
        B, T, _ = X_trgt.shape
        n_z = z_samples.size(0)
        return self._rep_rows(z_samples, R, B).reshape(n_z, B, 1, self.r_dim).expand(n_z, B, T, self.r_dim)

    def _target_suffstat(self, Xc_pt, z_samples, R, Xt_pt, B, C, T):
        n_z = z_samples.size(0)
        return self._decode_taskvec(Xt_pt, self._rep_rows(z_samples, R, B), B, T, n_z * B)


def _rows_mlp_linear(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    lead = x.shape[:-1]
    rows = int(math.prod(lead))
    ch = Chain(1, rows, x.device)
    ch.input_pt(FN.pack_pt(x.reshape(1, rows, x.shape[-1])), x.shape[-1]).linear(lin.weight, lin.bias).output_pt()
    return FN.unpack_pt(ch.run()[0], rows, lin.out_features).reshape(*lead, lin.out_features)


class AttnCNP(NeuralProcessFamily):
    """Attentive conditional neural process (npf/neuralproc/attnnp.py:27-131)."""

    _valid_paths = ["deterministic"]
    _attentive = True

    def __init__(self, x_dim, y_dim, XYEncoder=None, attention="scaledot", attention_kwargs={},
                 self_attention_kwargs={}, is_self_attn=False, **kwargs):
        kwargs["encoded_path"] = kwargs.get("encoded_path", "deterministic")
        super().__init__(x_dim, y_dim, **kwargs)
        self.is_self_attn = is_self_attn
        if is_self_attn:
            # attnnp.py:88-91: relu(x + resizer(y)) followed by self-attention layers over the context
            XYEncoder = merge_flat_input(SelfAttention, is_sum_merge=True, **self_attention_kwargs)
        elif XYEncoder is None:
            XYEncoder = self.dflt_Modules["XYEncoder"]
        self.xy_encoder = XYEncoder(self.x_transf_dim, self.y_dim, self.r_dim)
        if not isinstance(self.xy_encoder, MergeFlatInputs):
            raise NotImplementedError("the HIP path needs the stock merge_flat_input(MLP) XYEncoder")
        self.attender = get_attender(attention, self.x_transf_dim, self.r_dim, self.r_dim, **attention_kwargs)
        if not isinstance(self.attender, (DotAttender, MultiheadAttender)):
            raise NotImplementedError("the HIP path implements attention = 'scaledot', 'multihead', 'transformer'")

    dflt_Modules = CNP.dflt_Modules

    def _fused_target_side(self, C, T) -> bool:
        from . import x6

        return type(self) is AttnCNP and self.encoded_path == "deterministic" and x6.target_side_usable(self, C, T)

    # reference stage API
    def encode_globally(self, X_cntxt, Y_cntxt):
        B, C, _ = X_cntxt.shape
        if C == 0:
            return torch.zeros(B, 0, self.r_dim, device=X_cntxt.device)
        return self.xy_encoder(X_cntxt, Y_cntxt)

    def trgt_dependent_representation(self, X_cntxt, _, R, X_trgt):
        B, C, _ = X_cntxt.shape
        if C == 0:
            R_trgt = torch.zeros(B, X_trgt.size(1), self.r_dim, device=R.device)
        else:
            R_trgt = self.attender(X_cntxt, X_trgt, R)
        return R_trgt.unsqueeze(0)

    # fused path
    def _encode_globally_pt(self, X_enc, Y, B, P):
        """-> R_cntxt [B, P, r] as a PTensor (attnnp.py:105-116); None when there is no context."""
        if P == 0:
            return None
        return self._xyenc_pt(X_enc, Y)

    def _pool_pt(self, R_pts: PTensor, B):
        """attnnp.py:105-116: no pooling, one representation per context point."""
        return R_pts

    def _attend_into(self, ch, Xc_pt, R, Xt_pt, C, T, tap_x1: bool = False):
        """cur of ``ch`` <- attention of the targets over the context (attnnp.py:118-131): fused into
        the chain while a score row fits the registers, blocked (attention_long.py) beyond that."""
        from . import chain as _chain

        if not isinstance(self.attender, DotAttender):  # learned projections: its own launches
            ch.input_pt(self.attender.attend_pt(Xt_pt.t, Xc_pt.t, R.t, C, T, queries_proj=Xt_pt.proj), self.r_dim)
        elif _chain.COMPUTE_DTYPE == "bf16":
            if Xc_pt.img is not None and R.img is not None and self.attender.fits_fused(C):
                # bf16 compute mode with bf16 images of keys / values: attention and decoder in one bf16 chain
                ch.input_pt(Xt_pt.t, self.x_transf_dim)
                self.attender.append_to(ch, Xc_pt.t, R.t, C, keys_tr=Xc_pt.tr, values_tr=R.tr, keys_img=Xc_pt.img,
                                        values_img=R.img)
            else:  # the attention keeps an fp32 launch, the decoder chain behind it is bf16
                ch.input_pt(self.attender.attend_pt(Xt_pt.t, Xc_pt.t, R.t, C, T, keys_tr=Xc_pt.tr, values_tr=R.tr), self.r_dim)
        elif self.attender.fits_fused(C):
            ch.input_pt(Xt_pt.t, self.x_transf_dim)
            if tap_x1:
                # the encoded targets are the attention's queries here and the decoder's x1 on the split kernel next: handed on
                # through this chain, both gradients meet in its dgrad launch (MergeFlatInputs.finish_rows)
                ch.tap(alias_input=True)
                ch.x1_tapped = True
            self.attender.append_to(ch, Xc_pt.t, R.t, C, keys_tr=Xc_pt.tr, values_tr=R.tr)
        else:
            ch.input_pt(self.attender.attend_pt(Xt_pt.t, Xc_pt.t, R.t, C, T, keys_tr=Xc_pt.tr, values_tr=R.tr), self.r_dim)
        return ch

    def _target_suffstat(self, Xc_pt, z_samples, R, Xt_pt, B, C, T):
        if Xt_pt is None:  # (forward left the target side to the fused x6 program: x-encoder, attention, decoder in one launch)
            from . import x6

            return x6.target_side(self, self._X_trgt_raw, Xc_pt, R)
        if C > 0 and not isinstance(self.attender, DotAttender):
            from . import x6

            if x6.decoder_side_usable(self, T):
                # learned projections (multihead / transformer attention): its own launches, then the decoder as one x6 program
                return x6.decoder_side(self, self.attender.attend_pt(Xt_pt.t, Xc_pt.t, R.t, C, T, queries_proj=Xt_pt.proj), Xt_pt.t, T)
        ch = Chain(B, T, Xt_pt.t.device, wg_per_task=True)
        if C == 0:
            ch.input_pt(torch.zeros(pt_shape(B, T, self.r_dim), device=Xt_pt.t.device), self.r_dim)
        else:
            self._attend_into(ch, Xc_pt, R, Xt_pt, C, T, tap_x1=self.decoder.x6_resizer_ok())
        return self.decoder.finish_rows(ch, x1_pt=Xt_pt.t)


class AttnLNP(LatentNeuralProcessFamily, AttnCNP):
    """Attentive latent neural process: deterministic attention path + mean-pooled latent
    path, merged per target (npf/neuralproc/attnnp.py:134-202)."""

    _valid_paths = ["both"]

    def __init__(self, x_dim, y_dim, **kwargs):
        super().__init__(x_dim, y_dim, encoded_path="both", **kwargs)

    @property
    def dflt_Modules(self):
        d = AttnCNP.dflt_Modules.__get__(self)
        d.update(LatentNeuralProcessFamily.dflt_Modules.__get__(self))
        return d

    def rep_to_lat_input(self, R):
        B, C, _ = R.shape
        if C == 0:
            return torch.zeros(B, 1, self.r_dim, device=R.device)
        return FN.mean_agg(FN.pack_pt(R), C, self.r_dim)[:, : self.r_dim].reshape(B, 1, self.r_dim)

    def _lat_input(self, R: Optional[PTensor], B):
        """attnnp.py:172-181: the latent path pools the per-point representation (its own point count)."""
        if R is None:
            return torch.zeros(B, 1, self.r_dim, device=self.r_z_merger.weight.device)
        return FN.mean_agg(R.t, R.pts, self.r_dim)[:, : self.r_dim].reshape(B, 1, self.r_dim)

    def trgt_dependent_representation(self, X_cntxt, z_samples, R, X_trgt):
        B, T, _ = X_trgt.shape
        n_z = z_samples.size(0)
        z = z_samples.expand(n_z, B, T, self.z_dim)
        R_det = AttnCNP.trgt_dependent_representation(self, X_cntxt, None, R, X_trgt).squeeze(0)
        return self.merge_r_z(R_det, z)

    def _fused_target_side(self, C, T) -> bool:
        """One latent sample, scaled-dot attention over <= 256 context points, 256-wide layers: the whole target side --
        x-encoder, attention, merge_r_z, decoder -- is one x6 program (x6.target_side with the latent merge)."""
        from . import x6

        return (type(self) is AttnLNP and self.n_z_samples == 1 and self.z_dim == self.r_dim
                and x6.target_side_usable(self, C, T, latent_merge=True))

    def _target_suffstat(self, Xc_pt, z_samples, R, Xt_pt, B, C, T):
        n_z = z_samples.size(0)
        dev = z_samples.device
        W, b, r = self.r_z_merger.weight, self.r_z_merger.bias, self.r_dim
        # the latent half of merge_r_z is constant per (z-sample, task): a per-task bias
        rows = n_z * B
        chz = Chain(1, rows, dev)
        chz.input_pt(FN.pack_pt(z_samples.reshape(1, rows, self.z_dim)), self.z_dim).linear(W[:, r:], b).output_pt()
        zb = FN.unpack_pt(chz.run()[0], rows, pad32(r)).reshape(rows, pad32(r))
        if Xt_pt is None:  # (the fused target side: one launch forward, one for its dgrad)
            from . import x6

            return x6.target_side(self, self._X_trgt_raw, Xc_pt, R, zb=zb)
        if n_z == 1 and C > 0 and not isinstance(self.attender, DotAttender) and self.z_dim == self.r_dim:
            from . import x6

            if x6.decoder_side_usable(self, T):  # (attention with learned projections, then merge_r_z + decoder as one program)
                R_t = self.attender.attend_pt(Xt_pt.t, Xc_pt.t, R.t, C, T, queries_proj=Xt_pt.proj)
                return x6.decoder_side(self, R_t, Xt_pt.t, T, zb=zb[:, :r].contiguous())
        if n_z == 1:
            ch = Chain(B, T, dev, wg_per_task=True)
            if C == 0:
                ch.input_pt(torch.zeros(pt_shape(B, T, r), device=dev), r)
            else:
                self._attend_into(ch, Xc_pt, R, Xt_pt, C, T)
            mod = 0
        else:
            if C == 0:
                R_det = torch.zeros(pt_shape(B, T, r), device=dev)
            else:
                cha = Chain(B, T, dev, wg_per_task=True)
                self._attend_into(cha, Xc_pt, R, Xt_pt, C, T).output_pt()
                (R_det,) = cha.run()
            ch = Chain(rows, T, dev, wg_per_task=True)
            ch.input_pt(R_det, r, modulus=B)
            mod = B
        ch.linear(W[:, :r], zb, relu=True, bias_per_task=True)
        return self.decoder.finish_rows(ch, x1_pt=Xt_pt.t, x1_modulus=mod)
