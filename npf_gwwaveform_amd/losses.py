"""Training objectives of the neural-process family with the reference's interface
(npf/losses.py): ``CNPFLoss``, ``ELBOLossLNPF``, ``NLLLossLNPF``, ``SUMOLossLNPF``.

``sum_log_prob`` uses the log-likelihood that the Gaussian-head kernel already summed over
the targets (``npf_gauss_head_fwd``) whenever the predictive distribution comes from this
package's models; the remaining arithmetic is on [n_z, B]-sized tensors.
"""
from __future__ import annotations

import abc
import math

import torch
import torch.nn as nn
from torch.distributions.kl import kl_divergence

from . import functional as FN

__all__ = ["CNPFLoss", "ELBOLossLNPF", "NLLLossLNPF", "SUMOLossLNPF", "LightTailPareto", "sum_log_prob"]


def sum_from_nth_dim(t, dim):
    return t.view(*t.shape[:dim], -1).sum(-1)


def sum_log_prob(prob, sample):
    """``sum_log_prob`` (npf/losses.py:18-24): log-probability summed over everything but
    the z-sample and batch dims -> [n_z, B]."""
    cached = getattr(prob, "_npf_sum_log_prob", None)
    if cached is not None and cached[0] is sample:
        return cached[1]
    head = getattr(prob, "_npf_suff", None)
    if head is not None:
        suff, dy, homosk = head
        n_z_B = suff.shape[0]
        _, _, slp = FN.gauss_head(suff, sample.contiguous(), dy, homosk)
        return slp.view(n_z_B // sample.shape[0], sample.shape[0])
    return sum_from_nth_dim(prob.log_prob(sample), 2)  # small latent distributions


class BaseLossNPF(nn.Module, abc.ABC):
    """npf/losses.py:27-109."""

    def __init__(self, reduction="mean", is_force_mle_eval=True):
        super().__init__()
        self.reduction = reduction
        self.is_force_mle_eval = is_force_mle_eval

    def forward(self, pred_outputs, Y_trgt):
        p_yCc, z_samples, q_zCc, q_zCct = pred_outputs
        if self.training:
            loss = self.get_loss(p_yCc, z_samples, q_zCc, q_zCct, Y_trgt)
        else:
            if self.is_force_mle_eval:
                q_zCct = None
            loss = NLLLossLNPF.get_loss(self, p_yCc, z_samples, q_zCc, q_zCct, Y_trgt)
        if self.reduction is None:
            return loss
        elif self.reduction == "mean":
            return loss.mean(0)
        elif self.reduction == "sum":
            return loss.sum(0)
        raise ValueError(f"Unknown {self.reduction}")

    @abc.abstractmethod
    def get_loss(self, p_yCc, z_samples, q_zCc, q_zCct, Y_trgt):
        pass


class CNPFLoss(BaseLossNPF):
    """npf/losses.py:112-123."""

    def get_loss(self, p_yCc, _, q_zCc, ___, Y_trgt):
        assert q_zCc is None
        return -sum_log_prob(p_yCc, Y_trgt).squeeze(0)


class ELBOLossLNPF(BaseLossNPF):
    """npf/losses.py:126-150."""

    def get_loss(self, p_yCc, _, q_zCc, q_zCct, Y_trgt):
        E_z_sum_log_p_yCz = sum_log_prob(p_yCc, Y_trgt).mean(0)
        E_z_kl = sum_from_nth_dim(kl_divergence(q_zCct, q_zCc), 1)
        return -(E_z_sum_log_p_yCz - E_z_kl)


class NLLLossLNPF(BaseLossNPF):
    """npf/losses.py:153-203."""

    def get_loss(self, p_yCc, z_samples, q_zCc, q_zCct, Y_trgt):
        n_z_samples = p_yCc.batch_shape[0]
        sum_log_w_k = sum_log_prob(p_yCc, Y_trgt)
        if q_zCct is not None:
            sum_log_w_k = sum_log_w_k + sum_from_nth_dim(q_zCc.log_prob(z_samples), 2) \
                - sum_from_nth_dim(q_zCct.log_prob(z_samples), 2)
        return -(torch.logsumexp(sum_log_w_k, 0) - math.log(n_z_samples))


def _importance_log_weights(p_yCc, z_samples, q_zCc, q_zCct, Y_trgt):
    sum_log_w_k = sum_log_prob(p_yCc, Y_trgt)
    if q_zCct is not None:
        sum_log_w_k = sum_log_w_k + sum_from_nth_dim(q_zCc.log_prob(z_samples), 2) \
            - sum_from_nth_dim(q_zCct.log_prob(z_samples), 2)
    return sum_log_w_k


def _light_tail_pareto():
    from scipy.stats import rv_discrete
    import numpy as np

    class LightTailPareto(rv_discrete):
        """Number-of-samples distribution of SUMO (npf/utils/helpers.py:36-53): P(K >= k) ~ 1/k up to
        ``alpha``, geometric (0.9) beyond, shifted so that at least ``a`` samples are drawn."""

        def _cdf(self, k, alpha):
            m = self.a
            k = np.clip(k + 1 - m, a_min=1, a_max=None)
            alpha = alpha - m
            return 1 - np.where(k < alpha, 1 / k, (1 / alpha) * (0.9) ** (k - alpha))

    return LightTailPareto


def LightTailPareto(*args, **kwargs):
    """Factory with the reference's calling convention: ``LightTailPareto(a=5).freeze(85)``."""
    return _light_tail_pareto()(*args, **kwargs)


class SUMOLossLNPF(BaseLossNPF):
    """Negative log likelihood estimated with SUMO (npf/losses.py:207-276): the k-sample importance
    weighted bounds (a running logsumexp over the z-samples) combined with the inverse tail
    probabilities of the number-of-samples distribution."""

    def __init__(self, p_n_z_samples=None, **kwargs):
        super().__init__(**kwargs)
        self.p_n_z_samples = LightTailPareto(a=5).freeze(85) if p_n_z_samples is None else p_n_z_samples

    def get_loss(self, p_yCc, z_samples, q_zCc, q_zCct, Y_trgt):
        import numpy as np

        n_z_samples = p_yCc.batch_shape[0]
        sum_log_w_k = _importance_log_weights(p_yCc, z_samples, q_zCc, q_zCct, Y_trgt)  # [n_z, B]
        ks = torch.arange(1, n_z_samples + 1).unsqueeze(-1)
        cum_iwae = torch.logcumsumexp(sum_log_w_k, 0) - ks.float().log().to(sum_log_w_k.device)
        inv_weights = torch.from_numpy(1 - self.p_n_z_samples.cdf((ks - 1).numpy())).to(sum_log_w_k.device)
        m = self.p_n_z_samples.support()[0]
        sumo = cum_iwae[m - 1] + (inv_weights[m:] * (cum_iwae[m:] - cum_iwae[m - 1:-1])).sum(0)
        return -sumo
