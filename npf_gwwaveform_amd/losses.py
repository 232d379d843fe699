"""Training objectives of the neural-process family with the reference's interface
(npf/losses.py): ``CNPFLoss``, ``ELBOLossLNPF``, ``NLLLossLNPF``.

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

__all__ = ["CNPFLoss", "ELBOLossLNPF", "NLLLossLNPF", "sum_log_prob"]


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
