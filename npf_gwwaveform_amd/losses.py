"""Training and evaluation objectives of the neural-process family behind the reference's class names and
constructor arguments (npf/losses.py): ``CNPFLoss``, ``ELBOLossLNPF``, ``NLLLossLNPF``, ``SUMOLossLNPF``.

Every objective here is "a per-task estimate of log p(y_T | C) from the log weights of the latent samples":

    log_w[k, b] = sum_t log p(y_t | z_k)          fused into the Gaussian-head kernel, loss-only launch
                + log q(z_k | C) - log q(z_k | C, T)   when the samples were drawn from q(z | C, T)

reduced over k by ``npf_mc_objective_fwd`` (mean / log-mean-exp / SUMO, one thread per task, running logsumexp).
Nothing of size [n_z, B, T, y_dim] is written on the way: the predictive distribution the models return
(:class:`~npf_gwwaveform_amd.neuralproc.HeadDistribution`) only materialises ``loc`` / ``scale`` when somebody
looks at them.  What each class adds is one line: which samples count (``is_force_mle_eval``), which estimator,
which regulariser.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.distributions.kl import kl_divergence

from . import functional as FN

__all__ = ["CNPFLoss", "ELBOLossLNPF", "NLLLossLNPF", "SUMOLossLNPF", "LightTailPareto", "sum_log_prob"]


def sum_log_prob(prob, sample):
    """Log-probability of ``sample`` summed over everything but the first two (z-sample, task) dimensions -> [n_z, B]
    (npf/losses.py:18-24).  Predictive distributions of this package answer from the fused head kernel."""
    fused = getattr(prob, "sum_log_prob", None)
    if fused is not None:
        return fused(sample)
    lp = prob.log_prob(sample)  # the small latent distributions: [n_z, B, 1]
    return lp.reshape(lp.shape[0], lp.shape[1], -1).sum(-1)


def latent_log_ratio(z_samples, q_zCc, q_zCct):
    """log q(z | C) - log q(z | C, T) per (sample, task): the importance correction of samples drawn from q(z | C, T)."""
    return sum_log_prob(q_zCc, z_samples) - sum_log_prob(q_zCct, z_samples)


class _TaskLogLikelihood(nn.Module):
    """Shared shell: ``forward(pred_outputs, Y_trgt)`` -> loss, reduced over the tasks by ``reduction``
    ("mean" | "sum" | None).  Outside training the objective is the log-mean-exp estimate over the latent samples,
    without importance weights when ``is_force_mle_eval`` (npf/losses.py:45-81)."""

    estimator = FN.MC_LOGMEANEXP   # how the latent samples are combined in training
    uses_importance_weights = True  # does training use samples from q(z | C, T) with their importance correction

    def __init__(self, reduction="mean", is_force_mle_eval=True):
        super().__init__()
        if reduction not in ("mean", "sum", None):
            raise ValueError(f"Unknown {reduction}")
        self.reduction = reduction
        self.is_force_mle_eval = is_force_mle_eval

    # -- pieces -------------------------------------------------------------------------------------------------
    def log_weights(self, p_yCc, z_samples, q_zCc, q_zCct, Y_trgt):
        log_w = sum_log_prob(p_yCc, Y_trgt)
        if q_zCct is not None:
            log_w = log_w + latent_log_ratio(z_samples, q_zCc, q_zCct)
        return log_w

    def estimate(self, log_w):
        """Per-task log-likelihood estimate from the [n_z, B] log weights (training)."""
        return FN.mc_objective(log_w, self.estimator)

    def regulariser(self, q_zCc, q_zCct):
        return None

    # -- the nn.Module contract ---------------------------------------------------------------------------------
    def get_loss(self, p_yCc, z_samples, q_zCc, q_zCct, Y_trgt):
        """Per-task training loss [B] (the reference's hook of the same name)."""
        if not self.uses_importance_weights:
            z_samples = q_for_weights = None
        else:
            q_for_weights = q_zCct
        loss = -self.estimate(self.log_weights(p_yCc, z_samples, q_zCc, q_for_weights, Y_trgt))
        reg = self.regulariser(q_zCc, q_zCct)
        return loss if reg is None else loss + reg

    def forward(self, pred_outputs, Y_trgt):
        p_yCc, z_samples, q_zCc, q_zCct = pred_outputs
        if self.training:
            loss = self.get_loss(p_yCc, z_samples, q_zCc, q_zCct, Y_trgt)
        else:
            if self.is_force_mle_eval:
                q_zCct = None
            loss = -FN.mc_objective(_TaskLogLikelihood.log_weights(self, p_yCc, z_samples, q_zCc, q_zCct, Y_trgt),
                                    FN.MC_LOGMEANEXP)
        if self.reduction is None:
            return loss
        return loss.mean(0) if self.reduction == "mean" else loss.sum(0)


class CNPFLoss(_TaskLogLikelihood):
    """Conditional members (no latent): minus the log-likelihood of the targets (npf/losses.py:112-123)."""

    def get_loss(self, p_yCc, _, q_zCc, ___, Y_trgt):
        assert q_zCc is None
        return -sum_log_prob(p_yCc, Y_trgt).squeeze(0)


class ELBOLossLNPF(_TaskLogLikelihood):
    """Evidence lower bound (npf/losses.py:126-150): mean over the samples of q(z | C, T) of the target
    log-likelihood, minus KL(q(z | C, T) || q(z | C))."""

    estimator = FN.MC_MEAN
    uses_importance_weights = False

    def regulariser(self, q_zCc, q_zCct):
        kl = kl_divergence(q_zCct, q_zCc)
        return kl.reshape(kl.shape[0], -1).sum(-1)


class NLLLossLNPF(_TaskLogLikelihood):
    """Approximate maximum likelihood (npf/losses.py:153-203): log-mean-exp over the latent samples, importance
    weighted when they come from q(z | C, T)."""


def _light_tail_pareto_class():
    from scipy.stats import rv_discrete

    class LightTailPareto(rv_discrete):
        """Number-of-samples distribution of SUMO (npf/utils/helpers.py:36-53): P(K >= k) decays like 1 / k up to
        ``alpha`` and geometrically (0.9 per step) beyond; ``a`` = the smallest number of samples."""

        def _cdf(self, k, alpha):
            shifted = np.clip(k + 1 - self.a, a_min=1, a_max=None)
            knee = alpha - self.a
            survival = np.where(shifted < knee, 1 / shifted, (1 / knee) * 0.9 ** (shifted - knee))
            return 1 - survival

    return LightTailPareto


def LightTailPareto(*args, **kwargs):
    """``LightTailPareto(a=5).freeze(85)``, the reference's calling convention."""
    return _light_tail_pareto_class()(*args, **kwargs)


class SUMOLossLNPF(_TaskLogLikelihood):
    """SUMO estimate of the log marginal likelihood (npf/losses.py:207-276): the running k-sample bounds combined
    with the inverse tail probabilities P(K >= k) of the number-of-samples distribution."""

    estimator = FN.MC_SUMO

    def __init__(self, p_n_z_samples=None, **kwargs):
        super().__init__(**kwargs)
        self.p_n_z_samples = LightTailPareto(a=5).freeze(85) if p_n_z_samples is None else p_n_z_samples
        self._tail_cache = {}

    def _tail_probabilities(self, n_z, device):
        key = (n_z, str(device))
        if key not in self._tail_cache:
            survival = 1.0 - self.p_n_z_samples.cdf(np.arange(n_z))  # P(K >= k), k = 1 .. n_z
            self._tail_cache[key] = torch.as_tensor(survival, dtype=torch.float32, device=device).contiguous()
        return self._tail_cache[key]

    def estimate(self, log_w):
        n_z = log_w.shape[0]
        m = int(self.p_n_z_samples.support()[0])
        if m > n_z:
            raise ValueError(f"SUMO needs at least {m} latent samples, got {n_z}")
        return FN.mc_objective(log_w, FN.MC_SUMO, self._tail_probabilities(n_z, log_w.device), m)
