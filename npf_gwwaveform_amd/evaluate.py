"""Test log-likelihood per function, the evaluation protocol of the reference (utils/evaluate.py:9-28) without
its skorch harness: seed, evaluation mode, per-task (``reduction=None``) log-likelihoods of every batch in order."""
from __future__ import annotations

import random
from typing import Iterable

import numpy as np
import torch

__all__ = ["eval_loglike"]


def eval_loglike(model: torch.nn.Module, criterion: torch.nn.Module, batches: Iterable[dict], seed: int = 123) -> np.ndarray:
    """Log-likelihood of every task of every batch, in order, as one numpy vector.

    ``batches`` yields dicts ``X_cntxt, Y_cntxt, X_trgt, Y_trgt`` of device tensors (what ``CntxtTrgtGetter``
    produces).  Like the reference: the seed is set first (same latent noise and -- if the batches are generated
    lazily -- the same context / target draws on every call), model and criterion run in evaluation mode (so the
    criterion is the log-mean-exp over ``n_z_samples_test`` latent samples), the criterion's reduction is switched
    off for the call and restored afterwards, and the sign is flipped (log-likelihood, not loss)."""
    # utils/helpers.py:49-55 (set_seed): torch (+ cuda), random, numpy -- GetRandomIndcs draws the context size with
    # Python's ``random`` (npf/utils/datasplit.py:68,74)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    was_training = (model.training, criterion.training)
    old_reduction = criterion.reduction
    criterion.reduction = None
    model.eval()
    criterion.eval()
    out = []
    try:
        with torch.no_grad():
            for batch in batches:
                pred = model(batch["X_cntxt"], batch["Y_cntxt"], batch["X_trgt"], batch["Y_trgt"])
                out.append(-criterion(pred, batch["Y_trgt"]))
    finally:
        criterion.reduction = old_reduction
        model.train(was_training[0])
        criterion.train(was_training[1])
    return torch.cat(out, dim=0).detach().cpu().numpy()
