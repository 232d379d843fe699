"""Context / target split on the device: the step right before the path.

Mirrors, for 1-D / set-structured data, the reference's ``npf/utils/datasplit.py``:
``get_all_indcs`` (:30-34), ``GetRangeIndcs`` (:37-45), ``GetRandomIndcs`` (:60-145) and
``CntxtTrgtGetter`` (:148-255), with the same constructor arguments and call signatures.  The
reference samples indices with numpy on the host (one ``np.random.shuffle`` per batch row) and
gathers on whatever device ``X`` lives; here the per-row random subsets are drawn on the device
(argsort of uniform noise: every row an independent uniformly random subset, same distribution,
no host round trip) and the gather is one HIP launch for X and y together
(``npf_gather_points``).  Explicit ``context_indcs`` / ``target_indcs`` give bit-identical
selections to the reference's ``torch.gather``.  The *number* of points is a host-side draw
(``random.randint`` / ``scipy.stats.betabinom``), as in the reference.  The grid / mask getters
(images) are out of scope.
"""
from __future__ import annotations

import random
from typing import Optional

import numpy as np
import torch

from . import _lib as L

__all__ = ["get_all_indcs", "GetRangeIndcs", "GetRandomIndcs", "CntxtTrgtGetter"]


def _ratio_to_int(percentage, max_val):
    """npf/utils/helpers.py:99-108."""
    if 1 <= percentage <= max_val:
        out = percentage
    elif 0 <= percentage < 1:
        out = percentage * max_val
    else:
        raise ValueError("percentage={} outside of [0,{}].".format(percentage, max_val))
    return int(out)


def get_all_indcs(batch_size, n_possible_points, device=None):
    """All indices for every batch element (datasplit.py:30-34)."""
    return torch.arange(n_possible_points, device=device).expand(batch_size, n_possible_points)


class GetRangeIndcs:
    """All indices in a range (datasplit.py:37-45)."""

    def __init__(self, arange):
        self.arange = arange

    def __call__(self, batch_size, n_possible_points, device=None):
        indcs = torch.arange(*self.arange, device=device)
        return indcs.expand(batch_size, len(indcs))


class GetRandomIndcs:
    """Random subset of indices (datasplit.py:60-145), drawn on ``device``."""

    def __init__(self, a=0.1, b=0.5, is_batch_share=False, range_indcs=None, is_ensure_one=False,
                 is_beta_binomial=False, proba_uniform=0):
        self.a, self.b = a, b
        self.is_batch_share = is_batch_share
        self.range_indcs = range_indcs
        self.is_ensure_one = is_ensure_one
        self.is_beta_binomial = is_beta_binomial
        self.proba_uniform = proba_uniform

    def n_indcs(self, n_possible_points: int) -> int:
        if np.random.uniform(size=1) < self.proba_uniform:
            n = random.randint(0, n_possible_points)
        elif self.is_beta_binomial:
            from scipy.stats import betabinom

            n = int(betabinom(n_possible_points, self.a, self.b).rvs())
        else:
            n = random.randint(_ratio_to_int(self.a, n_possible_points), _ratio_to_int(self.b, n_possible_points))
        if self.is_ensure_one and n < 1:
            n = 1
        return n

    def __call__(self, batch_size, n_possible_points, device=None, generator: Optional[torch.Generator] = None):
        if self.range_indcs is not None:
            n_possible_points = self.range_indcs[1] - self.range_indcs[0]
        n = self.n_indcs(n_possible_points)
        if self.is_batch_share:
            indcs = torch.randperm(n_possible_points, device=device, generator=generator)[:n]
            indcs = indcs.unsqueeze(0).expand(batch_size, n)
        else:
            # an independent uniformly random subset (in random order) per row
            noise = torch.rand(batch_size, n_possible_points, device=device, generator=generator)
            indcs = noise.argsort(dim=1)[:, :n]
        if self.range_indcs is not None:
            indcs = indcs + self.range_indcs[0]
        return indcs


class CntxtTrgtGetter:
    """Split (X, y) into context and target points by indices (datasplit.py:148-255)."""

    def __init__(self, contexts_getter=GetRandomIndcs(), targets_getter=get_all_indcs, is_add_cntxts_to_trgts=False):
        self.contexts_getter = contexts_getter
        self.targets_getter = targets_getter
        self.is_add_cntxts_to_trgts = is_add_cntxts_to_trgts

    def __call__(self, X, y=None, context_indcs=None, target_indcs=None, is_return_indcs=False):
        batch_size, num_points = self.getter_inputs(X)
        given = context_indcs is not None or target_indcs is not None  # caller-supplied indices get range-checked
        if context_indcs is None:
            context_indcs = self._draw(self.contexts_getter, batch_size, num_points, X.device)
        if target_indcs is None:
            target_indcs = self._draw(self.targets_getter, batch_size, num_points, X.device)
        if self.is_add_cntxts_to_trgts:
            target_indcs = self.add_cntxts_to_trgts(num_points, target_indcs, context_indcs)
        X_pre_cntxt = self.preprocess_context(X)
        if is_return_indcs:
            return context_indcs, X_pre_cntxt, target_indcs, X
        X_cntxt, Y_cntxt = self.select(X_pre_cntxt, y, context_indcs, validate=given)
        X_trgt, Y_trgt = self.select(X, y, target_indcs, validate=given)
        return X_cntxt, Y_cntxt, X_trgt, Y_trgt

    @staticmethod
    def _draw(getter, batch_size, num_points, device):
        try:
            return getter(batch_size, num_points, device=device)
        except TypeError:  # a reference-style getter without the device argument
            return getter(batch_size, num_points)

    def preprocess_context(self, X):
        return X

    def add_cntxts_to_trgts(self, num_points, target_indcs, context_indcs):
        target_indcs = torch.cat([torch.as_tensor(target_indcs), torch.as_tensor(context_indcs).to(target_indcs.device)], dim=-1)
        return target_indcs[:, :num_points]

    def getter_inputs(self, X):
        batch_size, num_points, x_dim = X.shape
        return batch_size, num_points

    def select(self, X, y, indcs, validate=True):
        """``torch.gather`` of X and y along the points with the same indices (datasplit.py:246-255):
        one ``npf_gather_points`` launch.  ``validate``: range-check the indices (a host sync; the
        getters' own draws are in range by construction and skip it)."""
        if not X.is_cuda:
            raise RuntimeError("the HIP path takes device tensors only (got a CPU tensor); there is no CPU fallback")
        batch_size, num_points, x_dim = X.shape
        y_dim = y.size(-1)
        indcs = torch.as_tensor(indcs).to(device=X.device, dtype=torch.int64)
        if indcs.dim() != 2 or indcs.shape[0] != batch_size:
            raise ValueError(f"indices must be [batch_size, n_indcs], got {tuple(indcs.shape)}")
        indcs = indcs.contiguous()  # (materialises expanded / shared index rows)
        n_sel = indcs.shape[1]
        if validate and n_sel and (int(indcs.min()) < 0 or int(indcs.max()) >= num_points):
            raise IndexError("context / target index out of range")
        Xc, yc = X.contiguous().float(), y.contiguous().float()
        out_x = torch.empty(batch_size, n_sel, x_dim, dtype=torch.float32, device=X.device)
        out_y = torch.empty(batch_size, n_sel, y_dim, dtype=torch.float32, device=X.device)
        if n_sel:
            L.check(L.load().npf_gather_points(L.ptr(Xc), L.ptr(yc), indcs.data_ptr(), batch_size, num_points, n_sel,
                                               x_dim, y_dim, L.ptr(out_x), L.ptr(out_y), L.stream_ptr()),
                    "npf_gather_points")
        return out_x, out_y
