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


def _point_count(spec, n_points: int) -> int:
    """A number of points given either as a count (1 <= spec <= n_points) or as a fraction of ``n_points``
    (0 <= spec < 1), the convention of the reference's getters (npf/utils/helpers.py:99-108)."""
    if 0 <= spec < 1:
        return int(spec * n_points)
    if 1 <= spec <= n_points:
        return int(spec)
    raise ValueError("percentage={} outside of [0,{}].".format(spec, n_points))


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
            n = random.randint(_point_count(self.a, n_possible_points), _point_count(self.b, n_possible_points))
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
    """Split (X, y) into context and target points (datasplit.py:148-255): ``getter(X, y)`` ->
    ``X_cntxt, Y_cntxt, X_trgt, Y_trgt``.  Same constructor arguments, call signature and overridable hooks
    (``preprocess_context``, ``add_cntxts_to_trgts``, ``getter_inputs``, ``select``) as the reference; the work is
    two steps: :meth:`indices` decides which points go where (device-side draws unless the caller supplies them),
    :meth:`select` moves them (one gather launch per side)."""

    def __init__(self, contexts_getter=GetRandomIndcs(), targets_getter=get_all_indcs, is_add_cntxts_to_trgts=False):
        self.contexts_getter = contexts_getter
        self.targets_getter = targets_getter
        self.is_add_cntxts_to_trgts = is_add_cntxts_to_trgts

    def indices(self, X, context_indcs=None, target_indcs=None):
        """(context indices, target indices, were any supplied by the caller) for the batch ``X``."""
        batch_size, num_points = self.getter_inputs(X)
        supplied = not (context_indcs is None and target_indcs is None)
        drawn = []
        for given, getter in ((context_indcs, self.contexts_getter), (target_indcs, self.targets_getter)):
            if given is not None:
                drawn.append(given)
                continue
            try:
                drawn.append(getter(batch_size, num_points, device=X.device))
            except TypeError:  # a reference-style getter without the device argument
                drawn.append(getter(batch_size, num_points))
        ctx, trg = drawn
        if self.is_add_cntxts_to_trgts:
            trg = self.add_cntxts_to_trgts(num_points, trg, ctx)
        return ctx, trg, supplied

    def __call__(self, X, y=None, context_indcs=None, target_indcs=None, is_return_indcs=False):
        ctx, trg, supplied = self.indices(X, context_indcs, target_indcs)
        X_for_context = self.preprocess_context(X)
        if is_return_indcs:
            return ctx, X_for_context, trg, X
        # caller-supplied indices are range-checked (one host sync); the getters' own draws are in range by construction
        return (*self.select(X_for_context, y, ctx, validate=supplied), *self.select(X, y, trg, validate=supplied))

    # ---- hooks of the reference ---------------------------------------------------------------------------------
    def preprocess_context(self, X):
        """What the context side sees of X (identity; the reference's subclasses mask or crop here)."""
        return X

    def add_cntxts_to_trgts(self, num_points, target_indcs, context_indcs):
        """Targets followed by the context points, cut to ``num_points`` columns (datasplit.py:225-232)."""
        trg = torch.as_tensor(target_indcs)
        both = torch.cat([trg, torch.as_tensor(context_indcs).to(trg.device)], dim=-1)
        return both[:, :num_points]

    def getter_inputs(self, X):
        """(batch size, number of points) handed to the index getters."""
        return X.shape[0], X.shape[1]

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
