"""Helpers shared by the tests: build this package's models for a golden case."""
import warnings
from functools import partial

import numpy as np
import torch
from torch.distributions import Independent, Normal

import specs


class EpsIndependent(Independent):
    """Independent(Normal) whose rsample uses an injected eps (the same device-side
    replacement of the global-RNG draw that make_golden.py applies to the reference)."""

    eps = None

    def rsample(self, sample_shape=torch.Size()):
        e = type(self).eps
        assert e is not None and e.shape[0] == sample_shape[0]
        return self.base_dist.loc + e * self.base_dist.scale


def eps_latent_dist(loc, scale):
    return EpsIndependent(Normal(loc, scale, validate_args=False), 1)


def build_model(case: dict, device="cuda:0", params=None):
    import npf_gwwaveform_amd as A

    r = case["r"]
    res, drop = case.get("is_res", False), case.get("dropout", 0)
    kw = dict(
        r_dim=r, is_heteroskedastic=case.get("is_heteroskedastic", True),
        XYEncoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=case["L_xy"], is_force_hid_smaller=True,
                                             hidden_size=r, is_res=res, dropout=drop),
                                     is_sum_merge=case.get("is_sum_merge", True)),
        Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=case["L_dec"], hidden_size=r, is_res=res, dropout=drop),
                                   is_sum_merge=True),
    )
    if "x_transf_dim" in case:
        kw["x_transf_dim"] = case["x_transf_dim"]
    kind = case["kind"]
    if kind in ("LNP", "AttnLNP"):
        n_z = case.get("n_z", 1)
        kw.update(is_q_zCct=case.get("is_q_zCct", False), n_z_samples_train=n_z, n_z_samples_test=n_z,
                  LatentDistribution=eps_latent_dist)
    if kind == "LNP":
        kw["encoded_path"] = case["encoded_path"]
    if "attention" in case:
        kw["attention"] = case["attention"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = getattr(A, kind)(case["dx"], case["dy"], **kw)
    m.load_state_dict(params if params is not None else specs.make_params(case), strict=True)
    return m.to(device)


def build_loss(case: dict):
    import npf_gwwaveform_amd as A

    return {"cnpf": A.CNPFLoss, "elbo": A.ELBOLossLNPF, "nll": A.NLLLossLNPF,
            "sumo": A.SUMOLossLNPF}[specs.loss_name(case)]()


def assert_close(got, ref, tol=1e-5, what=""):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, dtype=np.float64)
    ref = ref.detach().cpu().double().numpy() if torch.is_tensor(ref) else np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    m = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max()
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    assert err <= tol * m, f"{what}: max|d|={err:.3e} > {tol:.0e} * max|ref|={m:.3e}"
