"""Models and global batches of the data-parallel tests (shared by the pytest process and its rank children)."""
import warnings
from functools import partial

import torch

CASES = {
    # the c2 / c4 model at a reduced batch: AttnCNP scaledot r = 256, 4-layer encoder / decoder
    "attncnp_r256": dict(kind="AttnCNP", r=256, L=4, B=4, C=64, T=160, steps=2),
    # the same through ``Trainer(use_graph=True)``: three eager steps (bucketed, overlapped all-reduce), then forward +
    # backward replayed from a HIP graph, ONE all-reduce of the flat gradient, Adam replayed from a second graph
    "attncnp_r256_graph": dict(kind="AttnCNP", r=256, L=4, B=4, C=64, T=160, steps=6, use_graph=True),
    # latent model with the target-side encode (q_zCct) and injected noise, ragged point counts
    "attnlnp_r64": dict(kind="AttnLNP", r=64, L=2, B=6, C=37, T=70, steps=2),
}


def build(case, seed, device="cuda:0"):
    import npf_gwwaveform_amd as A
    from helpers import eps_latent_dist

    torch.manual_seed(seed)
    r, L = case["r"], case["L"]
    kw = dict(r_dim=r,
              XYEncoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, is_force_hid_smaller=True, hidden_size=r),
                                           is_sum_merge=True),
              Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, hidden_size=r), is_sum_merge=True))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if case["kind"] == "AttnCNP":
            m, crit = A.AttnCNP(1, 2, attention="scaledot", **kw), A.CNPFLoss()
        else:
            m = A.AttnLNP(1, 2, attention="scaledot", is_q_zCct=True, n_z_samples_train=1, n_z_samples_test=1,
                          LatentDistribution=eps_latent_dist, **kw)
            crit = A.ELBOLossLNPF()
    return m.to(device), crit


def global_batch(case, device="cuda:0"):
    from npf_gwwaveform_amd.train import synthetic_waveform_batch

    batch = synthetic_waveform_batch(case["B"], case["C"], case["T"], 4321, device)
    if case["kind"] == "AttnLNP":
        g = torch.Generator(device=device).manual_seed(7)
        batch["eps"] = torch.randn(1, case["B"], 1, case["r"], generator=g, device=device)
    return batch
