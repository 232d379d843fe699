"""Shared definitions of the golden cases: model configs, seeded inputs, seeded weights.

Used by ``make_golden.py`` (which runs the *reference* on them, in the build container
only) and by the tests (which run the oracle / the HIP path on the very same inputs).
Everything here is this project's own code; nothing is taken from the reference.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import npf_oracle as O  # noqa: E402

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))

# name -> dict(kind, r, L_xy, L_dec, dx, dy, B, C, T, extra cfg kwargs)
CASES = {
    # G1: config-1 tiny CNP, bit-for-bit gate
    "g1_cnp_c1": dict(kind="CNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=8, C=32, T=64),
    # G2: config-1 tiny LNP (both paths), injected eps
    "g2_lnp_both_c1": dict(kind="LNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=8, C=32, T=64,
                           encoded_path="both", is_q_zCct=True, n_z=2),
    "g2_lnp_latent_c1": dict(kind="LNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=8, C=32, T=64,
                             encoded_path="latent", is_q_zCct=False, n_z=3),
    # mid-size attentive models with full gradients stored
    "g3s_attncnp_r64": dict(kind="AttnCNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=4, C=64, T=96),
    "g4s_attnlnp_r64": dict(kind="AttnLNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=4, C=64, T=96,
                            is_q_zCct=True, n_z=2),
    "g4s_attnlnp_r64_noqzcct": dict(kind="AttnLNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=4, C=64, T=96,
                                    is_q_zCct=False, n_z=1),
    # G3/G4: config-2 model at reduced batch (weights regenerated from seed, not stored)
    "g3_attncnp_c2": dict(kind="AttnCNP", r=256, L_xy=4, L_dec=4, dx=1, dy=2, B=2, C=256, T=1024),
    "g4_attnlnp_c2": dict(kind="AttnLNP", r=256, L_xy=4, L_dec=4, dx=1, dy=2, B=2, C=256, T=1024,
                          is_q_zCct=True, n_z=1),
    # G6: edge cases (ragged C/T, C=1, homoskedastic, wider x/y)
    "g6_attncnp_ragged": dict(kind="AttnCNP", r=32, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=5, T=7),
    "g6_attncnp_c1pt": dict(kind="AttnCNP", r=32, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=1, T=33),
    "g6_cnp_homosk": dict(kind="CNP", r=32, L_xy=2, L_dec=2, dx=2, dy=3, B=3, C=9, T=40,
                          is_heteroskedastic=False),
    "g6_cnp_c0": dict(kind="CNP", r=32, L_xy=2, L_dec=2, dx=1, dy=2, B=2, C=0, T=16),
    "g6_attncnp_c0": dict(kind="AttnCNP", r=32, L_xy=2, L_dec=2, dx=1, dy=2, B=2, C=0, T=16),
    "g6_attncnp_r128": dict(kind="AttnCNP", r=128, L_xy=2, L_dec=4, dx=1, dy=1, B=2, C=50, T=128),
    # G8: learned-projection attention, what the reference's notebooks and shipped
    # Attn* checkpoints use
    "g8_attncnp_multihead": dict(kind="AttnCNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=20, T=45, attention="multihead"),
    "g8_attncnp_transformer": dict(kind="AttnCNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=20, T=45,
                                   attention="transformer"),
    # G10: multi-sample objectives -- SUMO with importance weights, 8 latent samples
    "g10_lnp_sumo": dict(kind="LNP", r=32, L_xy=2, L_dec=2, dx=1, dy=2, B=4, C=10, T=24, encoded_path="latent",
                         is_q_zCct=True, n_z=8, loss="sumo"),
    "g10_attnlnp_nll_nz8": dict(kind="AttnLNP", r=32, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=12, T=20, is_q_zCct=True, n_z=8,
                                loss="nll"),
    # G13: layers wider than 256 features train too (the reference's MLPs have no width limit, mlp.py:44-93): the
    # config-5 decoder width as a train step (weight gradients in 256 x 256 blocks, 32-block chain instance)
    "g13_cnp_r512": dict(kind="CNP", r=512, L_xy=2, L_dec=2, dx=1, dy=2, B=2, C=40, T=100),
    "g13_attncnp_r512": dict(kind="AttnCNP", r=512, L_xy=1, L_dec=2, dx=1, dy=2, B=2, C=40, T=70),
    "g8_attnlnp_transformer": dict(kind="AttnLNP", r=128, L_xy=2, L_dec=2, dx=1, dy=2, B=2, C=33, T=40,
                                   attention="transformer", is_q_zCct=True, n_z=2),
}

# G14: the reference's other MLP / merge options on the path's own files (mlp.py:100-104 ``is_res``, encoders.py:163-183
# concatenating merge -- as XY-encoder: as a decoder the reference's own torch.cat of a 3-d and a 4-d tensor raises --,
# base.py:126-131 ``x_transf_dim`` != ``r_dim``).  Parameters: the reference's own seeded
# construction with perturbed biases, stored in the fixture (``param/...``).
VARIANT_CASES = {
    "g14_cnp_res": dict(kind="CNP", r=32, L_xy=3, L_dec=3, dx=1, dy=2, B=3, C=9, T=20, is_res=True),
    "g14_attncnp_res": dict(kind="AttnCNP", r=64, L_xy=3, L_dec=4, dx=1, dy=2, B=2, C=20, T=45, is_res=True),
    "g14_cnp_xt": dict(kind="CNP", r=32, L_xy=2, L_dec=2, dx=2, dy=1, B=3, C=9, T=40, x_transf_dim=64),
    "g14_lnp_xt": dict(kind="LNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=12, T=33, encoded_path="both", is_q_zCct=True,
                       n_z=2, x_transf_dim=40),
    "g14_attncnp_xt": dict(kind="AttnCNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=17, T=40, x_transf_dim=32),
    "g14_attnlnp_xt": dict(kind="AttnLNP", r=32, L_xy=2, L_dec=2, dx=1, dy=2, B=2, C=35, T=40, is_q_zCct=True, n_z=2,
                           x_transf_dim=96),
    "g14_cnp_cat": dict(kind="CNP", r=32, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=9, T=40, is_sum_merge=False),
    "g14_lnp_cat": dict(kind="LNP", r=32, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=9, T=40, encoded_path="latent", n_z=3,
                        is_sum_merge=False),
    "g14_attncnp_cat": dict(kind="AttnCNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=20, T=45, is_sum_merge=False),
    # dropout in the XY-encoder / decoder MLPs (mlp.py:81,98,105): the masks the reference drew are stored (dropmask/i)
    "g14_cnp_drop": dict(kind="CNP", r=32, L_xy=2, L_dec=3, dx=1, dy=2, B=3, C=9, T=40, dropout=0.25),
    "g14_attncnp_drop_res": dict(kind="AttnCNP", r=64, L_xy=3, L_dec=3, dx=1, dy=2, B=2, C=20, T=45, dropout=0.4, is_res=True),
    # concatenating merge WITH dropout: the first dropout sits behind the layer that runs as two accumulating halves
    "g14_cnp_cat_drop": dict(kind="CNP", r=32, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=9, T=40, is_sum_merge=False, dropout=0.3),
    "g14_attncnp_cat_drop": dict(kind="AttnCNP", r=64, L_xy=3, L_dec=2, dx=1, dy=2, B=2, C=20, T=45, is_sum_merge=False,
                                 dropout=0.2, is_res=True),
    "g14_attnlnp_all": dict(kind="AttnLNP", r=64, L_xy=3, L_dec=3, dx=1, dy=2, B=2, C=20, T=45, is_q_zCct=True, n_z=2,
                            is_sum_merge=False, is_res=True, x_transf_dim=32),
}

# G12: evaluation protocol (utils/evaluate.py:9-28): 32 latent samples at test time, per-task log-likelihoods
EVAL_CASES = {
    "attnlnp": dict(kind="AttnLNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=5, C=21, T=50, is_q_zCct=True, n_z=32),
    "lnp": dict(kind="LNP", r=32, L_xy=2, L_dec=2, dx=1, dy=2, B=4, C=10, T=33, encoded_path="latent", is_q_zCct=True, n_z=32),
    "cnp": dict(kind="CNP", r=32, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=9, T=20),
}

# G5: decode-only, config-5 decoder (r=512, L=4) at reduced batch
DECODE_CASE = dict(r=512, L_dec=4, dx=1, dy=2, B=2, T=4096)


def cfg_of(case: dict) -> O.OracleConfig:
    return O.OracleConfig(
        kind=case["kind"], x_dim=case["dx"], y_dim=case["dy"], r_dim=case["r"],
        encoded_path=case.get("encoded_path"), is_heteroskedastic=case.get("is_heteroskedastic", True),
        is_q_zCct=case.get("is_q_zCct", False), attention=case.get("attention", "scaledot"),
        x_transf_dim=case.get("x_transf_dim"), is_sum_merge=case.get("is_sum_merge", True), is_res=case.get("is_res", False),
        dropout=case.get("dropout", 0.0),
    )


def golden_dropout_masks(g: dict):
    """The keep masks of a fixture in the order the reference drew them (empty for cases without dropout)."""
    n = sum(1 for k in g if k.startswith("dropmask/"))
    return [torch.from_numpy(g[f"dropmask/{i}"].astype("float32")) for i in range(n)]


def golden_params(g: dict) -> dict:
    """The ``param/...`` entries of a fixture as a state dict."""
    return {k[len("param/"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")}


def make_params(case: dict, seed: int = 0):
    cfg = cfg_of(case)
    p = O.init_params(cfg, seed=seed, n_layers_xy=case["L_xy"], n_layers_dec=case["L_dec"])
    return O.perturb_biases(p, seed=seed + 1)


def make_inputs(case: dict, seed: int = 1234):
    """X ~ U[-1,1], Y ~ N(0,1), eps ~ N(0,1) from a numpy Philox stream."""
    rng = np.random.Generator(np.random.Philox(seed))
    B, C, T, dx, dy = case["B"], case["C"], case["T"], case["dx"], case["dy"]
    f = lambda a: torch.from_numpy(a.astype("float32"))  # noqa: E731
    out = dict(
        X_cntxt=f(rng.uniform(-1, 1, size=(B, C, dx))),
        Y_cntxt=f(rng.standard_normal(size=(B, C, dy))),
        X_trgt=f(rng.uniform(-1, 1, size=(B, T, dx))),
        Y_trgt=f(rng.standard_normal(size=(B, T, dy))),
    )
    if case["kind"] in ("LNP", "AttnLNP"):
        out["eps"] = f(rng.standard_normal(size=(case.get("n_z", 1), B, 1, case["r"])))
    return out


def make_decode_inputs(case: dict = DECODE_CASE, seed: int = 77):
    """Inputs of the decode-only case: an encoded X_trgt and an attention-style R_trgt."""
    rng = np.random.Generator(np.random.Philox(seed))
    B, T, r = case["B"], case["T"], case["r"]
    f = lambda a: torch.from_numpy(a.astype("float32"))  # noqa: E731
    return dict(
        X_trgt_enc=f(rng.standard_normal(size=(B, T, r)) * 0.5),
        R_trgt=f(rng.standard_normal(size=(1, B, T, r)) * 0.5),
    )


def make_decode_params(case: dict = DECODE_CASE, seed: int = 5):
    cfg = O.OracleConfig(kind="CNP", x_dim=case["dx"], y_dim=case["dy"], r_dim=case["r"])
    p = O.init_params(cfg, seed=seed, n_layers_xy=2, n_layers_dec=case["L_dec"])
    p = O.perturb_biases(p, seed=seed + 1)
    return cfg, {k: v for k, v in p.items() if k.startswith("decoder.")}


def load_golden(name: str) -> dict:
    with np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def loss_name(case: dict) -> str:
    if "loss" in case:
        return case["loss"]
    if case["kind"] in ("CNP", "AttnCNP"):
        return "cnpf"
    return "elbo" if case.get("is_q_zCct", False) else "nll"
