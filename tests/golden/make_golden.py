"""Generate the golden vectors by running the REFERENCE (``/root/reference/npf``).

Runs in the build container only (the reference does not exist on the GPU box); the
``.npz`` files it writes next to itself are committed.  Usage::

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it does per case (see ``specs.CASES``): builds the reference model class with the
reference's own factories, loads this project's seeded parameter dict into it
(``load_state_dict(strict=True)`` -- which also proves the state_dict key contract),
runs forward + loss + backward (+ one Adam step for G1) on the seeded inputs and stores
inputs-independent results: loc, scale, latent stats, loss, gradients.
"""
from __future__ import annotations

import os
import sys
import warnings
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import specs  # noqa: E402

REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore", category=UserWarning)

import npf  # noqa: E402  (the reference)
from npf.architectures import MLP, merge_flat_input  # noqa: E402
from torch.distributions import Independent, Normal  # noqa: E402

torch.set_num_threads(8)


class _EpsIndependent(Independent):
    """Independent(Normal) whose rsample uses an injected eps instead of the global RNG
    (passed to the reference through its ``LatentDistribution`` constructor argument)."""

    eps = None

    def rsample(self, sample_shape=torch.Size()):
        e = type(self).eps
        assert e is not None and e.shape[0] == sample_shape[0]
        return self.base_dist.loc + e * self.base_dist.scale


def _latent_dist(loc, scale):
    return _EpsIndependent(Normal(loc, scale), 1)


def build_reference(case: dict):
    r = case["r"]
    sm, res, drop = case.get("is_sum_merge", True), case.get("is_res", False), case.get("dropout", 0)
    kw = dict(
        r_dim=r,
        is_heteroskedastic=case.get("is_heteroskedastic", True),
        XYEncoder=merge_flat_input(
            partial(MLP, n_hidden_layers=case["L_xy"], is_force_hid_smaller=True, hidden_size=r, is_res=res, dropout=drop),
            is_sum_merge=sm
        ),
        Decoder=merge_flat_input(partial(MLP, n_hidden_layers=case["L_dec"], hidden_size=r, is_res=res, dropout=drop),
                                 is_sum_merge=True),
    )
    if "x_transf_dim" in case:
        kw["x_transf_dim"] = case["x_transf_dim"]
    kind = case["kind"]
    if kind in ("LNP", "AttnLNP"):
        n_z = case.get("n_z", 1)
        kw.update(is_q_zCct=case.get("is_q_zCct", False), n_z_samples_train=n_z, n_z_samples_test=n_z,
                  LatentDistribution=_latent_dist)
    if kind == "CNP":
        m = npf.CNP(case["dx"], case["dy"], **kw)
    elif kind == "LNP":
        m = npf.LNP(case["dx"], case["dy"], encoded_path=case["encoded_path"], **kw)
    elif kind == "AttnCNP":
        m = npf.AttnCNP(case["dx"], case["dy"], attention=case.get("attention", "scaledot"), **kw)
    else:
        m = npf.AttnLNP(case["dx"], case["dy"], attention=case.get("attention", "scaledot"), **kw)
    return m


def ref_loss(case: dict):
    return {"cnpf": npf.CNPFLoss, "elbo": npf.ELBOLossLNPF, "nll": npf.NLLLossLNPF,
            "sumo": npf.SUMOLossLNPF}[specs.loss_name(case)]()


def run_case(name: str, case: dict, store_params: bool, store_full_grads: bool, adam_step: bool = False,
             own_init: bool = False):
    inp = specs.make_inputs(case)
    if own_init:
        # the reference's own (seeded) construction, biases moved off zero; stored in the fixture
        torch.manual_seed(14)
        model = build_reference(case)
        rng = np.random.Generator(np.random.Philox(14))
        with torch.no_grad():
            for k, p in model.named_parameters():
                if k.endswith(".bias"):
                    p.copy_(torch.from_numpy(rng.uniform(-0.05, 0.05, tuple(p.shape)).astype("float32")))
        params = {k: v.clone() for k, v in model.state_dict().items()}
    else:
        params = specs.make_params(case)
        model = build_reference(case)
        missing = model.load_state_dict(params, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    model.train()
    if "eps" in inp:
        _EpsIndependent.eps = inp["eps"]
    crit = ref_loss(case)
    crit.train()
    # the keep masks nn.Dropout draws from torch's global generator, in call order (where its input is zero the mask
    # cannot be read off the output and does not matter: the unit contributes nothing either way)
    drawn = []
    hooks = [m.register_forward_hook(lambda mod, i, o: drawn.append((o != 0).to(torch.uint8).numpy()))
             for m in model.modules() if isinstance(m, torch.nn.Dropout)]
    torch.manual_seed(2014)
    out = model(inp["X_cntxt"], inp["Y_cntxt"], inp["X_trgt"], inp["Y_trgt"])
    for h in hooks:
        h.remove()
    p_yCc, z_samples, q_zCc, q_zCct = out
    loss = crit(out, inp["Y_trgt"])
    loss.backward()

    res = {
        "loc": p_yCc.base_dist.loc.detach().numpy(),
        "scale": p_yCc.base_dist.scale.detach().numpy(),
        "loss": loss.detach().numpy(),
        "n_params": np.array(sum(p.numel() for p in model.parameters())),
    }
    for i_m, m_ in enumerate(drawn):
        res[f"dropmask/{i_m}"] = m_
    if z_samples is not None:
        res["z_samples"] = z_samples.detach().numpy()
        res["q_zCc_loc"] = q_zCc.base_dist.loc.detach().numpy()
        res["q_zCc_scale"] = q_zCc.base_dist.scale.detach().numpy()
        if q_zCct is not None:
            res["q_zCct_loc"] = q_zCct.base_dist.loc.detach().numpy()
            res["q_zCct_scale"] = q_zCct.base_dist.scale.detach().numpy()
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        if store_full_grads:
            res[f"grad/{k}"] = g.numpy().copy()
        else:
            res[f"gradnorm/{k}"] = np.array(g.double().norm().item())
            res[f"gradhead/{k}"] = g.reshape(-1)[:64].numpy().copy()
    if store_params:
        for k, v in params.items():
            res[f"param/{k}"] = v.numpy()
    # eval-mode forward (losses.py:65-69 uses NLL for eval; n_z_samples_test)
    model.eval()
    with torch.no_grad():
        out_e = model(inp["X_cntxt"], inp["Y_cntxt"], inp["X_trgt"])
    res["eval_loc"] = out_e[0].base_dist.loc.numpy()
    res["eval_scale"] = out_e[0].base_dist.scale.numpy()
    if adam_step:
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        opt.step()
        for k, p in model.named_parameters():
            res[f"adam1/{k}"] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **res)
    print(f"{name}: loss={float(loss):.6f} n_params={int(res['n_params'])} keys={len(res)}")


def run_decode_case():
    case = specs.DECODE_CASE
    cfg, dparams = specs.make_decode_params(case)
    inp = specs.make_decode_inputs(case)
    model = npf.CNP(case["dx"], case["dy"], r_dim=case["r"],
                    Decoder=merge_flat_input(partial(MLP, n_hidden_layers=case["L_dec"], hidden_size=case["r"]),
                                             is_sum_merge=True))
    sd = model.state_dict()
    sd.update(dparams)
    model.load_state_dict(sd, strict=True)
    model.eval()
    with torch.no_grad():
        p = model.decode(inp["X_trgt_enc"], inp["R_trgt"])
    np.savez_compressed(os.path.join(HERE, "g5_decode_r512.npz"),
                        loc=p.base_dist.loc.numpy(), scale=p.base_dist.scale.numpy())
    print("g5_decode_r512: done", p.base_dist.loc.shape)


def run_stage_cases():
    """G6 stage level: MLP, MergeFlatInputs and the scaledot attender in isolation."""
    rng = np.random.Generator(np.random.Philox(99))
    f = lambda a: torch.from_numpy(np.asarray(a, dtype="float32"))  # noqa: E731
    res = {}
    # scaledot attention, ragged sizes
    from npf.architectures import get_attender
    for tag, (B, C, T, d, v) in {"a": (3, 37, 70, 64, 64), "b": (1, 96, 64, 256, 256), "c": (2, 1, 5, 32, 32)}.items():
        k, q, val = f(rng.standard_normal((B, C, d))), f(rng.standard_normal((B, T, d))), f(rng.standard_normal((B, C, v)))
        k.requires_grad_(), q.requires_grad_(), val.requires_grad_()
        att = get_attender("scaledot", d, v, v)
        o = att(k, q, val)
        w = f(rng.standard_normal(tuple(o.shape)))
        (o * w).sum().backward()
        res.update({f"attn_{tag}/keys": k.detach().numpy(), f"attn_{tag}/queries": q.detach().numpy(),
                    f"attn_{tag}/values": val.detach().numpy(), f"attn_{tag}/out": o.detach().numpy(),
                    f"attn_{tag}/w": w.numpy(), f"attn_{tag}/dkeys": k.grad.numpy(),
                    f"attn_{tag}/dqueries": q.grad.numpy(), f"attn_{tag}/dvalues": val.grad.numpy()})
    # MLP with the hidden clamp (in 128 -> hidden 256 -> out 128; and 2 -> 32 -> 64)
    for tag, (n_in, n_out, hid, nl, rows) in {"sq": (64, 64, 64, 3, 77), "clamp": (96, 48, 32, 2, 40),
                                               "skinny": (2, 64, 32, 1, 100), "wide": (128, 128, 256, 2, 65)}.items():
        m = MLP(n_in, n_out, hidden_size=hid, n_hidden_layers=nl)
        sd = {k: f(rng.uniform(-0.3, 0.3, tuple(v.shape))) for k, v in m.state_dict().items()}
        m.load_state_dict(sd)
        x = f(rng.standard_normal((rows, n_in)))
        x.requires_grad_()
        y = m(x)
        w = f(rng.standard_normal(tuple(y.shape)))
        (y * w).sum().backward()
        res.update({f"mlp_{tag}/x": x.detach().numpy(), f"mlp_{tag}/y": y.detach().numpy(), f"mlp_{tag}/w": w.numpy(),
                    f"mlp_{tag}/dx": x.grad.numpy()})
        for k, v in sd.items():
            res[f"mlp_{tag}/param/{k}"] = v.numpy()
        for k, p in m.named_parameters():
            res[f"mlp_{tag}/grad/{k}"] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g6_stages.npz"), **res)
    print("g6_stages: done", len(res))


def run_pretrained():
    """G7: shipped checkpoints (r=128, XY-encoder hidden 256) on seeded inputs.  Loaded with
    ``weights_only=True`` -- nothing from the file is executed."""
    rng = np.random.Generator(np.random.Philox(7))
    f = lambda a: torch.from_numpy(np.asarray(a, dtype="float32"))  # noqa: E731
    B, C, T = 4, 20, 128
    Xc, Yc, Xt = f(rng.uniform(-1, 1, (B, C, 1))), f(rng.standard_normal((B, C, 1))), f(rng.uniform(-1, 1, (B, T, 1)))
    res = {"X_cntxt": Xc.numpy(), "Y_cntxt": Yc.numpy(), "X_trgt": Xt.numpy()}
    path = os.path.join(REF, "results/pretrained/RBF_Kernel/CNP/run_0/params.pt")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    model = npf.CNP(1, 1, r_dim=128,
                    XYEncoder=merge_flat_input(partial(MLP, n_hidden_layers=2, hidden_size=256), is_sum_merge=True))
    model.load_state_dict(sd, strict=True)
    model.eval()
    with torch.no_grad():
        p, *_ = model(Xc, Yc, Xt)
    res["cnp_loc"], res["cnp_scale"] = p.base_dist.loc.numpy(), p.base_dist.scale.numpy()
    for k, v in sd.items():
        res[f"cnp_param/{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "g7_pretrained_cnp.npz"), **res)
    print("g7_pretrained_cnp: done; n_params", sum(v.numel() for v in sd.values()))


def run_pretrained_attn():
    """G9: the shipped RBF_Kernel AttnCNP / AttnLNP checkpoints (transformer attention, r = 128) on
    seeded inputs, eval mode; AttnLNP with an injected eps (2 latent samples).  ``weights_only=True``."""
    rng = np.random.Generator(np.random.Philox(9))
    f = lambda a: torch.from_numpy(np.asarray(a, dtype="float32"))  # noqa: E731
    B, C, T, r = 3, 25, 90, 128
    Xc, Yc, Xt = f(rng.uniform(-1, 1, (B, C, 1))), f(rng.standard_normal((B, C, 1))), f(rng.uniform(-1, 1, (B, T, 1)))
    eps = f(rng.standard_normal((2, B, 1, r)))
    res = {"X_cntxt": Xc.numpy(), "Y_cntxt": Yc.numpy(), "X_trgt": Xt.numpy(), "eps": eps.numpy()}
    kw = dict(r_dim=r, attention="transformer",
              XYEncoder=merge_flat_input(partial(MLP, n_hidden_layers=2, hidden_size=r), is_sum_merge=True),
              Decoder=merge_flat_input(partial(MLP, n_hidden_layers=4, hidden_size=r), is_sum_merge=True))
    for tag, cls, extra in (("attncnp", npf.AttnCNP, {}),
                            ("attnlnp", npf.AttnLNP, dict(n_z_samples_test=2, LatentDistribution=_latent_dist))):
        sd = torch.load(os.path.join(REF, f"results/pretrained/RBF_Kernel/{cls.__name__}/run_0/params.pt"),
                        map_location="cpu", weights_only=True)
        model = cls(1, 1, **kw, **extra)
        model.load_state_dict(sd, strict=True)
        model.eval()
        _EpsIndependent.eps = eps
        with torch.no_grad():
            p, *_ = model(Xc, Yc, Xt)
        res[f"{tag}_loc"], res[f"{tag}_scale"] = p.base_dist.loc.numpy(), p.base_dist.scale.numpy()
        # the same model evaluated by the reference in float64: trained weights make this forward pass
        # ill-conditioned (the fp32 result is ~1e-4 of max|loc| away from the exact one), so parity
        # tests on these checkpoints are stated against the fp64 value
        _EpsIndependent.eps = eps.double()
        with torch.no_grad():
            p64, *_ = model.double()(Xc.double(), Yc.double(), Xt.double())
        res[f"{tag}_loc64"], res[f"{tag}_scale64"] = p64.base_dist.loc.numpy(), p64.base_dist.scale.numpy()
        for k, v in sd.items():
            res[f"{tag}_param/{k}"] = v.numpy()
        print(f"g9 {tag}: n_params", sum(v.numel() for v in sd.values()), "loc", p.base_dist.loc.shape)
    np.savez_compressed(os.path.join(HERE, "g9_pretrained_attn.npz"), **res)


def run_selfattn_case():
    """G11: AttnCNP with the self-attention XY-encoder (attnnp.py:88-91,
    selfattn.py:10-100) and transformer cross attention.  The reference's own seeded init is stored in
    the fixture (no oracle restatement of this variant: the fixture pins the HIP path directly)."""
    torch.manual_seed(11)
    rng = np.random.Generator(np.random.Philox(11))
    f = lambda a: torch.from_numpy(np.asarray(a, dtype="float32"))  # noqa: E731
    B, C, T, r = 3, 14, 30, 32
    Xc, Yc = f(rng.uniform(-1, 1, (B, C, 1))), f(rng.standard_normal((B, C, 2)))
    Xt, Yt = f(rng.uniform(-1, 1, (B, T, 1))), f(rng.standard_normal((B, T, 2)))
    model = npf.AttnCNP(1, 2, r_dim=r, attention="transformer", is_self_attn=True)
    with torch.no_grad():  # non-trivial biases / LayerNorm weights
        for k, p in model.named_parameters():
            if k.endswith(".bias"):
                p.copy_(f(rng.uniform(-0.05, 0.05, tuple(p.shape))))
            elif "layer_norm" in k:
                p.copy_(f(rng.uniform(0.5, 1.5, tuple(p.shape))))
    res = {"X_cntxt": Xc.numpy(), "Y_cntxt": Yc.numpy(), "X_trgt": Xt.numpy(), "Y_trgt": Yt.numpy()}
    for k, v in model.state_dict().items():
        res[f"param/{k}"] = v.numpy().copy()
    model.train()
    crit = npf.CNPFLoss()
    out = model(Xc, Yc, Xt, Yt)
    loss = crit(out, Yt)
    loss.backward()
    res["loc"], res["scale"], res["loss"] = out[0].base_dist.loc.detach().numpy(), out[0].base_dist.scale.detach().numpy(), loss.detach().numpy()
    for k, p in model.named_parameters():
        res[f"grad/{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g11_attncnp_selfattn.npz"), **res)
    print("g11_attncnp_selfattn: loss", float(loss), "n_params", sum(p.numel() for p in model.parameters()))


def run_eval_case():
    """G12: the evaluation protocol of utils/evaluate.py:9-28 on the reference's modules --
    evaluation mode, ``reduction=None``, 32 latent samples at test time (``n_z_samples_test=32``), the per-task
    log-likelihood (minus the criterion) of two batches, concatenated.  Training criterion ELBO (so that the
    evaluation really exercises losses.py:65-69: NLL with the importance weights dropped) and SUMO."""
    res = {}
    for tag, case, crit_cls in (("attnlnp", specs.EVAL_CASES["attnlnp"], npf.ELBOLossLNPF),
                                ("lnp", specs.EVAL_CASES["lnp"], npf.SUMOLossLNPF),
                                ("cnp", specs.EVAL_CASES["cnp"], npf.CNPFLoss)):
        model = build_reference(case)
        model.load_state_dict(specs.make_params(case), strict=True)
        model.eval()
        crit = crit_cls()
        crit.reduction = None  # (as eval_loglike does; the reference's SUMOLossLNPF.__init__ drops its keyword arguments)
        crit.eval()
        ll = []
        for i in range(2):
            inp = specs.make_inputs(case, seed=5000 + i)
            if "eps" in inp:
                _EpsIndependent.eps = inp["eps"]
            with torch.no_grad():
                out = model(inp["X_cntxt"], inp["Y_cntxt"], inp["X_trgt"], inp["Y_trgt"])
                ll.append(-crit(out, inp["Y_trgt"]))
        res[f"{tag}_loglike"] = torch.cat(ll, 0).numpy()
        print(f"g12 {tag}: log-likelihoods", res[f"{tag}_loglike"])
    np.savez_compressed(os.path.join(HERE, "g12_eval_loglike.npz"), **res)


if __name__ == "__main__":
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    if only:  # e.g. `make_golden.py g8_ g9` regenerates the matching cases only
        for name, case in specs.CASES.items():
            if any(name.startswith(o) for o in only):
                run_case(name, case, store_params=False, store_full_grads=case["r"] < 256)
        if any(o.startswith("g9") for o in only):
            run_pretrained_attn()
        if any(o.startswith("g11") for o in only):
            run_selfattn_case()
        if any(o.startswith("g12") for o in only):
            run_eval_case()
        for name, case in specs.VARIANT_CASES.items():
            if any(name.startswith(o) for o in only):
                run_case(name, case, store_params=True, store_full_grads=True, own_init=True)
        sys.exit(0)
    small_full = {"g1_cnp_c1", "g2_lnp_both_c1", "g2_lnp_latent_c1", "g3s_attncnp_r64", "g4s_attnlnp_r64",
                  "g4s_attnlnp_r64_noqzcct"}
    for name, case in specs.CASES.items():
        big = case["r"] >= 256
        run_case(name, case, store_params=(name == "g1_cnp_c1"), store_full_grads=not big,
                 adam_step=(name == "g1_cnp_c1"))
    run_decode_case()
    run_stage_cases()
    run_pretrained()
    run_pretrained_attn()
    run_selfattn_case()
    run_eval_case()
    for name, case in specs.VARIANT_CASES.items():
        run_case(name, case, store_params=True, store_full_grads=True, own_init=True)
