"""Pins the CPU oracle (oracle/npf_oracle.py) against golden vectors produced by the
reference itself (tests/golden/make_golden.py).  Bit-exact on the small cases, stated
fp32 tolerance on the config-2 sized ones."""
import numpy as np
import pytest
import torch

import specs
from oracle import npf_oracle as O

torch.set_num_threads(8)

SMALL = [n for n, c in specs.CASES.items() if c["r"] < 256]
BIG = [n for n, c in specs.CASES.items() if c["r"] >= 256]
LOSSES = {"cnpf": O.cnpf_loss, "elbo": O.elbo_loss, "nll": O.nll_loss, "sumo": O.sumo_loss}


def run_oracle(case, training=True, with_grad=True, params=None):
    cfg = specs.cfg_of(case)
    params = specs.make_params(case) if params is None else params
    params = {k: v.clone().requires_grad_(with_grad) for k, v in params.items()}
    inp = specs.make_inputs(case)
    out = O.forward(cfg, params, inp["X_cntxt"], inp["Y_cntxt"], inp["X_trgt"],
                    inp["Y_trgt"] if training else None, eps=inp.get("eps"), n_z=case.get("n_z", 1),
                    training=training)
    loss = None
    if training:
        loss = LOSSES[specs.loss_name(case)](out, inp["Y_trgt"])
        if with_grad:
            loss.backward()
    return params, out, loss


@pytest.mark.parametrize("name", SMALL)
def test_oracle_bit_exact_small(name):
    case = specs.CASES[name]
    g = specs.load_golden(name)
    params, out, loss = run_oracle(case)
    assert np.array_equal(out["loc"].detach().numpy(), g["loc"])
    assert np.array_equal(out["scale"].detach().numpy(), g["scale"])
    assert np.array_equal(loss.detach().numpy(), g["loss"])
    if "z_samples" in g:
        assert np.array_equal(out["z_samples"].detach().numpy(), g["z_samples"])
        assert np.array_equal(out["q_zCc"][0].detach().numpy(), g["q_zCc_loc"])
        assert np.array_equal(out["q_zCc"][1].detach().numpy(), g["q_zCc_scale"])
    if "q_zCct_loc" in g:
        assert np.array_equal(out["q_zCct"][1].detach().numpy(), g["q_zCct_scale"])
    for k, p in params.items():
        ref = g[f"grad/{k}"]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(ref)
        # gradients: same op sequence, same autograd formulas -> tight, but the loss
        # restatement sums in a slightly different association, so allow last-ulp noise
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6 * max(1e-30, np.abs(ref).max()), err_msg=k)


@pytest.mark.parametrize("name", list(specs.VARIANT_CASES))
def test_oracle_bit_exact_variants(name):
    """G14: residual MLPs, the concatenating XY-encoder merge, x_transf_dim != r_dim -- on the parameters the
    reference itself constructed (stored in the fixture)."""
    case = specs.VARIANT_CASES[name]
    g = specs.load_golden(name)
    O.DROPOUT_MASKS = iter(specs.golden_dropout_masks(g))
    try:
        params, out, loss = run_oracle(case, params=specs.golden_params(g))
    finally:
        O.DROPOUT_MASKS = None
    assert np.array_equal(out["loc"].detach().numpy(), g["loc"])
    assert np.array_equal(out["scale"].detach().numpy(), g["scale"])
    assert np.array_equal(loss.detach().numpy(), g["loss"])
    if "z_samples" in g:
        assert np.array_equal(out["z_samples"].detach().numpy(), g["z_samples"])
    for k, p in params.items():
        ref = g[f"grad/{k}"]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(ref)
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6 * max(1e-30, np.abs(ref).max()), err_msg=k)
    _, out_e, _ = run_oracle(case, training=False, with_grad=False, params=specs.golden_params(g))
    assert np.array_equal(out_e["loc"].detach().numpy(), g["eval_loc"])
    assert np.array_equal(out_e["scale"].detach().numpy(), g["eval_scale"])


@pytest.mark.parametrize("name", SMALL)
def test_oracle_eval_mode(name):
    case = specs.CASES[name]
    g = specs.load_golden(name)
    _, out, _ = run_oracle(case, training=False, with_grad=False)
    assert np.array_equal(out["loc"].detach().numpy(), g["eval_loc"])
    assert np.array_equal(out["scale"].detach().numpy(), g["eval_scale"])


@pytest.mark.parametrize("name", BIG)
def test_oracle_config2_tolerance(name):
    """fp32 tolerance of SURVEY.md 8c: max|d| <= 1e-5 max|ref| and allclose(rtol 1e-5,
    atol 1e-6 max|ref|); sigma element-wise rel <= 1e-5."""
    case = specs.CASES[name]
    g = specs.load_golden(name)
    params, out, loss = run_oracle(case)
    for key in ("loc", "scale"):
        got, ref = out[key].detach().numpy(), g[key]
        m = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 1e-5 * m
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6 * m)
    np.testing.assert_allclose(out["scale"].detach().numpy(), g["scale"], rtol=1e-5)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    for k, p in params.items():
        n_ref = float(g[f"gradnorm/{k}"])
        n_got = p.grad.double().norm().item()
        assert abs(n_got - n_ref) <= 1e-4 * max(n_ref, 1e-12), k
        head = g[f"gradhead/{k}"]
        np.testing.assert_allclose(p.grad.reshape(-1)[:64].numpy(), head, rtol=1e-3,
                                   atol=1e-4 * np.abs(head).max() + 1e-12, err_msg=k)


def test_oracle_adam_step_g1():
    case = specs.CASES["g1_cnp_c1"]
    g = specs.load_golden("g1_cnp_c1")
    params, _, _ = run_oracle(case)
    leaves = list(params.values())
    opt = torch.optim.Adam(leaves, lr=1e-3)
    opt.step()
    for k, p in params.items():
        np.testing.assert_allclose(p.detach().numpy(), g[f"adam1/{k}"], rtol=1e-6, atol=1e-7, err_msg=k)


def test_stored_params_match_seeded():
    g = specs.load_golden("g1_cnp_c1")
    params = specs.make_params(specs.CASES["g1_cnp_c1"])
    for k, v in params.items():
        assert np.array_equal(v.numpy(), g[f"param/{k}"]), k
    assert int(g["n_params"]) == 35876


def test_param_counts_match_reference_notebooks():
    # c1 CNP 35 876; c1 LNP 56 612; c2 AttnCNP 799 588; c2 AttnLNP 1 128 292
    for name, n in [("g1_cnp_c1", 35876), ("g2_lnp_both_c1", 56612), ("g3_attncnp_c2", 799588),
                    ("g4_attnlnp_c2", 1128292)]:
        p = specs.make_params(specs.CASES[name])
        assert sum(v.numel() for v in p.values()) == n
        assert int(specs.load_golden(name)["n_params"]) == n


def test_oracle_decode_r512():
    g = specs.load_golden("g5_decode_r512")
    cfg, dparams = specs.make_decode_params()
    inp = specs.make_decode_inputs()
    loc, scale = O.decode(cfg, dparams, inp["X_trgt_enc"], inp["R_trgt"])
    for got, ref in ((loc.numpy(), g["loc"]), (scale.numpy(), g["scale"])):
        m = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 1e-5 * m


def test_oracle_stage_attention_and_mlp():
    g = specs.load_golden("g6_stages")
    for tag in ("a", "b", "c"):
        k, q, v = (torch.from_numpy(g[f"attn_{tag}/{n}"]).requires_grad_() for n in ("keys", "queries", "values"))
        o = O.scaledot_attend(k, q, v)
        assert np.array_equal(o.detach().numpy(), g[f"attn_{tag}/out"])
        (o * torch.from_numpy(g[f"attn_{tag}/w"])).sum().backward()
        np.testing.assert_allclose(k.grad.numpy(), g[f"attn_{tag}/dkeys"], rtol=1e-5, atol=1e-6)
    for tag in ("sq", "clamp", "skinny", "wide"):
        params = {kk.split("/param/")[1]: torch.from_numpy(vv) for kk, vv in g.items()
                  if kk.startswith(f"mlp_{tag}/param/")}
        params = {f"m.{kk}": vv for kk, vv in params.items()}
        x = torch.from_numpy(g[f"mlp_{tag}/x"])
        assert np.array_equal(O.mlp(params, "m", x).numpy(), g[f"mlp_{tag}/y"])


def test_oracle_pretrained_cnp():
    g = specs.load_golden("g7_pretrained_cnp")
    params = {k.split("cnp_param/")[1]: torch.from_numpy(v) for k, v in g.items() if k.startswith("cnp_param/")}
    cfg = O.OracleConfig(kind="CNP", x_dim=1, y_dim=1, r_dim=128)
    out = O.forward(cfg, params, torch.from_numpy(g["X_cntxt"]), torch.from_numpy(g["Y_cntxt"]),
                    torch.from_numpy(g["X_trgt"]), training=False)
    assert np.array_equal(out["loc"].numpy(), g["cnp_loc"])
    assert np.array_equal(out["scale"].numpy(), g["cnp_scale"])


def test_oracle_pretrained_attn_checkpoints():
    """G9: the shipped RBF_Kernel AttnCNP / AttnLNP checkpoints (transformer attention, r = 128),
    eval mode, against the reference's own outputs."""
    g = specs.load_golden("g9_pretrained_attn")
    Xc, Yc, Xt = (torch.from_numpy(g[k]) for k in ("X_cntxt", "Y_cntxt", "X_trgt"))
    for tag, kind in (("attncnp", "AttnCNP"), ("attnlnp", "AttnLNP")):
        params = {k[len(tag) + 7:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(f"{tag}_param/")}
        cfg = O.OracleConfig(kind=kind, x_dim=1, y_dim=1, r_dim=128, attention="transformer")
        out = O.forward(cfg, params, Xc, Yc, Xt, None, eps=torch.from_numpy(g["eps"]) if kind == "AttnLNP" else None,
                        n_z=2, training=False)
        for key in ("loc", "scale"):
            got, ref = out[key].detach().numpy(), g[f"{tag}_{key}"]
            assert got.shape == ref.shape
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6 * np.abs(ref).max(), err_msg=f"{tag} {key}")


def test_oracle_eval_loglike_protocol_g12():
    """G12: the reference's evaluation protocol (utils/evaluate.py:9-28: evaluation mode, reduction None, 32 latent
    samples at test time, NLL without importance weights, sign flipped) restated with the oracle's forward + nll_loss."""
    g = specs.load_golden("g12_eval_loglike")
    for tag, case in specs.EVAL_CASES.items():
        cfg = specs.cfg_of(case)
        params = specs.make_params(case)
        ll = []
        for i in range(2):
            inp = specs.make_inputs(case, seed=5000 + i)
            with torch.no_grad():
                out = O.forward(cfg, params, inp["X_cntxt"], inp["Y_cntxt"], inp["X_trgt"], inp["Y_trgt"], eps=inp.get("eps"),
                                n_z=case.get("n_z", 1), training=False)
                out = dict(out, q_zCct=None)  # is_force_mle_eval (losses.py:65-69)
                ll.append(-O.nll_loss(out, inp["Y_trgt"], reduction=None))
        np.testing.assert_allclose(torch.cat(ll).numpy(), g[f"{tag}_loglike"], rtol=2e-6, err_msg=tag)


def test_oracle_bf16_mode_backward_formulas():
    """The hand-written backward of the bf16 emulation (``_LinearBf16``): with operands and upstream gradient that are
    exactly representable in bf16 the rounding is the identity, and the result must equal plain autograd."""
    g = torch.Generator().manual_seed(0)
    r16 = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    x, W, b = r16(torch.randn(7, 12, generator=g)), r16(torch.randn(5, 12, generator=g)), torch.randn(5, generator=g)
    dy = r16(torch.randn(7, 5, generator=g))
    ref_in = [t.clone().requires_grad_(True) for t in (x, W, b)]
    torch.nn.functional.linear(*ref_in).backward(dy)
    got_in = [t.clone().requires_grad_(True) for t in (x, W, b)]
    with O.matmul_mode("bf16"):
        O.linear(*got_in).backward(dy)
    for a, c in zip(got_in, ref_in):
        torch.testing.assert_close(a.grad, c.grad, rtol=1e-6, atol=1e-6)
    # and it really rounds: an input that is not representable changes the product
    with O.matmul_mode("bf16"):
        y16 = O.linear(x + 1e-3, W, b)
    assert not torch.allclose(y16, torch.nn.functional.linear(x + 1e-3, W, b), rtol=0, atol=1e-6)
