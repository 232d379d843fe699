"""GPU parity of the full models (forward, loss, backward) against the golden vectors the
reference produced (tests/golden/*.npz) -- through this package's drop-in classes, i.e.
through the C ABI.  Tolerances (fp32): loc/scale max|d| <= 1e-5 max|ref| and
sigma element-wise rel <= 1e-5; gradients max|d| <= 1e-4 max|ref| (they are sums of up to
2.6e5 fp32 terms whose order differs from the CPU BLAS)."""
import numpy as np
import pytest
import torch

import specs
from helpers import EpsIndependent, assert_close, build_loss, build_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(case, train=True, params=None):
    model = build_model(case, DEV, params=params)
    inp = {k: v.to(DEV) for k, v in specs.make_inputs(case).items()}
    if "eps" in inp:
        EpsIndependent.eps = inp["eps"]
    crit = build_loss(case)
    model.train(train)
    crit.train(train)
    if train:
        out = model(inp["X_cntxt"], inp["Y_cntxt"], inp["X_trgt"], inp["Y_trgt"])
        loss = crit(out, inp["Y_trgt"])
        loss.backward()
    else:
        with torch.no_grad():
            out = model(inp["X_cntxt"], inp["Y_cntxt"], inp["X_trgt"])
        loss = None
    return model, out, loss


@pytest.mark.parametrize("name", list(specs.CASES))
def test_model_train_step_parity(name):
    case = specs.CASES[name]
    g = specs.load_golden(name)
    model, out, loss = _run(case)
    p_yCc, z_samples, q_zCc, q_zCct = out
    assert_close(p_yCc.base_dist.loc, g["loc"], what="loc")
    assert_close(p_yCc.base_dist.scale, g["scale"], what="scale")
    np.testing.assert_allclose(p_yCc.base_dist.scale.detach().cpu().numpy(), g["scale"], rtol=1e-5)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=2e-5)
    if "z_samples" in g:
        assert_close(z_samples, g["z_samples"], what="z_samples")
        assert_close(q_zCc.base_dist.loc, g["q_zCc_loc"], what="q_zCc.loc")
        assert_close(q_zCc.base_dist.scale, g["q_zCc_scale"], what="q_zCc.scale")
    if "q_zCct_loc" in g:
        assert_close(q_zCct.base_dist.loc, g["q_zCct_loc"], what="q_zCct.loc")
    full = f"grad/x_encoder.out.weight" in g
    for k, p in model.named_parameters():
        grad = p.grad if p.grad is not None else torch.zeros_like(p)
        if full:
            assert_close(grad, g[f"grad/{k}"], tol=1e-4, what=f"grad {k}")
        else:
            n_ref = float(g[f"gradnorm/{k}"])
            assert abs(grad.double().norm().item() - n_ref) <= 1e-4 * max(n_ref, 1e-12), k
            head = g[f"gradhead/{k}"]
            np.testing.assert_allclose(grad.reshape(-1)[:64].cpu().numpy(), head, rtol=1e-3,
                                       atol=1e-4 * np.abs(head).max() + 1e-12, err_msg=k)


@pytest.mark.parametrize("name", list(specs.VARIANT_CASES))
def test_model_variants_parity(name):
    """G14: ``MLP(is_res=True)`` (mlp.py:100-104), the concatenating XY-encoder merge (encoders.py:180-181) and
    ``x_transf_dim`` != ``r_dim`` (base.py:126-131) against the reference's outputs, loss, every gradient and its
    evaluation-mode outputs -- on the parameters the reference constructed (stored in the fixture)."""
    import npf_gwwaveform_amd as A

    case = specs.VARIANT_CASES[name]
    g = specs.load_golden(name)
    A.MLP.mask_source = iter(specs.golden_dropout_masks(g))  # (the masks the reference drew; none without dropout)
    try:
        model, out, loss = _run(case, params=specs.golden_params(g))
    finally:
        A.MLP.mask_source = None
    p_yCc, z_samples, q_zCc, q_zCct = out
    assert_close(p_yCc.base_dist.loc, g["loc"], what="loc")
    assert_close(p_yCc.base_dist.scale, g["scale"], what="scale")
    np.testing.assert_allclose(p_yCc.base_dist.scale.detach().cpu().numpy(), g["scale"], rtol=1e-5)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=2e-5)
    if "z_samples" in g:
        assert_close(z_samples, g["z_samples"], what="z_samples")
        assert_close(q_zCc.base_dist.scale, g["q_zCc_scale"], what="q_zCc.scale")
    for k, p in model.named_parameters():
        grad = p.grad if p.grad is not None else torch.zeros_like(p)
        assert_close(grad, g[f"grad/{k}"], tol=1e-4, what=f"grad {k}")
    _, out_e, _ = _run(case, train=False, params=specs.golden_params(g))
    assert_close(out_e[0].base_dist.loc, g["eval_loc"], what="eval loc")
    assert_close(out_e[0].base_dist.scale, g["eval_scale"], what="eval scale")


@pytest.mark.parametrize("name", ["g3_attncnp_c2", "g4_attnlnp_c2"])
def test_model_parity_without_split_mlp_stacks(name, monkeypatch):
    """NPF_NO_MLP_X6=1: the 256 -> 256 layers of the flat MLPs inside the fp32 chains (``v_mfma_f32_16x16x4_f32``) instead
    of on ``npf_mlp_x6_run`` (the default since round 2: fp32 products as three exact bf16 terms on the bf16 pipe,
    csrc/mlp_x6_kernel.hip) -- the reference's outputs, loss and gradients at the same fp32 tolerances on either path."""
    from npf_gwwaveform_amd import mlp_x6

    monkeypatch.setattr(mlp_x6, "ENABLED", False)
    test_model_train_step_parity(name)


def test_mlp_dropout_with_device_masks():
    """``MLP(dropout=p)`` with masks drawn on the device (no injected masks): evaluation mode is the identity, training
    mode zeroes about p of the hidden units' contributions and rescales by 1 / (1 - p) (mean preserved), the backward
    pass uses the same mask (a unit dropped in the forward pass gets no gradient through it)."""
    import npf_gwwaveform_amd as A

    torch.manual_seed(5)
    p = 0.5
    m = A.MLP(16, 8, hidden_size=64, n_hidden_layers=1, dropout=p).to(DEV)
    ref = A.MLP(16, 8, hidden_size=64, n_hidden_layers=1).to(DEV)
    ref.load_state_dict(m.state_dict())
    x = torch.randn(4000, 16, device=DEV)
    with torch.no_grad():
        assert torch.equal(m.eval()(x), ref.eval()(x))
        y_ref = ref(x)
        m.train()
        y1, y2 = m(x), m(x)
    assert not torch.equal(y1, y2)                                   # fresh masks per call
    # out layer is linear in the (masked, rescaled) hidden units: E[y] = y_ref - b + b
    assert_close(y1.mean(0), y_ref.mean(0), tol=5e-2, what="dropout preserves the mean")
    # exact check against a mask reconstructed from the module's own hook point: one hidden layer, so
    # y = W_out (keep * h / (1 - p)) + b for some 0/1 keep; solve for keep on a few rows via the gradient
    x1 = x[:64].clone().requires_grad_(True)
    torch.manual_seed(11)
    y = m(x1)
    y.sum().backward()
    torch.manual_seed(11)
    with torch.no_grad():
        y_again = m(x1)
    assert torch.equal(y, y_again)                                   # same generator state -> same masks
    h = torch.relu(x1.detach() @ m.to_hidden.weight.t() + m.to_hidden.bias)
    g_h = (m.out.weight.sum(0) / (1 - p)).expand_as(h)               # d sum(y) / d h where kept
    # the input gradient is W_hid^T (keep * relu' * g_h): recover keep * relu' by least squares is overkill -- check the
    # two extreme hypotheses instead: all kept would give gx_all; the actual gradient must be a masked version of it
    gx_all = ((h > 0).float() * g_h) @ m.to_hidden.weight
    assert not torch.allclose(x1.grad, gx_all, rtol=1e-3, atol=1e-5)
    frac = float(x1.grad.norm() / gx_all.detach().norm())
    assert 0.4 < frac < 1.1, frac                                    # about sqrt(1 - p) of the units' share survives


@pytest.mark.parametrize("name", ["g1_cnp_c1", "g2_lnp_both_c1", "g3s_attncnp_r64", "g4s_attnlnp_r64", "g6_cnp_homosk",
                                  "g6_attncnp_ragged", "g3_attncnp_c2"])
def test_model_eval_parity(name):
    case = specs.CASES[name]
    g = specs.load_golden(name)
    _, out, _ = _run(case, train=False)
    assert_close(out[0].base_dist.loc, g["eval_loc"], what="eval loc")
    assert_close(out[0].base_dist.scale, g["eval_scale"], what="eval scale")


def test_adam_step_matches_reference_g1():
    case = specs.CASES["g1_cnp_c1"]
    g = specs.load_golden("g1_cnp_c1")
    model, _, _ = _run(case)
    torch.optim.Adam(model.parameters(), lr=1e-3).step()
    for k, p in model.named_parameters():
        # Adam's first step moves every weight by lr * sign(grad): compare the moved weights
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"adam1/{k}"], rtol=1e-5, atol=2e-6, err_msg=k)


def test_stage_api_matches_forward():
    """x_encoder / encode_globally / trgt_dependent_representation / decode called one by
    one (as utils/ntbks_helpers.py:485-518 does) give the same result as forward()."""
    case = specs.CASES["g3s_attncnp_r64"]
    g = specs.load_golden("g3s_attncnp_r64")
    model = build_model(case, DEV).eval()
    inp = {k: v.to(DEV) for k, v in specs.make_inputs(case).items()}
    with torch.no_grad():
        Xc = model.x_encoder(inp["X_cntxt"])
        Xt = model.x_encoder(inp["X_trgt"])
        R = model.encode_globally(Xc, inp["Y_cntxt"])
        R_trgt = model.trgt_dependent_representation(Xc, None, R, Xt)
        p = model.decode(Xt, R_trgt)
    assert_close(p.base_dist.loc, g["eval_loc"], what="stage loc")
    assert_close(p.base_dist.scale, g["eval_scale"], what="stage scale")


def test_pretrained_cnp_checkpoint():
    """Real weights shipped by the reference (results/pretrained/RBF_Kernel/CNP/run_0)."""
    from functools import partial

    import npf_gwwaveform_amd as A

    g = specs.load_golden("g7_pretrained_cnp")
    params = {k.split("cnp_param/")[1]: torch.from_numpy(v) for k, v in g.items() if k.startswith("cnp_param/")}
    model = A.CNP(1, 1, r_dim=128, XYEncoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=2, hidden_size=256),
                                                                is_sum_merge=True))
    model.load_state_dict(params, strict=True)
    model = model.to(DEV).eval()
    with torch.no_grad():
        p, *_ = model(torch.from_numpy(g["X_cntxt"]).to(DEV), torch.from_numpy(g["Y_cntxt"]).to(DEV),
                      torch.from_numpy(g["X_trgt"]).to(DEV))
    assert_close(p.base_dist.loc, g["cnp_loc"], what="pretrained loc")
    assert_close(p.base_dist.scale, g["cnp_scale"], what="pretrained scale")


def test_out_of_range_features_raise_in_training():
    case = specs.CASES["g1_cnp_c1"]
    model = build_model(case, DEV).train()
    inp = {k: v.to(DEV) for k, v in specs.make_inputs(case).items()}
    with pytest.raises(ValueError, match=r"\[-1,1\]"):
        model(inp["X_cntxt"] * 3, inp["Y_cntxt"], inp["X_trgt"], inp["Y_trgt"])


def test_unknown_paths_raise_like_reference():
    import npf_gwwaveform_amd as A

    with pytest.raises(ValueError, match="Unknown encoded_path"):
        A.CNP(1, 1, encoded_path="nope")
    with pytest.raises(ValueError, match="Unknown attention"):
        A.get_attender("nope", 8, 8, 8)


def test_decode_only_r512_matches_reference():
    """BASELINE config 5 shape at reduced batch: decode(X_trgt_enc, R_trgt) with a 512-wide,
    4-layer decoder and 4096 targets (base.py:327-367) -- the <= 512-feature variant of the chain
    kernel -- against the reference's output (tests/golden/g5_decode_r512.npz)."""
    import warnings
    from functools import partial

    import npf_gwwaveform_amd as A

    g = specs.load_golden("g5_decode_r512")
    cfg, dparams = specs.make_decode_params()
    inp = specs.make_decode_inputs()
    case = specs.DECODE_CASE
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = A.CNP(case["dx"], case["dy"], r_dim=case["r"],
                      Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=case["L_dec"], hidden_size=case["r"]),
                                                 is_sum_merge=True))
    sd = model.state_dict()
    sd.update(dparams)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    with torch.no_grad():
        p = model.decode(inp["X_trgt_enc"].to(DEV), inp["R_trgt"].to(DEV))
    assert_close(p.base_dist.loc, g["loc"], what="decode r512 loc")
    assert_close(p.base_dist.scale, g["scale"], what="decode r512 scale")
    np.testing.assert_allclose(p.base_dist.scale.cpu().numpy(), g["scale"], rtol=1e-5)


def test_training_wider_than_256_features():
    """A 512-wide MLP trains (the reference has no width limit, npf/architectures/mlp.py:44-93): forward on the 32-block
    chain instance, input gradient by its dgrad, weight gradients as 256 x 256 block jobs of the wgrad kernel."""
    import npf_gwwaveform_amd as A

    torch.manual_seed(0)
    m = A.MLP(512, 512, hidden_size=512, n_hidden_layers=2).to(DEV)
    x = torch.randn(70, 512, device=DEV, requires_grad=True)
    w = torch.randn(70, 512, device=DEV)
    (m(x) * w).sum().backward()
    xr = x.detach().clone().requires_grad_(True)
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in m.named_parameters()}
    F = torch.nn.functional
    h = torch.relu(F.linear(xr, ps["to_hidden.weight"], ps["to_hidden.bias"]))
    h = torch.relu(F.linear(h, ps["linears.0.weight"], ps["linears.0.bias"]))
    (F.linear(h, ps["out.weight"], ps["out.bias"]) * w).sum().backward()
    assert_close(x.grad, xr.grad, tol=1e-4, what="dx of a 512-wide MLP")
    for k, p in m.named_parameters():
        assert_close(p.grad, ps[k].grad, tol=1e-4, what=f"grad {k}")


def test_training_inputs_outside_unit_range_raise_like_the_reference():
    """base.py:241-247: X outside [-1, 1] (or NaN) raises ValueError in training mode only."""
    case = specs.CASES["g1_cnp_c1"]
    model = build_model(case, DEV).train()
    inp = {k: v.to(DEV) for k, v in specs.make_inputs(case).items()}
    model(inp["X_cntxt"], inp["Y_cntxt"], inp["X_trgt"], inp["Y_trgt"])
    bad = inp["X_trgt"].clone()
    bad[0, 0, 0] = 1.5
    with pytest.raises(ValueError, match=r"\[-1,1\]"):
        model(inp["X_cntxt"], inp["Y_cntxt"], bad, inp["Y_trgt"])
    nan = inp["X_cntxt"].clone()
    nan[1, 2, 0] = float("nan")
    with pytest.raises(ValueError):
        model(nan, inp["Y_cntxt"], inp["X_trgt"], inp["Y_trgt"])
    model.eval()
    with torch.no_grad():
        model(inp["X_cntxt"], inp["Y_cntxt"], bad, inp["Y_trgt"])  # no check outside training


def _c2_model_and_batch(B, seed=7):
    """BASELINE config 2 at full size: AttnCNP r=256 L=4, 256 context / 1024 target points."""
    from npf_gwwaveform_amd.train import synthetic_waveform_batch

    case = dict(specs.CASES["g3_attncnp_c2"], B=B)
    model = build_model(case, DEV).train()
    return model, synthetic_waveform_batch(B, case["C"], case["T"], seed, DEV)


@pytest.mark.parametrize("B", [256, 1024], ids=["c2_256_tasks", "c4_1024_tasks_per_rank"])
def test_full_size_config2_properties(B):
    """Size-independent properties at BASELINE config 2's full size (256 tasks x 1024 targets) and at config 4's per-rank
    size (1024 tasks x 1024 targets, fp32: what one of the 8 ranks of the data-parallel run computes), where
    the CPU oracle is too slow to be the checker:
      * tasks are independent (base.py:177-239 has no cross-task op before the loss mean): a task
        evaluated alone gives the same loc / sigma as inside the full batch;
      * a target's prediction does not depend on the other targets of its task;
      * permuting the context points leaves the predictions unchanged (attention sums over them);
      * the gradient of the batch-mean loss is the mean of the two half-batch gradients."""
    import npf_gwwaveform_amd as A

    model, batch = _c2_model_and_batch(B)
    crit = A.CNPFLoss()
    p = model(batch["X_cntxt"], batch["Y_cntxt"], batch["X_trgt"], batch["Y_trgt"])[0]
    loc, scale = p.base_dist.loc.detach(), p.base_dist.scale.detach()
    assert torch.isfinite(loc).all() and torch.isfinite(scale).all() and (scale >= 0.01).all()
    for i in (0, 101, B - 1):
        one = {k: v[i:i + 1] for k, v in batch.items()}
        pi = model(one["X_cntxt"], one["Y_cntxt"], one["X_trgt"], one["Y_trgt"])[0]
        assert_close(pi.base_dist.loc, loc[:, i:i + 1], tol=1e-6, what=f"task {i} alone: loc")
        assert_close(pi.base_dist.scale, scale[:, i:i + 1], tol=1e-6, what=f"task {i} alone: scale")
    sub = slice(100, 357)
    ps = model(batch["X_cntxt"], batch["Y_cntxt"], batch["X_trgt"][:, sub], batch["Y_trgt"][:, sub])[0]
    assert_close(ps.base_dist.loc, loc[:, :, sub], tol=1e-6, what="target subset: loc")
    assert_close(ps.base_dist.scale, scale[:, :, sub], tol=1e-6, what="target subset: scale")
    perm = torch.randperm(batch["X_cntxt"].shape[1], device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    pp = model(batch["X_cntxt"][:, perm], batch["Y_cntxt"][:, perm], batch["X_trgt"], batch["Y_trgt"])[0]
    assert_close(pp.base_dist.loc, loc, tol=1e-5, what="context permutation: loc")
    assert_close(pp.base_dist.scale, scale, tol=1e-5, what="context permutation: scale")

    def grads(lo, hi):
        model.zero_grad(set_to_none=True)
        b = {k: v[lo:hi] for k, v in batch.items()}
        out = model(b["X_cntxt"], b["Y_cntxt"], b["X_trgt"], b["Y_trgt"])
        loss = crit(out, b["Y_trgt"])
        loss.backward()
        return loss.item(), {k: q.grad.clone() for k, q in model.named_parameters()}

    l_full, g_full = grads(0, B)
    l_a, g_a = grads(0, B // 2)
    l_b, g_b = grads(B // 2, B)
    np.testing.assert_allclose(l_full, 0.5 * (l_a + l_b), rtol=1e-5)
    for k in g_full:
        assert_close(g_full[k], 0.5 * (g_a[k] + g_b[k]), tol=1e-4, what=f"grad linearity {k}")


def test_full_size_config5_decode_properties():
    """BASELINE config 5 per-GPU size (512 waveforms x 4096 targets, r = 512): decoding a subset of the
    waveforms / targets gives the same loc / sigma as the corresponding part of the full decode."""
    import warnings
    from functools import partial

    import npf_gwwaveform_amd as A

    r, L, B, T = 512, 4, 512, 4096
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = A.CNP(1, 2, r_dim=r, Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, hidden_size=r),
                                                                  is_sum_merge=True)).to(DEV).eval()
    g = torch.Generator(device=DEV).manual_seed(5)
    Xt = torch.randn(B, T, r, device=DEV, generator=g) * 0.5
    R = torch.randn(1, B, T, r, device=DEV, generator=g) * 0.5
    with torch.no_grad():
        p = model.decode(Xt, R)
        loc, scale = p.base_dist.loc, p.base_dist.scale
        assert torch.isfinite(loc).all() and (scale >= 0.01).all()
        ps = model.decode(Xt[7:9, 1000:1100].contiguous(), R[:, 7:9, 1000:1100].contiguous())
    assert_close(ps.base_dist.loc, loc[:, 7:9, 1000:1100], tol=1e-6, what="decode subset: loc")
    assert_close(ps.base_dist.scale, scale[:, 7:9, 1000:1100], tol=1e-6, what="decode subset: scale")


def test_pretrained_attn_checkpoints_match_reference():
    """G9: the shipped RBF_Kernel AttnCNP / AttnLNP checkpoints (transformer
    attention, r = 128) loaded with ``strict=True`` into this package's classes, eval mode, against
    the reference's outputs on the same seeded inputs (tests/golden/g9_pretrained_attn.npz)."""
    import warnings
    from functools import partial

    import npf_gwwaveform_amd as A
    from helpers import eps_latent_dist

    g = specs.load_golden("g9_pretrained_attn")
    Xc, Yc, Xt = (torch.from_numpy(g[k]).to(DEV) for k in ("X_cntxt", "Y_cntxt", "X_trgt"))
    r = 128
    kw = dict(r_dim=r, attention="transformer",
              XYEncoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=2, hidden_size=r), is_sum_merge=True),
              Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=4, hidden_size=r), is_sum_merge=True))
    for tag, cls, extra in (("attncnp", A.AttnCNP, {}),
                            ("attnlnp", A.AttnLNP, dict(n_z_samples_test=2, LatentDistribution=eps_latent_dist))):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = cls(1, 1, **kw, **extra)
        sd = {k[len(tag) + 7:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(f"{tag}_param/")}
        model.load_state_dict(sd, strict=True)
        model = model.to(DEV).eval()
        EpsIndependent.eps = torch.from_numpy(g["eps"]).to(DEV)
        with torch.no_grad():
            p, *_ = model(Xc, Yc, Xt)
        # Trained weights make this forward pass ill-conditioned: the reference's own fp32 output is
        # ~1e-4 max|loc| away from its float64 evaluation (stored in the fixture).  The bar here is
        # therefore "as close to the exact result as the reference is" (factor 2 + the 1e-5 floor),
        # not 1e-5 of a value that itself carries 1e-4 of rounding noise.
        for key, got in (("loc", p.base_dist.loc), ("scale", p.base_dist.scale)):
            exact, ref32 = g[f"{tag}_{key}64"], g[f"{tag}_{key}"].astype(np.float64)
            m = np.abs(exact).max()
            ref_err = np.abs(ref32 - exact).max()
            err = np.abs(got.cpu().double().numpy() - exact).max()
            assert err <= 2.0 * ref_err + 1e-5 * m, f"{tag} {key}: |hip - fp64| = {err:.3e}, |reference fp32 - fp64| = {ref_err:.3e}"


def test_context_target_getter_matches_reference_selection():
    """CntxtTrgtGetter (npf/utils/datasplit.py:148-255): with explicit indices the selection equals
    the reference's torch.gather formulation bit for bit; drawn indices give a valid split that the
    model trains on; is_add_cntxts_to_trgts appends the context indices."""
    import npf_gwwaveform_amd as A

    g = torch.Generator().manual_seed(21)
    B, N, dx, dy = 5, 64, 2, 3
    X, Y = torch.rand(B, N, dx, generator=g) * 2 - 1, torch.randn(B, N, dy, generator=g)
    ci = torch.stack([torch.randperm(N, generator=g)[:17] for _ in range(B)])
    ti = torch.arange(N).expand(B, N)
    getter = A.CntxtTrgtGetter()
    Xc, Yc, Xt, Yt = getter(X.to(DEV), Y.to(DEV), context_indcs=ci, target_indcs=ti)
    ref = lambda t, i: torch.gather(t, 1, i.unsqueeze(-1).expand(B, -1, t.shape[-1]))  # noqa: E731
    assert torch.equal(Xc.cpu(), ref(X, ci)) and torch.equal(Yc.cpu(), ref(Y, ci))
    assert torch.equal(Xt.cpu(), X) and torch.equal(Yt.cpu(), Y)
    with pytest.raises(IndexError):
        getter(X.to(DEV), Y.to(DEV), context_indcs=ci + N, target_indcs=ti)
    add = A.CntxtTrgtGetter(contexts_getter=A.GetRandomIndcs(a=4, b=4), targets_getter=A.GetRandomIndcs(a=10, b=10),
                            is_add_cntxts_to_trgts=True)
    Xc, Yc, Xt, Yt = add(X.to(DEV), Y.to(DEV))
    assert Xc.shape == (B, 4, dx) and Xt.shape == (B, 14, dx) and torch.equal(Xt[:, 10:], Xc) and torch.equal(Yt[:, 10:], Yc)
    # a drawn split feeds the model
    case = dict(specs.CASES["g6_cnp_homosk"])
    model = build_model(case, DEV).train()
    Xc, Yc, Xt, Yt = A.CntxtTrgtGetter()(X.to(DEV), Y.to(DEV))
    assert 6 <= Xc.shape[1] <= 32 and Xt.shape[1] == N
    loss = A.CNPFLoss()(model(Xc, Yc, Xt, Yt), Yt)
    loss.backward()
    assert torch.isfinite(loss)


def test_self_attention_encoder_matches_reference():
    """G11: AttnCNP(is_self_attn=True, attention="transformer") -- relu(x + resizer(y))
    followed by two transformer self-attention layers over the context (selfattn.py:10-100), then
    transformer cross attention -- against the reference's forward / loss / gradients on its own
    seeded weights (tests/golden/g11_attncnp_selfattn.npz)."""
    import warnings

    import npf_gwwaveform_amd as A

    g = specs.load_golden("g11_attncnp_selfattn")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = A.AttnCNP(1, 2, r_dim=32, attention="transformer", is_self_attn=True)
    model.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")}, strict=True)
    model = model.to(DEV).train()
    Xc, Yc, Xt, Yt = (torch.from_numpy(g[k]).to(DEV) for k in ("X_cntxt", "Y_cntxt", "X_trgt", "Y_trgt"))
    out = model(Xc, Yc, Xt, Yt)
    loss = A.CNPFLoss()(out, Yt)
    loss.backward()
    assert_close(out[0].base_dist.loc, g["loc"], what="loc")
    assert_close(out[0].base_dist.scale, g["scale"], what="scale")
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=2e-5)
    for k, p in model.named_parameters():
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        assert_close(got, g[f"grad/{k}"], tol=1e-4, what=f"grad {k}")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_training_reduces_the_loss(dtype):
    """A few dozen Trainer steps on smooth synthetic waveforms lower the loss, in the fp32 path and in
    the bf16 compute mode (end-to-end sanity of forward, backward, gradient gather and Adam)."""
    import warnings
    from functools import partial

    import npf_gwwaveform_amd as A
    from npf_gwwaveform_amd.train import Trainer, synthetic_waveform_batch

    A.set_compute_dtype(dtype)
    try:
        torch.manual_seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = A.AttnCNP(1, 2, r_dim=64, attention="scaledot",
                              XYEncoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=2, hidden_size=64), is_sum_merge=True),
                              Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=2, hidden_size=64), is_sum_merge=True)).to(DEV)
        trainer = Trainer(model, A.CNPFLoss(), lr=2e-3, world=1)
        batches = [synthetic_waveform_batch(32, 24, 64, 100 + i, DEV) for i in range(4)]
        losses = [float(trainer.step(batches[i % 4])) for i in range(60)]
    finally:
        A.set_compute_dtype("fp32")
    assert all(np.isfinite(losses))
    assert np.mean(losses[-8:]) < np.mean(losses[:8]) - 5.0, (losses[:8], losses[-8:])


def test_graph_captured_step_equals_eager_step():
    """Trainer(use_graph=True): the step replayed from a captured HIP graph leaves the same parameters as
    the eager step (same kernels in the same order), for an attentive deterministic model (AttnCNP); the
    latent case follows below."""
    import warnings

    import npf_gwwaveform_amd as A
    from npf_gwwaveform_amd.train import Trainer, synthetic_waveform_batch

    def run(use_graph):
        torch.manual_seed(3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = A.AttnCNP(1, 2, r_dim=64).to(DEV)
        tr = Trainer(model, A.CNPFLoss(), lr=1e-3, world=1, use_graph=use_graph)
        losses = [float(tr.step(synthetic_waveform_batch(8, 20, 50, 500 + i, DEV))) for i in range(9)]
        return losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, tr

    l_e, p_e, _ = run(False)
    l_g, p_g, tr = run(True)
    assert tr._graph is not None
    np.testing.assert_allclose(l_g, l_e, rtol=1e-6)
    for k in p_e:
        assert torch.allclose(p_g[k], p_e[k], rtol=1e-6, atol=1e-8), k


def test_attnlnp_varying_context_sizes_track_the_oracle_every_step():
    """One AttnLNP (is_q_zCct=True) trained for 20 steps with a different number of context AND target
    points every step, garbage collection and allocator churn in between: q_zCc, q_zCct and the loss must
    equal the oracle's at every step.  The latent path pools each per-point representation over ITS OWN
    point count (attnnp.py:172-181) -- the count travels with the tensor (chain.PTensor), not through a
    lookup keyed by the identity of a temporary."""
    import gc

    from oracle import npf_oracle as O

    case = dict(kind="AttnLNP", r=64, L_xy=2, L_dec=2, dx=1, dy=2, B=3, C=8, T=16, is_q_zCct=True, n_z=1)
    model = build_model(case, DEV, params=specs.make_params(case, seed=21))
    crit = build_loss(case)
    model.train()
    crit.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    cfg = specs.cfg_of(case)
    rng = np.random.Generator(np.random.Philox(5))
    for step in range(20):
        c = dict(case, C=int(rng.integers(1, 70)), T=int(rng.integers(2, 90)))
        inp = specs.make_inputs(c, seed=900 + step)
        dinp = {k: v.to(DEV) for k, v in inp.items()}
        EpsIndependent.eps = dinp["eps"]
        opt.zero_grad(set_to_none=True)
        out = model(dinp["X_cntxt"], dinp["Y_cntxt"], dinp["X_trgt"], dinp["Y_trgt"])
        loss = crit(out, dinp["Y_trgt"])
        params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        ref = O.forward(cfg, params, inp["X_cntxt"], inp["Y_cntxt"], inp["X_trgt"], inp["Y_trgt"], eps=inp["eps"], n_z=1)
        ref_loss = O.elbo_loss(ref, inp["Y_trgt"])
        what = f"step {step} (C={c['C']}, T={c['T']})"
        assert_close(out[2].base_dist.loc, ref["q_zCc"][0], what=f"q_zCc.loc {what}")
        assert_close(out[2].base_dist.scale, ref["q_zCc"][1], what=f"q_zCc.scale {what}")
        assert_close(out[3].base_dist.loc, ref["q_zCct"][0], what=f"q_zCct.loc {what}")
        assert_close(out[0].base_dist.loc, ref["loc"], what=f"loc {what}")
        np.testing.assert_allclose(loss.item(), ref_loss.item(), rtol=2e-5, err_msg=what)
        loss.backward()
        opt.step()
        del out, loss
        gc.collect()
        junk = [torch.empty(int(rng.integers(1, 64)) * 1024, device=DEV) for _ in range(8)]  # allocator churn
        del junk


def test_graph_captured_step_equals_eager_step_latent_model():
    """The same for an AttnLNP with the target-side latent encode (is_q_zCct=True, one latent sample): the
    captured step replays the latent path (two mean aggregations with different point counts, the noise
    draw of rsample) and leaves the parameters of the eager run, given the same noise."""
    import warnings

    import npf_gwwaveform_amd as A
    from helpers import eps_latent_dist
    from npf_gwwaveform_amd.train import Trainer, synthetic_waveform_batch

    eps = torch.randn(1, 8, 1, 64, generator=torch.Generator().manual_seed(1)).to(DEV)

    def run(use_graph):
        torch.manual_seed(3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = A.AttnLNP(1, 2, r_dim=64, is_q_zCct=True, n_z_samples_train=1, n_z_samples_test=1,
                              LatentDistribution=eps_latent_dist).to(DEV)
        EpsIndependent.eps = eps
        tr = Trainer(model, A.ELBOLossLNPF(), lr=1e-3, world=1, use_graph=use_graph)
        losses = [float(tr.step(synthetic_waveform_batch(8, 20, 50, 500 + i, DEV))) for i in range(9)]
        return losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, tr

    l_e, p_e, _ = run(False)
    l_g, p_g, tr = run(True)
    assert tr._graph is not None
    np.testing.assert_allclose(l_g, l_e, rtol=1e-6)
    for k in p_e:
        assert torch.allclose(p_g[k], p_e[k], rtol=1e-6, atol=1e-8), k


def test_graph_step_keeps_the_input_range_check_on_the_device():
    """Trainer(use_graph=True): a replayed step cannot stop for the host, so the range check of base.py:241-247 stays in
    the step as a device reduction and ``check_inputs()`` raises the reference's ValueError at the next sync point."""
    import warnings

    import npf_gwwaveform_amd as A
    from npf_gwwaveform_amd.train import Trainer, synthetic_waveform_batch

    torch.manual_seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = A.AttnCNP(1, 2, r_dim=32).to(DEV)
    tr = Trainer(model, A.CNPFLoss(), lr=1e-3, world=1, use_graph=True)
    for i in range(6):
        tr.step(synthetic_waveform_batch(4, 10, 20, 700 + i, DEV))
    assert tr._graph is not None
    tr.check_inputs()  # all batches were in range
    bad = synthetic_waveform_batch(4, 10, 20, 800, DEV)
    bad["X_trgt"][1, 3, 0] = 1.7
    tr.step(bad)
    tr.step(synthetic_waveform_batch(4, 10, 20, 801, DEV))  # (a good batch afterwards does not hide it)
    with pytest.raises(ValueError, match=r"\[-1,1\]"):
        tr.check_inputs()
    tr.step(synthetic_waveform_batch(4, 10, 20, 802, DEV))
    tr.check_inputs()  # the record was reset


def test_checkpoint_load_under_a_captured_graph_resumes_like_the_eager_run(tmp_path):
    """A captured step holds the addresses of Adam's moment / step tensors: ``load_checkpoint`` must write into them (and
    drop the graph) -- resuming from a checkpoint in graph mode continues exactly like the eager resume, and a range
    verdict recorded before a re-capture is not lost."""
    import warnings

    import npf_gwwaveform_amd as A
    from npf_gwwaveform_amd.train import Trainer, synthetic_waveform_batch

    def make(use_graph):
        torch.manual_seed(3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = A.AttnCNP(1, 2, r_dim=64).to(DEV)
        return Trainer(model, A.CNPFLoss(), lr=1e-3, world=1, use_graph=use_graph)

    batches = [synthetic_waveform_batch(8, 20, 50, 900 + i, DEV) for i in range(14)]
    src = make(False)
    for b in batches[:4]:
        src.step(b)
    src.save_checkpoint(str(tmp_path / "ck"))

    def resume(use_graph):
        tr = make(use_graph)
        for b in batches[4:10]:  # (diverge first: in graph mode this captures a graph on the pre-load state)
            tr.step(b)
        if use_graph:
            assert tr._graph is not None
            bad = {k: v.clone() for k, v in batches[0].items()}
            bad["X_cntxt"][0, 0, 0] = -3.0
            tr.step(bad)  # out of range, verdict not read yet
        tr.load_checkpoint(str(tmp_path / "ck"))
        assert tr._graph is None
        losses = [float(tr.step(b)) for b in batches[8:14]]
        return losses, tr.flat.flat.detach().clone(), tr

    l_e, w_e, _ = resume(False)
    l_g, w_g, tr = resume(True)
    assert tr._graph is not None  # captured again after the load
    np.testing.assert_allclose(l_g, l_e, rtol=1e-6)
    assert torch.allclose(w_g, w_e, rtol=1e-6, atol=1e-8)
    with pytest.raises(ValueError, match=r"\[-1,1\]"):
        tr.check_inputs()


def test_graph_step_refuses_a_random_number_of_latent_samples():
    """A scipy random variable as n_z_samples_train is drawn on the host every forward (base.py:478-486): a
    captured graph would freeze one draw, so Trainer(use_graph=True) refuses such a model."""
    import warnings

    import scipy.stats

    import npf_gwwaveform_amd as A
    from npf_gwwaveform_amd.train import Trainer

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = A.AttnLNP(1, 2, r_dim=32, n_z_samples_train=scipy.stats.randint(1, 4)).to(DEV)
    with pytest.raises(ValueError, match="random"):
        Trainer(model, A.NLLLossLNPF(), use_graph=True)
