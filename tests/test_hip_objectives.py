"""The objectives as kernels: the loss-only Gaussian head, the Monte-Carlo reductions over the
latent samples (``npf_mc_objective_fwd/bwd``: mean / log-mean-exp / SUMO) and the evaluation protocol
(utils/evaluate.py:9-28) against torch, the oracle and the reference's golden vectors."""
import math

import numpy as np
import pytest
import torch

import specs
from helpers import EpsIndependent, assert_close, build_model
from oracle import npf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sumo_reference(log_w, a=5, alpha=85):
    """The oracle's SUMO formula (npf/losses.py:207-276 restated) on a [n_z, B] tensor, float64, with autograd."""
    n_z = log_w.shape[0]
    ks = torch.arange(1, n_z + 1).unsqueeze(-1)
    cum = torch.cat([torch.logsumexp(log_w[:i], dim=0, keepdim=True) for i in range(1, n_z + 1)], dim=0) - ks.double().log()
    kk = (ks - 1 + 1 - a).clamp(min=1).double()
    al = float(alpha - a)
    tail = torch.where(kk < al, 1.0 / kk, (1.0 / al) * 0.9 ** (kk - al))
    return cum[a - 1] + (tail[a:] * (cum[a:] - cum[a - 1:-1])).sum(0)


@pytest.mark.parametrize("n_z,B", [(1, 7), (5, 3), (8, 130), (32, 17), (100, 5)])
def test_mc_objective_kernels_match_float64_torch(n_z, B):
    from npf_gwwaveform_amd import functional as FN
    from npf_gwwaveform_amd.losses import SUMOLossLNPF

    g = torch.Generator().manual_seed(n_z * 1000 + B)
    lw = torch.randn(n_z, B, generator=g) * 8 - 100          # log-likelihood sized values, far from zero
    dout = torch.randn(B, generator=g)
    cases = [(FN.MC_MEAN, lambda t: t.mean(0)), (FN.MC_LOGMEANEXP, lambda t: torch.logsumexp(t, 0) - math.log(n_z))]
    if n_z >= 5:
        cases.append((FN.MC_SUMO, _sumo_reference))
    for mode, ref_fn in cases:
        ref_in = lw.double().requires_grad_(True)
        ref = ref_fn(ref_in)
        ref.backward(dout.double())
        x = lw.to(DEV).requires_grad_(True)
        if mode == FN.MC_SUMO:
            crit = SUMOLossLNPF()
            out = crit.estimate(x)
        else:
            out = FN.mc_objective(x, mode)
        out.backward(dout.to(DEV))
        assert_close(out, ref, tol=2e-6, what=f"mode {mode} forward")
        assert_close(x.grad, ref_in.grad, tol=2e-5, what=f"mode {mode} backward")


@pytest.mark.parametrize("homosk", [False, True])
def test_loss_only_head_equals_the_full_head(homosk):
    """``npf_gauss_head_fwd`` with loc = scale = NULL writes only the summed log-likelihood; its backward recomputes
    loc / scale from the raw decoder output: same numbers as the launch that materialises them."""
    from npf_gwwaveform_amd import functional as FN

    g = torch.Generator().manual_seed(5)
    rows, B, T, dy = 6, 3, 77, 2
    suff = torch.randn(rows, T, 2 * dy, generator=g)
    Y = torch.randn(B, T, dy, generator=g).to(DEV)
    gs = torch.randn(rows, generator=g).to(DEV)
    a = suff.to(DEV).requires_grad_(True)
    b = suff.to(DEV).requires_grad_(True)
    loc, scale, slp_full = FN.gauss_head(a, Y, dy, homosk)
    e0, e1, slp_only = FN.gauss_head(b, Y, dy, homosk, want_dist=False)
    assert e0.numel() == 0 and e1.numel() == 0 and loc.shape == (rows, T, dy)
    assert torch.equal(slp_full, slp_only)
    slp_full.backward(gs)
    slp_only.backward(gs)
    assert_close(b.grad, a.grad, tol=1e-6, what="d_suff of the loss-only launch")


def test_training_objective_never_materialises_loc_and_scale():
    case = specs.CASES["g10_attnlnp_nll_nz8"]
    model = build_model(case, DEV).train()
    inp = {k: v.to(DEV) for k, v in specs.make_inputs(case).items()}
    EpsIndependent.eps = inp["eps"]
    import npf_gwwaveform_amd as A

    out = model(inp["X_cntxt"], inp["Y_cntxt"], inp["X_trgt"], inp["Y_trgt"])
    loss = A.NLLLossLNPF().train()(out, inp["Y_trgt"])
    loss.backward()
    assert isinstance(out[0], A.HeadDistribution) and out[0]._base is None
    assert out[0].batch_shape == (8, case["B"], case["T"]) and out[0].event_shape == (case["dy"],)
    g = specs.load_golden("g10_attnlnp_nll_nz8")
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=2e-5)
    assert_close(out[0].base_dist.loc, g["loc"], what="loc (materialised on demand)")
    assert out[0]._base is not None
    assert_close(out[0].mean, g["loc"], what="mean")


def test_eval_loglike_matches_the_reference_protocol():
    """G12: per-task test log-likelihoods with 32 latent samples (utils/evaluate.py:9-28 + losses.py:65-69), two
    batches concatenated, against the reference's numbers; reduction and training flags are restored."""
    import npf_gwwaveform_amd as A

    g = specs.load_golden("g12_eval_loglike")
    for tag, crit in (("attnlnp", A.ELBOLossLNPF()), ("lnp", A.SUMOLossLNPF()), ("cnp", A.CNPFLoss())):
        case = specs.EVAL_CASES[tag]
        model = build_model(case, DEV).train()
        crit.train()

        def batches():
            for i in range(2):
                inp = {k: v.to(DEV) for k, v in specs.make_inputs(case, seed=5000 + i).items()}
                if "eps" in inp:
                    EpsIndependent.eps = inp.pop("eps")
                yield inp

        ll = A.eval_loglike(model, crit, batches(), seed=123)
        assert ll.shape == g[f"{tag}_loglike"].shape
        np.testing.assert_allclose(ll, g[f"{tag}_loglike"], rtol=2e-5, err_msg=tag)
        assert crit.reduction == "mean" and model.training and crit.training


def test_eval_loglike_reseeds_lazily_drawn_batches():
    """utils/evaluate.py:9-28 seeds before iterating: batches drawn lazily through ``CntxtTrgtGetter(GetRandomIndcs)`` --
    context size from Python's ``random`` (datasplit.py:68,74), subsets from the device generator -- are the same draws on
    every call, so two calls return the same per-task log-likelihoods."""
    import npf_gwwaveform_amd as A
    from npf_gwwaveform_amd.datasplit import CntxtTrgtGetter, GetRandomIndcs, get_all_indcs

    case = specs.EVAL_CASES["cnp"]
    model = build_model(case, DEV).eval()
    crit = A.CNPFLoss()
    getter = CntxtTrgtGetter(contexts_getter=GetRandomIndcs(a=3, b=20), targets_getter=get_all_indcs)
    g = torch.Generator(device="cpu").manual_seed(11)
    X = (torch.rand(4, 40, case["dx"], generator=g) * 2 - 1).to(DEV)
    Y = torch.randn(4, 40, case["dy"], generator=g).to(DEV)
    sizes = []

    def batches():
        for _ in range(3):
            Xc, Yc, Xt, Yt = getter(X, Y)
            sizes.append(Xc.shape[1])
            yield dict(X_cntxt=Xc, Y_cntxt=Yc, X_trgt=Xt, Y_trgt=Yt)

    a = A.eval_loglike(model, crit, batches(), seed=123)
    b = A.eval_loglike(model, crit, batches(), seed=123)
    assert sizes[:3] == sizes[3:] and len(set(sizes[:3])) > 1, sizes
    np.testing.assert_array_equal(a, b)
    c = A.eval_loglike(model, crit, batches(), seed=124)
    assert not np.array_equal(a, c)
