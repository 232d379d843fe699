"""GPU parity tests of the x6 programs (csrc/x6_kernel.hip, npf_gwwaveform_amd/x6.py) through the C ABI: the fused target side
of an attentive model (x-encoder, scaled-dot cross attention, decoder, output layer; npf/neuralproc/attnnp.py:118-131,
npf/architectures/attention.py:129-164,204-220, encoders.py:175-183, mlp.py:95-109) and its autograd against a float64
evaluation of the same fp32 parameters -- the arithmetic is fp32 (three exact bf16 terms per operand), so the gates are the
fp32 ones of SURVEY.md 8c: 1e-5 of max|ref| on outputs, 1e-4 on gradients."""
import math
import warnings
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def assert_close(got, ref, tol=1e-5, what=""):
    got = got.detach().cpu().double().numpy()
    ref = ref.detach().cpu().double().numpy()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    m = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max()
    assert err <= tol * m, f"{what}: max|d|={err:.3e} > {tol:.0e} * max|ref|={m:.3e}"


def _unpermute(img):
    """[..., rows, K] k-permuted image -> natural column order (position 8 g + i <- column 4 g + i, 8 g + 4 + i <- 16 + 4 g + i)."""
    K = img.shape[-1]
    idx = torch.empty(K, dtype=torch.long)
    for q in range(K):
        grp, r = divmod(q, 32)
        gq, i = divmod(r, 8)
        idx[q] = 32 * grp + (4 * gq + i if i < 4 else 16 + 4 * gq + (i - 4))
    out = torch.empty_like(img)
    out[..., idx] = img
    return out


@pytest.mark.parametrize("pts", [256, 200, 130])
def test_task_images_are_exact_three_term_splits(pts):
    """npf_x6_task_images: the sum of the three bf16 terms is the fp32 value (to 2^-24), rows / columns beyond the task's points
    are zero, both orientations."""
    from npf_gwwaveform_amd import functional as FN
    from npf_gwwaveform_amd import x6

    g = torch.Generator().manual_seed(pts)
    rows = torch.randn(3, pts, 256, generator=g) * torch.logspace(-3, 3, 256)
    ri, ti = x6.task_images(FN.pack_pt(rows.to(DEV)), pts)
    for img, want in ((ri, rows), (ti, rows.transpose(1, 2))):
        full = torch.zeros(3, 256, 256, dtype=torch.float64)
        if want.shape[1] == pts:
            full[:, :pts] = want.double()
        else:
            full[:, :, :pts] = want.double()
        terms = _unpermute(img.cpu().double())  # [3 tasks, 3 terms, 256, 256]
        got = terms.sum(1)
        assert float((got - full).abs().max()) <= 2.0 ** -23 * float(full.abs().max())
        assert float(((got - full).abs() / full.abs().clamp_min(1e-30)).max()) <= 2.0 ** -22
        # the leading term is the bf16 rounding of the value
        assert torch.equal(terms[:, 0].float(), full.float().to(torch.bfloat16).float())


def _build(r=256, L=4, dx=1, dy=2, seed=0):
    import npf_gwwaveform_amd as A

    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = A.AttnCNP(dx, dy, attention="scaledot", r_dim=r,
                      XYEncoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, is_force_hid_smaller=True, hidden_size=r),
                                                   is_sum_merge=True),
                      Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, hidden_size=r), is_sum_merge=True))
    with torch.no_grad():  # (biases are zero at initialisation: give them values)
        for k, p in m.named_parameters():
            if k.endswith(".bias"):
                p.uniform_(-0.05, 0.05)
    return m.to(DEV)


def _target_side_f64(model, X, K, V):
    """float64 statement of the target side with the model's fp32 parameters."""
    d = lambda t: t.detach().double().cpu()  # noqa: E731
    P = {k: d(v).requires_grad_(True) for k, v in model.named_parameters()}
    lin = lambda x, pre: torch.nn.functional.linear(x, P[pre + ".weight"], P[pre + ".bias"])  # noqa: E731

    def mlp(x, pre, n_lin):
        h = torch.relu(lin(x, pre + ".to_hidden"))
        for i in range(n_lin):
            h = torch.relu(lin(h, f"{pre}.linears.{i}"))
        return lin(h, pre + ".out")

    Xt = mlp(X, "x_encoder", len(model.x_encoder.linears))
    logits = torch.einsum("bkd,bqd->bqk", K, Xt) / math.sqrt(Xt.shape[-1])
    R = torch.bmm(logits.softmax(-1), V)
    x2 = mlp(R, "decoder.resizer", len(model.decoder.resizer.linears))
    out = mlp(torch.relu(Xt + x2), "decoder.flat_module", len(model.decoder.flat_module.linears))
    return out, P


@pytest.mark.parametrize("variant", [1, 2, 3], ids=["16_points_per_wave", "32_points_per_wave", "8_waves_one_slab_ring"])
@pytest.mark.parametrize("B,C,T,L,dx,dy,r", [(2, 256, 64, 4, 1, 2, 256), (3, 200, 96, 2, 2, 1, 256), (1, 129, 32, 1, 1, 2, 256),
                                             (2, 250, 70, 2, 1, 2, 256), (3, 256, 288, 1, 1, 2, 256),
                                             # r = 128: the reference's default width (base.py:104-117), few context points
                                             (3, 128, 64, 2, 1, 2, 128), (2, 37, 70, 4, 1, 2, 128), (4, 5, 33, 1, 2, 1, 128)])
def test_fused_target_side_matches_float64(B, C, T, L, dx, dy, r, variant, monkeypatch):
    from npf_gwwaveform_amd import chain as CH
    from npf_gwwaveform_amd import functional as FN
    from npf_gwwaveform_amd import x6

    if r == 128 and variant != 1:
        pytest.skip("one instance at 128 features")
    monkeypatch.setattr(x6, "VARIANT", variant)  # (every instance of the 256-wide program kernel, whatever the default is)

    model = _build(r=r, L=L, dx=dx, dy=dy, seed=B * 7 + C)
    assert x6.target_side_usable(model, C, T)
    g = torch.Generator().manual_seed(C + T)
    X = torch.rand(B, T, dx, generator=g) * 2 - 1
    K = torch.randn(B, C, r, generator=g) * 0.5
    V = torch.randn(B, C, r, generator=g) * 0.5
    w = torch.randn(B, T, 2 * dy, generator=g)
    Kd, Vd = K.to(DEV).requires_grad_(True), V.to(DEV).requires_grad_(True)
    rows = x6.target_side(model, X.to(DEV), CH.PTensor(FN.pack_pt(Kd), C, r), CH.PTensor(FN.pack_pt(Vd), C, r))
    assert tuple(rows.shape) == (B, T, 2 * dy)
    (rows * w.to(DEV)).sum().backward()
    Kr, Vr = K.double().requires_grad_(True), V.double().requires_grad_(True)
    ref, P = _target_side_f64(model, X.double(), Kr, Vr)
    (ref * w.double()).sum().backward()
    assert_close(rows, ref, tol=1e-5, what="target side rows")
    assert_close(Kd.grad, Kr.grad, tol=1e-4, what="dK")
    assert_close(Vd.grad, Vr.grad, tol=1e-4, what="dV")
    for k, p in model.named_parameters():
        if k.startswith("xy_encoder"):
            continue
        assert p.grad is not None, k
        assert_close(p.grad, P[k].grad, tol=1e-4, what=f"grad {k}")


def test_fused_target_side_inference_equals_training_forward():
    """Without gradients the program stores nothing but what it re-reads: same rows."""
    from npf_gwwaveform_amd import chain as CH
    from npf_gwwaveform_amd import functional as FN
    from npf_gwwaveform_amd import x6

    model = _build(L=2, seed=5)
    g = torch.Generator().manual_seed(9)
    X = (torch.rand(2, 64, 1, generator=g) * 2 - 1).to(DEV)
    K = FN.pack_pt((torch.randn(2, 256, 256, generator=g) * 0.5).to(DEV))
    V = FN.pack_pt((torch.randn(2, 256, 256, generator=g) * 0.5).to(DEV))
    a = x6.target_side(model, X, CH.PTensor(K, 256, 256), CH.PTensor(V, 256, 256))
    with torch.no_grad():
        b = x6.target_side(model, X, CH.PTensor(K, 256, 256), CH.PTensor(V, 256, 256))
    assert torch.equal(a, b)


def _context_side_f64(model, X, Y):
    d = lambda t: t.detach().double().cpu()  # noqa: E731
    P = {k: d(v).requires_grad_(True) for k, v in model.named_parameters()}
    lin = lambda x, pre: torch.nn.functional.linear(x, P[pre + ".weight"], P[pre + ".bias"])  # noqa: E731

    def mlp(x, pre, n_lin):
        h = torch.relu(lin(x, pre + ".to_hidden"))
        for i in range(n_lin):
            h = torch.relu(lin(h, f"{pre}.linears.{i}"))
        return lin(h, pre + ".out")

    Xc = mlp(X, "x_encoder", len(model.x_encoder.linears))
    x2 = mlp(Y, "xy_encoder.resizer", 0)
    R = mlp(torch.relu(Xc + x2), "xy_encoder.flat_module", len(model.xy_encoder.flat_module.linears))
    return Xc, R, P


@pytest.mark.parametrize("variant", [1, 2, 3], ids=["16_points_per_wave", "32_points_per_wave", "8_waves_one_slab_ring"])
@pytest.mark.parametrize("B,C,L,dx,dy,r", [(2, 256, 4, 1, 2, 256), (3, 64, 2, 2, 1, 256), (1, 32, 1, 1, 3, 256), (2, 45, 2, 1, 2, 256),
                                           (5, 96, 1, 1, 2, 256), (3, 50, 2, 1, 2, 128), (2, 128, 4, 2, 1, 128)])
def test_fused_context_side_matches_float64(B, C, L, dx, dy, r, variant, monkeypatch):
    from npf_gwwaveform_amd import functional as FN
    from npf_gwwaveform_amd import x6

    if r == 128 and variant != 1:
        pytest.skip("one instance at 128 features")
    monkeypatch.setattr(x6, "VARIANT", variant)

    model = _build(r=r, L=L, dx=dx, dy=dy, seed=B * 11 + C)
    assert x6.context_side_usable(model, C)
    g = torch.Generator().manual_seed(C + L)
    X = torch.rand(B, C, dx, generator=g) * 2 - 1
    Y = torch.randn(B, C, dy, generator=g)
    wk, wr = torch.randn(B, C, r, generator=g), torch.randn(B, C, r, generator=g)
    Xc, R = x6.context_side(model, X.to(DEV), Y.to(DEV))
    (FN.unpack_pt(Xc.t, C, r) * wk.to(DEV)).sum().backward(retain_graph=True)
    gk_only = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    (FN.unpack_pt(R.t, C, r) * wr.to(DEV)).sum().backward()
    Xc_r, R_r, P = _context_side_f64(model, X.double(), Y.double())
    ((Xc_r * wk.double()).sum() + (R_r * wr.double()).sum()).backward()
    assert_close(FN.unpack_pt(Xc.t, C, r), Xc_r, tol=1e-5, what="Xc_enc")
    assert_close(FN.unpack_pt(R.t, C, r), R_r, tol=1e-5, what="R")
    assert set(gk_only) == {k for k in P if k.startswith("x_encoder")}  # (a gradient for the keys alone stays in the x-encoder)
    for k, p in model.named_parameters():
        if k.startswith("decoder"):
            continue
        assert p.grad is not None, k
        assert_close(p.grad, P[k].grad, tol=1e-4, what=f"grad {k}")


@pytest.mark.parametrize("variant", [0, 1], ids=["default_instance", "variant_1"])
@pytest.mark.parametrize("F,n,T,L,dy", [(512, 2, 70, 4, 2), (512, 1, 4096, 4, 2), (256, 3, 100, 2, 1), (128, 2, 45, 3, 2), (512, 3, 33, 1, 2),
                                        (512, 5, 64, 2, 2)])
def test_decode_rows_from_row_major_inputs_matches_float64(F, n, T, L, dy, variant, monkeypatch):
    """``decode(X_trgt_enc, R_trgt)`` at inference (base.py:327-367 -> encoders.py:175-183 -> mlp.py:95-109) as one x6 program from
    the ROW-MAJOR tensors the reference's signature takes, at the widths the program kernel has instances for -- 512 is
    BASELINE config 5's decoder -- with ragged target counts: against float64 at the fp32 tolerance (1e-5 of max|ref|)."""
    import npf_gwwaveform_amd as A
    from npf_gwwaveform_amd import x6

    # (512 features: the default splits the contraction over pairs of waves, variant 1 keeps all 512 features in one wave)
    monkeypatch.setattr(x6, "VARIANT", variant)
    torch.manual_seed(F + T)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        dec = A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, hidden_size=F), is_sum_merge=True)(F, F, 2 * dy).to(DEV)
    with torch.no_grad():
        for k, p in dec.named_parameters():
            if k.endswith(".bias"):
                p.uniform_(-0.05, 0.05)
    g = torch.Generator().manual_seed(T)
    x1 = torch.randn(n, T, F, generator=g) * 0.5
    x2 = torch.randn(n, T, F, generator=g) * 0.5
    assert x6.decode_rows_usable(dec, x1.to(DEV), x2.to(DEV))
    with torch.no_grad():
        got = dec(x1.to(DEV), x2.to(DEV))
    assert tuple(got.shape) == (n, T, 2 * dy)
    P = {k: v.detach().double().cpu() for k, v in dec.named_parameters()}
    lin = lambda x, pre: torch.nn.functional.linear(x, P[pre + ".weight"], P[pre + ".bias"])  # noqa: E731

    def mlp(x, pre, n_lin):
        h = torch.relu(lin(x, pre + ".to_hidden"))
        for i in range(n_lin):
            h = torch.relu(lin(h, f"{pre}.linears.{i}"))
        return lin(h, pre + ".out")

    ref = mlp(torch.relu(x1.double() + mlp(x2.double(), "resizer", len(dec.resizer.linears))), "flat_module", len(dec.flat_module.linears))
    assert_close(got, ref, tol=1e-5, what=f"decode rows F={F}")


@pytest.mark.parametrize("B,T,L,dy", [(3, 64, 4, 2), (2, 70, 2, 1), (1, 257, 1, 2)])
def test_decoder_side_matches_float64(B, T, L, dy):
    """The decoder alone as one program each way (x6.decoder_side: r = 128 models whose attention is not the fused scaled-dot one,
    e.g. the transformer attention of the shipped checkpoints): rows, the gradients wrt both inputs and every dW / db."""
    from npf_gwwaveform_amd import functional as FN
    from npf_gwwaveform_amd import x6

    r = 128
    model = _build(r=r, L=L, dx=1, dy=dy, seed=T)
    assert x6.decoder_side_usable(model, T)
    g = torch.Generator().manual_seed(T)
    R, X1 = torch.randn(B, T, r, generator=g) * 0.5, torch.randn(B, T, r, generator=g) * 0.5
    w = torch.randn(B, T, 2 * dy, generator=g)
    Rd, Xd = R.to(DEV).requires_grad_(True), X1.to(DEV).requires_grad_(True)
    rows = x6.decoder_side(model, FN.pack_pt(Rd), FN.pack_pt(Xd), T)
    assert tuple(rows.shape) == (B, T, 2 * dy)
    (rows * w.to(DEV)).sum().backward()
    d = lambda t: t.detach().double().cpu()  # noqa: E731
    P = {k: d(v).requires_grad_(True) for k, v in model.named_parameters()}
    lin = lambda x, pre: torch.nn.functional.linear(x, P[pre + ".weight"], P[pre + ".bias"])  # noqa: E731

    def mlp(x, pre, n_lin):
        h = torch.relu(lin(x, pre + ".to_hidden"))
        for i in range(n_lin):
            h = torch.relu(lin(h, f"{pre}.linears.{i}"))
        return lin(h, pre + ".out")

    Rr, Xr = R.double().requires_grad_(True), X1.double().requires_grad_(True)
    ref = mlp(torch.relu(Xr + mlp(Rr, "decoder.resizer", len(model.decoder.resizer.linears))), "decoder.flat_module",
              len(model.decoder.flat_module.linears))
    (ref * w.double()).sum().backward()
    assert_close(rows, ref, tol=1e-5, what="decoder rows")
    assert_close(Rd.grad, Rr.grad, tol=1e-4, what="dR")
    assert_close(Xd.grad, Xr.grad, tol=1e-4, what="dX1")
    for k, p in model.named_parameters():
        if k.startswith("decoder"):
            assert p.grad is not None, k
            assert_close(p.grad, P[k].grad, tol=1e-4, what=f"grad {k}")


@pytest.mark.parametrize("B,C,r", [(3, 128, 128), (2, 37, 128), (2, 200, 256)])
def test_pair_linear_matches_float64(B, C, r):
    """Two bias-free r x r projections of two PT32 tensors as one launch each way (x6.pair_linear: the key / value projections of
    MultiheadAttender, attention.py:397-404)."""
    from npf_gwwaveform_amd import functional as FN
    from npf_gwwaveform_amd import x6

    torch.manual_seed(C)
    la, lb = torch.nn.Linear(r, r, bias=False).to(DEV), torch.nn.Linear(r, r, bias=False).to(DEV)
    assert x6.pair_linear_usable(la, lb)
    g = torch.Generator().manual_seed(B + C)
    a, b = torch.randn(B, C, r, generator=g), torch.randn(B, C, r, generator=g)
    wa, wb = torch.randn(B, C, r, generator=g), torch.randn(B, C, r, generator=g)
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    oa, ob = x6.pair_linear(FN.pack_pt(ad), FN.pack_pt(bd), C, la, lb)
    oa, ob = FN.unpack_pt(oa, C, r), FN.unpack_pt(ob, C, r)
    ((oa * wa.to(DEV)).sum() + (ob * wb.to(DEV)).sum()).backward()
    Wa, Wb = la.weight.detach().double().cpu().requires_grad_(True), lb.weight.detach().double().cpu().requires_grad_(True)
    ar, br = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ra, rb = ar @ Wa.t(), br @ Wb.t()
    ((ra * wa.double()).sum() + (rb * wb.double()).sum()).backward()
    assert_close(oa, ra, tol=1e-5, what="first projection")
    assert_close(ob, rb, tol=1e-5, what="second projection")
    assert_close(ad.grad, ar.grad, tol=1e-4, what="d first input")
    assert_close(bd.grad, br.grad, tol=1e-4, what="d second input")
    assert_close(la.weight.grad, Wa.grad, tol=1e-4, what="dW first")
    assert_close(lb.weight.grad, Wb.grad, tol=1e-4, what="dW second")


@pytest.mark.parametrize("B,T,L,dx,r", [(3, 64, 2, 1, 128), (2, 70, 1, 2, 128), (2, 96, 3, 1, 256)])
def test_xenc_proj_matches_float64(B, T, L, dx, r):
    """x-encoder + a projection with bias of its output as one launch each way (x6.xenc_proj: the targets in front of a
    multihead / transformer attention, mlp.py:95-109 + MultiheadAttender.query_transform)."""
    from npf_gwwaveform_amd import functional as FN
    from npf_gwwaveform_amd import x6

    model = _build(r=r, L=L, dx=dx, dy=2, seed=T)
    torch.manual_seed(T)
    lq = torch.nn.Linear(r, r, bias=True).to(DEV)
    assert x6.xenc_proj_usable(model, lq, T)
    g = torch.Generator().manual_seed(T + B)
    X = torch.rand(B, T, dx, generator=g) * 2 - 1
    we, wq = torch.randn(B, T, r, generator=g), torch.randn(B, T, r, generator=g)
    Xe, Q = x6.xenc_proj(model, X.to(DEV), lq)
    xe, q = FN.unpack_pt(Xe.t, T, r), FN.unpack_pt(Q, T, r)
    ((xe * we.to(DEV)).sum() + (q * wq.to(DEV)).sum()).backward()
    d = lambda t: t.detach().double().cpu()  # noqa: E731
    P = {k: d(v).requires_grad_(True) for k, v in model.named_parameters() if k.startswith("x_encoder")}
    Wq, bq = d(lq.weight).requires_grad_(True), d(lq.bias).requires_grad_(True)
    lin = lambda x, pre: torch.nn.functional.linear(x, P[pre + ".weight"], P[pre + ".bias"])  # noqa: E731
    h = torch.relu(lin(X.double(), "x_encoder.to_hidden"))
    for i in range(len(model.x_encoder.linears)):
        h = torch.relu(lin(h, f"x_encoder.linears.{i}"))
    re = lin(h, "x_encoder.out")
    rq = re @ Wq.t() + bq
    ((re * we.double()).sum() + (rq * wq.double()).sum()).backward()
    assert_close(xe, re, tol=1e-5, what="encoded points")
    assert_close(q, rq, tol=1e-5, what="projection")
    assert_close(lq.weight.grad, Wq.grad, tol=1e-4, what="dW projection")
    assert_close(lq.bias.grad, bq.grad, tol=1e-4, what="db projection")
    for k, p in model.named_parameters():
        if k.startswith("x_encoder"):
            assert_close(p.grad, P[k].grad, tol=1e-4, what=f"grad {k}")


@pytest.mark.parametrize("B,T,L,r", [(3, 64, 1, 128), (2, 70, 3, 128), (2, 33, 1, 256)])
def test_mlp_pt_matches_float64(B, T, L, r):
    """An MLP with r x r layers on a PT32 tensor as one launch each way (x6.mlp_pt: the MLP block of TransformerAttender,
    attention.py:576-588; MLP.forward mlp.py:95-109)."""
    import npf_gwwaveform_amd as A
    from npf_gwwaveform_amd import functional as FN
    from npf_gwwaveform_amd import x6

    torch.manual_seed(T + L)
    mlp = A.MLP(r, r, hidden_size=r, n_hidden_layers=L).to(DEV)
    with torch.no_grad():
        for k, p in mlp.named_parameters():
            if k.endswith(".bias"):
                p.uniform_(-0.05, 0.05)
    assert x6.mlp_pt_usable(mlp)
    g = torch.Generator().manual_seed(B + T)
    x, w = torch.randn(B, T, r, generator=g) * 0.7, torch.randn(B, T, r, generator=g)
    xd = x.to(DEV).requires_grad_(True)
    y = FN.unpack_pt(x6.mlp_pt(mlp, FN.pack_pt(xd), T), T, r)
    (y * w.to(DEV)).sum().backward()
    P = {k: v.detach().double().cpu().requires_grad_(True) for k, v in mlp.named_parameters()}
    lin = lambda t, pre: torch.nn.functional.linear(t, P[pre + ".weight"], P[pre + ".bias"])  # noqa: E731
    xr = x.double().requires_grad_(True)
    h = torch.relu(lin(xr, "to_hidden"))
    for i in range(len(mlp.linears)):
        h = torch.relu(lin(h, f"linears.{i}"))
    ref = lin(h, "out")
    (ref * w.double()).sum().backward()
    assert_close(y, ref, tol=1e-5, what="MLP output")
    assert_close(xd.grad, xr.grad, tol=1e-4, what="dx")
    for k, p in mlp.named_parameters():
        assert_close(p.grad, P[k].grad, tol=1e-4, what=f"grad {k}")


@pytest.mark.parametrize("B,T,L", [(3, 64, 2), (2, 70, 1)])
def test_decoder_side_with_latent_merge_matches_float64(B, T, L):
    """x6.decoder_side with AttnLNP's merge_r_z in front (base.py:554-575 as relu(W_R R + zb[task]), the latent half a per-task
    bias): rows, dR, dX1, d zb and every dW / db against float64."""
    import npf_gwwaveform_amd as A
    from npf_gwwaveform_amd import functional as FN
    from npf_gwwaveform_amd import x6

    r = 128
    torch.manual_seed(T)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = A.AttnLNP(1, 2, attention="transformer", r_dim=r,
                          Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, hidden_size=r), is_sum_merge=True)).to(DEV)
    assert x6.decoder_side_usable(model, T)
    g = torch.Generator().manual_seed(T + 1)
    R, X1, zb = torch.randn(B, T, r, generator=g) * 0.5, torch.randn(B, T, r, generator=g) * 0.5, torch.randn(B, r, generator=g) * 0.3
    w = torch.randn(B, T, 4, generator=g)
    Rd, Xd, zd = (t.to(DEV).requires_grad_(True) for t in (R, X1, zb))
    rows = x6.decoder_side(model, FN.pack_pt(Rd), FN.pack_pt(Xd), T, zb=zd)
    (rows * w.to(DEV)).sum().backward()
    d = lambda t: t.detach().double().cpu()  # noqa: E731
    P = {k: d(v).requires_grad_(True) for k, v in model.named_parameters()}
    lin = lambda x, pre: torch.nn.functional.linear(x, P[pre + ".weight"], P[pre + ".bias"])  # noqa: E731

    def mlp(x, pre, n_lin):
        h = torch.relu(lin(x, pre + ".to_hidden"))
        for i in range(n_lin):
            h = torch.relu(lin(h, f"{pre}.linears.{i}"))
        return lin(h, pre + ".out")

    Rr, Xr, zr = R.double().requires_grad_(True), X1.double().requires_grad_(True), zb.double().requires_grad_(True)
    Rm = torch.relu(Rr @ P["r_z_merger.weight"][:, :r].t() + zr[:, None, :])
    ref = mlp(torch.relu(Xr + mlp(Rm, "decoder.resizer", len(model.decoder.resizer.linears))), "decoder.flat_module",
              len(model.decoder.flat_module.linears))
    (ref * w.double()).sum().backward()
    assert_close(rows, ref, tol=1e-5, what="rows")
    assert_close(Rd.grad, Rr.grad, tol=1e-4, what="dR")
    assert_close(Xd.grad, Xr.grad, tol=1e-4, what="dX1")
    assert_close(zd.grad, zr.grad, tol=1e-4, what="d zb")
    assert_close(model.r_z_merger.weight.grad[:, :r], P["r_z_merger.weight"].grad[:, :r], tol=1e-4, what="dW_R")
    for k, p in model.named_parameters():
        if k.startswith("decoder"):
            assert_close(p.grad, P[k].grad, tol=1e-4, what=f"grad {k}")
