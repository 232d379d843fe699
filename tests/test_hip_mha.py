"""GPU parity test of the fused multihead attention kernel (csrc/mha_kernel.hip, ``npf_mha_fwd`` / ``npf_mha_bwd``) through the C
ABI: per-head scaled-dot attention with 16-feature heads (MultiheadAttender.forward between the projections and the concatenation,
npf/architectures/attention.py:505-527; DotAttender :204-220 per head with the head size in the scale) against a float64
evaluation -- fp32 arithmetic, so the fp32 gates of SURVEY.md 8c: 1e-5 of max|ref| on the output, 1e-4 on gradients.  The model
level (multihead / transformer goldens G8 / G9 / G11, the sweep) runs on this kernel too whenever the heads are 16 wide."""
import math

import pytest
import torch

from test_hip_x6 import DEV, assert_close

pytestmark = pytest.mark.gpu


def _ref(Q, K, V, H):
    B, T, F = Q.shape
    d = F // H
    heads = lambda x: x.view(B, -1, H, d).permute(0, 2, 1, 3)  # noqa: E731
    S = heads(Q) @ heads(K).transpose(-1, -2) / math.sqrt(d)
    return (S.softmax(-1) @ heads(V)).permute(0, 2, 1, 3).reshape(B, T, F)


@pytest.mark.parametrize("B,C,T,H,D", [(3, 128, 256, 8, 16), (2, 37, 70, 2, 16), (2, 200, 33, 8, 16), (1, 256, 100, 4, 16), (2, 5, 1, 1, 16),
                                       (4, 50, 128, 8, 16), (2, 64, 257, 3, 16),
                                       (2, 128, 96, 8, 32), (3, 37, 70, 2, 32), (2, 100, 33, 1, 32), (1, 64, 257, 3, 32)])
def test_mha_matches_float64(B, C, T, H, D):
    from npf_gwwaveform_amd import functional as FN

    F = D * H
    assert FN.mha_usable(D, D, C)
    g = torch.Generator().manual_seed(B * 1000 + C + T)
    Q, K, V = (torch.randn(B, n, F, generator=g) * s for n, s in ((T, 1.5), (C, 1.5), (C, 1.0)))
    w = torch.randn(B, T, F, generator=g)
    Qd, Kd, Vd = (x.to(DEV).requires_grad_(True) for x in (Q, K, V))
    out = FN.unpack_pt(FN.mha(FN.pack_pt(Qd), FN.pack_pt(Kd), FN.pack_pt(Vd), B, C, T, H, D), T, F)
    (out * w.to(DEV)).sum().backward()
    Qr, Kr, Vr = (x.double().requires_grad_(True) for x in (Q, K, V))
    ref = _ref(Qr, Kr, Vr, H)
    (ref * w.double()).sum().backward()
    assert_close(out, ref, tol=1e-5, what="attention output")
    assert_close(Qd.grad, Qr.grad, tol=1e-4, what="dQ")
    assert_close(Kd.grad, Kr.grad, tol=1e-4, what="dK")
    assert_close(Vd.grad, Vr.grad, tol=1e-4, what="dV")


def test_mha_inference_equals_training_forward():
    from npf_gwwaveform_amd import functional as FN

    g = torch.Generator().manual_seed(0)
    Q, K, V = (torch.randn(2, n, 128, generator=g).to(DEV) for n in (96, 50, 50))
    with torch.no_grad():
        a = FN.mha(FN.pack_pt(Q), FN.pack_pt(K), FN.pack_pt(V), 2, 50, 96, 8)
    b = FN.mha(FN.pack_pt(Q.requires_grad_(True)), FN.pack_pt(K), FN.pack_pt(V), 2, 50, 96, 8)
    assert torch.equal(a, b.detach())


def test_mha_rejects_other_head_sizes_and_too_many_keys():
    from npf_gwwaveform_amd import _lib as L
    from npf_gwwaveform_amd import chain as CH

    x = CH.pt_empty(1, 32, 128, DEV)
    lib = L.load()
    assert lib.npf_mha_fwd(L.ptr(x), L.ptr(x), L.ptr(x), 1, 2, 32, 32, 128, L.ptr(x), None, None) == -1   # 64-feature heads
    assert lib.npf_mha_fwd(L.ptr(x), L.ptr(x), L.ptr(x), 1, 4, 129, 32, 128, L.ptr(x), None, None) == -1  # 32-feature heads: <= 128 keys
    assert lib.npf_mha_fwd(L.ptr(x), L.ptr(x), L.ptr(x), 1, 8, 257, 32, 128, L.ptr(x), None, None) == -1  # keys > 256


@pytest.mark.parametrize("B,T,F", [(3, 256, 128), (2, 70, 128), (2, 33, 32), (1, 5, 256), (2, 64, 48)])
def test_add_layernorm_matches_float64(B, T, F):
    """LayerNorm(a + b) on PT32 tensors (npf_add_layernorm_fwd / _bwd; TransformerAttender.forward's first LayerNorm,
    attention.py:566-575) against torch.nn.functional.layer_norm in float64: output, both input gradients, dgamma, dbeta."""
    from npf_gwwaveform_amd import functional as FN

    torch.manual_seed(T)
    ln = torch.nn.LayerNorm(F).to(DEV)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.uniform_(-0.5, 0.5)
    g = torch.Generator().manual_seed(B + T)
    a, b, w = (torch.randn(B, T, F, generator=g) for _ in range(3))
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = FN.unpack_pt(FN.add_layernorm(FN.pack_pt(ad), FN.pack_pt(bd), ln, B, T), T, F)
    (y * w.to(DEV)).sum().backward()
    ar, br = a.double().requires_grad_(True), b.double().requires_grad_(True)
    gam, bet = ln.weight.detach().double().cpu().requires_grad_(True), ln.bias.detach().double().cpu().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(ar + br, (F,), gam, bet, ln.eps)
    (ref * w.double()).sum().backward()
    assert_close(y, ref, tol=1e-5, what="LayerNorm(a + b)")
    assert_close(ad.grad, ar.grad, tol=1e-4, what="da")
    assert_close(bd.grad, br.grad, tol=1e-4, what="db")
    assert_close(ln.weight.grad, gam.grad, tol=1e-4, what="dgamma")
    assert_close(ln.bias.grad, bet.grad, tol=1e-4, what="dbeta")
