"""GPU parity tests of the individual HIP kernels, through the C ABI (ctypes), against the
stage-level golden vectors of the reference (tests/golden/g6_stages.npz) and plain fp32
torch on the CPU.  Tolerance (fp32): max|d| <= 1e-5 * max|ref| per tensor."""
import math

import numpy as np
import pytest
import torch

import specs

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _mods():
    from npf_gwwaveform_amd import chain as CH
    from npf_gwwaveform_amd import functional as FN
    return CH, FN


def assert_close(got, ref, tol=1e-5, what=""):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, dtype=np.float64)
    ref = ref.detach().cpu().double().numpy() if torch.is_tensor(ref) else np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    m = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max()
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    assert err <= tol * m, f"{what}: max|d|={err:.3e} > {tol:.0e} * max|ref|={m:.3e}"


def to_pt_ref(rows: torch.Tensor) -> torch.Tensor:
    """Pure-index statement of the PT32 layout (include/npf_hip.h)."""
    n_tasks, pts, F = rows.shape
    tiles, Fp = (pts + 31) // 32, (F + 31) // 32 * 32
    buf = torch.zeros(n_tasks, tiles * 32, Fp)
    buf[:, :pts, :F] = rows
    return buf.view(n_tasks, tiles, 32, Fp // 4, 4).permute(0, 1, 3, 2, 4).contiguous()


@pytest.mark.parametrize("shape", [(3, 70, 64), (1, 32, 32), (2, 5, 2), (4, 129, 100), (2, 256, 256)])
def test_pack_unpack(shape):
    CH, FN = _mods()
    g = torch.Generator().manual_seed(0)
    rows = torch.randn(*shape, generator=g)
    pt = FN.pack_pt(rows.to(DEV))
    assert tuple(pt.shape) == CH.pt_shape(*shape)
    assert torch.equal(pt.cpu(), to_pt_ref(rows))
    back = FN.unpack_pt(pt, shape[1], shape[2])
    assert torch.equal(back.cpu(), rows)


@pytest.mark.parametrize("K,N", [(64, 64), (256, 256), (32, 256), (256, 4), (96, 48), (128, 256), (40, 17), (256, 32)])
@pytest.mark.parametrize("per_task", [False, True])
def test_single_linear(K, N, per_task):
    CH, FN = _mods()
    g = torch.Generator().manual_seed(K * 1000 + N)
    n_tasks, pts = 3, 70
    x = torch.randn(n_tasks, pts, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    ref = torch.relu(torch.nn.functional.linear(x, W, b))
    ch = CH.Chain(n_tasks, pts, DEV, wg_per_task=per_task)
    ch.input_pt(FN.pack_pt(x.to(DEV)), K).linear(W.to(DEV), b.to(DEV), relu=True).output_pt()
    (y_pt,) = ch.run()
    y = FN.unpack_pt(y_pt, pts, N)
    assert_close(y, ref, what=f"linear {K}->{N}")
    # padding features of the PT output must be exactly zero (they feed later layers)
    full = FN.unpack_pt(y_pt, pts, CH.pad32(N)).cpu()
    assert torch.count_nonzero(full[..., N:]) == 0


def test_rows_io_and_no_bias():
    CH, FN = _mods()
    g = torch.Generator().manual_seed(5)
    n_tasks, pts = 2, 45
    x = torch.rand(n_tasks, pts, 2, generator=g) * 2 - 1
    W0, W1 = torch.randn(32, 2, generator=g), torch.randn(4, 32, generator=g)
    ref = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x, W0)), W1)
    ch = CH.Chain(n_tasks, pts, DEV)
    ch.input_rows(x.to(DEV), 2).linear(W0.to(DEV), None, relu=True).linear(W1.to(DEV), None).output_rows()
    (y,) = ch.run()
    assert_close(y, ref, what="rows io")


def _mlp_params(g, tag):
    pre = f"mlp_{tag}/param/"
    return {k[len(pre):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre)}


def _layers(params):
    names = ["to_hidden"]
    i = 0
    while f"linears.{i}.weight" in params:
        names.append(f"linears.{i}")
        i += 1
    names.append("out")
    return names


@pytest.mark.parametrize("tag", ["sq", "clamp", "skinny", "wide"])
def test_mlp_chain_fwd_bwd_vs_reference_fixture(tag):
    """MLP.forward + autograd (npf/architectures/mlp.py:95-109) on the reference's vectors."""
    CH, FN = _mods()
    g = specs.load_golden("g6_stages")
    params = {k: v.to(DEV).requires_grad_() for k, v in _mlp_params(g, tag).items()}
    x = torch.from_numpy(g[f"mlp_{tag}/x"]).to(DEV).requires_grad_()
    rows, n_in = x.shape
    names = _layers(params)
    ch = CH.Chain(1, rows, DEV)
    ch.input_pt(FN.pack_pt(x.view(1, rows, n_in)), n_in)
    for j, nm in enumerate(names):
        ch.linear(params[f"{nm}.weight"], params[f"{nm}.bias"], relu=(j < len(names) - 1))
    ch.output_pt()
    (y_pt,) = ch.run()
    n_out = params["out.weight"].shape[0]
    y = FN.unpack_pt(y_pt, rows, n_out).view(rows, n_out)
    assert_close(y, g[f"mlp_{tag}/y"], what=f"mlp {tag} y")
    (y * torch.from_numpy(g[f"mlp_{tag}/w"]).to(DEV)).sum().backward()
    assert_close(x.grad, g[f"mlp_{tag}/dx"], what=f"mlp {tag} dx")
    for k, p in params.items():
        assert_close(p.grad, g[f"mlp_{tag}/grad/{k}"], what=f"mlp {tag} d{k}")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_scaledot_attention_fwd_bwd_vs_reference_fixture(tag):
    """DotAttender (npf/architectures/attention.py:129-164,204-220) on the reference's vectors."""
    CH, FN = _mods()
    g = specs.load_golden("g6_stages")
    k = torch.from_numpy(g[f"attn_{tag}/keys"]).to(DEV).requires_grad_()
    q = torch.from_numpy(g[f"attn_{tag}/queries"]).to(DEV).requires_grad_()
    v = torch.from_numpy(g[f"attn_{tag}/values"]).to(DEV).requires_grad_()
    B, C, d = k.shape
    T, r = q.shape[1], v.shape[2]
    ch = CH.Chain(B, T, DEV, wg_per_task=True)
    ch.input_pt(FN.pack_pt(q), d).attn_scores(FN.pack_pt(k), C).softmax(1.0 / math.sqrt(d)).attn_values(FN.pack_pt(v), r)
    ch.output_pt()
    (o_pt,) = ch.run()
    o = FN.unpack_pt(o_pt, T, r)
    assert_close(o, g[f"attn_{tag}/out"], what=f"attn {tag} out")
    (o * torch.from_numpy(g[f"attn_{tag}/w"]).to(DEV)).sum().backward()
    assert_close(q.grad, g[f"attn_{tag}/dqueries"], what=f"attn {tag} dq")
    assert_close(k.grad, g[f"attn_{tag}/dkeys"], what=f"attn {tag} dk")
    assert_close(v.grad, g[f"attn_{tag}/dvalues"], what=f"attn {tag} dv")


def test_merge_addend_and_taskvec():
    """relu(x1 + resizer(x2)) with a PT addend (encoders.py:175-183) and a per-task vector."""
    CH, FN = _mods()
    g = torch.Generator().manual_seed(11)
    n_tasks, pts, r = 3, 50, 64
    x1 = torch.randn(n_tasks, pts, r, generator=g).requires_grad_()
    x2 = torch.randn(n_tasks, pts, r, generator=g).requires_grad_()
    vec = torch.randn(n_tasks, r, generator=g).requires_grad_()
    W = (torch.randn(r, r, generator=g) / 8).requires_grad_()
    b = (torch.randn(r, generator=g) * 0.1).requires_grad_()
    w_out = torch.randn(n_tasks, pts, r, generator=g)
    ref = torch.relu(torch.relu(x1 + torch.nn.functional.linear(x2, W, b)) + vec[:, None, :])
    (ref * w_out).sum().backward()
    refs = [t.grad.clone() for t in (x1, x2, vec, W, b)]
    d = [t.detach().to(DEV).requires_grad_() for t in (x1, x2, vec, W, b)]
    ch = CH.Chain(n_tasks, pts, DEV)
    ch.input_pt(FN.pack_pt(d[1]), r).linear(d[3], d[4], relu=True, addend=FN.pack_pt(d[0])).add_taskvec(d[2], relu=True)
    ch.output_pt()
    (y_pt,) = ch.run()
    y = FN.unpack_pt(y_pt, pts, r)
    assert_close(y, ref, what="merge fwd")
    (y * w_out.to(DEV)).sum().backward()
    for name, t, rf in zip(("x1", "x2", "vec", "W", "b"), d, refs):
        assert_close(t.grad, rf, what=f"merge d{name}")


@pytest.mark.parametrize("homosk", [False, True])
@pytest.mark.parametrize("dy", [1, 2, 3])
def test_gauss_head(homosk, dy):
    CH, FN = _mods()
    g = torch.Generator().manual_seed(3)
    rows, B, pts = 6, 3, 77
    suff = torch.randn(rows, pts, 2 * dy, generator=g).requires_grad_()
    suff.data[0, 0, dy] = 25.0  # softplus threshold branch
    Y = torch.randn(B, pts, dy, generator=g)
    loc, raw = suff.split(dy, dim=-1)
    scale = 0.01 + 0.99 * torch.nn.functional.softplus(raw)
    if homosk:
        scale = scale.mean(1, keepdim=True).expand(rows, pts, dy)
    dist = torch.distributions.Independent(torch.distributions.Normal(loc, scale), 1)
    slp = dist.log_prob(Y.repeat(rows // B, 1, 1)).sum(-1)
    wl, ws, wp = (torch.randn(*s, generator=g) for s in (loc.shape, scale.shape, slp.shape))
    ((loc * wl).sum() + (scale * ws).sum() + (slp * wp).sum()).backward()
    s2 = suff.detach().to(DEV).requires_grad_()
    loc2, scale2, slp2 = FN.gauss_head(s2, Y.to(DEV), dy, homosk)
    assert_close(loc2, loc, what="loc")
    assert_close(scale2, scale, what="scale")
    assert_close(slp2, slp, what="sum_logp")
    ((loc2 * wl.to(DEV)).sum() + (scale2 * ws.to(DEV)).sum() + (slp2 * wp.to(DEV)).sum()).backward()
    assert_close(s2.grad, suff.grad, tol=2e-5, what="d_suff")


def test_mean_agg():
    CH, FN = _mods()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(5, 41, 64, generator=g).requires_grad_()
    w = torch.randn(5, 64, generator=g)
    (x.mean(1) * w).sum().backward()
    x2 = x.detach().to(DEV).requires_grad_()
    m = FN.mean_agg(FN.pack_pt(x2), 41, 64)
    assert_close(m, x.mean(1), what="mean")
    (m * w.to(DEV)).sum().backward()
    assert_close(x2.grad, x.grad, what="dmean")


def test_cpu_tensor_is_rejected_loudly():
    CH, FN = _mods()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FN.pack_pt(torch.zeros(1, 4, 4))


def test_layernorm_chain_step_matches_torch():
    """NPF_OP_LAYERNORM / NPF_OP_LAYERNORM_BWD against torch.nn.functional.layer_norm (float64):
    output, input gradient, gamma / beta gradients; feature counts that are not multiples of 32."""
    from npf_gwwaveform_amd import functional as FN
    from npf_gwwaveform_amd.chain import Chain

    g = torch.Generator().manual_seed(12)
    for n_tasks, pts, F in ((3, 45, 64), (2, 33, 100), (1, 70, 256), (2, 5, 24)):
        x = (torch.randn(n_tasks, pts, F, generator=g) * 2 + 0.5).requires_grad_(True)
        gam = (torch.rand(F, generator=g) + 0.5).requires_grad_(True)
        bet = (torch.randn(F, generator=g) * 0.1).requires_grad_(True)
        w = torch.randn(n_tasks, pts, F, generator=g)
        ref = torch.nn.functional.layer_norm(x.double(), (F,), gam.double(), bet.double(), 1e-5)
        gx, gg, gb = torch.autograd.grad((ref * w.double()).sum(), (x, gam, bet))
        xd, gd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, gam, bet))
        ch = Chain(n_tasks, pts, DEV)
        ch.input_pt(FN.pack_pt(xd), F).layernorm(gd, bd, 1e-5).output_pt()
        y = FN.unpack_pt(ch.run()[0], pts, F)
        (y * w.to(DEV)).sum().backward()
        assert_close(y, ref, what=f"layernorm F={F}")
        assert_close(xd.grad, gx, tol=2e-5, what=f"layernorm dx F={F}")
        assert_close(gd.grad, gg, tol=2e-5, what=f"layernorm dgamma F={F}")
        assert_close(bd.grad, gb, tol=2e-5, what=f"layernorm dbeta F={F}")


def test_split_merge_heads_roundtrip():
    """npf_split_heads / npf_merge_heads against the reference's view/permute formulation
    (attention.py:505-527)."""
    from npf_gwwaveform_amd import functional as FN

    g = torch.Generator().manual_seed(13)
    for B, P, F, H in ((3, 40, 128, 8), (2, 33, 64, 8), (1, 70, 256, 8), (2, 9, 96, 4)):
        x = torch.randn(B, P, F, generator=g)
        hs = F // H
        ref = x.view(B, P, H, hs).permute(2, 0, 1, 3).contiguous().view(B * H, P, hs)
        xd = x.to(DEV)
        sp = FN.split_heads(FN.pack_pt(xd), B, P, F, H)
        assert torch.equal(FN.unpack_pt(sp, P, hs).cpu(), ref)
        back = FN.merge_heads(sp, B, P, F, H)
        assert torch.equal(FN.unpack_pt(back, P, F).cpu(), x)


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def test_bf16_linear_chain_matches_bf16_emulation():
    """bf16 compute mode of the chain kernel (NPF_OP_LINEAR on npf_cast_bf16_weights images,
    v_mfma_f32_16x16x32_bf16): every layer equals fp32-accumulated products of the bf16-rounded input
    and bf16-rounded weights.  Widths that are / are not multiples of 32, skinny first and last layers."""
    CH, FN = _mods()
    g = torch.Generator().manual_seed(31)
    for n_tasks, pts, dims in ((3, 70, (256, 256, 256)), (2, 45, (2, 64, 100, 36, 4)), (1, 33, (128, 128)), (2, 40, (32, 256, 32))):
        x = torch.randn(n_tasks, pts, dims[0], generator=g)
        Ws = [torch.randn(o, i, generator=g) / i ** 0.5 for i, o in zip(dims[:-1], dims[1:])]
        bs = [torch.randn(o, generator=g) * 0.1 for o in dims[1:]]
        ref = x.double()
        for j, (W, b) in enumerate(zip(Ws, bs)):
            ref = _bf16_round(ref.float()).double() @ _bf16_round(W).double().t() + b.double()
            if j < len(Ws) - 1:
                ref = torch.relu(ref)
        prog = CH.Program(n_tasks, pts, False)
        xp = FN.pack_pt(x.to(DEV))
        out = CH.pt_empty(n_tasks, pts, dims[-1], DEV)
        prog.load_pt(xp, dims[0])
        for j, (W, b) in enumerate(zip(Ws, bs)):
            prog.linear_bf16(CH.cast_bf16_weights(W.to(DEV)), W.shape[1], W.shape[0], bias=b.to(DEV), relu=j < len(Ws) - 1)
        prog.store_pt(out, dims[-1])
        prog.launch()
        got = FN.unpack_pt(out, pts, dims[-1])
        # the only difference to the emulation: fp32 (not fp64) accumulation, and a bf16 rounding that may
        # flip when the fp32 intermediate sits next to a rounding boundary
        assert_close(got, ref, tol=2e-3, what=f"bf16 chain {dims}")


def test_bf16_weight_image_layout():
    """npf_cast_bf16_weights: values rounded to nearest-even bf16, columns permuted inside groups of 32,
    zero padding; transposed images."""
    CH, _ = _mods()
    g = torch.Generator().manual_seed(32)
    W = torch.randn(37, 70, generator=g)
    for transposed in (False, True):
        M = W.t().contiguous() if transposed else W
        img = CH.cast_bf16_weights(W.to(DEV), transposed=transposed).cpu().float()
        rows, cols = M.shape
        Kp = (cols + 31) // 32 * 32
        assert img.shape == (rows, Kp)
        ref = torch.zeros(rows, Kp)
        ref[:, :cols] = M.to(torch.bfloat16).float()
        q = torch.arange(Kp)
        grp, gg, i = q // 32, (q % 32) // 8, q % 8
        src = 32 * grp + torch.where(i < 4, 4 * gg + i, 16 + 4 * gg + (i - 4))
        assert torch.equal(img, ref[:, src])


def test_batched_weight_preparation_equals_the_single_launches():
    """npf_prepare_weights (W^T, bf16 image, bf16 image of W^T for many matrices in one launch) is bit-identical to
    npf_transpose / npf_cast_bf16_weights on each; strided rows, odd shapes, more than 32 jobs."""
    CH, _ = _mods()
    g = torch.Generator().manual_seed(61)
    specs, refs = [], []
    for n, (N, K) in enumerate([(256, 256), (37, 70), (4, 256), (256, 3), (128, 96), (1, 1)] * 7):
        big = torch.randn(N, K + 5, generator=g).to(DEV)
        W = big[:, :K] if n % 2 else big[:, :K].contiguous()      # (row stride K + 5 or K)
        kind = n % 3
        specs.append((W, kind))
        refs.append(CH.transpose(W.contiguous()) if kind == 0 else CH.cast_bf16_weights(W, transposed=kind == 2))
    outs = CH.prepare_weights(specs)
    assert len(outs) == len(refs) == 42
    for (W, kind), o, r in zip(specs, outs, refs):
        assert o.dtype == r.dtype and o.shape == r.shape
        assert torch.equal(o.view(torch.int16) if kind else o, r.view(torch.int16) if kind else r), (tuple(W.shape), kind)


def test_bf16_wgrad_matches_bf16_emulation():
    """wgrad kernel in the bf16 compute mode (NPF_WGRAD_BF16, v_mfma_f32_16x16x16_bf16): dW and db equal the
    contraction of the bf16-rounded operands (fp32 accumulation; db sums the unrounded dZ)."""
    CH, FN = _mods()
    g = torch.Generator().manual_seed(41)
    for n_tasks, pts, N, K in ((3, 70, 256, 256), (2, 45, 100, 36), (1, 200, 32, 256), (4, 33, 64, 2)):
        dz, a = torch.randn(n_tasks, pts, N, generator=g), torch.randn(n_tasks, pts, K, generator=g)
        ref = torch.einsum("bpn,bpk->nk", _bf16_round(dz).double(), _bf16_round(a).double())
        dW = torch.empty(N, K, device=DEV)
        db = torch.empty(N, device=DEV)
        CH.COMPUTE_DTYPE = "bf16"
        try:
            CH.run_wgrad([dict(dZ=FN.pack_pt(dz.to(DEV)), A=FN.pack_pt(a.to(DEV)), N=N, K=K, dW=dW, db=db)], n_tasks, pts, DEV)
        finally:
            CH.COMPUTE_DTYPE = "fp32"
        assert_close(dW, ref, tol=1e-4, what=f"bf16 wgrad {N}x{K}")
        assert_close(db, dz.double().sum((0, 1)), tol=1e-5, what="bf16 wgrad db")


def test_wgrad_split_bf16_products_equal_fp32():
    """NPF_WGRAD_F32X6 (wgrad_x6_kernel): the fp32 contraction with every operand split exactly into three bf16 terms and six
    cross products per term on the bf16 matrix pipe.  Against a float64 contraction of the same fp32 operands -- entries
    spread over six decades -- its error stays at the native fp32 kernel's (fp32 summation noise), far below one bf16 or
    even one two-term (bf16 x 2) rounding; bias gradients, per-task (key / value) jobs, accumulation into dW, blocks of
    operands wider than 256 features and widths that are not multiples of 32 included."""
    CH, FN = _mods()
    g = torch.Generator().manual_seed(47)

    def operands(n_tasks, pts, N, K):
        spread = lambda *sh: torch.randn(*sh, generator=g) * 10.0 ** (6 * torch.rand(*sh, generator=g) - 3)  # noqa: E731
        return spread(n_tasks, pts, N), spread(n_tasks, pts, K)

    def run(x6, jobs_of):
        old = CH.WGRAD_X6
        CH.WGRAD_X6 = x6
        try:
            return jobs_of()
        finally:
            CH.WGRAD_X6 = old

    for n_tasks, pts, N, K in ((3, 70, 256, 256), (2, 45, 100, 36), (1, 200, 32, 256), (4, 33, 64, 2), (2, 300, 4, 256),
                               (2, 64, 512, 384), (5, 31, 200, 130)):
        dz, a = operands(n_tasks, pts, N, K)
        ref = torch.einsum("bpn,bpk->nk", dz.double(), a.double())
        mag = torch.einsum("bpn,bpk->nk", dz.double().abs(), a.double().abs())  # what fp32 noise scales with
        ref_b = dz.double().sum((0, 1))
        errs = {}
        for x6 in (True, False):
            def jobs_of():
                dW0 = torch.randn(N, K, generator=g)
                dW, db = dW0.to(DEV), torch.zeros(N, device=DEV)
                CH.run_wgrad([dict(dZ=FN.pack_pt(dz.to(DEV)), A=FN.pack_pt(a.to(DEV)), N=N, K=K, dW=dW, db=db, accumulate=True)],
                             n_tasks, pts, DEV)
                return dW.cpu().double() - dW0.double(), db.cpu().double()
            dW, db = run(x6, jobs_of)
            errs[x6] = float(((dW - ref).abs() / mag).max())
            assert float((db - ref_b).abs().max()) <= 2e-6 * float(dz.double().abs().sum((0, 1)).max()), (x6, N, K)
        # relative to sum |dz| |a|: fp32 accumulation of ~n_tasks * pts terms; one bf16 rounding would be 4e-3, bf16 x 2 1.5e-5
        assert errs[True] <= 2e-6, (N, K, errs)
        assert errs[True] <= 4 * errs[False] + 2e-7, (N, K, errs)

    # per-task jobs (attention: dK[c][d] = sum_t dS[t][c] q[t][d]) -> PT32 outputs
    for n_tasks, pts, N, K in ((3, 130, 256, 256), (2, 64, 37, 96)):
        dz, a = operands(n_tasks, pts, N, K)
        ref = torch.einsum("bpn,bpk->bnk", dz.double(), a.double())
        mag = torch.einsum("bpn,bpk->bnk", dz.double().abs(), a.double().abs())
        for x6 in (True, False):
            def jobs_of():
                out = CH.pt_empty(n_tasks, N, K, DEV)
                CH.run_wgrad([dict(dZ=FN.pack_pt(dz.to(DEV)), A=FN.pack_pt(a.to(DEV)), N=N, K=K, dW=out, per_task=True)],
                             n_tasks, pts, DEV)
                return FN.unpack_pt(out, N, K).cpu().double()
            got = run(x6, jobs_of)
            assert float(((got - ref).abs() / mag).max()) <= 2e-6, (x6, N, K)


def test_wgrad_h16_split_once_kernel_matches_float64(monkeypatch):
    """wgrad_h16_kernel (256 x 256 jobs: every operand value split once per workgroup, half tiles of 16 points, transposed LDS
    reads, v_mfma_f32_32x32x16_bf16): dW, db and per-task outputs against float64 at fp32 summation noise -- one tile, odd tile
    counts, several jobs per launch, blocks of 512-wide tensors (strided tiles), accumulation -- and against the kernel it
    replaces (chain.WGRAD_H16 = False) to the same noise."""
    CH, FN = _mods()
    g = torch.Generator().manual_seed(53)
    spread = lambda *sh: torch.randn(*sh, generator=g) * 10.0 ** (4 * torch.rand(*sh, generator=g) - 2)  # noqa: E731
    for n_tasks, pts, N, K, n_jobs in ((1, 32, 256, 256, 1), (3, 70, 256, 256, 3), (2, 97, 512, 512, 1), (5, 33, 256, 512, 2)):
        ops = [(spread(n_tasks, pts, N), spread(n_tasks, pts, K)) for _ in range(n_jobs)]
        res = {}
        for old in (False, True):
            monkeypatch.setattr(CH, "WGRAD_H16", not old)
            dWs = [torch.zeros(N, K, device=DEV) for _ in ops]
            dbs = [torch.zeros(N, device=DEV) for _ in ops]
            CH.run_wgrad([dict(dZ=FN.pack_pt(dz.to(DEV)), A=FN.pack_pt(a.to(DEV)), N=N, K=K, dW=dW, db=db)
                          for (dz, a), dW, db in zip(ops, dWs, dbs)], n_tasks, pts, DEV)
            res[old] = ([d.cpu().double() for d in dWs], [d.cpu().double() for d in dbs])
        for j, (dz, a) in enumerate(ops):
            ref = torch.einsum("bpn,bpk->nk", dz.double(), a.double())
            mag = torch.einsum("bpn,bpk->nk", dz.double().abs(), a.double().abs())
            for old in (False, True):
                assert float(((res[old][0][j] - ref).abs() / mag).max()) <= 2e-6, (old, n_tasks, pts, N, K, j)
                assert float((res[old][1][j] - dz.double().sum((0, 1))).abs().max()) <= 2e-6 * float(dz.double().abs().sum((0, 1)).max())
    # per-task outputs (key / value gradients), accumulated into
    monkeypatch.setattr(CH, "WGRAD_H16", True)
    for n_tasks, pts in ((3, 130), (1, 16), (2, 1024)):
        dz, a = spread(n_tasks, pts, 256), spread(n_tasks, pts, 256)
        ref = torch.einsum("bpn,bpk->bnk", dz.double(), a.double())
        mag = torch.einsum("bpn,bpk->bnk", dz.double().abs(), a.double().abs())
        base = torch.randn(n_tasks, 256, 256, generator=g)
        out = FN.pack_pt(base.to(DEV)).contiguous()
        CH.run_wgrad([dict(dZ=FN.pack_pt(dz.to(DEV)), A=FN.pack_pt(a.to(DEV)), N=256, K=256, dW=out, per_task=True, accumulate=True)],
                     n_tasks, pts, DEV)
        got = FN.unpack_pt(out, 256, 256).cpu().double() - base.double()
        assert float(((got - ref).abs() / (mag + base.double().abs())).max()) <= 2e-6, (n_tasks, pts)


def test_mlp_x6_stack_forward_and_backward_match_float64():
    """npf_mlp_x6_run (csrc/mlp_x6_kernel.hip): a stack of 256 -> 256 layers with the fp32 products on the bf16 pipe (three exact
    bf16 terms per operand, six cross products).  Forward values, the input gradient and every dW / db against a float64
    evaluation of the same fp32 parameters, at the tolerances of the fp32 chain kernel (1e-5 / 1e-4 of max|ref|); odd tile
    counts (a wave without a tile), ReLU and linear last layers, more layers than one launch takes."""
    from npf_gwwaveform_amd import mlp_x6

    CH, FN = _mods()
    g = torch.Generator().manual_seed(53)
    for n_tasks, pts, n_layers, last_relu, add_at in ((3, 70, 3, False, -1), (1, 33, 1, True, -1), (2, 64, 10, True, -1),
                                                      (2, 45, 4, True, 1), (1, 40, 10, False, 8)):
        lins = [torch.nn.Linear(256, 256) for _ in range(n_layers)]
        for lin in lins:
            lin.weight.data = torch.randn(256, 256, generator=g) * 0.09
            lin.bias.data = torch.randn(256, generator=g) * 0.1
        relus = [True] * (n_layers - 1) + [last_relu]
        x = torch.randn(n_tasks, pts, 256, generator=g)
        w = torch.randn(n_tasks, pts, 256, generator=g)
        addend = torch.randn(n_tasks, pts, 256, generator=g) if add_at >= 0 else None
        # float64 reference
        ref_lins = [torch.nn.Linear(256, 256).double() for _ in lins]
        for a, b in zip(ref_lins, lins):
            a.load_state_dict({k: v.double() for k, v in b.state_dict().items()})
        xr = x.double().requires_grad_(True)
        ar = addend.double().requires_grad_(True) if addend is not None else None
        h = xr
        for i, (lin, r) in enumerate(zip(ref_lins, relus)):
            h = lin(h) + (ar if i == add_at else 0.0)
            h = torch.relu(h) if r else h
        (h * w.double()).sum().backward()
        # HIP
        dev_lins = [lin.to(DEV) for lin in lins]
        xd = x.to(DEV).requires_grad_(True)
        ad = addend.to(DEV).requires_grad_(True) if addend is not None else None
        y_pt = mlp_x6.run_stack(FN.pack_pt(xd), pts, dev_lins, relus, addend=FN.pack_pt(ad) if ad is not None else None,
                                add_at=max(add_at, 0))
        y = FN.unpack_pt(y_pt, pts, 256)
        (y * w.to(DEV)).sum().backward()
        assert_close(y, h, tol=1e-5, what=f"x6 stack forward ({n_layers} layers)")
        assert_close(xd.grad, xr.grad, tol=1e-4, what="x6 stack dx")
        if ad is not None:
            assert_close(ad.grad, ar.grad, tol=1e-4, what="x6 stack d(addend)")
        for i, (a, b) in enumerate(zip(dev_lins, ref_lins)):
            assert_close(a.weight.grad, b.weight.grad, tol=1e-4, what=f"x6 stack dW[{i}]")
            assert_close(a.bias.grad, b.bias.grad, tol=1e-4, what=f"x6 stack db[{i}]")


def test_mlp_x6_split_edge_values():
    """Edge semantics of the three-term split (csrc/mlp_x6_kernel.hip, x6m_split) against fp32 ``F.linear``:
    (1) huge finite inputs up to 3.38e38 (just under the largest bf16, 3.3895e38): exact like any other value;
    (2) subnormal-range and zero inputs: zero to within a denormal flush;
    (3) a non-finite input (inf / nan) makes that point's outputs non-finite, as the reference's fp32 addmm does (inf * w, and
        inf - inf = nan across mixed signs), and leaves every other point untouched;
    (4) DOCUMENTED DEVIATION: a finite input in (3.3895e38, FLT_MAX] rounds to bf16 infinity in the first term, so its point
        comes out non-finite where fp32 arithmetic would still be finite -- the last 0.4 % of the fp32 range, where the
        reference's own next layer overflows."""
    from npf_gwwaveform_amd import mlp_x6

    CH, FN = _mods()
    g = torch.Generator().manual_seed(61)
    lin = torch.nn.Linear(256, 256)
    lin.weight.data = torch.randn(256, 256, generator=g) * 1e-3  # (products and sums stay finite)
    lin.bias.data.zero_()
    W = lin.weight.detach().clone()
    pts = 64
    x = torch.randn(1, pts, 256, generator=g)
    x[0, 1, :] = torch.where(torch.rand(256, generator=g) < 0.5, 3.38e38, -3.38e38)   # (1)
    x[0, 2, ::2] = 1e-39                                                               # (2) subnormal
    x[0, 2, 1::2] = 0.0
    x[0, 3, 7] = float("inf")                                                          # (3)
    x[0, 4, 9] = float("nan")
    x[0, 5, 11] = 3.40e38                                                              # (4)
    with torch.no_grad():
        ref64 = torch.nn.functional.linear(x.double(), W.double())
        y = FN.unpack_pt(mlp_x6.run_stack(FN.pack_pt(x.to(DEV)), pts, [lin.to(DEV)], [False]), pts, 256).cpu()
    finite_pts = [p for p in range(pts) if p not in (3, 4, 5)]
    assert torch.isfinite(y[0, finite_pts]).all()
    for p in (0, 1, 2, 6):
        m = (x[0, p].double().abs() @ W.double().abs().T).max()
        assert float((y[0, p].double() - ref64[0, p]).abs().max()) <= 1e-6 * float(m) + 1e-36, p  # (+ denormal flush)
    for p in (3, 4, 5):
        assert not torch.isfinite(y[0, p]).any(), p
    assert torch.isfinite(torch.nn.functional.linear(x[0, 5], W)).all()  # (fp32 itself is still finite there)


def test_mlp_x6_stack_with_output_layer_rows_matches_float64():
    """The decoder's 256 -> 4 output layer riding on the stack (``run_stack(tail=...)``): rows forward through a chain launch,
    its dgrad formed inside the stack's dgrad launch (npf_mlp_x6_run_rows), its dW / db from the rows as a PT32 operand --
    rows, dx, d(addend) and every dW / db against float64, at the tolerances of the fp32 kernels; odd tile counts, more
    layers than one launch takes."""
    from npf_gwwaveform_amd import mlp_x6

    CH, FN = _mods()
    g = torch.Generator().manual_seed(59)
    for n_tasks, pts, n_layers, add_at in ((3, 32, 2, -1), (1, 96, 6, 1), (2, 64, 10, -1)):
        lins = [torch.nn.Linear(256, 256) for _ in range(n_layers)] + [torch.nn.Linear(256, 4)]
        for lin in lins:
            lin.weight.data = torch.randn(lin.weight.shape, generator=g) * 0.09
            lin.bias.data = torch.randn(lin.bias.shape, generator=g) * 0.1
        assert mlp_x6.tail_usable(lins[-1], pts) and not mlp_x6.tail_usable(lins[-1], pts + 1)
        x = torch.randn(n_tasks, pts, 256, generator=g)
        w = torch.randn(n_tasks, pts, 4, generator=g)
        addend = torch.randn(n_tasks, pts, 256, generator=g) if add_at >= 0 else None
        ref_lins = [torch.nn.Linear(l.in_features, l.out_features).double() for l in lins]
        for a, b in zip(ref_lins, lins):
            a.load_state_dict({k: v.double() for k, v in b.state_dict().items()})
        xr = x.double().requires_grad_(True)
        ar = addend.double().requires_grad_(True) if addend is not None else None
        h = xr
        for i, lin in enumerate(ref_lins[:-1]):
            h = torch.relu(lin(h) + (ar if i == add_at else 0.0))
        rows_ref = ref_lins[-1](h)
        (rows_ref * w.double()).sum().backward()
        dev_lins = [lin.to(DEV) for lin in lins]
        xd = x.to(DEV).requires_grad_(True)
        ad = addend.to(DEV).requires_grad_(True) if addend is not None else None
        rows = mlp_x6.run_stack(FN.pack_pt(xd), pts, dev_lins[:-1], [True] * n_layers,
                                addend=FN.pack_pt(ad) if ad is not None else None, add_at=max(add_at, 0), tail=dev_lins[-1])
        assert tuple(rows.shape) == (n_tasks, pts, 4)
        (rows * w.to(DEV)).sum().backward()
        assert_close(rows, rows_ref, tol=1e-5, what=f"x6 stack + output rows ({n_layers} layers)")
        assert_close(xd.grad, xr.grad, tol=1e-4, what="x6 stack + rows dx")
        if ad is not None:
            assert_close(ad.grad, ar.grad, tol=1e-4, what="x6 stack + rows d(addend)")
        for i, (a, b) in enumerate(zip(dev_lins, ref_lins)):
            assert_close(a.weight.grad, b.weight.grad, tol=1e-4, what=f"x6 stack + rows dW[{i}]")
            assert_close(a.bias.grad, b.bias.grad, tol=1e-4, what=f"x6 stack + rows db[{i}]")


def _pack_pt16_reference(x):
    """Row-major [B, P, F] -> PT16 (bf16 tiles) with plain torch ops (the layout of chain.pt16_shape)."""
    B, P, F = x.shape
    Fp, tiles = (F + 31) // 32 * 32, (P + 31) // 32
    xp = torch.zeros(B, tiles * 32, Fp)
    xp[:, :P, :F] = x
    # feature f = 32 s + 16 h + 4 g + i  ->  [B, tiles, s, g, p, (h, i)]
    v = xp.view(B, tiles, 32, Fp // 32, 2, 4, 4).permute(0, 1, 3, 5, 2, 4, 6).contiguous()
    return v.view(B, tiles, Fp // 8, 32, 8).to(torch.bfloat16)


def _unpack_pt16_reference(t, P, F):
    """PT16 tensor -> row-major [B, P, F] float (inverse of _pack_pt16_reference)."""
    B, tiles, rows, _, _ = t.shape
    Fp = rows * 8
    v = t.float().cpu().view(B, tiles, Fp // 32, 4, 32, 2, 4).permute(0, 1, 4, 2, 5, 3, 6).contiguous()
    return v.view(B, tiles * 32, Fp)[:, :P, :F]


def test_pt16_operands_of_the_chain_ops():
    """NPF_F_P16 on every chain op that takes it (bf16 instance): LOAD_PT, a LINEAR's addend (generic slab loop:
    a PT16 addend keeps a layer off the pipelined path) and its fused relu-backward mask (pipelined path at
    256 -> 256), ADD_PT (+relu), MASK_POS, ROWDOT_PT + SOFTMAX_BWD, STORE_PT -- against the same operations on the
    bf16-rounded tensors."""
    CH, FN = _mods()
    g = torch.Generator().manual_seed(51)
    for n_tasks, pts, K, N in ((2, 70, 256, 256), (3, 45, 64, 100), (1, 33, 32, 256)):
        x, a1, a2, m = (torch.randn(n_tasks, pts, f, generator=g) for f in (K, N, N, N))
        W = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g) * 0.1
        r = _bf16_round
        lin = r(x).double() @ r(W).double().t() + b.double()
        y1 = torch.relu(torch.relu(lin + r(a1).double()) + r(a2).double())     # addend, then add_pt with relu
        y2 = torch.where(r(m) > 0, lin, torch.zeros_like(lin))                  # fused mask
        y3 = torch.where(r(m) > 0, y1, torch.zeros_like(y1))                    # mask_pos as its own op
        P_ = torch.softmax(torch.randn(n_tasks, pts, N, generator=g), -1)
        y4 = 0.5 * r(P_).double() * (lin - (lin * r(P_).double()).sum(-1, keepdim=True))  # softmax backward
        x16, a16, b16, m16, p16 = (_pack_pt16_reference(t).to(DEV) for t in (x, a1, a2, m, P_))
        img = CH.cast_bf16_weights(W.to(DEV))
        outs = []
        for variant in range(4):
            prog = CH.Program(n_tasks, pts, False)
            prog.load_pt(x16, K)
            if variant == 0:
                prog.linear_bf16(img, K, N, bias=b.to(DEV), relu=True, addend=a16)
                prog.add_pt(b16, N, relu=True)
            elif variant == 1:
                prog.linear_bf16(img, K, N, bias=b.to(DEV))
                prog.mask_pos(m16, N)                       # (fuses into the layer above)
            elif variant == 2:
                prog.linear_bf16(img, K, N, bias=b.to(DEV), relu=True, addend=a16)
                prog.add_pt(b16, N, relu=True)
                prog.mask_pos(m16, N)                       # (the layer has an addend: stays its own op)
            else:
                prog.linear_bf16(img, K, N, bias=b.to(DEV))
                prog.rowdot_pt(p16, N)
                prog.softmax_bwd(p16, N, 0.5)
            o16 = CH.pt16_empty(n_tasks, pts, N, DEV)
            o32 = CH.pt_empty(n_tasks, pts, N, DEV)
            prog.store_pt(o16, N)
            prog.store_pt(o32, N)
            prog.launch()
            outs.append((o16, o32))
        for (o16, o32), ref, what in zip(outs, (y1, y2, y3, y4), ("addend + add_pt", "fused mask", "mask_pos", "softmax bwd")):
            got32 = FN.unpack_pt(o32, pts, N)
            assert_close(got32, ref, tol=2e-3, what=f"pt16 {what} {K}->{N}")
            # the PT16 store is the bf16 rounding of the fp32 store, element for element
            assert torch.equal(_unpack_pt16_reference(o16, pts, N), got32.cpu().to(torch.bfloat16).float()), what


def test_bf16_wgrad_with_pt16_operands():
    """wgrad kernel, bf16 variant, with one or both operands given as PT16 tensors (bf16 tiles): same result
    as with the fp32 tiles of the same bf16-rounded values."""
    CH, FN = _mods()
    g = torch.Generator().manual_seed(43)
    for n_tasks, pts, N, K in ((3, 70, 256, 256), (2, 45, 100, 36), (1, 200, 32, 256), (2, 33, 64, 160)):
        dz, a = _bf16_round(torch.randn(n_tasks, pts, N, generator=g)), _bf16_round(torch.randn(n_tasks, pts, K, generator=g))
        ref = torch.einsum("bpn,bpk->nk", dz.double(), a.double())
        for z16, a16 in ((True, True), (True, False), (False, True)):
            dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
            opz = _pack_pt16_reference(dz).to(DEV) if z16 else FN.pack_pt(dz.to(DEV))
            opa = _pack_pt16_reference(a).to(DEV) if a16 else FN.pack_pt(a.to(DEV))
            assert tuple(opz.shape) == (CH.pt16_shape(n_tasks, pts, N) if z16 else CH.pt_shape(n_tasks, pts, N))
            CH.COMPUTE_DTYPE = "bf16"
            try:
                CH.run_wgrad([dict(dZ=opz, A=opa, N=N, K=K, dW=dW, db=db)], n_tasks, pts, DEV)
            finally:
                CH.COMPUTE_DTYPE = "fp32"
            assert_close(dW, ref, tol=1e-5, what=f"pt16 wgrad {N}x{K} z16={z16} a16={a16}")
            assert_close(db, dz.double().sum((0, 1)), tol=1e-5, what="pt16 wgrad db")
