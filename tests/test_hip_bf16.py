"""Parity gate of the bf16 compute mode (BASELINE configs 3 and 4-bf16): the HIP path
with ``set_compute_dtype("bf16")`` against the oracle's bf16 emulation (``oracle.npf_oracle.matmul_mode``:
the pinned fp32 restatement with every contraction's operands rounded to bfloat16 exactly where the kernels
round them -- weights, layer inputs, keys / values, probabilities, and in the backward pass dZ, dO, dS and
the saved activations).  Forward outputs, latent statistics, the loss and EVERY gradient tensor in full must
agree to 2e-3 of max|ref| per tensor (what remains is fp32 summation order plus the rare bf16 rounding it
flips); a wrong-but-correlated bf16 backward cannot pass this.  The fp32 reference itself is ~1e-2 away
(reported by test_hip_models.py::test_bf16_compute_mode_tracks_fp32_reference), so the gate is 5x tighter
than the mode's own error."""
import numpy as np
import pytest
import torch

import specs
from helpers import EpsIndependent, assert_close, build_loss, build_model
from test_hip_sweep import SWEEP, _oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-3

CASES = {k: specs.CASES[k] for k in ("g1_cnp_c1", "g2_lnp_both_c1", "g2_lnp_latent_c1", "g3s_attncnp_r64", "g4s_attnlnp_r64",
                                     "g4s_attnlnp_r64_noqzcct", "g6_attncnp_ragged", "g6_attncnp_c1pt", "g6_cnp_homosk",
                                     "g6_attncnp_r128", "g10_attnlnp_nll_nz8")}
CASES.update({k: SWEEP[k] for k in ("cnp_r48", "cnp_r100_dx3_dy1", "cnp_r200_L1", "lnp_latent_nz3_r40", "lnp_both_nz2_r72",
                                    "attncnp_r96", "attncnp_r160_c255", "attncnp_r256_c256_t100", "attncnp_r44",
                                    "attnlnp_nz3_r64", "attnlnp_nz2_r104_noq",
                                    # BASELINE config 3's model and point counts (batch 2): full gradients
                                    "attncnp_c2_full", "attnlnp_c2_full")})


def _hip_bf16(case, inp, params):
    import npf_gwwaveform_amd as A

    model = build_model(case, DEV, params=params)
    dinp = {k: v.to(DEV) for k, v in inp.items()}
    if "eps" in dinp:
        EpsIndependent.eps = dinp["eps"]
    crit = build_loss(case)
    model.train()
    crit.train()
    A.set_compute_dtype("bf16")
    try:
        out = model(dinp["X_cntxt"], dinp["Y_cntxt"], dinp["X_trgt"], dinp["Y_trgt"])
        loss = crit(out, dinp["Y_trgt"])
        loss.backward()
    finally:
        A.set_compute_dtype("fp32")
    return model, out, loss


@pytest.mark.parametrize("name", list(CASES))
def test_bf16_mode_matches_the_bf16_oracle(name):
    case = CASES[name]
    params = specs.make_params(case, seed=11)
    inp = specs.make_inputs(case, seed=4321)
    ref_p, ref_out, ref_loss = _oracle(case, inp, params, mode="bf16")
    fp_p, fp_out, _ = _oracle(case, inp, params, mode="fp32")
    model, out, loss = _hip_bf16(case, inp, params)

    assert_close(out[0].base_dist.loc, ref_out["loc"], tol=TOL, what="loc")
    assert_close(out[0].base_dist.scale, ref_out["scale"], tol=TOL, what="scale")
    np.testing.assert_allclose(loss.item(), ref_loss.item(), rtol=TOL)
    if out[1] is not None:
        assert_close(out[1], ref_out["z_samples"], tol=TOL, what="z_samples")
        assert_close(out[2].base_dist.loc, ref_out["q_zCc"][0], tol=TOL, what="q_zCc.loc")
        assert_close(out[2].base_dist.scale, ref_out["q_zCc"][1], tol=TOL, what="q_zCc.scale")
    worst = 0.0
    for k, p in model.named_parameters():
        ref = ref_p[k].grad if ref_p[k].grad is not None else torch.zeros_like(ref_p[k])
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        assert_close(got, ref, tol=TOL, what=f"grad {k}")
        if float(ref.abs().max()) > 0:
            worst = max(worst, float((got.cpu() - ref).abs().max() / ref.abs().max()))
    # the bf16 instances really ran: the result is NOT the fp32 one
    d32 = float((out[0].base_dist.loc.detach().cpu() - fp_out["loc"]).abs().max() / fp_out["loc"].abs().max())
    assert d32 > 1e-5, d32
    print(f"{name}: worst gradient error {worst:.2e} of max|ref| (bf16 oracle); loc vs fp32 oracle {d32:.2e}")


def test_full_size_config3_properties():
    """BASELINE config 3 at full size (bf16 mode, 1024 tasks x 1024 targets): the size-independent properties
    of test_hip_models.py::test_full_size_config2_properties, in the bf16 compute mode.  Task and target
    independence hold exactly in exact arithmetic and to fp32 rounding here (the rounding points do not depend
    on the batch); gradient linearity over the batch to the bf16 gate."""
    import npf_gwwaveform_amd as A
    from npf_gwwaveform_amd.train import synthetic_waveform_batch

    B = 1024
    case = dict(specs.CASES["g3_attncnp_c2"], B=B)
    model = build_model(case, DEV).train()
    batch = synthetic_waveform_batch(B, case["C"], case["T"], 7, DEV)
    crit = A.CNPFLoss()
    A.set_compute_dtype("bf16")
    try:
        p = model(batch["X_cntxt"], batch["Y_cntxt"], batch["X_trgt"], batch["Y_trgt"])[0]
        loc, scale = p.base_dist.loc.detach(), p.base_dist.scale.detach()
        assert torch.isfinite(loc).all() and torch.isfinite(scale).all() and (scale >= 0.01).all()
        for i in (0, 517, 1023):
            one = {k: v[i:i + 1] for k, v in batch.items()}
            pi = model(one["X_cntxt"], one["Y_cntxt"], one["X_trgt"], one["Y_trgt"])[0]
            assert_close(pi.base_dist.loc, loc[:, i:i + 1], tol=1e-5, what=f"task {i} alone: loc")
            assert_close(pi.base_dist.scale, scale[:, i:i + 1], tol=1e-5, what=f"task {i} alone: scale")
        sub = slice(100, 357)
        ps = model(batch["X_cntxt"], batch["Y_cntxt"], batch["X_trgt"][:, sub], batch["Y_trgt"][:, sub])[0]
        assert_close(ps.base_dist.loc, loc[:, :, sub], tol=1e-5, what="target subset: loc")
        perm = torch.randperm(batch["X_cntxt"].shape[1], device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
        pp = model(batch["X_cntxt"][:, perm], batch["Y_cntxt"][:, perm], batch["X_trgt"], batch["Y_trgt"])[0]
        assert_close(pp.base_dist.loc, loc, tol=TOL, what="context permutation: loc")

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
    finally:
        A.set_compute_dtype("fp32")
    np.testing.assert_allclose(l_full, 0.5 * (l_a + l_b), rtol=1e-5)
    for k in g_full:
        assert_close(g_full[k], 0.5 * (g_a[k] + g_b[k]), tol=1e-4, what=f"grad linearity {k}")
