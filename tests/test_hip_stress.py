"""Randomised shape sweep of the HIP models against the CPU oracle (forward, loss, all gradients).
Opt-in (NPF_STRESS=<number of cases>): the fixed sweep of test_hip_sweep.py is the gate, this is the
net for shapes nobody thought of."""
import os
import random

import numpy as np
import pytest
import torch

import specs
from helpers import EpsIndependent, assert_close, build_loss, build_model
from oracle import npf_oracle as O
from test_hip_sweep import _oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_CASES = int(os.environ.get("NPF_STRESS", "0"))
# "bf16": the bf16 compute mode against the oracle's bf16 emulation (oracle.matmul_mode), with the relative-L2 / max-norm
# gate of tests/test_hip_bf16.py (what that tolerance is made of is explained there)
DTYPE = os.environ.get("NPF_STRESS_DTYPE", "fp32")


FUSED_ONLY = os.environ.get("NPF_STRESS_FUSED", "0") == "1"  # draw only what the fused x6 sides take (x6.py)


def _random_case(rng: random.Random) -> dict:
    if FUSED_ONLY:
        # attentive models with scaled-dot attention at the widths the x6 programs have instances for, any point counts
        kind = rng.choice(["AttnCNP", "AttnLNP"])
        r = rng.choice([128, 256])
        case = dict(kind=kind, r=r, L_xy=rng.randint(1, 4), L_dec=rng.randint(1, 4), dx=rng.randint(1, 3), dy=rng.randint(1, 2),
                    B=rng.randint(1, 5), C=rng.choice([1, 2, 17, 31, 32, 33, 64, 100, 127, 128, 129, 200, 255, 256]),
                    T=rng.choice([1, 3, 16, 31, 32, 33, 65, 128, 200, 257]), is_heteroskedastic=rng.random() < 0.8)
        if kind == "AttnLNP":
            case.update(n_z=rng.choice([1, 1, 2]), is_q_zCct=rng.random() < 0.5)
        return case
    kind = rng.choice(["CNP", "LNP", "AttnCNP", "AttnLNP"])
    r = rng.choice([8, 12, 20, 32, 40, 64, 72, 96, 100, 128, 160, 200, 256])
    case = dict(kind=kind, r=r, L_xy=rng.randint(1, 3), L_dec=rng.randint(1, 4), dx=rng.randint(1, 3), dy=rng.randint(1, 3),
                B=rng.randint(1, 4), C=rng.choice([1, 2, 5, 17, 31, 32, 33, 64, 100, 255, 256, 257, 300]),
                T=rng.choice([1, 3, 16, 31, 32, 33, 65, 128, 200]),
                is_heteroskedastic=rng.random() < 0.8)
    if kind in ("LNP", "AttnLNP"):
        case.update(n_z=rng.randint(1, 3), is_q_zCct=rng.random() < 0.5)
    if kind == "LNP":
        case["encoded_path"] = rng.choice(["latent", "both"])
    if kind in ("AttnCNP", "AttnLNP") and r % 32 == 0 and rng.random() < 0.3:
        case["attention"] = rng.choice(["multihead", "transformer"])
    # the reference's other MLP / merge options
    if rng.random() < 0.25:
        case["is_res"] = True
    if case.get("attention", "scaledot") == "scaledot" and rng.random() < 0.25:
        case["x_transf_dim"] = rng.choice([8, 20, 32, 64, 100, 128, 256])
    if rng.random() < 0.2:
        case["is_sum_merge"] = False
    return case


def _bf16_end_to_end(model, out, loss, ref_p, ref_out, ref_loss, sanity=False):
    from test_hip_bf16 import SANITY_L2, SANITY_MAX, TOL_GRAD_L2, TOL_MAX, TOL_OUT_L2, _check

    if sanity:  # (the bound that holds even with flipped roundings on record)
        TOL_GRAD_L2, TOL_MAX = SANITY_L2, SANITY_MAX

    _check(out[0].base_dist.loc, ref_out["loc"], TOL_OUT_L2, TOL_MAX, "loc")
    _check(out[0].base_dist.scale, ref_out["scale"], TOL_OUT_L2, TOL_MAX, "scale")
    np.testing.assert_allclose(loss.item(), ref_loss.item(), rtol=1e-3)
    for k, p in model.named_parameters():
        ref = ref_p[k].grad if ref_p[k].grad is not None else torch.zeros_like(ref_p[k])
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        if float(ref.abs().max()) > 0:
            _check(got, ref, TOL_GRAD_L2, TOL_MAX, f"grad {k}")


@pytest.mark.skipif(N_CASES == 0, reason="set NPF_STRESS=<n> to run the randomised sweep")
def test_random_shapes_match_oracle():
    rng = random.Random(int(os.environ.get("NPF_STRESS_SEED", "0")))
    failures, ties, flipped = [], 0, 0
    for i in range(N_CASES):
        case = _random_case(rng)
        if DTYPE == "bf16" and (case.get("attention", "scaledot") != "scaledot" or case["C"] > 256
                                or not case.get("is_sum_merge", True)):
            continue  # (the bf16 emulation models the fused scaled-dot family; the rest keeps fp32 attention launches)
        try:
            params = specs.make_params(case, seed=100 + i)
            inp = specs.make_inputs(case, seed=200 + i)
            O.RELU_MARGINS = []
            ref_p, ref_out, ref_loss = _oracle(case, inp, params, mode=DTYPE)
            margin, O.RELU_MARGINS = min(O.RELU_MARGINS, default=1.0), None
            model = build_model(case, DEV, params=params)
            dinp = {k: v.to(DEV) for k, v in inp.items()}
            if "eps" in dinp:
                EpsIndependent.eps = dinp["eps"]
            crit = build_loss(case)
            model.train()
            crit.train()
            import npf_gwwaveform_amd as A

            from npf_gwwaveform_amd import chain as CH

            A.set_compute_dtype(DTYPE)
            trace = [] if DTYPE == "bf16" else None
            CH.TRACE = trace
            try:
                out = model(dinp["X_cntxt"], dinp["Y_cntxt"], dinp["X_trgt"], dinp["Y_trgt"])
                loss = crit(out, dinp["Y_trgt"])
                loss.backward()
            finally:
                A.set_compute_dtype("fp32")
                CH.TRACE = None
            if DTYPE == "bf16":
                import teacher

                # the gate proper: every step against the launch's own stored inputs; the end-to-end comparison below is
                # the sanity bound, and a case beyond it counts as explained only by flipped roundings on record
                rep = teacher.check_trace(trace)
                assert not rep.failures(), f"teacher-forced: {rep.failures()[:5]}"
                try:
                    _bf16_end_to_end(model, out, loss, ref_p, ref_out, ref_loss)
                except AssertionError as e:
                    if rep.n_flips == 0:
                        raise
                    _bf16_end_to_end(model, out, loss, ref_p, ref_out, ref_loss, sanity=True)
                    flipped += 1
                    print(f"case {i}: beyond the end-to-end bound ({repr(e)[:120]}) with {rep.n_flips} flipped roundings, "
                          f"teacher-forced gate clean ({rep.summary()}); e.g. {rep.flips[:2]}")
                continue
            tol_out, tol_loss, tol_grad = 1e-5, 5e-5, 2e-4
            assert_close(out[0].base_dist.loc, ref_out["loc"], tol=tol_out, what="loc")
            assert_close(out[0].base_dist.scale, ref_out["scale"], tol=tol_out, what="scale")
            np.testing.assert_allclose(loss.item(), ref_loss.item(), rtol=tol_loss)
            for k, p in model.named_parameters():
                ref = ref_p[k].grad if ref_p[k].grad is not None else torch.zeros_like(ref_p[k])
                got = p.grad if p.grad is not None else torch.zeros_like(p)
                assert_close(got, ref, tol=tol_grad, what=f"grad {k}")
        except Exception as e:  # collect, report all
            if isinstance(e, AssertionError) and "grad" in repr(e) and margin < (2e-5 if DTYPE == "bf16" else 2e-7):
                # a ReLU pre-activation within fp32 rounding of zero (bf16 mode: within the shift one flipped bf16
                # rounding upstream causes): its derivative is decided by rounding noise, both gradients are
                # valid results (forward outputs and loss were checked above)
                ties += 1
                continue
            failures.append((i, case, repr(e)[:300]))
        finally:
            O.RELU_MARGINS = None
    print(f"{N_CASES} cases, {ties} skipped for a ReLU tie, {flipped} beyond the end-to-end bound with flipped roundings listed "
          f"(teacher-forced gate clean), {len(failures)} failures")
    assert not failures, "\n".join(f"case {i}: {c}\n   {e}" for i, c, e in failures)
