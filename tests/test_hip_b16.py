"""GPU parity tests of the b16 programs (csrc/b16_kernel.hip, ``npf_b16_run``): the fused context / target sides in the bf16
compute mode (BASELINE config 3).  The reference has no bf16 path (npf/architectures/mlp.py:95-109 and attention.py:129-164 are
fp32), so the arithmetic under test is the pinned fp32 restatement plus the rounding points of DESIGN.md 4, and the gate is the
teacher-forced one of tests/teacher.py: every op of every launch against the launch's OWN stored tensors at fp32-accumulation
tolerances (2e-6 of max|ref| per op, 1e-5 for sums over all points), bf16 stores accepted within half an ulp.  The end-to-end
comparison against the bf16 oracle (tests/test_hip_bf16.py) runs on these launches too."""
import ctypes as C

import pytest
import torch

from test_hip_x6 import DEV, _build, _unpermute

pytestmark = pytest.mark.gpu


@pytest.fixture
def bf16_mode():
    import npf_gwwaveform_amd as A

    A.set_compute_dtype("bf16")
    yield
    A.set_compute_dtype("fp32")


@pytest.mark.parametrize("pts,width", [(256, 256), (200, 256), (37, 128)])
def test_b16_task_images_are_the_rounded_values(pts, width):
    from npf_gwwaveform_amd import functional as FN
    from npf_gwwaveform_amd import x6

    g = torch.Generator().manual_seed(pts)
    M = torch.randn(3, pts, width, generator=g)
    row, tr = x6.task_images(FN.pack_pt(M.to(DEV)), pts, width=width, bf16=True)
    assert tuple(row.shape) == (3, width, width) and row.dtype == torch.bfloat16
    want = torch.zeros(3, width, width)
    want[:, :pts] = M
    r16 = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    assert torch.equal(_unpermute(row.cpu()).float(), r16(want))
    assert torch.equal(_unpermute(tr.cpu()).float(), r16(want.transpose(1, 2)))


@pytest.mark.parametrize("variant", [1, 2], ids=["8_waves_64_row_slabs", "4_waves_32_row_slabs"])
@pytest.mark.parametrize("B,C,T,L,dx,dy,r", [(2, 256, 64, 4, 1, 2, 256), (3, 200, 96, 2, 2, 1, 256), (1, 129, 33, 1, 1, 2, 256),
                                             (9, 250, 288, 2, 1, 2, 256), (3, 128, 64, 2, 1, 2, 128), (4, 5, 70, 1, 2, 1, 128)])
def test_b16_fused_sides_teacher_forced(B, C, T, L, dx, dy, r, variant, monkeypatch, bf16_mode):
    """Context side + target side + their backward launches on a whole model step; every op of every program and every gradient
    job from the launches' own stored tensors."""
    import npf_gwwaveform_amd as A
    import teacher
    from npf_gwwaveform_amd import chain as CH
    from npf_gwwaveform_amd import x6

    monkeypatch.setattr(x6, "B16_VARIANT", variant)
    model = _build(r=r, L=L, dx=dx, dy=dy, seed=B + C).train()
    assert x6.target_side_usable(model, C, T) and x6.context_side_usable(model, C)
    g = torch.Generator().manual_seed(C + T)
    Xc, Xt = torch.rand(B, C, dx, generator=g) * 2 - 1, torch.rand(B, T, dx, generator=g) * 2 - 1
    Yc, Yt = torch.randn(B, C, dy, generator=g), torch.randn(B, T, dy, generator=g)
    trace = []
    monkeypatch.setattr(CH, "TRACE", trace)
    out = model(Xc.to(DEV), Yc.to(DEV), Xt.to(DEV), Yt.to(DEV))
    A.CNPFLoss()(out, Yt.to(DEV)).backward()
    torch.cuda.synchronize()
    monkeypatch.setattr(CH, "TRACE", None)
    progs = [rec for rec in trace if rec[0] == "prog"]
    assert len(progs) == 4 and all(rec[1].bf16 for rec in progs), [rec[0] for rec in trace]
    assert sum(rec[0] == "wgrad" for rec in trace) == 2
    rep = teacher.check_trace(trace)
    bad = rep.failures()
    assert rep.rows and not bad, "teacher-forced checks failed:\n" + "\n".join(f"  {w}: {e:.3e} > {t:.0e}" for w, e, t in bad[:20])
    assert rep.unforced == 0  # (a training step stores what every op produces)
    print(rep.summary())
    for p in model.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()


def test_b16_run_rejects_what_it_does_not_take():
    """Row-major operands, PT32 masks, other widths and unknown variants are NPF_EINVAL -- nothing is launched."""
    from npf_gwwaveform_amd import _lib as L
    from npf_gwwaveform_amd import chain as CH

    EINVAL = -1  # include/npf_hip.h NPF_EINVAL
    lib = L.load()
    x = CH.pt_empty(1, 32, 256, DEV)
    img = torch.zeros(256, 256, dtype=torch.bfloat16, device=DEV)

    def run(width=256, variant=0, **kw):
        arr = (L.NpfX6Op * 1)()
        arr[0].in_pt, arr[0].w_img = x.data_ptr(), img.data_ptr()
        for k, v in kw.items():
            if k == "flags":
                arr[0].reserved[0] = v
            else:
                setattr(arr[0], k, v)
        return lib.npf_b16_run(arr, 1, None, None, None, 1, 1, 0, width, variant, None)

    assert run(width=512) == EINVAL
    assert run(variant=3) == EINVAL
    assert run(flags=L.X6_IN_RM) == EINVAL
    assert run(flags=L.X6_ADD_RM, addend=x.data_ptr()) == EINVAL
    assert run(mask=x.data_ptr()) == EINVAL
    assert run(w_task_stride=256 * 256 * 2) == EINVAL  # (per-task weights in a flat launch)
    assert isinstance(C.sizeof(L.NpfX6Op), int)
