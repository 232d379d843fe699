"""Development aid: accuracy of the split-product weight-gradient kernel (NPF_WGRAD_F32X6) and of the fp32-MFMA kernel against a
float64 contraction of the same fp32 operands (entries spread over six decades); run on the GPU box."""
import sys, os, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from npf_gwwaveform_amd import chain as CH, functional as FN
DEV = "cuda:0"
g = torch.Generator().manual_seed(47)
for n_tasks, pts, N, K in ((1, 32, 32, 32), (3, 70, 256, 256), (2, 45, 100, 36)):
    dz = torch.randn(n_tasks, pts, N, generator=g) * 10.0 ** (6 * torch.rand(n_tasks, pts, N, generator=g) - 3)
    a = torch.randn(n_tasks, pts, K, generator=g) * 10.0 ** (6 * torch.rand(n_tasks, pts, K, generator=g) - 3)
    ref = torch.einsum("bpn,bpk->nk", dz.double(), a.double())
    mag = torch.einsum("bpn,bpk->nk", dz.double().abs(), a.double().abs())
    refb = dz.double().sum((0, 1)); magb = dz.double().abs().sum((0, 1))
    for x6 in (True, False):
        CH.WGRAD_X6 = x6
        dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
        CH.run_wgrad([dict(dZ=FN.pack_pt(dz.to(DEV)), A=FN.pack_pt(a.to(DEV)), N=N, K=K, dW=dW, db=db)], n_tasks, pts, DEV)
        e = ((dW.cpu().double() - ref).abs() / mag)
        eb = ((db.cpu().double() - refb).abs() / magb)
        print(f"{N}x{K} x6={x6}: dW err/mag max {e.max():.3e} mean {e.mean():.3e}; db err/mag max {eb.max():.3e}")
    # which split terms matter: emulate on CPU
    def split(x):
        h = x.to(torch.bfloat16).float(); r = x - h; m = r.to(torch.bfloat16).float(); r2 = r - m; l = r2.to(torch.bfloat16).float()
        return h, m, l
    zh, zm, zl = split(dz)
    for name, v in (("hi", zh), ("hi+mid", zh + zm), ("hi+mid+lo", zh.double() + zm.double() + zl.double())):
        print("   db emu", name, float(((v.double().sum((0, 1)) - refb).abs() / magb).max()))
