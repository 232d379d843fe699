import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import test_hip_x6 as tx
from npf_gwwaveform_amd import chain as CH, functional as FN, x6
x6.VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, C, T, L, dx, dy = 2, 256, 64, 4, 1, 2
model = tx._build(L=L, dx=dx, dy=dy, seed=B * 7 + C)
g = torch.Generator().manual_seed(C + T)
X = torch.rand(B, T, dx, generator=g) * 2 - 1
K = torch.randn(B, C, 256, generator=g) * 0.5
V = torch.randn(B, C, 256, generator=g) * 0.5
w = torch.randn(B, T, 2 * dy, generator=g)
DEV = "cuda:0"
Kd, Vd = K.to(DEV).requires_grad_(True), V.to(DEV).requires_grad_(True)
rows = x6.target_side(model, X.to(DEV), CH.PTensor(FN.pack_pt(Kd), C, 256), CH.PTensor(FN.pack_pt(Vd), C, 256))
(rows * w.to(DEV)).sum().backward()
Kr, Vr = K.double().requires_grad_(True), V.double().requires_grad_(True)
ref, P = tx._target_side_f64(model, X.double(), Kr, Vr)
(ref * w.double()).sum().backward()
def err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max())
print("rows", err(rows, ref), "dK", err(Kd.grad, Kr.grad), "dV", err(Vd.grad, Vr.grad))
for k, p in model.named_parameters():
    if not k.startswith("xy_encoder"):
        print(k, err(p.grad, P[k].grad))
