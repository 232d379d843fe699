import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
from test_hip_models import _c2_model_and_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model, batch = _c2_model_and_batch(B)
p = model(batch["X_cntxt"], batch["Y_cntxt"], batch["X_trgt"], batch["Y_trgt"])[0]
loc = p.base_dist.loc.detach()
for lo, hi in ((0, 320), (0, 224), (0, 96), (0, 160), (0, 288), (0, 352), (0,1024)):
    sub = slice(lo, hi)
    for rep in range(2):
        ps = model(batch["X_cntxt"], batch["Y_cntxt"], batch["X_trgt"][:, sub], batch["Y_trgt"][:, sub])[0]
        d = (ps.base_dist.loc.detach() - loc[:, :, sub]).abs()[0]  # [B, T', dy]
        bad = (d.amax(-1) > 1e-6 * loc.abs().max()).nonzero()
        print(f"subset {lo}:{hi} rep {rep}: max|d| {float(d.max()):.3e}, bad points {bad.shape[0]}; tasks {sorted(set(bad[:,0].tolist()))[:12]}; "
              f"targets {sorted(set(bad[:,1].tolist()))[:40]}")
