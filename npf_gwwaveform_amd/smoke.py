"""One small invocation of the hot path on cuda:0, checked against the CPU oracle
(``__graft_entry__.smoke()``).  The oracle is used here as the checker only."""
from __future__ import annotations

import os
import sys
import warnings
from functools import partial

import torch


def run() -> None:
    """Two small train steps against the CPU oracle: r = 64 (chain kernel) and the flagship width r = 256 with 130 context points
    (the fused x6 programs and the split weight-gradient kernels of BASELINE configs 2 / 4)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    _one(64, 2, 4, 64, 96)
    _one(256, 2, 2, 130, 64)


def _one(r, L, B, C, T) -> None:
    import npf_gwwaveform_amd as A
    from oracle import npf_oracle as O

    dev = "cuda:0"
    cfg = O.OracleConfig(kind="AttnCNP", x_dim=1, y_dim=2, r_dim=r)
    params = O.perturb_biases(O.init_params(cfg, seed=3, n_layers_xy=L, n_layers_dec=L), seed=4)
    g = torch.Generator().manual_seed(0)
    Xc, Xt = torch.rand(B, C, 1, generator=g) * 2 - 1, torch.rand(B, T, 1, generator=g) * 2 - 1
    Yc, Yt = torch.randn(B, C, 2, generator=g), torch.randn(B, T, 2, generator=g)

    ref_p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.forward(cfg, ref_p, Xc, Yc, Xt, Yt)
    ref_loss = O.cnpf_loss(ref, Yt)
    ref_loss.backward()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = A.AttnCNP(1, 2, r_dim=r, attention="scaledot",
                          XYEncoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, is_force_hid_smaller=True,
                                                               hidden_size=r), is_sum_merge=True),
                          Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, hidden_size=r),
                                                     is_sum_merge=True))
    model.load_state_dict(params, strict=True)
    model = model.to(dev).train()
    out = model(Xc.to(dev), Yc.to(dev), Xt.to(dev), Yt.to(dev))
    loss = A.CNPFLoss()(out, Yt.to(dev))
    loss.backward()
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a.detach().cpu() - b.detach()).abs().max() / b.detach().abs().max())

    e_loc = rel(out[0].base_dist.loc, ref["loc"])
    e_scale = rel(out[0].base_dist.scale, ref["scale"])
    e_loss = abs(loss.item() - ref_loss.item()) / abs(ref_loss.item())
    e_grad = max(rel(p.grad, ref_p[k].grad) for k, p in model.named_parameters())
    print(f"smoke r={r}: loc {e_loc:.2e} scale {e_scale:.2e} loss {e_loss:.2e} grads {e_grad:.2e} (rel, vs CPU oracle)")
    assert e_loc <= 1e-5 and e_scale <= 1e-5 and e_loss <= 1e-5 and e_grad <= 1e-4, "HIP path disagrees with the oracle"


if __name__ == "__main__":
    run()
