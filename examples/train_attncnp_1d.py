#!/usr/bin/env python3
"""The reference's 1-D workflow on the MI355X path, end to end: synthetic waveform-like functions,
random context / target split on the device, AttnCNP with transformer attention (what the
reference's notebooks and shipped checkpoints use), Adam, checkpoint in skorch's layout.

    python examples/train_attncnp_1d.py [--steps 200] [--dtype bf16]

Only the import line differs from a script written against the reference:
    from npf import AttnCNP, CNPFLoss                      # reference
    from npf_gwwaveform_amd import AttnCNP, CNPFLoss       # this package
"""
import argparse
import os
import sys
import time
import warnings
from functools import partial

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import npf_gwwaveform_amd as A  # noqa: E402
from npf_gwwaveform_amd.train import Trainer  # noqa: E402


def functions(n_tasks, n_points, device, seed):
    """Smooth random 1-D functions on sorted random inputs in [-1, 1] (amplitude / phase like)."""
    g = torch.Generator(device=device).manual_seed(seed)
    x, _ = torch.sort(torch.rand(n_tasks, n_points, 1, generator=g, device=device) * 2 - 1, dim=1)
    th = torch.rand(n_tasks, 1, 4, generator=g, device=device)
    y = torch.cat([(1.5 + x / 2).pow(-7 / 6) * (1 + th[..., :1]), torch.sin(6 * th[..., 1:2] * x + 6 * th[..., 2:3])], dim=-1)
    return x, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--tasks", type=int, default=64)
    ap.add_argument("--points", type=int, default=128)
    ap.add_argument("--r", type=int, default=128)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph (fixed batch shapes)")
    ap.add_argument("--out", default="/tmp/npf_example_ckpt")
    args = ap.parse_args()
    dev = "cuda:0"
    A.set_compute_dtype(args.dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = A.AttnCNP(1, 2, r_dim=args.r, attention="transformer",
                          XYEncoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=2, hidden_size=args.r), is_sum_merge=True),
                          Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=4, hidden_size=args.r), is_sum_merge=True)).to(dev)
    trainer = Trainer(model, A.CNPFLoss(), lr=1e-3, world=1, use_graph=args.graph)
    trainer.set_lr_decay(10, max(args.steps // 50, 1))
    # (a captured graph needs fixed shapes: a fixed number of context points per batch in that mode)
    n_ctx = dict(a=32, b=32) if args.graph else dict(a=0.1, b=0.5)
    split = A.CntxtTrgtGetter(contexts_getter=A.GetRandomIndcs(**n_ctx), targets_getter=A.get_all_indcs)
    t0 = time.perf_counter()
    for step in range(args.steps):
        X, Y = functions(args.tasks, args.points, dev, seed=step)
        Xc, Yc, Xt, Yt = split(X, Y)
        loss = trainer.step(dict(X_cntxt=Xc, Y_cntxt=Yc, X_trgt=Xt, Y_trgt=Yt))
        if (step + 1) % 50 == 0:
            print(f"step {step + 1:5d}  loss/task {loss.item():9.3f}  lr {trainer.end_epoch():.2e}  "
                  f"{(step + 1) * args.tasks * args.points / (time.perf_counter() - t0):,.0f} target-points/s")
    trainer.save_checkpoint(args.out, history=[{"steps": args.steps, "loss": float(loss)}])
    print("checkpoint:", sorted(os.listdir(args.out)))


if __name__ == "__main__":
    main()
