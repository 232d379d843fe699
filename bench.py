#!/usr/bin/env python3
"""Benchmark of the neural-process train step on MI355X (see DESIGN.md section "Measurement").

    python bench.py --gpus N --steps K --warmup W

A step = forward + loss + backward + (N > 1: bucketed RCCL gradient all-reduce) + Adam on one
synthetic batch of the BASELINE.json config-2 workload: AttnCNP (scaledot), r = 256, 4-layer
xy-encoder / decoder, 256 context and 1024 target frequency points, 256 tasks PER GPU, fp32
(weak scaling).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE
JSON line.  For N > 1 launch with ``python -m torch.distributed.run --nproc-per-node N``.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings
from functools import partial

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA (v_mfma_f32_16x16x4_f32) = vector peak
PEAK_HBM_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FLOP_PER_PT_TRAIN = {"attncnp": 4_148_544, "attnlnp": 6_952_768}  # SURVEY.md 8d (FlopCounterMode on the reference)


def build_model(kind: str, r: int, L: int, device):
    import npf_gwwaveform_amd as A

    torch.manual_seed(0)
    kw = dict(r_dim=r,
              XYEncoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, is_force_hid_smaller=True, hidden_size=r),
                                           is_sum_merge=True),
              Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, hidden_size=r), is_sum_merge=True))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if kind == "attncnp":
            m, crit = A.AttnCNP(1, 2, attention="scaledot", **kw), A.CNPFLoss()
        else:
            m = A.AttnLNP(1, 2, attention="scaledot", is_q_zCct=True, n_z_samples_train=1, n_z_samples_test=1, **kw)
            crit = A.ELBOLossLNPF()
    return m.to(device), crit


def cpu_baseline(kind: str, r: int, L: int, C: int, T: int, budget_s: float = 12.0):
    """The oracle (CPU restatement of the reference, oracle/npf_oracle.py) on the host cores:
    the same train step (forward, loss, backward, Adam lr 1e-3) at the same shapes, on a
    bounded sample (small batch, a few steps)."""
    from oracle import npf_oracle as O
    from npf_gwwaveform_amd.train import synthetic_waveform_batch

    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a 1-GPU box's CPU share is 16 cores (more threads than that oversubscribe the host)
    cores = max(1, min(avail, int(os.environ.get("NPF_CPU_BASELINE_THREADS", "16"))))
    torch.set_num_threads(cores)
    B = 8
    cfg = O.OracleConfig(kind="AttnCNP" if kind == "attncnp" else "AttnLNP", x_dim=1, y_dim=2, r_dim=r,
                         is_q_zCct=(kind != "attncnp"))
    params = {k: v.clone().requires_grad_(True) for k, v in O.init_params(cfg, 0, L, L).items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)
    batch = synthetic_waveform_batch(B, C, T, 99, "cpu")
    eps = torch.randn(1, B, 1, r) if kind != "attncnp" else None
    loss_fn = O.cnpf_loss if kind == "attncnp" else O.elbo_loss

    def step():
        opt.zero_grad(set_to_none=True)
        out = O.forward(cfg, params, batch["X_cntxt"], batch["Y_cntxt"], batch["X_trgt"], batch["Y_trgt"], eps=eps)
        loss_fn(out, batch["Y_trgt"]).backward()
        opt.step()

    step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    return {"value": B * T * n / el, "unit": "target-points/s", "cores": cores, "kind": "port",
            "sample": f"{n} train steps (fwd+loss+bwd+Adam) of the CPU oracle at the same shapes with batch {B} "
                      f"(C={C}, T={T}, r={r}, L={L}), {el:.1f} s, torch {torch.__version__} CPU, {cores} threads"}


FLOP_PER_PT_DECODE_R512_L4 = 3_149_824  # (L+2) 2 r^2 + 4 r dy at r = 512, L = 4, dy = 2


def decode_model(r: int, L: int, device):
    import npf_gwwaveform_amd as A

    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = A.CNP(1, 2, r_dim=r, Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, hidden_size=r), is_sum_merge=True))
    return m.to(device).eval()


def cpu_baseline_decode(r: int, L: int, T: int, budget_s: float = 12.0):
    """The oracle's decode(X_trgt_enc, R_trgt) (base.py:327-367 restated) on the host cores, batch 2."""
    from oracle import npf_oracle as O

    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = max(1, min(avail, int(os.environ.get("NPF_CPU_BASELINE_THREADS", "16"))))
    torch.set_num_threads(cores)
    B = 2
    cfg = O.OracleConfig(kind="CNP", x_dim=1, y_dim=2, r_dim=r)
    params = O.init_params(cfg, 0, 2, L)
    g = torch.Generator().manual_seed(5)
    Xt, R = torch.randn(B, T, r, generator=g) * 0.5, torch.randn(1, B, T, r, generator=g) * 0.5
    with torch.no_grad():
        O.decode(cfg, params, Xt, R)
        n, t0 = 0, time.perf_counter()
        while True:
            O.decode(cfg, params, Xt, R)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 200:
                break
    return {"value": B * T * n / el, "unit": "target-points/s", "cores": cores, "kind": "port",
            "sample": f"{n} decode passes of the CPU oracle with batch {B} (T={T}, r={r}, L={L}), {el:.1f} s, "
                      f"torch {torch.__version__} CPU, {cores} threads"}


def main_decode(args, rank, world, dev):
    """BASELINE config 5 per GPU: decode(X_trgt_enc, R_trgt) only, 512-wide 4-layer decoder, 4096
    target points per waveform, 4096 / 8 = 512 waveforms per GPU, encoder outputs resident in HBM."""
    from npf_gwwaveform_amd import chain as CH

    r, L, T = args.r, args.layers, args.trgt
    B = args.batch
    model = decode_model(r, L, dev)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    Xt = torch.randn(B, T, r, device=dev, generator=g) * 0.5
    R = torch.randn(1, B, T, r, device=dev, generator=g) * 0.5

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        with torch.no_grad():
            p = model.decode(Xt, R)
        return p.base_dist.loc

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loc = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * B * T * args.steps / elapsed
    roofline, kernels = None, {}
    if rank == 0 and not args.no_roofline:
        CH.PROFILE = []
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        agg = {}
        for name, flops, e0, e1, nbytes in CH.PROFILE:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += e0.elapsed_time(e1) * 1e-3
            a[3] += nbytes
        CH.PROFILE = None
        for name, (n, fl, sec, nb) in agg.items():
            kernels[name] = {"launches_per_step": n / 2, "avg_launch_ms": sec / n * 1e3,
                             "algorithmic_gflop_per_launch": fl / n * 1e-9, "achieved_tflops": fl / sec * 1e-12,
                             "algorithmic_hbm_gb_per_launch": nb / n * 1e-9}
        name, (n, fl, sec, nb) = max(agg.items(), key=lambda kv: kv[1][2])
        ach = fl / sec * 1e-12
        roofline = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_F32_TFLOPS, "traffic": None}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_decode(r, L, T)
    if rank == 0:
        flop_pt = FLOP_PER_PT_DECODE_R512_L4 if (r, L) == (512, 4) else None
        print(json.dumps({
            "metric": "waveform target-points/sec (decode only)", "value": value, "unit": "target-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config 5: decode(X_trgt_enc, R_trgt) only, {r}-wide {L}-layer decoder, {T} target "
                                   f"points per waveform, {B} waveforms per GPU, fp32, encoder outputs resident in HBM "
                                   f"(row-major [B,T,r] as the reference's decode takes them)",
                       "tasks_per_gpu": B, "global_tasks": B * world, "target_points": T, "r_dim": r,
                       "parallelism": f"replicas x{world} (no collective)", "checksum_loc": float(loc.double().sum().item()),
                       "decode_tflops_algorithmic": (value * flop_pt * 1e-12) if flop_pt else None, "kernels": kernels},
            "roofline": roofline, "cpu_baseline": cpu}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="train", choices=["train", "decode"],
                    help="train = BASELINE config 2 (the headline metric); decode = config 5 (decode-only, r=512, T=4096)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="attncnp", choices=["attncnp", "attnlnp"])
    ap.add_argument("--batch", type=int, default=None, help="tasks per GPU (train: 256, decode: 512)")
    ap.add_argument("--ctx", type=int, default=256)
    ap.add_argument("--trgt", type=int, default=None, help="target points per task (train: 1024, decode: 4096)")
    ap.add_argument("--r", type=int, default=None, help="feature width (train: 256, decode: 512)")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="fp32 = BASELINE config 2 (headline); bf16 = config 3's compute mode (bf16 MFMA in the MLP stacks, "
                         "fp32 attention / accumulation / weight gradients), 1024 tasks per GPU by default")
    ap.add_argument("--graph", action="store_true",
                    help="replay the train step from a captured HIP graph (Trainer(use_graph=True); single rank only; "
                         "skips the per-step host-side input range check) -- not the default measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    dflt = {"train": (1024 if args.dtype == "bf16" else 256, 1024, 256), "decode": (512, 4096, 512)}[args.workload]
    args.batch = dflt[0] if args.batch is None else args.batch
    args.trgt = dflt[1] if args.trgt is None else args.trgt
    args.r = dflt[2] if args.r is None else args.r

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # rehearsal on a one-GPU box (not a measurement): NPF_BENCH_REHEARSAL=1 puts every rank on
    # device 0 and exchanges gradients over gloo, to exercise the N > 1 control flow end to end
    rehearsal = os.environ.get("NPF_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.workload == "decode":
        main_decode(args, rank, world, dev)
        if world > 1:
            dist.destroy_process_group()
        return

    from npf_gwwaveform_amd import chain as CH
    from npf_gwwaveform_amd.train import Trainer, synthetic_waveform_batch

    B, C, T = args.batch, args.ctx, args.trgt
    if args.dtype == "bf16":
        import npf_gwwaveform_amd as A

        A.set_compute_dtype("bf16")
    model, crit = build_model(args.model, args.r, args.layers, dev)
    n_params = sum(p.numel() for p in model.parameters())
    trainer = Trainer(model, crit, lr=1e-3, world=world, use_graph=args.graph)
    batches = [synthetic_waveform_batch(B, C, T, 1234 + rank * 10**6 + i, dev) for i in range(4)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        trainer.step(batches[i % len(batches)])
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = trainer.step(batches[i % len(batches)])
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * B * T * args.steps / elapsed
    loss_val = float(loss.item())

    roofline = None
    kernels = {}
    n_prof = 3
    if args.graph:
        args.no_roofline = True  # (HIP events per launch cannot be recorded inside a graph replay)
    if not args.no_roofline:
        # instrumented pass: HIP events around every kernel launch on the launch stream (rank 0);
        # every rank runs these steps because a step contains the gradient all-reduce
        if rank == 0:
            CH.PROFILE = []
        for i in range(n_prof):
            trainer.step(batches[i % len(batches)])
        sync()
    if rank == 0 and not args.no_roofline:
        agg = {}
        if os.environ.get("NPF_BENCH_VERBOSE"):
            per = len(CH.PROFILE) // n_prof
            for name, flops, e0, e1, nbytes in CH.PROFILE[-per:]:
                ms = e0.elapsed_time(e1)
                print(f"  {name:14s} {ms:8.3f} ms {flops * 1e-9:9.2f} GFLOP {flops / ms * 1e-9:7.1f} TF/s "
                      f"{nbytes * 1e-9:7.3f} GB (algorithmic) {nbytes / ms * 1e-6:7.0f} GB/s", file=sys.stderr)
        for name, flops, e0, e1, nbytes in CH.PROFILE:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += e0.elapsed_time(e1) * 1e-3
            a[3] += nbytes
        CH.PROFILE = None
        for name, (n, fl, sec, nb) in agg.items():
            kernels[name] = {"launches_per_step": n / n_prof, "avg_launch_ms": sec / n * 1e3,
                             "algorithmic_gflop_per_launch": fl / n * 1e-9, "achieved_tflops": fl / sec * 1e-12,
                             "algorithmic_hbm_gb_per_launch": nb / n * 1e-9}
        dom = max(agg.items(), key=lambda kv: kv[1][2])
        name, (n, fl, sec, nb) = dom
        ach = fl / sec * 1e-12
        roofline = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_F32_TFLOPS, "traffic": None}
        if args.dtype == "bf16" and name == "chain_kernel":
            # bf16 MLP layers have 1/16 of the MFMA cycles of their fp32 form: these launches are priced
            # against HBM by their algorithmic bytes (PT16 tensors counted at 2 bytes per value)
            gbps = nb / sec * 1e-9
            roofline = {"kernel": name, "bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": gbps / PEAK_HBM_GBPS, "traffic": None, "achieved_tflops_algorithmic": ach}
        # HBM bytes per launch of that kernel: PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
        # separate rocprofv3 --pmc passes of this same command), condensed by
        # tools/summarize_profiles.py into profiles/<round>_summary.json
        import glob
        profiled = {"fp32": (256, 256, 1024, 256), "bf16": (1024, 256, 1024, 256)}[args.dtype]  # the shapes the PMC passes ran
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")), reverse=True):
            if ("bf16" in os.path.basename(f)) != (args.dtype == "bf16"):
                continue
            try:
                k = json.load(open(f))["kernels"].get("npf::" + name)
                if k and "hbm_bytes_per_launch" in k and (B, C, T, args.r) == profiled and args.workload == "train":
                    roofline["traffic"] = k["hbm_bytes_per_launch"]
                    roofline["traffic_unit"] = "bytes/launch"
                    roofline["traffic_source"] = os.path.relpath(f, ROOT)
                    break
            except Exception:
                pass

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.model, args.r, args.layers, C, T)

    if rank == 0:
        flop_pt = FLOP_PER_PT_TRAIN[args.model] if (args.r, args.layers, C, T) == (256, 4, 256, 1024) else None
        line = {
            "metric": "waveform target-points/sec (train step)",
            "value": value,
            "unit": "target-points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.dtype == "fp32" else "bf16 (products in MLP stacks, attention and weight gradients; f32 accumulation, epilogues, outputs, optimizer)",
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU over gloo -- not a measurement)",
            "config": {
                "workload": f"BASELINE config {2 if args.dtype == 'fp32' else 3}: {'AttnCNP' if args.model == 'attncnp' else 'AttnLNP(is_q_zCct, n_z=1)'} "
                            f"scaledot, r={args.r}, {args.layers}-layer xy-encoder/decoder, {C} context / {T} target "
                            f"points, {B} tasks per GPU, {args.dtype} train step (fwd+loss+bwd+allreduce+Adam)",
                "tasks_per_gpu": B, "global_tasks": B * world, "context_points": C, "target_points": T,
                "r_dim": args.r, "n_params": n_params, "parallelism": f"dp{world}", "final_loss": loss_val,
                "hip_graph": bool(args.graph),
                "train_step_tflops_algorithmic": (value * flop_pt * 1e-12) if flop_pt else None,
                "kernels": kernels,
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
