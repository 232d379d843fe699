#!/usr/bin/env python3
"""Benchmark of the neural-process hot path on MI355X (see DESIGN.md section "Measurement").

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5]

Default (= ``--config c2``): a step = forward + loss + backward + (N > 1: bucketed RCCL gradient
all-reduce) + Adam on one synthetic batch of the BASELINE.json config-2 workload: AttnCNP
(scaledot), r = 256, 4-layer xy-encoder / decoder, 256 context and 1024 target frequency points,
256 tasks PER GPU, fp32 (weak scaling).  ``c3`` = the same model with bf16 products, 1024 tasks per
GPU; ``c4`` = the data-parallel configuration, 1024 tasks per GPU (8192 global at N = 8), fp32 or
``--dtype bf16``; ``c5`` = decode only (r = 512 decoder, 4096 targets, 512 waveforms per GPU).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

N > 1: either launch the ranks yourself (``python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...``) or call ``python bench.py --gpus N`` bare: the process then starts N ranks
as children (before it has touched the GPU) and exits with their status.  A world size that differs
from ``--gpus`` is an error, never a silent downgrade.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time
import warnings
from functools import partial

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32 MFMA (v_mfma_f32_16x16x4_f32) = vector peak
PEAK_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense
PEAK_HBM_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FLOP_PER_PT_TRAIN = {"attncnp": 4_148_544, "attnlnp": 6_952_768}  # SURVEY.md 8d (FlopCounterMode on the reference)
FLOP_PER_PT_DECODE_R512_L4 = 3_149_824  # (L+2) 2 r^2 + 4 r dy at r = 512, L = 4, dy = 2

# --config presets: workload, dtype, tasks per GPU, target points, width, BASELINE.json config number
CONFIGS = {
    "c2": dict(workload="train", dtype="fp32", batch=256, trgt=1024, r=256, number=2),
    "c3": dict(workload="train", dtype="bf16", batch=1024, trgt=1024, r=256, number=3),
    "c4": dict(workload="train", dtype="fp32", batch=1024, trgt=1024, r=256, number=4),
    "c5": dict(workload="decode", dtype="fp32", batch=512, trgt=4096, r=512, number=5),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="BASELINE.json configuration (c2 = the headline metric and the default; c3 = bf16, 1024 tasks; "
                         "c4 = data parallel, 1024 tasks per GPU, fp32 unless --dtype bf16; c5 = decode only)")
    ap.add_argument("--workload", default=None, choices=["train", "decode"], help="(older spelling) train = c2, decode = c5")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="attncnp", choices=["attncnp", "attnlnp"])
    ap.add_argument("--batch", type=int, default=None, help="tasks per GPU (overrides the preset)")
    ap.add_argument("--ctx", type=int, default=256)
    ap.add_argument("--trgt", type=int, default=None, help="target points per task (overrides the preset)")
    ap.add_argument("--r", type=int, default=None, help="feature width (overrides the preset)")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--attention", default="scaledot", choices=["scaledot", "transformer"],
                    help="cross attention of the model (BASELINE configs: scaledot; transformer = what the reference's "
                         "shipped checkpoints use, an example workload, not a BASELINE config)")
    ap.add_argument("--dtype", default=None, choices=["fp32", "bf16"],
                    help="fp32 | bf16 (bf16 MFMA products in the MLP stacks, attention and weight gradients; fp32 accumulation)")
    ap.add_argument("--no-graph", action="store_true",
                    help="N = 1 replays the train step from a captured HIP graph by default (Trainer(use_graph=True): the same "
                         "kernels in the same order; the input range check stays in the step as a device reduction and is "
                         "read after the timed region); this flag launches every step eagerly instead.  N > 1 launches eagerly "
                         "either way unless --dp-graph is given")
    ap.add_argument("--dp-graph", action="store_true",
                    help="N > 1: replay the step from two HIP graphs around ONE all-reduce of the flat gradient instead of launching "
                         "eagerly with the bucketed all-reduce overlapped with the backward pass (the N > 1 default: with the input "
                         "range check kept on the device an eagerly launched step costs what a replayed one does -- 6.39 against "
                         "6.36 ms at N = 1 -- and hides the exchange)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args(argv)
    if args.config is None:
        if args.workload == "decode":
            args.config = "c5"
        else:
            args.config = "c3" if args.dtype == "bf16" else "c2"
    preset = CONFIGS[args.config]
    if args.workload is not None and args.workload != preset["workload"]:
        ap.error(f"--config {args.config} is a {preset['workload']} workload")
    args.workload = preset["workload"]
    args.dtype = preset["dtype"] if args.dtype is None else args.dtype
    if args.workload == "decode" and args.dtype != "fp32":
        ap.error("the decode-only workload is fp32")
    if args.config in ("c2",) and args.dtype != "fp32":
        ap.error("config c2 is fp32 (bf16 is c3)")
    args.preset = (args.batch is None and args.trgt is None and args.r is None and args.ctx == 256 and args.layers == 4
                   and args.attention == "scaledot")
    args.batch = preset["batch"] if args.batch is None else args.batch
    args.trgt = preset["trgt"] if args.trgt is None else args.trgt
    args.r = preset["r"] if args.r is None else args.r
    args.config_number = preset["number"]
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


# ---------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks as children BEFORE this process touches the GPU
# ---------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args) -> int:
    """``python bench.py --gpus N`` called bare: run ``torch.distributed.run`` with N ranks of this same
    command as a child process and return its exit status.  Nothing here initialises HIP
    (``device_count`` does not on this image), and no process that did is ever re-exec'ed."""
    import torch

    rehearsal = os.environ.get("NPF_BENCH_REHEARSAL") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not rehearsal:
        print(f"bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


# ---------------------------------------------------------------------------------------
# host description + CPU baseline (the oracle, timed on the host cores; rank 0, N = 1 only)
# ---------------------------------------------------------------------------------------
def host_cpu():
    """(model name, physical cores of the host, logical CPUs this process may run on)."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model == "unknown":
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None and cid is not None:
                phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    try:
        allowed = len(os.sched_getaffinity(0))
    except Exception:
        allowed = os.cpu_count() or 1
    # a container's CPU share is a cgroup quota, not an affinity mask: threads beyond it only time-slice
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), None))):
        try:
            quota, period = parse(open(path).read())
            if period is None:
                period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1"):
                allowed = max(1, min(allowed, int(float(quota) / float(period) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return model, (len(phys) or (os.cpu_count() or 1)), allowed


def thread_legs(cores: int):
    """Thread counts the CPU baseline is timed at: every core this process may use, 16 (the CPU share of a one-GPU
    box) and 8 -- the line reports the fastest leg as `value` and lists them all."""
    return list(dict.fromkeys(n for n in (cores, min(16, cores), min(8, cores)) if n >= 1))


def _time_steps(step, budget_s: float, max_n: int = 200):
    step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= max_n:
            return n, el


def cpu_baseline(kind: str, r: int, L: int, C: int, T: int, budget_s: float = 10.0):
    """The oracle (CPU restatement of the reference, oracle/npf_oracle.py) on the host cores: the same
    train step (forward, loss, backward, Adam lr 1e-3) at the same shapes with batch 32, once
    on every core this process may use (the box's CPU share; at most the host's physical cores) and once on
    8 threads."""
    import torch
    from oracle import npf_oracle as O
    from npf_gwwaveform_amd.train import synthetic_waveform_batch

    model, phys, allowed = host_cpu()
    cores = max(1, min(allowed, phys, int(os.environ.get("NPF_CPU_BASELINE_THREADS", "1024"))))
    B = 32
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

    legs = {}
    for n_thr in thread_legs(cores):
        torch.set_num_threads(n_thr)
        n, el = _time_steps(step, budget_s / 2 if n_thr != cores else budget_s)
        legs[n_thr] = (B * T * n / el, n, el)
    best = max(legs, key=lambda k: legs[k][0])
    v, n, el = legs[best]
    return {"value": v, "unit": "target-points/s", "cores": best, "kind": "port",
            "cpu_model": model, "host_physical_cores": phys, "cpus_allowed": allowed,
            "by_threads": {str(k): round(val[0], 1) for k, val in legs.items()},
            "sample": f"{n} train steps (fwd+loss+bwd+Adam) of the CPU oracle at the same shapes with batch {B} "
                      f"(C={C}, T={T}, r={r}, L={L}), {el:.1f} s, torch {torch.__version__} CPU, {best} threads = the fastest of "
                      f"{sorted(legs)} threads on {model} ({phys} physical cores on the host, {allowed} usable by this process)"}


def cpu_baseline_decode(r: int, L: int, T: int, budget_s: float = 10.0):
    """The oracle's decode(X_trgt_enc, R_trgt) (base.py:327-367 restated) on the host cores, batch 2."""
    import torch
    from oracle import npf_oracle as O

    model, phys, allowed = host_cpu()
    cores = max(1, min(allowed, phys, int(os.environ.get("NPF_CPU_BASELINE_THREADS", "1024"))))
    B = 2
    cfg = O.OracleConfig(kind="CNP", x_dim=1, y_dim=2, r_dim=r)
    params = O.init_params(cfg, 0, 2, L)
    g = torch.Generator().manual_seed(5)
    Xt, R = torch.randn(B, T, r, generator=g) * 0.5, torch.randn(1, B, T, r, generator=g) * 0.5
    legs = {}
    with torch.no_grad():
        for n_thr in thread_legs(cores):
            torch.set_num_threads(n_thr)
            n, el = _time_steps(lambda: O.decode(cfg, params, Xt, R), budget_s / 2 if n_thr != cores else budget_s)
            legs[n_thr] = (B * T * n / el, n, el)
    best = max(legs, key=lambda k: legs[k][0])
    v, n, el = legs[best]
    return {"value": v, "unit": "target-points/s", "cores": best, "kind": "port",
            "cpu_model": model, "host_physical_cores": phys, "cpus_allowed": allowed,
            "by_threads": {str(k): round(val[0], 1) for k, val in legs.items()},
            "sample": f"{n} decode passes of the CPU oracle with batch {B} (T={T}, r={r}, L={L}), {el:.1f} s, "
                      f"torch {torch.__version__} CPU, {best} threads = the fastest of {sorted(legs)} on {model}"}


# ---------------------------------------------------------------------------------------
# the measured workloads
# ---------------------------------------------------------------------------------------
def build_model(kind: str, r: int, L: int, device, attention: str = "scaledot"):
    import torch
    import npf_gwwaveform_amd as A

    torch.manual_seed(0)
    kw = dict(r_dim=r,
              XYEncoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, is_force_hid_smaller=True, hidden_size=r),
                                           is_sum_merge=True),
              Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, hidden_size=r), is_sum_merge=True))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if kind == "attncnp":
            m, crit = A.AttnCNP(1, 2, attention=attention, **kw), A.CNPFLoss()
        else:
            m = A.AttnLNP(1, 2, attention=attention, is_q_zCct=True, n_z_samples_train=1, n_z_samples_test=1, **kw)
            crit = A.ELBOLossLNPF()
    return m.to(device), crit


def decode_model(r: int, L: int, device):
    import torch
    import npf_gwwaveform_amd as A

    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = A.CNP(1, 2, r_dim=r, Decoder=A.merge_flat_input(partial(A.MLP, n_hidden_layers=L, hidden_size=r), is_sum_merge=True))
    return m.to(device).eval()


def timed_region(step, n_steps: int, sync):
    """EXACTLY ``n_steps`` steps between two barrier + device-synchronize brackets (wall clock), plus one
    HIP event per step boundary on the launch stream (no host sync inside the region) for the spread."""
    import torch

    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
    sync()
    t0 = time.perf_counter()
    evs[0].record()
    last = None
    for i in range(n_steps):
        last = step(i)
        evs[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(n_steps)]
    spread = {"min": min(per), "median": statistics.median(per), "max": max(per), "source": "HIP events between steps"}
    return elapsed, spread, last


def profile_launches(step, n_prof: int, rank: int, sync, CH):
    """Instrumented pass after the timed region: HIP events around every kernel launch on the launch
    stream (rank 0); every rank runs the steps because a train step contains the gradient all-reduce.
    Returns (agg, kernels, launches): per-kernel totals, their JSON form, and the launches of ONE step in
    order (kernel, what it computes, ms, algorithmic GFLOP / GB) -- the attention path is a row of its own there."""
    if rank == 0:
        CH.PROFILE = []
    for i in range(n_prof):
        step(i)
    sync()
    if rank != 0:
        return None, {}, []
    prof, CH.PROFILE = CH.PROFILE, None
    per = len(prof) // n_prof
    launches = []
    for rec in prof[-per:]:
        name, flops, e0, e1, nbytes = rec[:5]
        ms = e0.elapsed_time(e1)
        launches.append({"kernel": name, "what": rec[5] if len(rec) > 5 else "", "ms": round(ms, 4),
                         "algorithmic_gflop": round(flops * 1e-9, 3), "tflops": round(flops / ms * 1e-9, 1) if ms > 0 else 0.0,
                         "algorithmic_hbm_gb": round(nbytes * 1e-9, 4)})
        if os.environ.get("NPF_BENCH_VERBOSE"):
            print(f"  {name:22s} {ms:8.3f} ms {flops * 1e-9:9.2f} GFLOP {flops / ms * 1e-9:7.1f} TF/s "
                  f"{nbytes * 1e-9:7.3f} GB (algorithmic) {nbytes / ms * 1e-6:7.0f} GB/s  {launches[-1]['what']}", file=sys.stderr)
    agg = {}
    for rec in prof:
        name, flops, e0, e1, nbytes = rec[:5]
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += flops
        a[2] += e0.elapsed_time(e1) * 1e-3
        a[3] += nbytes
    kernels = {}
    for name, (n, fl, sec, nb) in agg.items():
        kernels[name] = {"launches_per_step": n / n_prof, "avg_launch_ms": sec / n * 1e3,
                         "algorithmic_gflop_per_launch": fl / n * 1e-9, "achieved_tflops": fl / sec * 1e-12,
                         "algorithmic_hbm_gb_per_launch": nb / n * 1e-9, "algorithmic_hbm_gbps": nb / sec * 1e-9}
    if "mlp_x6_kernel" in kernels:
        kernels["mlp_x6_kernel"]["note"] = ("npf_mlp_x6_run: the 256 -> 256 layers of the flat MLPs (decoder resizer + merge + hidden, "
                                            "XY-encoder flat module), fp32 operands as three exact bf16 terms, six bf16 MFMAs per "
                                            "product group -- fp32 results on the bf16 pipe (DESIGN.md 3.4); NPF_NO_MLP_X6=1 "
                                            "keeps them in the fp32 chains")
    if "x6_program_kernel" in kernels:
        kernels["x6_program_kernel"]["note"] = ("npf_x6_run: a whole side of the model per launch (x-encoder, scaled-dot attention, "
                                                "decoder / XY-encoder; forward or dgrad), every product an fp32 product on the bf16 "
                                                "pipe (three exact bf16 terms per operand, six v_mfma_f32_16x16x32_bf16 per product "
                                                "group, fp32 accumulation; DESIGN.md 3.1); NPF_NO_X6_FUSED=1 = the round-2 launches")
    if "b16_program_kernel" in kernels:
        kernels["b16_program_kernel"]["note"] = ("npf_b16_run: a whole side of the model per launch in the bf16 compute mode (one "
                                                 "v_mfma_f32_16x16x32_bf16 per product group, fp32 accumulation, backward-only "
                                                 "tensors as bf16 tiles; DESIGN.md 8.1); NPF_NO_B16_FUSED=1 = the bf16 chain launches")
    if "wgrad_kernel" in kernels and CH.COMPUTE_DTYPE != "bf16" and CH.WGRAD_X6:
        # (the rate can exceed the fp32 MFMA peak: these launches run on the bf16 matrix pipe)
        kernels["wgrad_kernel"]["note"] = ("wgrad_x6_kernel: fp32 operands split exactly into three bf16 terms, six "
                                           "v_mfma_f32_16x16x32_bf16 per product group, fp32 accumulation -- an fp32 result on "
                                           "the bf16 pipe (DESIGN.md 3.2); NPF_NO_WGRAD_X6=1 = the fp32-MFMA kernel")
    return agg, kernels, launches


# fp32 kernels that multiply on the bf16 matrix pipe (three exact bf16 terms per operand): priced against bf16 peak / 6
SPLIT_KERNELS = set()


def _pipe(name: str, dtype: str):
    """(peak TFLOP/s, its name) of the matrix pipe a kernel's products run on."""
    if dtype == "bf16":
        return PEAK_BF16_TFLOPS, "bf16 MFMA"
    if name in SPLIT_KERNELS:
        return PEAK_BF16_TFLOPS / 6.0, "bf16 MFMA / 6"
    return PEAK_F32_TFLOPS, "fp32 MFMA"


def roofline_of(agg, dtype: str, tag: str, preset: bool, n_steps_prof: int = 1, ms_per_step=None, launches=None):
    """The dominant kernel (most device time) against its roofline.  fp32 launches on v_mfma_f32_16x16x4_f32 are bound by the
    fp32 MFMA rate; fp32 launches that multiply three-term bf16 splits by the dense bf16 rate / 6.  bf16 launches have 1/16
    of the fp32 MFMA cycles and are priced against both rooflines; the line carries the larger fraction (the binding one).
    ``step_frac``: the time the step's algorithmic FLOPs would take with every kernel at the peak of its own pipe, over the
    measured step time."""
    name, (n, fl, sec, nb) = max(agg.items(), key=lambda kv: kv[1][2])
    tf, gbps = fl / sec * 1e-12, nb / sec * 1e-9
    peak, pipe = _pipe(name, dtype)
    if dtype == "bf16":
        f_m, f_h = tf / PEAK_BF16_TFLOPS, gbps / PEAK_HBM_GBPS
        if f_h >= f_m:
            roof = {"kernel": name, "bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": f_h}
        else:
            roof = {"kernel": name, "bound": "mfma", "achieved": tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": f_m}
        roof["frac_mfma_bf16"], roof["frac_hbm"] = f_m, f_h
        roof["achieved_tflops_algorithmic"], roof["achieved_gbps_algorithmic"] = tf, gbps
    else:
        roof = {"kernel": name, "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                "achieved_gbps_algorithmic": gbps, "frac_hbm_algorithmic": gbps / PEAK_HBM_GBPS}
        if name in SPLIT_KERNELS:
            # fp32 results from six bf16 MFMAs per product group: the pipe's dense bf16 rate / 6 is this kernel's roofline
            roof["peak_is"] = "dense bf16 MFMA rate / 6 (fp32 operands as three exact bf16 terms, six cross products; DESIGN.md 3.2)"
    # every kernel of the step against its own roofline (the block above is the one with the most device time)
    roof["by_kernel"] = {
        k: {"ms_per_step": round(v[2] / max(1, n_steps_prof) * 1e3, 3),
            "frac": round((v[1] / v[2] * 1e-12) / _pipe(k, dtype)[0], 3), "of": _pipe(k, dtype)[1],
            "frac_hbm_algorithmic": round(v[3] / v[2] * 1e-9 / PEAK_HBM_GBPS, 3)}
        for k, v in agg.items()}
    if ms_per_step:
        # seconds the step's algorithmic work takes at the peak of each kernel's pipe / measured step time
        t_min = sum(v[1] / max(1, n_steps_prof) / (_pipe(k, dtype)[0] * 1e12) for k, v in agg.items())
        roof["step_frac"] = round(t_min / (ms_per_step * 1e-3), 4)
        roof["step_frac_is"] = "sum over kernels of (algorithmic FLOPs / peak of the kernel's pipe) / measured step time"
    if launches:
        roof["launches"] = [dict(l, frac=round(l["tflops"] / _pipe(l["kernel"], dtype)[0], 3), of=_pipe(l["kernel"], dtype)[1])
                            for l in launches]
    roof["traffic"] = None
    # HBM bytes per launch of that kernel: PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate
    # rocprofv3 --pmc passes of this same command), condensed by tools/summarize_profiles.py into
    # profiles/<round>_<config>_summary.json -- only quoted when this run has the shapes those passes had
    if preset:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_summary.json")), reverse=True):
            try:
                k = json.load(open(f))["kernels"].get("npf::" + name)
            except Exception:
                continue
            if k and "hbm_bytes_per_launch" in k:
                roof["traffic"] = k["hbm_bytes_per_launch"]
                roof["traffic_unit"] = "bytes/launch"
                roof["traffic_source"] = os.path.relpath(f, ROOT)
                if "mfma_busy_frac" in k:
                    roof["mfma_busy_frac_pmc"] = k["mfma_busy_frac"]
                break
    return roof


def main_decode(args, rank, world, dev, sync):
    """BASELINE config 5 per GPU: decode(X_trgt_enc, R_trgt) only, 512-wide 4-layer decoder, 4096
    target points per waveform, 4096 / 8 = 512 waveforms per GPU, encoder outputs resident in HBM."""
    import torch
    import torch.distributed as dist
    from npf_gwwaveform_amd import chain as CH

    r, L, T, B = args.r, args.layers, args.trgt, args.batch
    model = decode_model(r, L, dev)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    Xt = torch.randn(B, T, r, device=dev, generator=g) * 0.5
    R = torch.randn(1, B, T, r, device=dev, generator=g) * 0.5

    def step(_i=0):
        with torch.no_grad():
            p = model.decode(Xt, R)
        return p.base_dist.loc

    for _ in range(args.warmup):
        step()
    elapsed, spread, loc = timed_region(step, args.steps, sync)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * B * T * args.steps / elapsed
    roofline, kernels = None, {}
    if not args.no_roofline:
        agg, kernels, launches = profile_launches(step, 2, rank, sync, CH)
        if rank == 0:
            SPLIT_KERNELS.update(("mlp_x6_kernel", "x6_program_kernel"))
            roofline = roofline_of(agg, "fp32", "c5", args.preset, 2, elapsed / args.steps * 1e3, launches)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_decode(r, L, T)
    if rank == 0:
        flop_pt = FLOP_PER_PT_DECODE_R512_L4 if (r, L) == (512, 4) else None
        print(json.dumps({
            "metric": "waveform target-points/sec (decode only)", "value": value, "unit": "target-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_spread": spread,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (fp32 operands as three exact bf16 terms, six v_mfma_f32_16x16x32_bf16 per product group, f32 accumulation "
                      "-- f32 results, DESIGN.md 3.1 / 3.2)" if "x6_program_kernel" in kernels else "f32"),
            "data": "synthetic",
            "config": {"workload": f"BASELINE config 5: decode(X_trgt_enc, R_trgt) only, {r}-wide {L}-layer decoder, {T} target "
                                   f"points per waveform, {B} waveforms per GPU, fp32, encoder outputs resident in HBM "
                                   f"(row-major [B,T,r] as the reference's decode takes them)",
                       "tasks_per_gpu": B, "global_tasks": B * world, "target_points": T, "r_dim": r,
                       "parallelism": f"replicas x{world} (no collective)", "checksum_loc": float(loc.double().sum().item()),
                       "decode_tflops_algorithmic": (value * flop_pt * 1e-12) if flop_pt else None, "kernels": kernels},
            "roofline": roofline, "cpu_baseline": cpu}))


def main_train(args, rank, world, dev, sync, rehearsal):
    import torch
    import torch.distributed as dist
    from npf_gwwaveform_amd import chain as CH
    from npf_gwwaveform_amd.train import Trainer, synthetic_waveform_batch

    B, C, T = args.batch, args.ctx, args.trgt
    if args.dtype == "bf16":
        import npf_gwwaveform_amd as A

        A.set_compute_dtype("bf16")
    model, crit = build_model(args.model, args.r, args.layers, dev, args.attention)
    n_params = sum(p.numel() for p in model.parameters())
    use_graph = not args.no_graph and (world == 1 or args.dp_graph)
    # (the input range check stays in every step as a device reduction -- replayed steps cannot stop for the host, and the eagerly
    # launched steps of the comparison below check the same way; the verdict is read at the sync point after the timed region)
    trainer = Trainer(model, crit, lr=1e-3, world=world, use_graph=use_graph, defer_input_check=True)
    batches = [synthetic_waveform_batch(B, C, T, 1234 + rank * 10**6 + i, dev) for i in range(4)]

    def step(i):
        return trainer.step(batches[i % len(batches)])

    if use_graph:
        # set-up, not warm-up: the Trainer launches its first three steps eagerly (allocator, lazy initialisations) and
        # captures the HIP graph on the fourth -- with --warmup < 4 the capture would otherwise land in the timed region
        for i in range(4):
            step(i)
        sync()
    for i in range(args.warmup):
        step(i)
    elapsed, spread, loss = timed_region(step, args.steps, sync)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * B * T * args.steps / elapsed
    loss_val = float(loss.item())
    trainer.check_inputs()  # (graph mode: the range check's verdict, deferred to this sync point)
    # beside the headline: (1) N > 1, graph mode: the all-reduce between the two replayed graphs, timed by HIP events over
    # another K steps; (2) the same K steps launched eagerly (bucketed all-reduce overlapped with the backward pass) with the
    # phases of the step timed -- how much of the exchange the backward pass hides there, and what the host costs
    def timed_with_phases(fn):
        trainer.phase_events = []
        el, _, _ = timed_region(fn, args.steps, sync)
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        ph = trainer.phase_times()
        trainer.phase_events = None
        return el / args.steps * 1e3, ph

    phases = {}
    if use_graph and world > 1:
        _, phases = timed_with_phases(step)
    ms_eager, phases_eager = timed_with_phases(lambda i: trainer.step(batches[i % len(batches)], eager=True))
    if not use_graph:
        phases = phases_eager
    trainer.check_inputs()

    roofline, kernels = None, {}
    if not args.no_roofline:
        # (HIP events per launch cannot be recorded inside a graph replay: the instrumented steps run eagerly)
        agg, kernels, launches = profile_launches(lambda i: trainer.step(batches[i % len(batches)], eager=True), 3, rank, sync, CH)
        if rank == 0:
            tag = args.config if not (args.config == "c4" and args.dtype == "bf16") else "c4bf16"
            if args.dtype != "bf16":
                SPLIT_KERNELS.update(("mlp_x6_kernel", "x6_program_kernel"))
                if CH.WGRAD_X6:
                    SPLIT_KERNELS.add("wgrad_kernel")
            roofline = roofline_of(agg, args.dtype, tag, args.preset and args.model == "attncnp", 3,
                                   elapsed / args.steps * 1e3, launches)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.model, args.r, args.layers, C, T)

    if rank == 0:
        flop_pt = FLOP_PER_PT_TRAIN[args.model] if (args.r, args.layers, C, T) == (256, 4, 256, 1024) else None
        model_name = "AttnCNP" if args.model == "attncnp" else "AttnLNP(is_q_zCct, n_z=1)"
        line = {
            "metric": "waveform target-points/sec (train step)",
            "value": value,
            "unit": "target-points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_spread": spread,
            "allreduce_ms_exposed": phases.get("allreduce_ms_exposed") if world > 1 else 0.0,
            "backward_ms": phases.get("backward_ms"),
            "eager": {"ms_per_step": ms_eager, "backward_ms": phases_eager.get("backward_ms"),
                      "allreduce_ms_exposed": phases_eager.get("allreduce_ms_exposed") if world > 1 else 0.0,
                      "note": "the same steps launched eagerly (Trainer(use_graph=False, defer_input_check=True)): bucketed "
                              "all-reduce overlapped with the backward pass -- what N > 1 runs unless --dp-graph; the N = 1 "
                              "headline replays the step from a HIP graph"},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32" if os.environ.get("NPF_NO_WGRAD_X6", "0") == "1" and os.environ.get("NPF_NO_MLP_X6", "0") == "1" else
                      "f32 (every contraction of the step: fp32 operands as three exact bf16 terms, six v_mfma_f32_16x16x32_bf16 per product group, f32 accumulation -- f32 results, DESIGN.md 3.1 / 3.2; the 1-4-wide first / last layers: f32 FMAs)"
                      if os.environ.get("NPF_NO_X6_FUSED", "0") != "1" and args.attention == "scaledot" and args.model == "attncnp" and args.r == 256 else
                      "f32 (attention / first / last layers: v_mfma_f32_16x16x4_f32; 256-wide MLP layers and weight gradients: fp32 operands as three exact bf16 terms, six v_mfma_f32_16x16x32_bf16 per product group, f32 accumulation)")
            if args.dtype == "fp32" else "bf16 (products in MLP stacks, attention and weight gradients; f32 accumulation, epilogues, outputs, optimizer)",
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU over gloo -- not a measurement)",
            "config": {
                "workload": f"{'BASELINE config ' + str(args.config_number) if args.attention == 'scaledot' else 'example (not a BASELINE config)'}: {model_name} {args.attention}, r={args.r}, {args.layers}-layer "
                            f"xy-encoder/decoder, {C} context / {T} target points, {B} tasks per GPU ({B * world} global), "
                            f"{args.dtype} train step (fwd+loss+bwd+allreduce+Adam)",
                "name": args.config, "tasks_per_gpu": B, "global_tasks": B * world, "context_points": C, "target_points": T,
                "r_dim": args.r, "n_params": n_params, "parallelism": f"dp{world}", "final_loss": loss_val,
                "hip_graph": bool(use_graph and trainer._graph is not None),
                "train_step_tflops_algorithmic": (value * flop_pt * 1e-12) if flop_pt else None,
                "kernels": kernels,
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) joined (WORLD_SIZE={world_env})")
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
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        if args.workload == "decode":
            main_decode(args, rank, world, dev, sync)
        else:
            main_train(args, rank, world, dev, sync, rehearsal)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
