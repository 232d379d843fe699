#!/usr/bin/env python3
"""Micro-benchmarks of the chain / wgrad kernels (development aid; not part of the product
contract).  Each case builds a program directly and times it with HIP events."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from npf_gwwaveform_amd import chain as CH  # noqa: E402

DEV = "cuda:0"
PEAK = 157.3


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def report(name, flops, sec):
    tf = flops / sec * 1e-12
    print(f"{name:55s} {sec*1e3:8.3f} ms  {tf:7.1f} TF/s  {100*tf/PEAK:5.1f}% of fp32 peak", flush=True)


def chain_case(name, n_tasks, pts, layers, store=False, relu=True, per_task=False):
    """layers: list of (K, N)."""
    K0 = layers[0][0]
    x = torch.randn(CH.pt_shape(n_tasks, pts, K0), device=DEV)
    Ws = [torch.randn(N, K, device=DEV) / math.sqrt(K) for K, N in layers]
    bs = [torch.randn(N, device=DEV) * 0.1 for K, N in layers]
    bufs = [CH.pt_empty(n_tasks, pts, N, DEV) for K, N in layers]
    prog = CH.Program(n_tasks, pts, per_task)
    prog.load_pt(x, K0)
    for (K, N), W, b, buf in zip(layers, Ws, bs, bufs):
        prog.linear(W, K, N, bias=b, relu=relu)
        if store:
            prog.store_pt(buf, N)
    if not store:
        prog.store_pt(bufs[-1], layers[-1][1])
    report(name, prog.flops(), timeit(prog._launch))


def chain_case_bf16(name, n_tasks, pts, layers, store=None, relu=True):
    """bf16 compute mode; store: None (last layer only), "pt32" or "pt16" (every layer, as a training forward)."""
    CH.set_compute_dtype("bf16")
    K0 = layers[0][0]
    x = torch.randn(CH.pt_shape(n_tasks, pts, K0), device=DEV)
    imgs = [CH.cast_bf16_weights(torch.randn(N, K, device=DEV) / math.sqrt(K)) for K, N in layers]
    bs = [torch.randn(N, device=DEV) * 0.1 for K, N in layers]
    mk = CH.pt16_empty if store == "pt16" else CH.pt_empty
    bufs = [mk(n_tasks, pts, N, DEV) for K, N in layers]
    prog = CH.Program(n_tasks, pts, False)
    prog.load_pt(x, K0)
    for (K, N), img, b, buf in zip(layers, imgs, bs, bufs):
        prog.linear_bf16(img, K, N, bias=b, relu=relu)
        if store:
            prog.store_pt(buf, N)
    if not store:
        prog.store_pt(bufs[-1], layers[-1][1])
    report(name, prog.flops(), timeit(prog._launch))
    CH.set_compute_dtype("fp32")


def attn_case(name, B, C, T, r):
    q = torch.randn(CH.pt_shape(B, T, r), device=DEV)
    k = torch.randn(CH.pt_shape(B, C, r), device=DEV)
    v = torch.randn(CH.pt_shape(B, C, r), device=DEV)
    o = CH.pt_empty(B, T, r, DEV)
    prog = CH.Program(B, T, True)
    prog.load_pt(q, r)
    prog.linear(k, r, C, mode=CH.L.W_PT_ROWS, w_tiles=k.shape[1])
    prog.softmax(C, 1 / math.sqrt(r))
    prog.linear(v, C, r, mode=CH.L.W_PT_COLS, w_tiles=v.shape[1])
    prog.store_pt(o, r)
    report(name, prog.flops(), timeit(prog._launch))
    vtr = torch.randn(B, r, CH.pad32(C), device=DEV)
    prog = CH.Program(B, T, True)
    prog.load_pt(q, r)
    prog.linear(k, r, C, mode=CH.L.W_PT_ROWS, w_tiles=k.shape[1])
    prog.softmax(C, 1 / math.sqrt(r))
    prog.linear(vtr, C, r, mode=CH.L.W_ROWMAJOR, ldw=vtr.shape[2], w_task_stride=r * vtr.shape[2])
    prog.store_pt(o, r)
    report(name + " [values feature-major]", prog.flops(), timeit(prog._launch))


def wgrad_case(name, n_tasks, pts, shapes, per_task=False):
    jobs = []
    for N, K in shapes:
        dz = torch.randn(CH.pt_shape(n_tasks, pts, N), device=DEV)
        a = torch.randn(CH.pt_shape(n_tasks, pts, K), device=DEV)
        if per_task:
            dW = torch.empty(CH.pt_shape(n_tasks, N, K), device=DEV)
            jobs.append(dict(dZ=dz, A=a, N=N, K=K, dW=dW, per_task=True))
        else:
            jobs.append(dict(dZ=dz, A=a, N=N, K=K, dW=torch.empty(N, K, device=DEV), db=torch.empty(N, device=DEV)))
    fl = sum(2 * N * K for N, K in shapes) * n_tasks * pts
    report(name, fl, timeit(lambda: CH._run_wgrad(jobs, n_tasks, pts, DEV)))


def wgrad_case_bf16(name, n_tasks, pts, shapes, pt16=True):
    """bf16 wgrad variant; operands PT16 (bf16 tiles) or fp32 PT32."""
    CH.set_compute_dtype("bf16")
    mk = (lambda F: torch.randn(CH.pt16_shape(n_tasks, pts, F), device=DEV).to(torch.bfloat16)) if pt16 else \
         (lambda F: torch.randn(CH.pt_shape(n_tasks, pts, F), device=DEV))
    ops = {}
    jobs = []
    for N, K in shapes:
        dz, a = ops.setdefault(("z", N), mk(N)), ops.setdefault(("a", K), mk(K))
        jobs.append(dict(dZ=dz, A=a, N=N, K=K, dW=torch.empty(N, K, device=DEV), db=torch.empty(N, device=DEV)))
    fl = sum(2 * N * K for N, K in shapes) * n_tasks * pts
    sec = timeit(lambda: CH._run_wgrad(jobs, n_tasks, pts, DEV))
    nbytes = sum((pad(N) + pad(K)) * (2 if pt16 else 4) for N, K in shapes) * n_tasks * pts
    report(name + f"  [{nbytes / sec * 1e-12:.2f} TB/s]", fl, sec)
    CH.set_compute_dtype("fp32")


def pad(v):
    return (v + 31) // 32 * 32


if __name__ == "__main__":
    B, T, C = 256, 1024, 256
    which = sys.argv[1:] or ["chain", "attn", "wgrad"]
    if "chain" in which:
        chain_case("8 x linear 256->256, relu, store last", B, T, [(256, 256)] * 8)
        chain_case("8 x linear 256->256, relu, store every layer", B, T, [(256, 256)] * 8, store=True)
        chain_case("1 x linear 256->256", B, T, [(256, 256)])
        chain_case("8 x linear 128->128", B, T, [(128, 128)] * 8)
        chain_case("8 x linear 64->64", B, T, [(64, 64)] * 8)
        chain_case("32->256 then 7 x 256->256", B, T, [(32, 256)] + [(256, 256)] * 7)
        chain_case("8 x [256->32]", B, T, [(256, 32), (32, 256)] * 4)
        chain_case("8 x linear 256->256 (ctx-sized grid)", B, C, [(256, 256)] * 8)
    if "wgbf16" in which:
        wgrad_case_bf16("bf16 wgrad 7 jobs 256x256, PT16 operands, 1M points", 1024, T, [(256, 256)] * 7)
        wgrad_case_bf16("bf16 wgrad 7 jobs 256x256, fp32 operands, 1M points", 1024, T, [(256, 256)] * 7, pt16=False)
        wgrad_case_bf16("bf16 wgrad 1 job 256x256, PT16 operands, 1M points", 1024, T, [(256, 256)])
    if "bf16q" in which:
        chain_case_bf16("bf16 8 x linear 256->256, store last", 1024, T, [(256, 256)] * 8)
        chain_case("fp32 8 x linear 256->256, relu, store last", B, T, [(256, 256)] * 8)
    if "bf16" in which:
        chain_case_bf16("bf16 8 x linear 256->256, store last", 1024, T, [(256, 256)] * 8)
        chain_case_bf16("bf16 8 x linear 256->256, store every layer (PT16)", 1024, T, [(256, 256)] * 8, store="pt16")
        chain_case_bf16("bf16 8 x linear 256->256, store every layer (PT32)", 1024, T, [(256, 256)] * 8, store="pt32")
        for bits, name in [(16, "generic loop"), (17, "generic, no slab DMA"), (24, "generic, no MFMA"), (20, "generic, no barrier")]:
            CH.DEBUG_ABLATE = bits
            chain_case_bf16(f"bf16 ablate[{name}] 8 x linear 256->256", 1024, T, [(256, 256)] * 8)
        CH.DEBUG_ABLATE = 0
    if "decode" in which:
        # BASELINE config 5 per GPU: decoder r=512, L=4, T=4096, 512 tasks (4096 / 8 GPUs)
        chain_case("decode c5: 6 x 512->512 + 512->4, B=128 T=4096", 128, 4096, [(512, 512)] * 6 + [(512, 4)])
    if "occ" in which:
        for nt in (8, 16, 32, 64, 128):
            chain_case(f"occupancy probe: {nt * 1024 // 64} WGs, 8 x linear 256->256", nt, 1024, [(256, 256)] * 8)
    if "wg" in which:
        for mode, name in [(1, "64-pt WGs"), (2, "128-pt paired WGs")]:
            CH.FORCE_WG = mode
            chain_case(f"[{name}] 8 x linear 256->256", B, T, [(256, 256)] * 8)
            chain_case(f"[{name}] 8 x linear 128->128", B, T, [(128, 128)] * 8)
            chain_case(f"[{name}] 8 x linear 256->256 (ctx-sized grid)", B, C, [(256, 256)] * 8)
            attn_case(f"[{name}] attention fwd B256 C256 T1024 r256", B, C, T, 256)
        CH.FORCE_WG = 0
    if "ablate" in which:
        for bits, name in [(0, "full"), (1, "no slab DMA"), (2, "no bias loads"), (4, "no barrier"), (8, "no MFMA"),
                           (9, "no MFMA, no DMA"), (15, "nothing"), (16, "no start skew")]:
            CH.DEBUG_ABLATE = bits
            chain_case(f"ablate[{name}] 8 x linear 256->256", B, T, [(256, 256)] * 8)
        CH.DEBUG_ABLATE = 0
    if "attn" in which:
        attn_case("attention fwd B256 C256 T1024 r256", B, C, T, 256)
        attn_case("attention fwd B256 C64 T1024 r64", B, 64, T, 64)
    if "wgrad" in which:
        wgrad_case("wgrad 1 job 256x256 (target pts)", B, T, [(256, 256)])
        wgrad_case("wgrad 7 jobs 256x256 (target pts)", B, T, [(256, 256)] * 7)
        wgrad_case("wgrad 7 jobs 256x256 (ctx pts)", B, C, [(256, 256)] * 7)
        wgrad_case("wgrad per-task 2 jobs 256x256 (attention dK,dV)", B, T, [(256, 256)] * 2, per_task=True)
        wgrad_case("wgrad 4x256 + 256x32 + 32x2", B, T, [(4, 256), (256, 32), (32, 2)])
