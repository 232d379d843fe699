#!/usr/bin/env python3
"""Microbenchmark: time per 256->256 bf16 layer of the chain kernel as the SLOPE over the number of layers (the load /
store of the chain's ends drops out), for four program shapes:

    bare         LINEAR + ReLU only
    fwd_store16  + the PT16 activation store a training forward does for the wgrad launch
    fwd_train    + the ReLU mask bits as well (what a training forward emits per layer)
    bwd_train    dgrad layer: mask bits in the epilogue, dZ stored as PT16

    python tools/bf16_layer_slope.py build [name ...]   # here (hipcc cross-compiles): tools/ab/libnpf_abl_<name>.so
    NPF_ABL_SHAPE=fwd_train python tools/bf16_layer_slope.py one full 0 4 12      # on the GPU box
    python tools/bf16_layer_slope.py layers                                       # bare shape, 0..16 layers

VARIANTS may carry extra -D flags for a timing-only build of the library (round 2 used NPF_ABL_* switches that removed the
MFMAs, the LDS reads, the DMA, the barrier or the whole stage; what they showed is in DESIGN.md Appendix A -- the switches
themselves are gone from the kernel).  Round 2, 1M points: bare 0.159 -> 0.126 ms per layer, fwd_train 0.234, bwd_train 0.218.
"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "ab")
VARIANTS = {
    "full": [],
}


def so_of(name):
    return os.path.join(OUT, f"libnpf_abl_{name}.so")


def build(only=()):
    from npf_gwwaveform_amd import _build
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for name, flags in VARIANTS.items():
        if only and name not in only:
            continue
        cmd = [_build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *flags,
               "-I", os.path.join(ROOT, "include"), "-I", _build.CSRC, *_build.sources(), "-o", so_of(name)]
        procs.append((name, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
        if len(procs) % 4 == 0:
            for _, p in procs[-4:]:
                p.wait()
    for name, p in procs:
        assert p.wait() == 0, name


def run_one(name, layer_counts=(8,)):
    import torch
    from npf_gwwaveform_amd import _build, _lib
    _build.LIB_PATH = so_of(name)
    _lib._lib = None
    from npf_gwwaveform_amd import chain as CH
    CH.set_compute_dtype("bf16")
    for L in layer_counts:
        _time(name, CH, torch, L)


SHAPE = os.environ.get("NPF_ABL_SHAPE", "bare")  # bare | fwd_train | bwd_train | fwd_store16


def _time(name, CH, torch, L):
    n_tasks, pts = 1024, 1024
    dev = "cuda:0"
    x = torch.randn(CH.pt_shape(n_tasks, pts, 256), device=dev)
    imgs = [CH.cast_bf16_weights(torch.randn(256, 256, device=dev) / 16) for _ in range(L)]
    bs = [torch.randn(256, device=dev) * 0.1 for _ in range(L)]
    out = CH.pt_empty(n_tasks, pts, 256, dev)
    prog = CH.Program(n_tasks, pts, False)
    prog.load_pt(x, 256)
    keep = []
    if SHAPE == "small_k":  # pairs of a 256->4 and a 4->256 layer (the generic slab loop: x-encoder / output-layer shapes)
        down = CH.cast_bf16_weights(torch.randn(4, 256, device=dev) / 16)
        up = CH.cast_bf16_weights(torch.randn(256, 4, device=dev) / 2)
        for b in bs:
            prog.linear_bf16(down, 256, 4, bias=None, relu=False)
            prog.linear_bf16(up, 4, 256, bias=b, relu=True)
        imgs, bs = [], []
        keep += [down, up]
    for img, b in zip(imgs, bs):
        if SHAPE == "bwd_train":  # dgrad layer: mask bits in the epilogue, dZ stored for the wgrad launch
            prog.linear_bf16(img, 256, 256, bias=None, relu=False)
            m = torch.randint(-2**31, 2**31 - 1, CH.ptm_empty(n_tasks, pts, 256, dev).shape, dtype=torch.int32, device=dev)
            prog.mask_bits(m, 256)
            dz = CH.pt16_empty(n_tasks, pts, 256, dev)
            prog.store_pt(dz, 256)
            keep += [m, dz]
        else:
            prog.linear_bf16(img, 256, 256, bias=b, relu=True)
            if SHAPE == "fwd_train":  # what a training forward stores per layer: ReLU bits + the wgrad operand
                m = CH.ptm_empty(n_tasks, pts, 256, dev)
                prog.store_mask(m, 256)
                a = CH.pt16_empty(n_tasks, pts, 256, dev)
                prog.store_pt(a, 256)
                keep += [m, a]
            elif SHAPE == "fwd_store16":
                a = CH.pt16_empty(n_tasks, pts, 256, dev)
                prog.store_pt(a, 256)
                keep.append(a)
    prog.store_pt(out, 256)
    for _ in range(3):
        prog._launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        prog._launch()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:18s} {SHAPE:12s} {L:3d} layers {e0.elapsed_time(e1) / 10:7.3f} ms", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "run":
        for name in VARIANTS:
            subprocess.run([sys.executable, os.path.abspath(__file__), "one", name], check=False)
    elif sys.argv[1] == "layers":
        for name in sys.argv[2:] or ("full",):
            subprocess.run([sys.executable, os.path.abspath(__file__), "one", name, "0", "1", "2", "4", "8", "16"], check=False)
    else:
        run_one(sys.argv[2], tuple(int(a) for a in sys.argv[3:]) or (8,))
