#!/usr/bin/env python3
"""Diagnostic: where does a bf16 ring layer (fast_layer_ring) spend its time?  Timing-only builds of the library with parts
of the stage removed by -DNPF_ABL_* (results are garbage, only the clock counts), each run on the bare 8-layer 256->256
bf16 chain over 1M points.

    python tools/ring_ablate.py build     # here (hipcc cross-compiles): tools/ab/libnpf_abl_<name>.so
    python tools/ring_ablate.py run       # on the GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "ab")
VARIANTS = {
    "full": [],
    "no_mfma": ["-DNPF_ABL_NO_MFMA"],
    "no_lds": ["-DNPF_ABL_NO_LDS"],
    "no_mfma_no_lds": ["-DNPF_ABL_NO_MFMA", "-DNPF_ABL_NO_LDS"],
    "no_dma": ["-DNPF_ABL_NO_DMA"],
    "no_bar": ["-DNPF_ABL_NO_BAR"],
    "no_epi": ["-DNPF_ABL_NO_EPI"],
    "boundary_only": ["-DNPF_ABL_BOUNDARY_ONLY"],
    "slim": ["-DNPF_SLIM"],
    "slim_boundary_only": ["-DNPF_SLIM", "-DNPF_ABL_BOUNDARY_ONLY"],
    "skeleton": ["-DNPF_ABL_NO_MFMA", "-DNPF_ABL_NO_LDS", "-DNPF_ABL_NO_DMA", "-DNPF_ABL_NO_EPI"],
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


def _time(name, CH, torch, L):
    n_tasks, pts = 1024, 1024
    x = torch.randn(CH.pt_shape(n_tasks, pts, 256), device="cuda:0")
    imgs = [CH.cast_bf16_weights(torch.randn(256, 256, device="cuda:0") / 16) for _ in range(L)]
    bs = [torch.randn(256, device="cuda:0") * 0.1 for _ in range(L)]
    out = CH.pt_empty(n_tasks, pts, 256, "cuda:0")
    prog = CH.Program(n_tasks, pts, False)
    prog.load_pt(x, 256)
    for img, b in zip(imgs, bs):
        prog.linear_bf16(img, 256, 256, bias=b, relu=True)
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
    print(f"{name:18s} {L:3d} layers {e0.elapsed_time(e1) / 10:7.3f} ms", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "run":
        for name in VARIANTS:
            subprocess.run([sys.executable, os.path.abspath(__file__), "one", name], check=False)
    elif sys.argv[1] == "layers":
        for name in sys.argv[2:] or ("full", "no_mfma_no_lds", "skeleton", "boundary_only"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "one", name, "0", "1", "2", "4", "8", "16"], check=False)
    else:
        run_one(sys.argv[2], tuple(int(a) for a in sys.argv[3:]) or (8,))
