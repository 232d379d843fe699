#!/usr/bin/env python3
"""Diagnostic: build the library with -DNPF_STAMPS into a private .so, run bf16 chains on the bf16 interpreter
(chain16_kernel.hip) and print where wave 0 of workgroup 0 spends its cycles.  The stamped build is never the
shipped library (shares are meaningful, absolute time is not)."""
import ctypes as C
import math
import os
import subprocess
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from npf_gwwaveform_amd import _build, _lib  # noqa: E402

so = "/tmp/libnpf_stamps16.so"
cmd = [_build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DNPF_STAMPS", *sys.argv[1:],
       "-I", os.path.join(ROOT, "include"), "-I", _build.CSRC, *_build.sources(), "-o", so]
subprocess.run(cmd, check=True)
_build.LIB_PATH = so
_lib._lib = None
from npf_gwwaveform_amd import chain as CH  # noqa: E402

lib = _lib.load()
lib.npf_debug_stamps16.argtypes = [C.POINTER(C.c_ulonglong)]
DEV = "cuda:0"
CH.set_compute_dtype("bf16")
NAMES = ["layer setup (pack)", "slab DMA issue", "mfma loop", "vmcnt wait", "barrier", "layer epilogue", "-", "other ops"]
for n_tasks, store in ((16, None), (1024, None), (1024, "pt16")):
    pts, L = 1024, 8
    x = torch.randn(CH.pt_shape(n_tasks, pts, 256), device=DEV)
    imgs = [CH.cast_bf16_weights(torch.randn(256, 256, device=DEV) / 16) for _ in range(L)]
    bs = [torch.randn(256, device=DEV) * 0.1 for _ in range(L)]
    bufs = [CH.pt16_empty(n_tasks, pts, 256, DEV) for _ in range(L)]
    out = CH.pt_empty(n_tasks, pts, 256, DEV)
    prog = CH.Program(n_tasks, pts, False)
    prog.load_pt(x, 256)
    for img, b, buf in zip(imgs, bs, bufs):
        prog.linear_bf16(img, 256, 256, bias=b, relu=True)
        if store:
            prog.store_pt(buf, 256)
    prog.store_pt(out, 256)
    for _ in range(3):
        prog._launch()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    assert lib.npf_debug_stamps16(buf) == 0
    v = list(buf)
    tot = sum(v)
    print(f"{n_tasks * pts} points, store={store}: wave 0 of workgroup 0: {tot} cycles for {L} layers = {L * 8} slabs")
    for n, c in zip(NAMES, v):
        if c:
            print(f"   {n:22s} {c / (L * 8):9.1f} cycles/slab  {100.0 * c / tot:5.1f}%")
