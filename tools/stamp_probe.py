#!/usr/bin/env python3
"""Diagnostic: build the chain kernel with -DNPF_STAMPS into a private .so, run an 8-layer
256->256 chain and print where wave 0 of workgroup 0 spends its cycles per slab iteration.
The stamped build is never the shipped library (shares are meaningful, absolute time is not)."""
import ctypes as C
import math
import os
import subprocess
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from npf_gwwaveform_amd import _build, _lib  # noqa: E402

so = "/tmp/libnpf_stamps.so"
cmd = [_build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DNPF_STAMPS", *sys.argv[1:],
       "-I", os.path.join(ROOT, "include"), "-I", _build.CSRC, *_build.sources(), "-o", so]
subprocess.run(cmd, check=True)
_build.LIB_PATH = so
_lib._lib = None
from npf_gwwaveform_amd import chain as CH  # noqa: E402

lib = _lib.load()
lib.npf_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
DEV = "cuda:0"
for n_tasks in (16, 256):
    pts, L = 1024, 8
    x = torch.randn(CH.pt_shape(n_tasks, pts, 256), device=DEV)
    Ws = [torch.randn(256, 256, device=DEV) / 16 for _ in range(L)]
    bs = [torch.randn(256, device=DEV) * 0.1 for _ in range(L)]
    out = CH.pt_empty(n_tasks, pts, 256, DEV)
    prog = CH.Program(n_tasks, pts, False)
    prog.load_pt(x, 256)
    for W, b in zip(Ws, bs):
        prog.linear(W, 256, 256, bias=b, relu=True)
    prog.store_pt(out, 256)
    buf0 = (C.c_ulonglong * 16)()
    for _ in range(3):
        prog._launch()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    assert lib.npf_debug_stamps(buf) == 0
    names = ["dma pieces", "addend+mfma loop", "barrier+dma wait", "epilogue", "barrier wait (group B)", "between slab loops",
             "loop top", "cursor advance"]
    for grp, v in (("wave 0", list(buf)[:8]), ("wave 4 (group B of a 128-point workgroup)", list(buf)[8:])):
        tot = sum(v)
        if not tot:
            continue
        print(f"{n_tasks * pts} points, {grp}: total {tot} cycles, per slab iteration (64 slabs):")
        for n, c in zip(names, v):
            if c:
                print(f"   {n:22s} {c / 64:9.1f} cycles/slab  {100.0 * c / tot:5.1f}%")
