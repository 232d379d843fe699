#!/usr/bin/env python3
"""Diagnostic: build the library with -DNPF_STAMPS into a private .so, run a bare 8-layer 256->256 bf16 chain and print
where wave 0 of workgroup 0 spends its cycles in the ring stages (fast_layer_ring).  Each stamp costs ~150 cycles
(s_memtime round trip): shares are meaningful, absolute time is not; the stamped build is never the shipped library."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from npf_gwwaveform_amd import _build, _lib  # noqa: E402

so = os.environ.get("NPF_STAMP_SO", "/tmp/libnpf_stamps_bf16.so")  # (prebuilt: tools/ab/..., compiled with -DNPF_STAMPS)
cmd = [_build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DNPF_STAMPS", *sys.argv[1:],
       "-I", os.path.join(ROOT, "include"), "-I", _build.CSRC, *_build.sources(), "-o", so]
if "NPF_STAMP_SO" not in os.environ:
    subprocess.run(cmd, check=True)
_build.LIB_PATH = so
_lib._lib = None
from npf_gwwaveform_amd import chain as CH  # noqa: E402

lib = _lib.load()
lib.npf_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
DEV = "cuda:0"
CH.set_compute_dtype("bf16")
NAMES = ["counted DMA wait", "the two halves of the stages (MFMA, DMA issue, epilogue)", "barrier", "loop back edge + opcode fetch",
         "LINEAR prologue (descriptor fields, addresses)", "pack + mask words", "last epilogue", "peek at the next LINEAR"]
for n_tasks in (16, 1024):
    pts, L = 1024, 8
    x = torch.randn(CH.pt_shape(n_tasks, pts, 256), device=DEV)
    imgs = [CH.cast_bf16_weights(torch.randn(256, 256, device=DEV) / 16) for _ in range(L)]
    bs = [torch.randn(256, device=DEV) * 0.1 for _ in range(L)]
    out = CH.pt_empty(n_tasks, pts, 256, DEV)
    prog = CH.Program(n_tasks, pts, False)
    prog.load_pt(x, 256)
    for img, b in zip(imgs, bs):
        prog.linear_bf16(img, 256, 256, bias=b, relu=True)
    prog.store_pt(out, 256)
    for _ in range(3):
        prog._launch()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    assert lib.npf_debug_stamps(buf) == 0
    v = list(buf)[:8]
    tot = sum(v)
    print(f"{n_tasks * pts} points: wave 0 of workgroup 0: {tot} cycles for {L} layers = {L * 8} slabs")
    for n, c in zip(NAMES, v):
        if c:
            print(f"   {n:50s} {c / (L * 8):9.1f} cycles/slab  {100.0 * c / tot:5.1f}%")
