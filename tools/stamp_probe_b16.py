#!/usr/bin/env python3
"""Diagnostic: build the library with -DNPF_STAMPS into a private .so, run config 3's train step once and print where wave 0 of
workgroup 0 of the LAST b16 program launch spent its cycles (csrc/b16_kernel.hip BP_STAMP).  Each stamp costs ~150 cycles
(s_memtime round trip): shares are meaningful, absolute time is not; the stamped build is never the shipped library."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from npf_gwwaveform_amd import _build, _lib  # noqa: E402

so = "/tmp/libnpf_stamps_b16.so"
subprocess.run([_build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DNPF_STAMPS", *sys.argv[1:],
                "-I", os.path.join(ROOT, "include"), "-I", _build.CSRC, *_build.sources(), "-o", so], check=True)
_build.LIB_PATH = so
_lib._lib = None
import npf_gwwaveform_amd as A  # noqa: E402
from npf_gwwaveform_amd import x6  # noqa: E402
import bench  # noqa: E402

lib = _lib.load()
lib.npf_debug_stamps_b16.argtypes = [C.POINTER(C.c_ulonglong)]
NAMES = ["op loop back edge", "input side", "pack + addend loads", "counted wait + barrier", "previous slab's stores + slab pieces",
         "reads, matrix instructions, epilogues", "last slab's stores", "softmax / bits behind the op", "F -> 4 layer"]
A.set_compute_dtype("bf16")
DEV = "cuda:0"
model, _crit = bench.build_model("attncnp", 256, 4, DEV)
B, Cn, T = 1024, 200, 1024
g = torch.Generator(device=DEV).manual_seed(0)
Xc, Xt = torch.rand(B, Cn, 1, device=DEV, generator=g) * 2 - 1, torch.rand(B, T, 1, device=DEV, generator=g) * 2 - 1
Yc, Yt = torch.randn(B, Cn, 2, device=DEV, generator=g), torch.randn(B, T, 2, device=DEV, generator=g)
model.train()
orig = x6.Program.launch


def launch(self):
    orig(self)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    assert lib.npf_debug_stamps_b16(buf) == 0
    v = list(buf)
    tot = v[10]
    n_mm = sum(1 for o in self.ops if o.get("img") is not None)
    print(f"{self.tag}: wave 0 of workgroup 0: {tot} cycles, {len(self.ops)} ops, {n_mm} multiplies = {4 * n_mm} slabs")
    for n, c in zip(NAMES, v[:9]):
        print(f"   {n:45s} {c:9d} cycles  {100.0 * c / max(tot, 1):5.1f}%")


x6.Program.launch = launch
for it in range(2):
    out = model(Xc, Yc, Xt, Yt)
    A.CNPFLoss()(out, Yt).backward()
    torch.cuda.synchronize()
    print("---- step", it)
