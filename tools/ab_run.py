#!/usr/bin/env python3
"""Development aid: time the launches of a train step under several builds of the library (tools/fastbuild.sh variants in
tools/variants/*.so, selected through NPF_HIP_LIB) -- one bench.py child per variant, the per-launch milliseconds side by side.
usage (GPU box): python3 tools/ab_run.py [bench.py arguments]"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
names = []
for so in sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "*.so"))):
    name = os.path.basename(so)[:-3]
    env = dict(os.environ, NPF_HIP_LIB=so)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "6", "--warmup", "2",
                          *sys.argv[1:]], env=env, capture_output=True, text=True)
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    if res.returncode != 0 or not line:
        print(name, "FAILED", res.stderr[-500:])
        continue
    d = json.loads(line[-1])
    names.append(name)
    rows.setdefault("step (graph replay)", {})[name] = d["ms_per_step"]
    for i, l in enumerate(d["roofline"]["launches"]):
        rows.setdefault(f"{i:2d} {l['kernel'][:18]} {l['what'][:34]}", {})[name] = l["ms"]
print(f"{'launch':58s}" + "".join(f"{n[:12]:>13s}" for n in names))
for k, v in rows.items():
    print(f"{k:58s}" + "".join(f"{v.get(n, float('nan')):13.3f}" for n in names))
