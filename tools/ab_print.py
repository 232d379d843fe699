#!/usr/bin/env python3
"""Development aid: print the per-launch table of a bench.py JSON line (file argument)."""
import json
import signal
import sys

signal.signal(signal.SIGPIPE, signal.SIG_DFL)  # (piped into head)

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], f"{d['value'] / 1e6:.1f} M", f"{d['ms_per_step']:.3f} ms")
for l in d["roofline"]["launches"]:
    print("  ", l["kernel"][:14].ljust(14), l["what"][:44].ljust(44), f"{l['ms']:.3f}", f"{l['frac']:.3f}")
