#!/bin/bash
# Development aid (GPU box): where the waves of each kernel spend their cycles -- parked (SQ_WAIT_ANY), issue-stalled
# (SQ_WAIT_INST_ANY), issuing (SQ_ACTIVE_INST_ANY), by instruction class.  usage: bash tools/pmc_wait.sh <outdir> [bench args]
set -e
OUT=${1:-gpurun_out/pmc_wait}; shift || true
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; export TMPDIR=/tmp; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $OUT/a --output-format csv -- python3 bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1 "$@" > $OUT/a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/b --output-format csv -- python3 bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1 "$@" > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "npf::" in n:
            agg[n + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print(k)
    print("   " + "  ".join(f"{n[3:]}={v / wc:.3f}" for n, v in m.items() if n.startswith("SQ_") and n != "SQ_WAVE_CYCLES" and not n.startswith("SQ_INSTS")))
    print("   " + "  ".join(f"{n[3:]}={v:.3g}" for n, v in m.items() if n.startswith("SQ_INSTS") or n == "SQ_WAVE_CYCLES"))
PY
