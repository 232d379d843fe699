#!/bin/bash
# Run on the GPU box from the repo root: for one BASELINE config the bench line plus the rocprofv3 passes whose
# summaries go to profiles/<tag>_<config>_* (tools/summarize_profiles.py).  PMC passes are separate runs with
# --kernel-trace only; the profiled program is `python3 bench.py ...` itself, directly after `--`.
# usage: bash tools/profile_round.sh <tag> <config> [extra bench.py arguments]
#        e.g. bash tools/profile_round.sh r02 c2 ; bash tools/profile_round.sh r02 c3 ; bash tools/profile_round.sh r02 c5
set -e
TAG=${1:-r02}
CFG=${2:-c2}
shift 2 || true
ARGS="--config $CFG $@"
NAME=${TAG}_${CFG}
OUT=gpurun_out/prof_$NAME
case " $ARGS " in *" --gpus "*) echo "profile_round.sh profiles the single-GPU run only (bench.py --gpus N spawns ranks: not under rocprofv3)"; exit 2;; esac
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
cd "$ROOT"
mkdir -p $OUT
python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"; cut -c1-200 $OUT/bench.json
P="--no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py $ARGS $P --steps 10 --warmup 3 > $OUT/stats.log 2>&1
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 bench.py $ARGS $P --steps 3 --warmup 1 > $OUT/fetch.log 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 bench.py $ARGS $P --steps 3 --warmup 1 > $OUT/write.log 2>&1
echo "write done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/mfma --output-format csv -- python3 bench.py $ARGS $P --steps 3 --warmup 1 > $OUT/mfma.log 2>&1
echo "mfma done"
python3 tools/summarize_profiles.py $OUT $NAME
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${NAME}_* gpurun_out/profiles_$TAG/
