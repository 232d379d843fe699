#!/bin/bash
# Run on the GPU box from the repo root: the full bench line plus the rocprofv3 passes whose summaries go
# to profiles/<tag>_* (tools/summarize_profiles.py).  PMC passes are separate runs with --kernel-trace only.
# usage: bash tools/profile_round.sh <tag> [bench.py arguments, e.g. --dtype bf16]
#        (writes gpurun_out/prof_<tag>/ and profiles/<tag>_*; without bench arguments also the bf16 / decode lines)
set -e
TAG=${1:-r01}
shift || true
ARGS="$@"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"; cut -c1-200 $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stats.log 2>&1
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
echo "write done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/mfma --output-format csv -- python3 bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline > $OUT/mfma.log 2>&1
echo "mfma done"
python3 tools/summarize_profiles.py $OUT $TAG
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
[ -n "$ARGS" ] && exit 0
python3 bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/profiles_$TAG/${TAG}_bench_bf16.json 2>> $OUT/bench.err
python3 bench.py --workload decode --no-cpu-baseline > gpurun_out/profiles_$TAG/${TAG}_bench_decode_nocpu.json 2>> $OUT/bench.err
cut -c1-200 gpurun_out/profiles_$TAG/${TAG}_bench_bf16.json
