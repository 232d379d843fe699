#!/bin/bash
# Development aid: rebuild ONE translation unit of csrc/ (optionally with -D switches) and link it against cached objects of
# the others -- seconds instead of a minute; run a variant with NPF_HIP_LIB=<OUT.so> (npf_gwwaveform_amd/_lib.py).
# usage: tools/fastbuild.sh <unit, e.g. x6_kernel> OUT.so [-DFLAG ...]   (flags apply to that unit only; cache: /tmp/npfobj)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
UNIT="$1"; OUT="$2"; shift 2
mkdir -p /tmp/npfobj
CS="$ROOT/npf_gwwaveform_amd/csrc"
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC"
OBJS=""
for src in "$CS"/*.hip; do
  f="$(basename "$src" .hip)"
  [ "$f" = "$UNIT" ] && continue
  OBJS="$OBJS /tmp/npfobj/$f.o"
  if [ ! -f /tmp/npfobj/$f.o ] || [ "$src" -nt /tmp/npfobj/$f.o ] || [ "$CS/npf_common.hpp" -nt /tmp/npfobj/$f.o ] || [ "$ROOT/include/npf_hip.h" -nt /tmp/npfobj/$f.o ]; then
    $HIPCC -I "$ROOT/include" -I "$CS" -c "$src" -o /tmp/npfobj/$f.o &
  fi
done
$HIPCC "$@" -I "$ROOT/include" -I "$CS" -c "$CS/$UNIT.hip" -o /tmp/npfobj/${UNIT}_$$.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $OBJS /tmp/npfobj/${UNIT}_$$.o -o "$OUT"
rm -f /tmp/npfobj/${UNIT}_$$.o
echo "built $OUT"
