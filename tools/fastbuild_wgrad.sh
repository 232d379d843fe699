#!/bin/bash
# Development aid: rebuild only wgrad_kernel.hip (optionally with -D switches) and link it against cached objects of the
# other translation units -- seconds instead of a minute; run a variant with NPF_HIP_LIB=<OUT.so>.
# usage: fastbuild.sh OUT.so [-DFLAG ...]   (flags apply to wgrad_kernel.hip only; other objects cached in /tmp/npfobj)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$1"; shift
mkdir -p /tmp/npfobj
CS="$ROOT/npf_gwwaveform_amd/csrc"
for f in chain_kernel head_kernels layout_kernels mlp_x6_kernel; do
  if [ ! -f /tmp/npfobj/$f.o ] || [ "$CS/$f.hip" -nt /tmp/npfobj/$f.o ] || [ "$CS/npf_common.hpp" -nt /tmp/npfobj/$f.o ] || [ "$ROOT/include/npf_hip.h" -nt /tmp/npfobj/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I "$ROOT/include" -I "$CS" -c "$CS/$f.hip" -o /tmp/npfobj/$f.o &
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -I "$ROOT/include" -I "$CS" -c "$CS/wgrad_kernel.hip" -o /tmp/npfobj/wgrad_$$.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared /tmp/npfobj/chain_kernel.o /tmp/npfobj/head_kernels.o /tmp/npfobj/layout_kernels.o /tmp/npfobj/mlp_x6_kernel.o /tmp/npfobj/wgrad_$$.o -o "$OUT"
rm -f /tmp/npfobj/wgrad_$$.o
echo "built $OUT"
