#!/bin/bash
# VGPR / SGPR / spill / LDS figures of every kernel instance (run after touching a .hip file: a change
# that makes hipcc spill does not fail any test, it only shows up as a slowdown).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP="$(mktemp -d)"
for f in "$ROOT"/npf_gwwaveform_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $NPF_EXTRA_FLAGS -I "$ROOT/include" -I "$ROOT/npf_gwwaveform_amd/csrc" \
    --cuda-device-only -S "$f" -o "$TMP/$(basename "$f").s" 2>/dev/null
  grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size)|^\s+\.name:" "$TMP/$(basename "$f").s" \
    | paste - - - - - | sed 's/  */ /g'
done
rm -rf "$TMP"
