// Shared device/host helpers of the gfx950 neural-process kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "npf_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace npf {

constexpr int kTilePts = 32;  // points per PT32 tile

// Offset (floats) of the float4 holding features [4*f4, 4*f4+4) of point p inside a PT32 tile.
__host__ __device__ inline int pt_off(int f4, int p) { return (f4 * 32 + p) * 4; }

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

#define NPF_CHECK_LAUNCH()                     \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return NPF_ELAUNCH; \
  } while (0)

}  // namespace npf
