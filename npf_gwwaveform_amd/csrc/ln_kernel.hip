// y = LayerNorm(a + b) over the features of every point, forward and backward, on PT32 tensors (npf_add_layernorm_fwd / _bwd):
// TransformerAttender.forward's `layer_norm1(context + queries)` (npf/architectures/attention.py:566-575; nn.LayerNorm: biased
// variance, eps inside the root, affine gamma / beta).  HBM-bound elementwise work: one workgroup per 32-point tile, a thread owns
// one point and F / 32 of its float4 columns, the eight threads of a point meet in LDS.  The backward pass recomputes x = a + b and
// its statistics (two floats per point are saved), writes dx (the gradient of both summands) and one partial of dgamma / dbeta per
// tile, which the caller sums over the tiles.
#include "npf_common.hpp"

namespace npf {

constexpr int kLnThreads = 256;

// sums of v over the eight threads (tid % 32 equal) that share a point
__device__ __forceinline__ float ln_sum8(float v, float* red, int tid) {
  red[tid] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += red[(tid & 31) + 32 * k];
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(kLnThreads) void add_layernorm_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                       float eps, int F, int Fp, float* __restrict__ y,
                                                                       float* __restrict__ stats) {
  __shared__ float red[kLnThreads];
  const int tid = threadIdx.x, p = tid & 31, c0 = tid >> 5;   // float4 columns c0, c0 + 8, ...
  const size_t tile = (size_t)blockIdx.x * (Fp * 32);
  const int ncol = F >> 2;
  f32x4 x[8];  // (F <= 256: at most eight columns per thread)
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int c = c0 + 8 * n;
    x[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < ncol) {
      const size_t at = tile + ((size_t)c * 32 + p) * 4;
      x[n] = *(const f32x4*)(a + at) + *(const f32x4*)(b + at);
      s += x[n][0] + x[n][1] + x[n][2] + x[n][3];
    }
  }
  const float mean = ln_sum8(s, red, tid) / (float)F;
  float v = 0.f;
#pragma unroll
  for (int n = 0; n < 8; ++n)
    if (c0 + 8 * n < ncol) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v += (x[n][e] - mean) * (x[n][e] - mean);
    }
  const float rstd = 1.0f / sqrtf(ln_sum8(v, red, tid) / (float)F + eps);
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int c = c0 + 8 * n;
    if (c < ncol) {
      const f32x4 g = *(const f32x4*)(gamma + 4 * c), bt = *(const f32x4*)(beta + 4 * c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (x[n][e] - mean) * rstd * g[e] + bt[e];
      *(f32x4*)(y + tile + ((size_t)c * 32 + p) * 4) = o;
    }
  }
  if (stats != nullptr && c0 == 0) {
    stats[((size_t)blockIdx.x * 32 + p) * 2] = mean;
    stats[((size_t)blockIdx.x * 32 + p) * 2 + 1] = rstd;
  }
}

__global__ __launch_bounds__(kLnThreads) void add_layernorm_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                       const float* __restrict__ gamma, const float* __restrict__ stats,
                                                                       const float* __restrict__ dy, int F, int Fp, int pts,
                                                                       int tiles_per_task, float* __restrict__ dx,
                                                                       float* __restrict__ partials) {
  __shared__ float red[kLnThreads];
  const int tid = threadIdx.x, p = tid & 31, c0 = tid >> 5;
  const size_t tile = (size_t)blockIdx.x * (Fp * 32);
  const int ncol = F >> 2;
  const bool live = (int)(blockIdx.x % tiles_per_task) * 32 + p < pts;  // (padding points: no gradient, nothing into dgamma / dbeta)
  const float mean = stats[((size_t)blockIdx.x * 32 + p) * 2], rstd = stats[((size_t)blockIdx.x * 32 + p) * 2 + 1];
  f32x4 xh[8], g[8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int c = c0 + 8 * n;
    xh[n] = g[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c >= ncol) continue;
    const size_t at = tile + ((size_t)c * 32 + p) * 4;
    const f32x4 x = *(const f32x4*)(a + at) + *(const f32x4*)(b + at);
    f32x4 d = *(const f32x4*)(dy + at);
    if (!live) d = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 gm = *(const f32x4*)(gamma + 4 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xh[n][e] = (x[e] - mean) * rstd;
      g[n][e] = d[e];                       // dy (for dgamma / dbeta)
      const float dxh = d[e] * gm[e];
      s1 += dxh;
      s2 += dxh * xh[n][e];
    }
  }
  const float m1 = ln_sum8(s1, red, tid) / (float)F, m2 = ln_sum8(s2, red, tid) / (float)F;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int c = c0 + 8 * n;
    if (c >= ncol) continue;  // (uniform over a half wave: c0 is)
    const f32x4 gm = *(const f32x4*)(gamma + 4 * c);
    f32x4 o, dg, db;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = live ? rstd * (g[n][e] * gm[e] - m1 - xh[n][e] * m2) : 0.f;
      dg[e] = g[n][e] * xh[n][e];
      db[e] = g[n][e];
    }
    *(f32x4*)(dx + tile + ((size_t)c * 32 + p) * 4) = o;
    // dgamma / dbeta of this column over the tile's 32 points: the 32 lanes of a half wave hold the 32 points of column c
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) {
        dg[e] += __shfl_xor(dg[e], off);
        db[e] += __shfl_xor(db[e], off);
      }
    }
    if (p == 0) {
      *(f32x4*)(partials + ((size_t)blockIdx.x * 2 + 0) * F + 4 * c) = dg;
      *(f32x4*)(partials + ((size_t)blockIdx.x * 2 + 1) * F + 4 * c) = db;
    }
  }
}

}  // namespace npf

static int ln_check(const void* a, const void* b, const void* c, const void* d, int32_t n_tasks, int32_t pts, int32_t F) {
  if (!a || !b || !c || !d || n_tasks <= 0 || pts <= 0 || F <= 0 || (F & 3) || F > 256) return NPF_EINVAL;
  if ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) return NPF_EINVAL;
  return NPF_OK;
}

extern "C" int npf_add_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, float eps, int32_t n_tasks,
                                     int32_t pts_per_task, int32_t F, float* y, float* stats, void* stream) {
  const int rc = ln_check(a, b, gamma, y, n_tasks, pts_per_task, F);
  if (rc != NPF_OK || !beta || (((uintptr_t)beta) & 15)) return NPF_EINVAL;
  const int tiles = (pts_per_task + 31) / 32;
  hipLaunchKernelGGL(npf::add_layernorm_fwd_kernel, dim3(n_tasks * tiles), dim3(npf::kLnThreads), 0, (hipStream_t)stream, a, b, gamma,
                     beta, eps, F, npf::round_up(F, 32), y, stats);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_add_layernorm_bwd(const float* a, const float* b, const float* gamma, const float* stats, const float* dy,
                                     int32_t n_tasks, int32_t pts_per_task, int32_t F, float* dx, float* partials, void* stream) {
  const int rc = ln_check(a, b, gamma, dx, n_tasks, pts_per_task, F);
  if (rc != NPF_OK || !stats || !dy || !partials || ((((uintptr_t)dy) | ((uintptr_t)partials)) & 15)) return NPF_EINVAL;
  const int tiles = (pts_per_task + 31) / 32;
  hipLaunchKernelGGL(npf::add_layernorm_bwd_kernel, dim3(n_tasks * tiles), dim3(npf::kLnThreads), 0, (hipStream_t)stream, a, b, gamma,
                     stats, dy, F, npf::round_up(F, 32), pts_per_task, tiles, dx, partials);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}
