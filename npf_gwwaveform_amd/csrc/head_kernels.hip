// Gaussian head of the decoder: split -> loc, scale = 0.01 + 0.99 softplus(raw)
// (npf/neuralproc/base.py:350-353,116), optional homoskedastic pooling of the scale over
// the targets (base.py:356-362, neuralproc/helpers.py:21-32) and, fused, the summed
// log-likelihood of the targets under Independent(Normal(loc, scale), 1)
// (npf/losses.py:18-24, npf/utils/helpers.py:125-129).  One workgroup per (z-sample, task)
// row; HBM-bound (a few floats per point), reductions by wavefront shuffles + LDS.
#include "npf_common.hpp"

namespace npf {

constexpr float kHalfLog2Pi = 0.91893853320467274178f;  // log(sqrt(2 pi))

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // red may still be read by a previous call
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float softplus_grad(float x) { return x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void gauss_head_fwd_kernel(const float* __restrict__ suff, int pts, int dy, int homosk,
                                                             const float* __restrict__ Y, int n_y_rows,
                                                             float* __restrict__ loc, float* __restrict__ scale,
                                                             float* __restrict__ sum_logp) {
  __shared__ float red[8];
  __shared__ float pooled[32];
  const size_t row = blockIdx.x;
  const float* s = suff + row * pts * (size_t)(2 * dy);
  float* lo = loc + row * pts * (size_t)dy;
  float* sc = scale + row * pts * (size_t)dy;
  const float* y = Y ? Y + (row % n_y_rows) * pts * (size_t)dy : nullptr;
  const int n = pts * dy;
  if (homosk) {
    for (int d = 0; d < dy; ++d) {
      float part = 0.f;
      for (int t = threadIdx.x; t < pts; t += blockDim.x) part += 0.01f + 0.99f * softplus_t(s[t * 2 * dy + dy + d]);
      const float tot = block_sum(part, red);
      if (threadIdx.x == 0) pooled[d] = tot / (float)pts;
    }
    __syncthreads();
  }
  float lp = 0.f;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const int t = e / dy, d = e - t * dy;
    const float mu = s[t * 2 * dy + d];
    const float sg = homosk ? pooled[d] : 0.01f + 0.99f * softplus_t(s[t * 2 * dy + dy + d]);
    lo[e] = mu;
    sc[e] = sg;
    if (y) {
      const float diff = y[e] - mu;
      lp += -(diff * diff) / (2.f * sg * sg) - logf(sg) - kHalfLog2Pi;
    }
  }
  if (sum_logp) {
    const float tot = block_sum(lp, red);
    if (threadIdx.x == 0) sum_logp[row] = tot;
  }
}

__global__ __launch_bounds__(256) void gauss_head_bwd_kernel(const float* __restrict__ suff, const float* __restrict__ loc,
                                                             const float* __restrict__ scale, int pts, int dy, int homosk,
                                                             const float* __restrict__ Y, int n_y_rows,
                                                             const float* __restrict__ d_loc, const float* __restrict__ d_scale,
                                                             const float* __restrict__ d_sum_logp, float* __restrict__ d_suff) {
  __shared__ float red[8];
  __shared__ float pooled[32];
  const size_t row = blockIdx.x;
  const size_t ebase = row * pts * (size_t)dy;
  const float* s = suff + row * pts * (size_t)(2 * dy);
  float* ds = d_suff + row * pts * (size_t)(2 * dy);
  const float* y = Y ? Y + (row % n_y_rows) * pts * (size_t)dy : nullptr;
  const float g = (d_sum_logp && y) ? d_sum_logp[row] : 0.f;
  const int n = pts * dy;
  if (homosk) {
    for (int d = 0; d < dy; ++d) {
      float part = 0.f;
      for (int t = threadIdx.x; t < pts; t += blockDim.x) {
        const int e = t * dy + d;
        const float sg = scale[ebase + e];
        float dsg = d_scale ? d_scale[ebase + e] : 0.f;
        if (y) {
          const float diff = y[e] - loc[ebase + e];
          dsg += g * (diff * diff / (sg * sg * sg) - 1.f / sg);
        }
        part += dsg;
      }
      const float tot = block_sum(part, red);
      if (threadIdx.x == 0) pooled[d] = tot / (float)pts;
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const int t = e / dy, d = e - t * dy;
    const float mu = loc[ebase + e], sg = scale[ebase + e];
    float dmu = d_loc ? d_loc[ebase + e] : 0.f;
    float dsg = d_scale ? d_scale[ebase + e] : 0.f;
    if (y) {
      const float diff = y[e] - mu;
      dmu += g * diff / (sg * sg);
      dsg += g * (diff * diff / (sg * sg * sg) - 1.f / sg);
    }
    if (homosk) dsg = pooled[d];
    ds[t * 2 * dy + d] = dmu;
    ds[t * 2 * dy + dy + d] = dsg * 0.99f * softplus_grad(s[t * 2 * dy + dy + d]);
  }
}

}  // namespace npf

extern "C" int npf_gauss_head_fwd(const float* suff, int32_t n_rows, int32_t pts, int32_t dy, int32_t homoskedastic,
                                  const float* Y, int32_t n_y_rows, float* loc, float* scale, float* sum_logp,
                                  void* stream) {
  if (!suff || !loc || !scale || n_rows <= 0 || pts <= 0 || dy <= 0 || dy > 16) return NPF_EINVAL;
  if (Y && n_y_rows <= 0) return NPF_EINVAL;
  if (sum_logp && !Y) return NPF_EINVAL;
  hipLaunchKernelGGL(npf::gauss_head_fwd_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, suff, pts, dy,
                     homoskedastic, Y, Y ? n_y_rows : 1, loc, scale, sum_logp);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_gauss_head_bwd(const float* suff, const float* loc, const float* scale, int32_t n_rows, int32_t pts,
                                  int32_t dy, int32_t homoskedastic, const float* Y, int32_t n_y_rows, const float* d_loc,
                                  const float* d_scale, const float* d_sum_logp, float* d_suff, void* stream) {
  if (!suff || !loc || !scale || !d_suff || n_rows <= 0 || pts <= 0 || dy <= 0 || dy > 16) return NPF_EINVAL;
  if (Y && n_y_rows <= 0) return NPF_EINVAL;
  hipLaunchKernelGGL(npf::gauss_head_bwd_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, suff, loc, scale, pts, dy,
                     homoskedastic, Y, Y ? n_y_rows : 1, d_loc, d_scale, d_sum_logp, d_suff);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}
