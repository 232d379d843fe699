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
    if (loc) {  // (NULL: loss-only launch, nothing of size [n_z, B, T, dy] is written)
      lo[e] = mu;
      sc[e] = sg;
    }
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
  // loc / scale NULL (the forward pass was a loss-only launch): recomputed from the raw decoder output
  __shared__ float pooled_sg[32];
  if (!scale && homosk) {
    for (int d = 0; d < dy; ++d) {
      float part = 0.f;
      for (int t = threadIdx.x; t < pts; t += blockDim.x) part += 0.01f + 0.99f * softplus_t(s[t * 2 * dy + dy + d]);
      const float tot = block_sum(part, red);
      if (threadIdx.x == 0) pooled_sg[d] = tot / (float)pts;
    }
    __syncthreads();
  }
  auto mu_of = [&](int e, int t, int d) { return loc ? loc[ebase + e] : s[t * 2 * dy + d]; };
  auto sg_of = [&](int e, int t, int d) {
    return scale ? scale[ebase + e] : (homosk ? pooled_sg[d] : 0.01f + 0.99f * softplus_t(s[t * 2 * dy + dy + d]));
  };
  if (homosk) {
    for (int d = 0; d < dy; ++d) {
      float part = 0.f;
      for (int t = threadIdx.x; t < pts; t += blockDim.x) {
        const int e = t * dy + d;
        const float sg = sg_of(e, t, d);
        float dsg = d_scale ? d_scale[ebase + e] : 0.f;
        if (y) {
          const float diff = y[e] - mu_of(e, t, d);
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
    const float mu = mu_of(e, t, d), sg = sg_of(e, t, d);
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

// ---- Monte-Carlo objectives over the latent samples ----------------------------------------------------------------
// lw[k][b]: log weight of latent sample k for task b (sum_t log p(y_t | z_k) [+ log q(z_k|C) - log q(z_k|C,T)]).
//   mode 0: mean_k lw                                    (first term of the ELBO,  npf/losses.py:126-150)
//   mode 1: logsumexp_k lw - log n_z                     (NPML / IWAE bound,       npf/losses.py:153-203)
//   mode 2: SUMO: c_k = logsumexp_{j <= k} lw_j - log(k + 1) (0-based k), estimate = c_{m-1} +
//           sum_{k >= m} inv_w[k] (c_k - c_{k-1})        (npf/losses.py:207-276; inv_w[k] = P(K >= k + 1 - ...) from the host)
// One thread per task (n_z is tens to hundreds, B thousands): sequential running logsumexp, no [n_z, B] temporaries.
__global__ void mc_objective_fwd_kernel(const float* __restrict__ lw, int n_z, int B, int mode, const float* __restrict__ inv_w,
                                        int m, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (mode == 0) {
    float sum = 0.f;
    for (int k = 0; k < n_z; ++k) sum += lw[(size_t)k * B + b];
    out[b] = sum / (float)n_z;
    return;
  }
  float mx = -INFINITY, acc = 0.f, est = 0.f, c_prev = 0.f;  // running logsumexp = mx + log(acc)
  for (int k = 0; k < n_z; ++k) {
    const float v = lw[(size_t)k * B + b];
    if (v > mx) {
      acc = acc * expf(mx - v) + 1.f;
      mx = v;
    } else {
      acc += expf(v - mx);
    }
    if (mode == 2) {
      const float c = mx + logf(acc) - logf((float)(k + 1));
      if (k == m - 1) est = c;
      else if (k >= m) est += inv_w[k] * (c - c_prev);
      c_prev = c;
    }
  }
  out[b] = mode == 1 ? mx + logf(acc) - logf((float)n_z) : est;
}

// d lw[k][b] = d_out[b] * d out / d lw:  mode 0: 1 / n_z;  mode 1: softmax_k;  mode 2: sum over the prefixes
// K >= k of coef[K] * exp(lw_k - lse_K), coef[K] = [K == m - 1] + [K >= m] inv_w[K] - [K + 1 >= m, K + 1 < n_z] inv_w[K + 1]
// (the prefix logsumexps lse_K are recomputed by a forward sweep, the sum by a backward sweep).
__global__ void mc_objective_bwd_kernel(const float* __restrict__ lw, int n_z, int B, int mode, const float* __restrict__ inv_w,
                                        int m, const float* __restrict__ d_out, float* __restrict__ d_lw,
                                        float* __restrict__ lse_ws) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float g = d_out[b];
  if (mode == 0) {
    for (int k = 0; k < n_z; ++k) d_lw[(size_t)k * B + b] = g / (float)n_z;
    return;
  }
  float mx = -INFINITY, acc = 0.f;
  for (int k = 0; k < n_z; ++k) {
    const float v = lw[(size_t)k * B + b];
    if (v > mx) {
      acc = acc * expf(mx - v) + 1.f;
      mx = v;
    } else {
      acc += expf(v - mx);
    }
    if (mode == 2) lse_ws[(size_t)k * B + b] = mx + logf(acc);
  }
  if (mode == 1) {
    const float lse = mx + logf(acc);
    for (int k = 0; k < n_z; ++k) d_lw[(size_t)k * B + b] = g * expf(lw[(size_t)k * B + b] - lse);
    return;
  }
  // SUMO: tail[k] = sum_{K >= k} coef[K] exp(-lse_K), accumulated relative to a running reference to stay in range
  float ref = -INFINITY, tail = 0.f;  // sum_{K >= k} coef[K] exp(-lse_K) = tail * exp(-ref)
  for (int k = n_z - 1; k >= 0; --k) {
    float coef = (k == m - 1 ? 1.f : 0.f) + (k >= m ? inv_w[k] : 0.f);
    if (k + 1 >= m && k + 1 < n_z) coef -= inv_w[k + 1];
    const float lse = lse_ws[(size_t)k * B + b];
    if (ref == -INFINITY) {
      ref = lse;
      tail = coef;
    } else {
      // lse_k <= lse_{k+1} <= ... : exp(-lse_k) is the largest term so far; re-reference to it
      tail = tail * expf(lse - ref) + coef;
      ref = lse;
    }
    d_lw[(size_t)k * B + b] = g * tail * expf(lw[(size_t)k * B + b] - ref);
  }
}

}  // namespace npf

extern "C" int npf_mc_objective_fwd(const float* log_w, int32_t n_z, int32_t n_tasks, int32_t mode, const float* inv_weights,
                                    int32_t m, float* out, void* stream) {
  if (!log_w || !out || n_z <= 0 || n_tasks <= 0 || mode < 0 || mode > 2) return NPF_EINVAL;
  if (mode == 2 && (!inv_weights || m < 1 || m > n_z)) return NPF_EINVAL;
  hipLaunchKernelGGL(npf::mc_objective_fwd_kernel, dim3((n_tasks + 127) / 128), dim3(128), 0, (hipStream_t)stream, log_w, n_z,
                     n_tasks, mode, inv_weights, m, out);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_mc_objective_bwd(const float* log_w, int32_t n_z, int32_t n_tasks, int32_t mode, const float* inv_weights,
                                    int32_t m, const float* d_out, float* d_log_w, float* workspace, void* stream) {
  if (!log_w || !d_out || !d_log_w || n_z <= 0 || n_tasks <= 0 || mode < 0 || mode > 2) return NPF_EINVAL;
  if (mode == 2 && (!inv_weights || !workspace || m < 1 || m > n_z)) return NPF_EINVAL;
  hipLaunchKernelGGL(npf::mc_objective_bwd_kernel, dim3((n_tasks + 127) / 128), dim3(128), 0, (hipStream_t)stream, log_w, n_z,
                     n_tasks, mode, inv_weights, m, d_out, d_log_w, workspace);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_gauss_head_fwd(const float* suff, int32_t n_rows, int32_t pts, int32_t dy, int32_t homoskedastic,
                                  const float* Y, int32_t n_y_rows, float* loc, float* scale, float* sum_logp,
                                  void* stream) {
  if (!suff || n_rows <= 0 || pts <= 0 || dy <= 0 || dy > 16) return NPF_EINVAL;
  if ((loc == nullptr) != (scale == nullptr)) return NPF_EINVAL;
  if (!loc && !sum_logp) return NPF_EINVAL;  // a launch that writes nothing
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
  if (!suff || !d_suff || n_rows <= 0 || pts <= 0 || dy <= 0 || dy > 16) return NPF_EINVAL;
  if ((loc == nullptr) != (scale == nullptr)) return NPF_EINVAL;
  if (!loc && (d_loc || d_scale)) return NPF_EINVAL;
  if (Y && n_y_rows <= 0) return NPF_EINVAL;
  hipLaunchKernelGGL(npf::gauss_head_bwd_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, suff, loc, scale, pts, dy,
                     homoskedastic, Y, Y ? n_y_rows : 1, d_loc, d_scale, d_sum_logp, d_suff);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}
