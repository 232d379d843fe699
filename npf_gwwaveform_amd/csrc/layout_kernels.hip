// Layout changes at the module boundary (row-major torch tensors <-> PT32), the weight
// transpose used by the dgrad chains, and the mean aggregation over a task's points
// (torch.mean(R_cntxt, dim=1): npf/neuralproc/np.py:95, attnnp.py:181).  All HBM-bound,
// one float4 per thread, coalesced on the PT32 side.
#include "npf_common.hpp"

namespace npf {

__global__ void pack_pt_kernel(const float* __restrict__ rows, int n_tasks, int pts, int Fv, int Fp, float* __restrict__ pt) {
  const int tiles = (pts + 31) / 32;
  const size_t total = (size_t)n_tasks * tiles * (Fp / 4) * 32;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int p = idx & 31;
    const size_t r = idx >> 5;
    const int f4 = r % (Fp / 4);
    const size_t tt = r / (Fp / 4);  // task * tiles + tile
    const int tile = tt % tiles;
    const size_t task = tt / tiles;
    const int point = tile * 32 + p;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (point < pts) {
      const float* src = rows + (task * pts + point) * (size_t)Fv + 4 * f4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * f4 + j < Fv) v[j] = src[j];
    }
    *(f32x4*)(pt + idx * 4) = v;
  }
}

__global__ void unpack_pt_kernel(const float* __restrict__ pt, int n_tasks, int pts, int Fv, int Fp, float* __restrict__ rows) {
  const int tiles = (pts + 31) / 32;
  const size_t total = (size_t)n_tasks * tiles * (Fp / 4) * 32;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int p = idx & 31;
    const size_t r = idx >> 5;
    const int f4 = r % (Fp / 4);
    const size_t tt = r / (Fp / 4);
    const int tile = tt % tiles;
    const size_t task = tt / tiles;
    const int point = tile * 32 + p;
    if (point < pts) {
      const f32x4 v = *(const f32x4*)(pt + idx * 4);
      float* dst = rows + (task * pts + point) * (size_t)Fv + 4 * f4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * f4 + j < Fv) dst[j] = v[j];
    }
  }
}

// Heads as tasks and back (attention.py:505-527), one float4 of the *head-split* tensor per thread.
// SPLIT: hs[h*B + b][p][f] = full[b][p][h*hsz + f];  !SPLIT: the inverse.  Padding quads of the
// destination are written as zeros.
template <bool SPLIT>
__global__ void heads_kernel(const float* __restrict__ src, int n_tasks, int pts, int Fp, int hsz, int n_heads,
                             float* __restrict__ dst) {
  const int tiles = (pts + 31) / 32;
  const int hq = ((hsz + 31) / 32) * 8;  // quads per point of the head tensor (padded to 32 features)
  const int fq = Fp / 4;
  if (SPLIT) {
    const size_t total = (size_t)n_heads * n_tasks * tiles * hq * 32;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
      const int p = idx & 31;
      const size_t r = idx >> 5;
      const int q = r % hq;
      const size_t tt = r / hq;
      const int tile = tt % tiles;
      const size_t ht = tt / tiles;  // h * n_tasks + b
      const int h = ht / n_tasks, b = ht - (size_t)h * n_tasks;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (4 * q < hsz) v = *(const f32x4*)(src + ((((size_t)b * tiles + tile) * fq + (h * hsz) / 4 + q) * 32 + p) * 4);
      *(f32x4*)(dst + idx * 4) = v;
    }
  } else {
    const size_t total = (size_t)n_tasks * tiles * fq * 32;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
      const int p = idx & 31;
      const size_t r = idx >> 5;
      const int q = r % fq;
      const size_t tt = r / fq;
      const int tile = tt % tiles;
      const size_t b = tt / tiles;
      const int h = (4 * q) / hsz, qh = q - (h * hsz) / 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (h < n_heads) v = *(const f32x4*)(src + (((((size_t)h * n_tasks + b) * tiles + tile) * hq + qh) * 32 + p) * 4);
      *(f32x4*)(dst + idx * 4) = v;
    }
  }
}

// CntxtTrgtGetter.select: one thread per selected (task, point); x_dim and y_dim are tiny (1-3 floats).
__global__ void gather_points_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                     const long long* __restrict__ idx, int n_tasks, int n_points, int n_sel, int x_dim,
                                     int y_dim, float* __restrict__ out_x, float* __restrict__ out_y) {
  const size_t total = (size_t)n_tasks * n_sel;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t b = t / n_sel;
    long long j = idx[t];
    j = j < 0 ? 0 : (j >= n_points ? n_points - 1 : j);  // (the host validates; never read out of bounds)
    const float* xs = x + (b * n_points + (size_t)j) * x_dim;
    for (int d = 0; d < x_dim; ++d) out_x[t * x_dim + d] = xs[d];
    if (y) {
      const float* ys = y + (b * n_points + (size_t)j) * y_dim;
      for (int d = 0; d < y_dim; ++d) out_y[t * y_dim + d] = ys[d];
    }
  }
}

// fp32 weights -> the bf16 image the bf16 chain instance streams: dst [rows][Kp] bf16 (Kp = roundup(cols, 32)),
// columns permuted inside every group of 32 so that the 8 bf16 of lane group g are the features
// {4g..4g+3} and {16+4g..16+4g+3} of the group -- the lane's own accumulator values of two adjacent
// 16-row blocks, which is what v_mfma_f32_16x16x32_bf16 gets as its B operand.  transposed: dst rows
// are the columns of src (the dgrad chains multiply by W^T).
__global__ void cast_bf16_weights_kernel(const float* __restrict__ src, int n_rows, int n_cols, int ld, int transposed,
                                         unsigned short* __restrict__ dst) {
  const int rows = transposed ? n_cols : n_rows, cols = transposed ? n_rows : n_cols;
  const int Kp = ((cols + 31) >> 5) * 32;
  const size_t total = (size_t)rows * Kp;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int r = idx / Kp, q = idx - (size_t)r * Kp;
    const int sgrp = q >> 5, g = (q & 31) >> 3, i = q & 7;
    const int c = 32 * sgrp + (i < 4 ? 4 * g + i : 16 + 4 * g + (i - 4));
    float v = 0.f;
    if (c < cols) v = transposed ? src[(size_t)c * ld + r] : src[(size_t)r * ld + c];
    const __bf16 b = (__bf16)v;
    dst[idx] = __builtin_bit_cast(unsigned short, b);
  }
}

// 32x32 LDS tile transpose
__global__ void transpose_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[(size_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) dst[(size_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// Batched weight preparation (npf_prepare_weights): blockIdx.y = job, the blocks of a job stride over its work.
struct WprepJobs {
  npf_wprep_job_t job[NPF_MAX_WPREP_JOBS];
};
__global__ void prepare_weights_kernel(const WprepJobs J) {
  __shared__ float tile[32][33];
  const npf_wprep_job_t& jb = J.job[blockIdx.y];
  const float* __restrict__ src = jb.src;
  if (jb.kind == 0) {
    float* __restrict__ dst = (float*)jb.dst;
    const int rows = jb.n_rows, cols = jb.n_cols, ld = jb.ld;
    const int tc = (cols + 31) / 32, tr = (rows + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tl = blockIdx.x; tl < tc * tr; tl += gridDim.x) {
      const int c0 = (tl % tc) * 32, r0 = (tl / tc) * 32;
      for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = src[(size_t)(r0 + i) * ld + c0 + tx];
      __syncthreads();
      for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) dst[(size_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
      __syncthreads();
    }
  } else {
    unsigned short* __restrict__ dst = (unsigned short*)jb.dst;
    const bool transposed = (jb.kind & 3) == 2;
    // kind bits 4-5: 0 = the rounded value; t = 1..3: term t - 1 of the exact three-term split (x0 = bf16(x), x1 = bf16(x - x0),
    // x2 = bf16(x - x0 - x1); the subtractions are exact in fp32) that the split-product kernels multiply with
    const int term = (jb.kind >> 4) & 3;
    const int rows = transposed ? jb.n_cols : jb.n_rows, cols = transposed ? jb.n_rows : jb.n_cols, ld = jb.ld;
    const int Kp = (jb.kind & 4) ? 256 : ((cols + 31) >> 5) * 32;  // kinds 5, 6: rows zero-padded to 256 inputs
    const size_t total = (size_t)rows * Kp;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
      const int r = idx / Kp, q = idx - (size_t)r * Kp;
      const int sgrp = q >> 5, g = (q & 31) >> 3, i = q & 7;
      const int c = 32 * sgrp + (i < 4 ? 4 * g + i : 16 + 4 * g + (i - 4));
      float v = 0.f;
      if (c < cols) v = transposed ? src[(size_t)c * ld + r] : src[(size_t)r * ld + c];
      for (int t = 1; t < term; ++t) v -= (float)(__bf16)v;
      const __bf16 b = (__bf16)v;
      dst[idx] = __builtin_bit_cast(unsigned short, b);
    }
  }
}

// out[task][f] = mean over valid points.  grid = (ceil(F/32), n_tasks); 256 threads = 8 feature quads x 32 points.
__global__ void mean_agg_fwd_kernel(const float* __restrict__ R, int pts, int F, float* __restrict__ out) {
  const int tiles = (pts + 31) / 32;
  const int p = threadIdx.x & 31, f4 = blockIdx.x * 8 + (threadIdx.x >> 5);
  const size_t task = blockIdx.y;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (f4 < F / 4) {
    const float* base = R + task * tiles * (size_t)(F * 32) + pt_off(f4, p);
    for (int t = 0; t < tiles; ++t)
      if (t * 32 + p < pts) s += *(const f32x4*)(base + (size_t)t * F * 32);
  }
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] += __shfl_xor(s[j], off);
  if (p == 0 && f4 < F / 4) *(f32x4*)(out + task * F + 4 * f4) = s * (1.f / (float)pts);
}

__global__ void mean_agg_bwd_kernel(const float* __restrict__ d_out, int n_tasks, int pts, int F, float* __restrict__ dR,
                                    int accumulate) {
  const int tiles = (pts + 31) / 32;
  const size_t total = (size_t)n_tasks * tiles * (F / 4) * 32;
  const float inv = 1.f / (float)pts;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int p = idx & 31;
    const size_t r = idx >> 5;
    const int f4 = r % (F / 4);
    const size_t tt = r / (F / 4);
    const int tile = tt % tiles;
    const size_t task = tt / tiles;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (tile * 32 + p < pts) v = *(const f32x4*)(d_out + task * F + 4 * f4) * inv;
    if (accumulate) v += *(const f32x4*)(dR + idx * 4);
    *(f32x4*)(dR + idx * 4) = v;
  }
}

static unsigned grid_for(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  if (g > 256u * 8u) g = 256u * 8u;  // grid-stride the rest (cdna guide, Guideline 11)
  if (g == 0) g = 1;
  return (unsigned)g;
}

}  // namespace npf

extern "C" int npf_pack_pt(const float* rows, int32_t n_tasks, int32_t pts, int32_t Fv, float* pt, void* stream) {
  if (!rows || !pt || n_tasks <= 0 || pts <= 0 || Fv <= 0) return NPF_EINVAL;
  const int Fp = npf::round_up(Fv, 32);
  const size_t total = (size_t)n_tasks * ((pts + 31) / 32) * (Fp / 4) * 32;
  hipLaunchKernelGGL(npf::pack_pt_kernel, dim3(npf::grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, rows, n_tasks,
                     pts, Fv, Fp, pt);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_unpack_pt(const float* pt, int32_t n_tasks, int32_t pts, int32_t Fv, float* rows, void* stream) {
  if (!rows || !pt || n_tasks <= 0 || pts <= 0 || Fv <= 0) return NPF_EINVAL;
  const int Fp = npf::round_up(Fv, 32);
  const size_t total = (size_t)n_tasks * ((pts + 31) / 32) * (Fp / 4) * 32;
  hipLaunchKernelGGL(npf::unpack_pt_kernel, dim3(npf::grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, pt,
                     n_tasks, pts, Fv, Fp, rows);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_transpose(const float* src, int32_t rows, int32_t cols, float* dst, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0) return NPF_EINVAL;
  hipLaunchKernelGGL(npf::transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                     src, rows, cols, dst);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

static int heads_launch(bool split, const float* src, int32_t n_tasks, int32_t pts, int32_t F, int32_t n_heads, float* dst,
                        void* stream) {
  if (!src || !dst || n_tasks <= 0 || pts <= 0 || F <= 0 || n_heads <= 0 || F % n_heads) return NPF_EINVAL;
  const int hsz = F / n_heads;
  if (hsz & 3) return NPF_EINVAL;
  const int Fp = npf::round_up(F, 32), tiles = (pts + 31) / 32;
  if (split) {
    const size_t total = (size_t)n_heads * n_tasks * tiles * (npf::round_up(hsz, 32) / 4) * 32;
    hipLaunchKernelGGL(npf::heads_kernel<true>, dim3(npf::grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       n_tasks, pts, Fp, hsz, n_heads, dst);
  } else {
    const size_t total = (size_t)n_tasks * tiles * (Fp / 4) * 32;
    hipLaunchKernelGGL(npf::heads_kernel<false>, dim3(npf::grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       n_tasks, pts, Fp, hsz, n_heads, dst);
  }
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_gather_points(const float* x, const float* y, const int64_t* idx, int32_t n_tasks, int32_t n_points,
                                 int32_t n_sel, int32_t x_dim, int32_t y_dim, float* out_x, float* out_y, void* stream) {
  if (!x || !idx || !out_x || n_tasks <= 0 || n_points <= 0 || n_sel < 0 || x_dim <= 0) return NPF_EINVAL;
  if (y && (!out_y || y_dim <= 0)) return NPF_EINVAL;
  if (n_sel == 0) return NPF_OK;
  const size_t total = (size_t)n_tasks * n_sel;
  hipLaunchKernelGGL(npf::gather_points_kernel, dim3(npf::grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                     (const long long*)idx, n_tasks, n_points, n_sel, x_dim, y_dim, out_x, out_y);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_cast_bf16_weights(const float* src, int32_t n_rows, int32_t n_cols, int32_t ld, int32_t transposed,
                                     void* dst, void* stream) {
  if (!src || !dst || n_rows <= 0 || n_cols <= 0 || ld < n_cols || (((uintptr_t)dst) & 15)) return NPF_EINVAL;
  const int rows = transposed ? n_cols : n_rows, cols = transposed ? n_rows : n_cols;
  const size_t total = (size_t)rows * npf::round_up(cols, 32);
  hipLaunchKernelGGL(npf::cast_bf16_weights_kernel, dim3(npf::grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     n_rows, n_cols, ld, transposed, (unsigned short*)dst);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_prepare_weights(const npf_wprep_job_t* jobs, int32_t n_jobs, void* stream) {
  if (!jobs || n_jobs <= 0 || n_jobs > NPF_MAX_WPREP_JOBS) return NPF_EINVAL;
  npf::WprepJobs J;
  size_t most = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const npf_wprep_job_t& b = jobs[j];
    const int base = b.kind & 15, term = b.kind >> 4;
    const bool kind_ok = b.kind >= 0 && ((term == 0 && (base <= 2 || base == 5 || base == 6)) || (term >= 1 && term <= 3 && (base == 1 || base == 2)));
    if (!b.src || !b.dst || b.n_rows <= 0 || b.n_cols <= 0 || b.ld < b.n_cols || !kind_ok) return NPF_EINVAL;
    if (b.kind != 0 && (((uintptr_t)b.dst) & 15)) return NPF_EINVAL;
    if ((b.kind == 5 && b.n_cols > 256) || (b.kind == 6 && b.n_rows > 256)) return NPF_EINVAL;  // (the image's inputs)
    J.job[j] = b;
    const size_t img_rows = (b.kind & 3) == 2 ? b.n_cols : b.n_rows;
    const size_t blocks = b.kind == 0   ? (size_t)((b.n_rows + 31) / 32) * ((b.n_cols + 31) / 32)
                          : (b.kind & 4) ? (img_rows * 256 + 255) / 256
                                         : ((size_t)b.n_rows * b.n_cols + 255) / 256;
    most = blocks > most ? blocks : most;
  }
  for (int j = n_jobs; j < NPF_MAX_WPREP_JOBS; ++j) J.job[j] = jobs[0];
  const unsigned gx = (unsigned)(most < 64 ? most : 64);
  hipLaunchKernelGGL(npf::prepare_weights_kernel, dim3(gx ? gx : 1, n_jobs), dim3(256), 0, (hipStream_t)stream, J);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_split_heads(const float* src, int32_t n_tasks, int32_t pts, int32_t F, int32_t n_heads, float* dst,
                               void* stream) {
  return heads_launch(true, src, n_tasks, pts, F, n_heads, dst, stream);
}

extern "C" int npf_merge_heads(const float* src, int32_t n_tasks, int32_t pts, int32_t F, int32_t n_heads, float* dst,
                               void* stream) {
  return heads_launch(false, src, n_tasks, pts, F, n_heads, dst, stream);
}

extern "C" int npf_mean_agg_fwd(const float* R_pt, int32_t n_tasks, int32_t pts, int32_t F, float* out, void* stream) {
  if (!R_pt || !out || n_tasks <= 0 || pts <= 0 || F <= 0 || (F & 31)) return NPF_EINVAL;
  hipLaunchKernelGGL(npf::mean_agg_fwd_kernel, dim3(F / 32, n_tasks), dim3(256), 0, (hipStream_t)stream, R_pt, pts, F, out);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_mean_agg_bwd(const float* d_out, int32_t n_tasks, int32_t pts, int32_t F, float* dR_pt,
                                int32_t accumulate, void* stream) {
  if (!d_out || !dR_pt || n_tasks <= 0 || pts <= 0 || F <= 0 || (F & 31)) return NPF_EINVAL;
  const size_t total = (size_t)n_tasks * ((pts + 31) / 32) * (F / 4) * 32;
  hipLaunchKernelGGL(npf::mean_agg_bwd_kernel, dim3(npf::grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, d_out,
                     n_tasks, pts, F, dR_pt, accumulate);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_version(void) { return 1; }
