// Weight / bias gradients of the chain layers:  dW[n][k] = sum_p dZ[n][p] A[k][p],
// db[n] = sum_p dZ[n][p], contraction over the points (autograd of nn.Linear on the path,
// npf/architectures/mlp.py:84-91) -- and, per task, the gradients of the attention keys and
// values (autograd of einsum / bmm, npf/architectures/attention.py:151,212), which are the
// same contraction with the task's targets as the points.
//
// Both operands are PT32 tensors ([F/4][32 points][4 features] per tile), so one 16-byte LDS
// read gives a lane 4 *features* of one point.  With v_mfma_f32_16x16x4_f32 the 4 k-slots are
// 4 consecutive points (lane group g), the lane's row/column is a feature *quad* and the 16
// MFMAs (m, m') of a step use element m of the dZ read and element m' of the A read: one pair
// of ds_read_b128 feeds 16 MFMAs = a 64 x 64 block of dW.  A workgroup (8 waves, 2 per SIMD)
// owns the whole <= 256 x 256 dW: wave w holds rows 64*(w>>1).. and columns 128*(w&1).. in
// 128 accumulator registers, streams its tiles through one LDS buffer (register-staged
// prefetch of the next tile) and writes one partial; a second tiny kernel sums the partials
// of the workgroups that split the points (deterministic, no atomics).
#include "npf_common.hpp"

namespace npf {

constexpr int kWgThreads = 512;
constexpr int kQStride = 136;              // floats per feature-quad row in LDS: 32 pts * 4 + 8 pad
constexpr int kQRows = NPF_MAX_FEATURES / 4;  // 64 quad rows per operand
constexpr int kMaxJobs = 16;

struct WgradJobs {
  npf_wgrad_job_t job[kMaxJobs];
  int32_t first_wg[kMaxJobs + 1];  // workgroup range of each job
  int32_t n_jobs;
  int32_t n_tasks;
  int32_t tiles_per_task;
  int32_t pad;
  int64_t part_off[kMaxJobs];      // float offset of each job's partial slabs (shared-weight jobs)
};

__global__ __launch_bounds__(kWgThreads, 2) void wgrad_kernel(const WgradJobs J, float* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kQRows * kQStride];
  float* ldz = lds;
  float* lda = lds + kQRows * kQStride;

  int j = 0;
  while (j + 1 < J.n_jobs && (int)blockIdx.x >= J.first_wg[j + 1]) ++j;
  const npf_wgrad_job_t& job = J.job[j];
  const int split = blockIdx.x - J.first_wg[j];
  const int n_split = J.first_wg[j + 1] - J.first_wg[j];
  const int Np = ((job.N + 31) >> 5) * 32, Kp = ((job.K + 31) >> 5) * 32;
  const long total_tiles = (long)J.n_tasks * J.tiles_per_task;
  long t0, t1;
  if (job.per_task) {
    t0 = (long)split * J.tiles_per_task;
    t1 = t0 + J.tiles_per_task;
  } else {
    t0 = total_tiles * split / n_split;
    t1 = total_tiles * (split + 1) / n_split;
  }

  const int tid = threadIdx.x, lane = tid & 63;
  const int i = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave >> 1, ch = wave & 1;
  const bool act_a = 64 * rg < Np;
  const bool act_b0 = act_a && (128 * ch < Kp);
  const bool act_b1 = act_a && (128 * ch + 64 < Kp);

  // zero the LDS once: quad rows beyond Np/4, Kp/4 stay zero for the whole kernel
  for (int x = tid; x < 2 * kQRows * kQStride; x += kWgThreads) lds[x] = 0.f;

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc0[4][4], acc1[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      acc0[m][n] = zero4;
      acc1[m][n] = zero4;
    }
  f32x4 dbacc = zero4;

  const int nz4 = 8 * Np, na4 = 8 * Kp;  // float4 per tile of each operand
  f32x4 sz[4], sa[4];
  auto stage_load = [&](long t) {
    const float* zsrc = job.dZ + (size_t)t * Np * 32;
    const float* asrc = job.A + (size_t)t * Kp * 32;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + u * kWgThreads;
      sz[u] = idx < nz4 ? *(const f32x4*)(zsrc + (size_t)idx * 4) : zero4;
      sa[u] = idx < na4 ? *(const f32x4*)(asrc + (size_t)idx * 4) : zero4;
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + u * kWgThreads;
      if (idx < nz4) *(f32x4*)(ldz + (idx >> 5) * kQStride + (idx & 31) * 4) = sz[u];
      if (idx < na4) *(f32x4*)(lda + (idx >> 5) * kQStride + (idx & 31) * 4) = sa[u];
    }
  };

  __syncthreads();
  if (t0 < t1) {
    stage_load(t0);
    stage_write();
  }
  __syncthreads();

  const float* za = ldz + (16 * rg + i) * kQStride + 4 * g;
  const float* ab0 = lda + (32 * ch + i) * kQStride + 4 * g;
  const float* ab1 = ab0 + 16 * kQStride;
  for (long t = t0; t < t1; ++t) {
    const bool more = t + 1 < t1;
    if (more) stage_load(t + 1);
    if (act_a) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const f32x4 a4 = *(const f32x4*)(za + 16 * s);  // 4 features of point 4s+g
        if (ch == 0) dbacc += a4;
        if (act_b0) {
          const f32x4 b4 = *(const f32x4*)(ab0 + 16 * s);
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
              acc0[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m], b4[n], acc0[m][n], 0, 0, 0);
        }
        if (act_b1) {
          const f32x4 b4 = *(const f32x4*)(ab1 + 16 * s);
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
              acc1[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m], b4[n], acc1[m][n], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) {
      stage_write();
      __syncthreads();
    }
  }

  // ---- write out --------------------------------------------------------------------
  // accX[m][n][e] on lane (i, g) = D[row 4*(16rg + 4g + e) + m][col 4*(32ch + 16x + i) + n]
  if (job.per_task) {
    // PT32 tensor, points = row index (n of dW), features = column index (k)
    float* out = job.dW + (size_t)split * ((Np >> 5) * Kp * 32);
    if (act_a) {
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        if (x == 0 ? act_b0 : act_b1) {
          const int kq = 32 * ch + 16 * x + i;  // feature quad of the output
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int row = 4 * (16 * rg + 4 * g + e) + m;
              if (row < Np && 4 * kq < Kp) {
                f32x4 v;
#pragma unroll
                for (int n = 0; n < 4; ++n) v[n] = x == 0 ? acc0[m][n][e] : acc1[m][n][e];
                float* dst = out + ((size_t)(row >> 5) * (Kp >> 2) + kq) * 128 + (row & 31) * 4;
                if (job.accumulate) v += *(const f32x4*)dst;
                *(f32x4*)dst = v;
              }
            }
        }
      }
    }
  } else {
    float* part = partials + J.part_off[j] + (size_t)split * ((size_t)Np * Kp + Np);
    if (act_a) {
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        if (x == 0 ? act_b0 : act_b1) {
          const int col = 4 * (32 * ch + 16 * x + i);
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int row = 4 * (16 * rg + 4 * g + e) + m;
              if (row < Np && col < Kp) {
                f32x4 v;
#pragma unroll
                for (int n = 0; n < 4; ++n) v[n] = x == 0 ? acc0[m][n][e] : acc1[m][n][e];
                *(f32x4*)(part + (size_t)row * Kp + col) = v;
              }
            }
        }
      }
      if (ch == 0) {
        // dbacc[m] on lane (i, g): sum over the points = g (mod 4) of feature 4*(16rg+i)+m
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          dbacc[m] += __shfl_xor(dbacc[m], 16);
          dbacc[m] += __shfl_xor(dbacc[m], 32);
        }
        if (g == 0 && 4 * (16 * rg + i) < Np) *(f32x4*)(part + (size_t)Np * Kp + 4 * (16 * rg + i)) = dbacc;
      }
    }
  }
}

// dW[n][k] (+)= sum_s partial[s][n][k];  db[n] (+)= sum_s partial_db[s][n]
__global__ void wgrad_reduce_kernel(const WgradJobs J, const float* __restrict__ partials) {
  const int j = blockIdx.y;
  const npf_wgrad_job_t& job = J.job[j];
  if (job.per_task) return;
  const int n_split = J.first_wg[j + 1] - J.first_wg[j];
  const int Np = ((job.N + 31) >> 5) * 32, Kp = ((job.K + 31) >> 5) * 32;
  const size_t slab = (size_t)Np * Kp + Np;
  const float* part = partials + J.part_off[j];
  const int total = Np * Kp + Np;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int sp = 0; sp < n_split; ++sp) s += part[(size_t)sp * slab + idx];
    if (idx < Np * Kp) {
      const int n = idx / Kp, k = idx - n * Kp;
      if (n < job.N && k < job.K) {
        float* d = job.dW + (size_t)n * job.ldw + k;
        *d = job.accumulate ? *d + s : s;
      }
    } else if (job.db) {
      const int n = idx - Np * Kp;
      if (n < job.N) job.db[n] = job.accumulate ? job.db[n] + s : s;
    }
  }
}

static int plan(const npf_wgrad_job_t* jobs, int n_jobs, int n_tasks, int tiles_per_task, WgradJobs* J) {
  if (!jobs || n_jobs <= 0 || n_jobs > kMaxJobs || n_tasks <= 0 || tiles_per_task <= 0) return NPF_EINVAL;
  const long total_tiles = (long)n_tasks * tiles_per_task;
  int n_shared = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const npf_wgrad_job_t& b = jobs[j];
    if (!b.dZ || !b.A || !b.dW || b.N <= 0 || b.K <= 0 || b.N > NPF_MAX_FEATURES || b.K > NPF_MAX_FEATURES) return NPF_EINVAL;
    if ((((uintptr_t)b.dZ) | ((uintptr_t)b.A)) & 15) return NPF_EINVAL;
    if (b.per_task && (((uintptr_t)b.dW) & 15)) return NPF_EINVAL;
    if (!b.per_task && b.ldw < b.K) return NPF_EINVAL;
    n_shared += b.per_task ? 0 : 1;
  }
  long splits = n_shared ? 256 / n_shared : 1;
  if (splits < 1) splits = 1;
  if (splits > total_tiles) splits = total_tiles;
  J->n_jobs = n_jobs;
  J->n_tasks = n_tasks;
  J->tiles_per_task = tiles_per_task;
  J->pad = 0;
  int wg = 0;
  int64_t off = 0;
  for (int j = 0; j < n_jobs; ++j) {
    J->job[j] = jobs[j];
    J->first_wg[j] = wg;
    J->part_off[j] = off;
    if (jobs[j].per_task) {
      wg += n_tasks;
    } else {
      const int Np = npf::round_up(jobs[j].N, 32), Kp = npf::round_up(jobs[j].K, 32);
      wg += (int)splits;
      off += (int64_t)splits * ((int64_t)Np * Kp + Np);
    }
  }
  J->first_wg[n_jobs] = wg;
  for (int j = n_jobs + 1; j <= kMaxJobs; ++j) J->first_wg[j] = wg;
  return (int)0;
}

}  // namespace npf

extern "C" int64_t npf_wgrad_partials_bytes(const npf_wgrad_job_t* jobs, int32_t n_jobs, int32_t n_tasks,
                                            int32_t tiles_per_task) {
  npf::WgradJobs J;
  if (npf::plan(jobs, n_jobs, n_tasks, tiles_per_task, &J) != 0) return -1;
  int64_t off = 0;
  for (int j = 0; j < n_jobs; ++j) {
    if (jobs[j].per_task) continue;
    const int Np = npf::round_up(jobs[j].N, 32), Kp = npf::round_up(jobs[j].K, 32);
    off += (int64_t)(J.first_wg[j + 1] - J.first_wg[j]) * ((int64_t)Np * Kp + Np);
  }
  return off * 4 + 16;
}

extern "C" int npf_wgrad_run(const npf_wgrad_job_t* jobs, int32_t n_jobs, int32_t n_tasks, int32_t tiles_per_task,
                             float* partials, int64_t partials_bytes, void* stream) {
  npf::WgradJobs J;
  const int rc = npf::plan(jobs, n_jobs, n_tasks, tiles_per_task, &J);
  if (rc != 0) return rc;
  const int64_t need = npf_wgrad_partials_bytes(jobs, n_jobs, n_tasks, tiles_per_task);
  bool any_shared = false;
  for (int j = 0; j < n_jobs; ++j) any_shared |= !jobs[j].per_task;
  if (any_shared && (!partials || partials_bytes < need || (((uintptr_t)partials) & 15))) return NPF_EINVAL;
  const int n_wg = J.first_wg[n_jobs];
  hipLaunchKernelGGL(npf::wgrad_kernel, dim3(n_wg), dim3(npf::kWgThreads), 0, (hipStream_t)stream, J, partials);
  NPF_CHECK_LAUNCH();
  if (any_shared) {
    hipLaunchKernelGGL(npf::wgrad_reduce_kernel, dim3(64, n_jobs), dim3(256), 0, (hipStream_t)stream, J,
                       (const float*)partials);
    NPF_CHECK_LAUNCH();
  }
  return NPF_OK;
}
