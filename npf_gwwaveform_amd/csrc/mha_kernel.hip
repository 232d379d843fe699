// Multihead scaled-dot attention with 16- or 32-feature heads, forward and backward, straight on the PT32 tensors the K / Q / V
// projections leave (npf_mha_fwd / npf_mha_bwd).
//
// What it computes, in the reference's terms: MultiheadAttender.forward between the projections and the concatenation
// (npf/architectures/attention.py:505-527: heads stacked as extra batches, DotAttender per head with the HEAD size in the scale
// :216-218, heads concatenated back) -- out[b, q, D h + :] = softmax_k(Q_h K_h^T / sqrt(D)) V_h with X_h = X[..., D h : D h + D],
// D = 16 (the reference's default r_dim = 128 with 8 heads: what TransformerAttender and every shipped Attn* checkpoint run; up to
// 256 keys) or 32 (r_dim = 256 with 8 heads; up to 128 keys).  No split / merge of heads is materialised: a head is a 16-feature slice of the PT32 tile.
//
// fp32 throughout (v_mfma_f32_16x16x4_f32, k-ordered accumulation, softmax with max subtraction, expf) -- same gates as the chain
// kernel's attention.  The softmax'ed scores never leave the registers: the first contraction is computed TRANSPOSED
// (S^T[key][query], accumulator rows = keys 4 g + i, column = the lane's query), so element j of the accumulator is directly the B
// operand of the second contraction's k-chunk {keys 4 g + j} (a sum over keys does not care about their order).
// The backward pass recomputes the probabilities from the saved log-sum-exp (one float per head and query) in both layouts:
// transposed for dQ, plain for dK / dV, which a workgroup owning one (task, head) accumulates over all its queries in registers.
#include "npf_common.hpp"

namespace npf {

// head size D = 16 (up to 256 keys) or 32 (up to 128 keys); LDS row stride of a head's keys / values = D + 1 floats: conflict-free
// for both operand roles
constexpr int kMhaMaxKeys = 256;

// feature f (a multiple of 4 -> one float4) of point p of a task's PT32 tensor with Fp (padded) features
__device__ __forceinline__ size_t mha_pt(int task, int tiles, int Fp, int p, int f) {
  return ((((size_t)task * tiles + (p >> 5)) * (Fp >> 2) + (f >> 2)) * 32 + (p & 31)) * 4 + (f & 3);
}

__device__ __forceinline__ f32x4 mha_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float mha_sum4(float v) {  // over the four lanes (g = 0..3) that share a column
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float mha_max4(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

// a head's keys and values into LDS: rows = keys (zero rows up to Cp), kMhaLd floats apart
template <int D>
__device__ __forceinline__ void mha_stage(const float* __restrict__ K, const float* __restrict__ V, int b, int h, int tilesC, int Fp,
                                          int C, int Cp, float* Ks, float* Vs, int tid, int n_threads) {
  constexpr int kMhaD = D, kMhaLd = D + 1, NKC = D / 4;
  for (int i = tid; i < Cp * NKC; i += n_threads) {
    const int key = i / NKC, kc = i % NKC;
    f32x4 k = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
    if (key < C) {
      const size_t at = mha_pt(b, tilesC, Fp, key, kMhaD * h + 4 * kc);
      k = *(const f32x4*)(K + at);
      v = *(const f32x4*)(V + at);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      Ks[key * kMhaLd + 4 * kc + e] = k[e];
      Vs[key * kMhaLd + 4 * kc + e] = v[e];
    }
  }
}

// One workgroup = one (task, head, up to 256 queries): the head's keys / values are staged once, a wave takes 16 queries per round.
template <int D>
__global__ __launch_bounds__(256) void mha_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                     const float* __restrict__ V, float* __restrict__ O, float* __restrict__ lse,
                                                     int n_tasks, int n_heads, int C, int T, int Fp, float scale) {
  constexpr int kMhaD = D, kMhaLd = D + 1, NKC = D / 4, NDT = D / 16, MaxKeys = D == 16 ? kMhaMaxKeys : kMhaMaxKeys / 2;
  __shared__ float Ks[MaxKeys * kMhaLd], Vs[MaxKeys * kMhaLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int qblocks = (T + 255) >> 8;
  const int qb = blockIdx.x % qblocks, bh = blockIdx.x / qblocks, h = bh % n_heads, b = bh / n_heads;
  const int tilesC = (C + 31) >> 5, tilesT = (T + 31) >> 5, Cp = (C + 15) & ~15, nblk = Cp >> 4;
  mha_stage<D>(K, V, b, h, tilesC, Fp, C, Cp, Ks, Vs, tid, 256);
  __syncthreads();
  for (int round = 0; round < 4; ++round) {
  const int q = qb * 256 + round * 64 + wave * 16 + c;
  if (qb * 256 + round * 64 + wave * 16 >= T) break;  // (wave-uniform)
  const bool live = q < T;
  // the lane's query as an operand: Q[q][4 kc + g]
  float Qq[NKC];
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) Qq[kc] = live ? Q[mha_pt(b, tilesT, Fp, q, kMhaD * h + 4 * kc) + g] : 0.f;
  // S^T[key = 16 blk + 4 g + i][q = c]
  f32x4 S[MaxKeys / 16];
  float mx = -INFINITY;
#pragma unroll
  for (int blk = 0; blk < MaxKeys / 16; ++blk) {
    if (blk < nblk) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) acc = mha_mfma(Ks[(16 * blk + c) * kMhaLd + 4 * kc + g], Qq[kc], acc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = (16 * blk + 4 * g + i < C) ? acc[i] * scale : -INFINITY;
        mx = fmaxf(mx, acc[i]);
      }
      S[blk] = acc;
    }
  }
  mx = mha_max4(mx);
  float sum = 0.f;
#pragma unroll
  for (int blk = 0; blk < MaxKeys / 16; ++blk) {
    if (blk < nblk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        S[blk][i] = expf(S[blk][i] - mx);  // (exp(-inf) = 0 for the padding keys)
        sum += S[blk][i];
      }
    }
  }
  sum = mha_sum4(sum);
  const float inv = 1.f / sum;
  // O^T[dv = 16 dt + 4 g + i][q = c] = sum_key V[key][dv] P^T[key][q], k-chunk j of block blk = keys 16 blk + 4 g + j
  f32x4 o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int blk = 0; blk < MaxKeys / 16; ++blk) {
    if (blk < nblk) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = mha_mfma(Vs[(16 * blk + 4 * g + j) * kMhaLd + 16 * dt + c], S[blk][j] * inv, o[dt]);
    }
  }
  if (live) {
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) *(f32x4*)(O + mha_pt(b, tilesT, Fp, q, kMhaD * h + 16 * dt + 4 * g)) = o[dt];
    if (g == 0 && lse != nullptr) lse[((size_t)b * n_heads + h) * T + q] = mx + logf(sum);
  }
  }
}

// One workgroup = one (task, head); its four waves take the 16-query blocks in turn and keep dK^T / dV^T of all keys in registers.
template <int D, int NBLK>  // head size; key blocks held (keys <= 16 NBLK)
__global__ __launch_bounds__(256) void mha_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                     const float* __restrict__ V, const float* __restrict__ O,
                                                     const float* __restrict__ dO, const float* __restrict__ lse,
                                                     float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV,
                                                     int n_tasks, int n_heads, int C, int T, int Fp, float scale) {
  constexpr int kMhaD = D, kMhaLd = D + 1, NKC = D / 4, NDT = D / 16;
  __shared__ float Ks[16 * NBLK * kMhaLd], Vs[16 * NBLK * kMhaLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int h = blockIdx.x % n_heads, b = blockIdx.x / n_heads;
  const int tilesC = (C + 31) >> 5, tilesT = (T + 31) >> 5, Cp = (C + 15) & ~15, nblk = Cp >> 4;
  mha_stage<D>(K, V, b, h, tilesC, Fp, C, Cp, Ks, Vs, tid, 256);
  __syncthreads();
  f32x4 aK[NBLK][NDT], aV[NBLK][NDT];  // dK^T / dV^T [feature 16 dt + 4 g + i][key = 16 blk + c]
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) aK[blk][dt] = aV[blk][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* lse_h = lse + ((size_t)b * n_heads + h) * T;
  for (int q0 = wave * 16; q0 < T; q0 += 64) {
    const int q = q0 + c;
    const bool live = q < T;
    // operands with (lane % 16 -> query, lane / 16 -> feature within the k-chunk): Q, dO, O at [q][4 kc + g]
    float Qq[NKC], Gq[NKC];
    float dsum = 0.f;
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      const size_t at = mha_pt(b, tilesT, Fp, live ? q : 0, kMhaD * h + 4 * kc) + g;
      Qq[kc] = live ? Q[at] : 0.f;
      Gq[kc] = live ? dO[at] : 0.f;
      dsum = fmaf(Gq[kc], live ? O[at] : 0.f, dsum);
    }
    const float Dc = mha_sum4(dsum);              // sum_dv dO[q][dv] O[q][dv] of the lane's query
    const float Lc = live ? lse_h[q] : 0.f;
    // ... and with (lane / 16, j -> query 4 g + j; lane % 16 -> feature): Q, dO at [q0 + 4 g + j][16 dt + c]
    float Qa[NDT][4], Ga[NDT][4], Dr[4], Lr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int qj = q0 + 4 * g + j;
      const bool lj = qj < T;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const size_t at = mha_pt(b, tilesT, Fp, lj ? qj : 0, kMhaD * h + 16 * dt + (c & ~3)) + (c & 3);
        Qa[dt][j] = lj ? Q[at] : 0.f;
        Ga[dt][j] = lj ? dO[at] : 0.f;
      }
      Dr[j] = __shfl(Dc, 4 * g + j);  // (lane 4 g + j holds query q0 + 4 g + j in its column role)
      Lr[j] = __shfl(Lc, 4 * g + j);
    }
    f32x4 dq[NDT];  // dQ^T[d = 16 dt + 4 g + i][q = c]
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      if (blk < nblk) {
        float Kr[NKC], Vr[NKC];  // K / V [key = 16 blk + c][4 kc + g]: A operand of the transposed products, B operand of the plain ones
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
          Kr[kc] = Ks[(16 * blk + c) * kMhaLd + 4 * kc + g];
          Vr[kc] = Vs[(16 * blk + c) * kMhaLd + 4 * kc + g];
        }
        // transposed: rows = keys 4 g + i, column = query c
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
          st = mha_mfma(Kr[kc], Qq[kc], st);
          dpt = mha_mfma(Vr[kc], Gq[kc], dpt);
        }
        f32x4 dst;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = (16 * blk + 4 * g + i < C) ? expf(st[i] * scale - Lc) : 0.f;
          dst[i] = scale * p * (dpt[i] - Dc);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) dq[dt] = mha_mfma(Ks[(16 * blk + 4 * g + j) * kMhaLd + 16 * dt + c], dst[j], dq[dt]);
        // plain: rows = queries 4 g + i, column = key c
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
          s = mha_mfma(Qq[kc], Kr[kc], s);
          dp = mha_mfma(Gq[kc], Vr[kc], dp);
        }
        f32x4 pr, ds;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pr[i] = (16 * blk + c < C && q0 + 4 * g + i < T) ? expf(s[i] * scale - Lr[i]) : 0.f;
          ds[i] = scale * pr[i] * (dp[i] - Dr[i]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            aV[blk][dt] = mha_mfma(Ga[dt][j], pr[j], aV[blk][dt]);   // dV^T[dv][key]: A[row = dv = 16 dt + c][k = query 4 g + j]
            aK[blk][dt] = mha_mfma(Qa[dt][j], ds[j], aK[blk][dt]);
          }
      }
    }
    if (live) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) *(f32x4*)(dQ + mha_pt(b, tilesT, Fp, q, kMhaD * h + 16 * dt + 4 * g)) = dq[dt];
    }
  }
  // the four waves' partial dK^T / dV^T meet in LDS (the keys / values are not needed any more)
  __syncthreads();
  float* red = Ks;  // [wave 1..3][64 lanes][4], one key block and 16-feature tile at a time
  static_assert(3 * 256 <= 16 * NBLK * kMhaLd, "reduction buffer");
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      if (blk < nblk) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const f32x4 mine = pass == 0 ? aK[blk][dt] : aV[blk][dt];
          if (wave > 0) *(f32x4*)(red + ((wave - 1) * 64 + lane) * 4) = mine;
          __syncthreads();
          if (wave == 0) {
            f32x4 t = mine;
#pragma unroll
            for (int w = 0; w < 3; ++w) t += *(const f32x4*)(red + (w * 64 + lane) * 4);
            const int key = 16 * blk + c;
            if (key < C) *(f32x4*)((pass == 0 ? dK : dV) + mha_pt(b, tilesC, Fp, key, kMhaD * h + 16 * dt + 4 * g)) = t;
          }
          __syncthreads();
        }
      }
    }
  }
}

}  // namespace npf

static int mha_check(const void* a, const void* b, const void* c, const void* d, int32_t n_tasks, int32_t n_heads, int32_t n_keys,
                     int32_t n_queries, int32_t F) {
  if (!a || !b || !c || !d || n_tasks <= 0 || n_heads <= 0 || n_keys <= 0 || n_queries <= 0) return NPF_EINVAL;
  const int D = F / n_heads;
  if (F != n_heads * D || (D != 16 && D != 32) || n_keys > (D == 16 ? npf::kMhaMaxKeys : npf::kMhaMaxKeys / 2)) return NPF_EINVAL;
  if ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) return NPF_EINVAL;
  return NPF_OK;
}

extern "C" int npf_mha_fwd(const float* q, const float* k, const float* v, int32_t n_tasks, int32_t n_heads, int32_t n_keys,
                           int32_t n_queries, int32_t F, float* out, float* lse, void* stream) {
  const int rc = mha_check(q, k, v, out, n_tasks, n_heads, n_keys, n_queries, F);
  if (rc != NPF_OK) return rc;
  const int Fp = npf::round_up(F, 32), D = F / n_heads;
  const dim3 grid(n_tasks * n_heads * ((n_queries + 255) / 256)), block(256);
  const float scale = 1.0f / sqrtf((float)D);
  if (D == 16)
    hipLaunchKernelGGL(npf::mha_fwd_kernel<16>, grid, block, 0, (hipStream_t)stream, q, k, v, out, lse, n_tasks, n_heads, n_keys,
                       n_queries, Fp, scale);
  else
    hipLaunchKernelGGL(npf::mha_fwd_kernel<32>, grid, block, 0, (hipStream_t)stream, q, k, v, out, lse, n_tasks, n_heads, n_keys,
                       n_queries, Fp, scale);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_mha_bwd(const float* q, const float* k, const float* v, const float* out, const float* d_out, const float* lse,
                           int32_t n_tasks, int32_t n_heads, int32_t n_keys, int32_t n_queries, int32_t F, float* d_q, float* d_k,
                           float* d_v, void* stream) {
  const int rc = mha_check(q, k, v, out, n_tasks, n_heads, n_keys, n_queries, F);
  if (rc != NPF_OK) return rc;
  if (!d_out || !lse || !d_q || !d_k || !d_v) return NPF_EINVAL;
  if ((((uintptr_t)d_out) | ((uintptr_t)d_q) | ((uintptr_t)d_k) | ((uintptr_t)d_v)) & 15) return NPF_EINVAL;
  const int Fp = npf::round_up(F, 32), D = F / n_heads;
  const float scale = 1.0f / sqrtf((float)D);
  const dim3 grid(n_tasks * n_heads), block(256);
  hipStream_t st = (hipStream_t)stream;
#define MHA_BWD(DD, N)                                                                                                              \
  hipLaunchKernelGGL((npf::mha_bwd_kernel<DD, N>), grid, block, 0, st, q, k, v, out, d_out, lse, d_q, d_k, d_v, n_tasks, n_heads, \
                     n_keys, n_queries, Fp, scale)
  if (D == 16) {
    if (n_keys <= 64) MHA_BWD(16, 4);
    else if (n_keys <= 128) MHA_BWD(16, 8);
    else MHA_BWD(16, 16);
  } else {
    if (n_keys <= 64) MHA_BWD(32, 4);
    else MHA_BWD(32, 8);
  }
#undef MHA_BWD
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}
