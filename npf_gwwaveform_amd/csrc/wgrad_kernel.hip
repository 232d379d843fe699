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
// of ds_read_b128 feeds 16 MFMAs = a 64 x 64 block of dW.  A workgroup (16 waves, 4 per SIMD)
// owns the whole <= 256 x 256 dW: wave w holds the 64 x 64 block (w>>2, w&3) in 64 accumulator
// registers, streams its tiles by LDS-DMA through a double-buffered, XOR-
// swizzled LDS image (one barrier per tile) and writes one partial; a second tiny kernel sums
// the partials of the workgroups that split the points (deterministic, no atomics).
#include "npf_common.hpp"

namespace npf {

constexpr int kWgThreads = 1024;
constexpr int kWgWaves = kWgThreads / 64;
constexpr int kWgMaxF = 256;            // widest layer side this kernel handles
constexpr int kQRows = kWgMaxF / 4;     // 64 quad rows per operand
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

__device__ __forceinline__ void wg_dma16(const float* src, float* lds_dst_wave_uniform) {
  // (aux = 2, nt, for these read-once tiles was measured: config 3 12.85 -> 12.95 ms; default policy kept)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

// LDS image of one tile of one operand: [F/4 feature-quad rows][32 points] 16-byte chunks,
// dense (512 B per row), chunk c of row r stored at chunk position c ^ (r & 15): the fragment
// read (16 quad rows x 4 consecutive points per ds_read_b128) is then bank-conflict free and
// a 1 KiB LDS-DMA piece (2 rows) lands linearly with the swizzle applied to its source address.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x4 pack4_bf16(float a, float b, float c, float d) {
  const bf16x2v lo = __builtin_convertvector((f32x2v{a, b}), bf16x2v), hi = __builtin_convertvector((f32x2v{c, d}), bf16x2v);
  bf16x4 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
  return r;
}

// The four bf16x4 operands of one 16-point step (MFMA lane i = one feature, its 4 k-slots = the 4 consecutive
// points 16 t2 + 4 g .. + 3) of the wave's 64-feature block `blk`, and which feature that is:
//   fp32 tile image: lane i = quad row 16 blk + i, operand m = feature 4 (16 blk + i) + m:
//     4 x ds_read_b128 (one per point), rounded here with v_cvt_pk_bf16_f32;
//   PT16 tile image ([F/8 rows][32 points][8 bf16], row 4 s + gg = features {32 s + 4 gg + j}, {32 s + 16 + 4 gg + j}):
//     one ds_read_b64_tr_b16 per operand -- the hardware transpose turns 4 points x 16 features (lane 4 q + p of a
//     16-lane group supplies point q, the 8 bytes = 4 features of column group p) into "lane c holds feature c at
//     the 4 points".  The 16 features of operand m are {32 s + 16 h + 4 m + j}: s in {s0, s0 + 2}, h, j -- rows
//     4 s0 + m and 4 (s0 + 2) + m differ in bit 3 of the row swizzle, which makes the read bank-conflict free
//     (32 lanes, 32 distinct bank pairs); the block owns s0 = 4 (blk >> 1) + (blk & 1) and s0 + 2.
// `want_sum`: `dsum` accumulates the sum over the points (bias gradient), element m = the lane's feature of operand m.
__device__ __forceinline__ int wg_s0(int blk) { return 4 * (blk >> 1) + (blk & 1); }

__device__ __forceinline__ void operand16(const float* img, bool fmt16, int blk, int i, int g, int t2, bf16x4 (&op)[4],
                                          bool want_sum, f32x4& dsum) {
  if (!fmt16) {
    // (inline asm, here and below: a C++ LDS load after the tile DMA was issued makes hipcc wait for vmcnt(0)
    // first -- it cannot tell the two LDS buffers apart -- and the double buffering is gone)
    f32x4 q[4];
    const unsigned lbase = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)img + (16 * blk + i) * 512;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int c = 16 * t2 + 4 * g + jj;
      asm volatile("ds_read_b128 %0, %1" : "=v"(q[jj]) : "v"(lbase + ((c ^ i) << 4)));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
    if (want_sum) dsum += (q[0] + q[1]) + (q[2] + q[3]);
#pragma unroll
    for (int m = 0; m < 4; ++m) op[m] = pack4_bf16(q[0][m], q[1][m], q[2][m], q[3][m]);
  } else {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int q = i >> 2, p = i & 3;
    const int row0 = 4 * (wg_s0(blk) + 2 * (p >> 1));  // + m
    const int point = 16 * t2 + 4 * g + q;
    const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)img + 8u * (p & 1);
    u32x2 r[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = row0 + m;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r[m]) : "v"(base + row * 512 + ((point ^ (row & 15)) << 4)));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      op[m] = __builtin_bit_cast(bf16x4, r[m]);
    }
    if (want_sum) {
      f32x4 sm;
#pragma unroll
      for (int m = 0; m < 4; ++m)
        sm[m] = (__builtin_bit_cast(float, r[m][0] << 16) + __builtin_bit_cast(float, r[m][0] & 0xffff0000u)) +
                (__builtin_bit_cast(float, r[m][1] << 16) + __builtin_bit_cast(float, r[m][1] & 0xffff0000u));
      dsum += sm;
    }
  }
}

// Feature index of (64-feature block, operand lane i, operand m) for the two tile formats.
__device__ __forceinline__ int wg_feature(bool fmt16, int blk, int i, int m) {
  if (!fmt16) return 4 * (16 * blk + i) + m;
  return 32 * (wg_s0(blk) + 2 * (i >> 3)) + 16 * ((i >> 2) & 1) + 4 * m + (i & 3);
}
// Does block `blk` hold any feature < Fp (Fp a multiple of 32)?
__device__ __forceinline__ bool wg_block_active(bool fmt16, int blk, int Fp) {
  return fmt16 ? 32 * wg_s0(blk) < Fp : 64 * blk < Fp;
}

// BF16: the same contraction with the operands rounded to bf16 at the MFMA input
// (v_mfma_f32_16x16x16_bf16, fp32 accumulation): the k-slots of a lane group are 4 consecutive points, so
// a lane reads its quad row at 4 points (fp32 tiles: 4 x ds_read_b128 per operand; PT16 tiles: 4 transposing
// reads, see operand16) and forms the 4 + 4 operands of the 16 (m, m') MFMAs of a 16-point step; 2 steps per
// tile instead of 8 -- HBM traffic (4 TB/s measured with PT16 operands), not the MFMAs, then sets the pace.
template <bool BF16>
__global__ __launch_bounds__(kWgThreads, 4) void wgrad_kernel(const WgradJobs J, float* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kQRows * 128];  // 2 buffers x (dZ, A)
  constexpr int kOp = kQRows * 128;  // floats per operand image

  int j = 0;
  while (j + 1 < J.n_jobs && (int)blockIdx.x >= J.first_wg[j + 1]) ++j;
  const npf_wgrad_job_t& job = J.job[j];
  const int split = blockIdx.x - J.first_wg[j];
  const int n_split = J.first_wg[j + 1] - J.first_wg[j];
  const int Np = ((job.N + 31) >> 5) * 32, Kp = ((job.K + 31) >> 5) * 32;
  const long total_tiles = (long)J.n_tasks * J.tiles_per_task;
  // A shared-weight job's tiles are dealt round-robin to its workgroups (tile split + n * n_split): the
  // workgroups advance together, so at any time they read one contiguous region (all HBM channels) instead
  // of n_split streams a fixed large stride apart.
  long t0, t1;
  int tstride = 1;
  if (job.per_task) {
    t0 = (long)split * J.tiles_per_task;
    t1 = t0 + J.tiles_per_task;
  } else {
    t0 = split;
    t1 = total_tiles;
    tstride = n_split;
  }

  const int tid = threadIdx.x, lane = tid & 63;
  const int i = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 64 x 64 block (rg, cg) of dW owned by this wave.  Consecutive waves sit on different SIMDs,
  // so the narrower block dimension goes to the slow-varying index: a skinny job (one active
  // block row or column) then has one active wave per SIMD instead of four on one SIMD.
  const bool wide_rows = ((Kp + 63) >> 6) < ((Np + 63) >> 6);
  const int rg = wide_rows ? (wave & 3) : (wave >> 2);
  const int cg = wide_rows ? (wave >> 2) : (wave & 3);
  const bool act = wg_block_active(BF16 && (job.accumulate & NPF_WGRAD_DZ16) != 0, rg, Np) &&
                   wg_block_active(BF16 && (job.accumulate & NPF_WGRAD_A16) != 0, cg, Kp);

  // zero the LDS once: quad rows beyond Np/4, Kp/4 are never written and must read as zero
  for (int x = tid; x < 2 * 2 * kOp; x += kWgThreads) lds[x] = 0.f;

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = zero4;
  f32x4 dbacc = zero4;

  // DMA: piece q of an operand = its quad rows 2q, 2q+1; wave w issues pieces w, w+16, ...: the
  // swizzle term (row & 15) = (2w + (lane >> 5)) & 15 is the same for all of them
  // (uniform 64-bit base + 32-bit lane offset = the SGPR-base form of the DMA instruction: no VALU
  // address arithmetic per piece; the operand pointers are pinned in SGPRs for the whole loop)
  const unsigned dma_lane = 4u * (unsigned)((lane >> 5) * 128 + (((lane & 31) ^ ((2 * wave + (lane >> 5)) & 15)) << 2));
  const char* dz_base = (const char*)job.dZ;
  const char* a_base = (const char*)job.A;
  asm volatile("" : "+s"(dz_base), "+s"(a_base));
  // PT16 operands (bf16 variant only, NPF_WGRAD_DZ16 / NPF_WGRAD_A16): a tile is [F/8 rows of 8 features][32
  // points] 16-byte chunks = half the bytes and half the rows of the fp32 tile; same 512-byte rows, same swizzle.
  const bool z16 = BF16 && (job.accumulate & NPF_WGRAD_DZ16) != 0, a16 = BF16 && (job.accumulate & NPF_WGRAD_A16) != 0;
  const int zsh = z16 ? 4 : 3, ash = a16 ? 4 : 3;  // pieces per tile = F >> sh, tile bytes = F << (10 - sh)
  // features per tile of the tensors the operands live in (a job on a block of <= 256 features of a wider tensor --
  // NPF_MAX_TRAIN_FEATURES = 512 is split into such jobs -- points at the block's first feature row and strides over
  // the whole tile)
  const int Zf = job.ldz > 0 ? job.ldz : Np, Af = job.lda > 0 ? job.lda : Kp;
  auto tile_dma = [&](long t, float* buf) {
    const char* zsrc = dz_base + (((size_t)t * Zf) << (10 - zsh));
    const char* asrc = a_base + (((size_t)t * Af) << (10 - ash));
    for (int q = wave; q < (Np >> zsh); q += kWgWaves)
      wg_dma16((const float*)(zsrc + (size_t)q * 1024 + (size_t)dma_lane), buf + q * 256);
    for (int q = wave; q < (Kp >> ash); q += kWgWaves)
      wg_dma16((const float*)(asrc + (size_t)q * 1024 + (size_t)dma_lane), buf + kOp + q * 256);
  };

  // Both operands PT16: a tile pair fits one buffer (each image is half of kOp), and two tiles go through
  // every barrier: the bytes in flight per workgroup of the fp32 tiles (one 32 KiB pair in flight per
  // workgroup is latency-bound at ~4 TB/s).
  [[maybe_unused]] const bool dual = z16 && a16;
  const long tstep = (BF16 && dual) ? 2 * (long)tstride : (long)tstride;
  auto step_dma = [&](long t, float* buf) {  // the tile(s) of one pipeline step
    tile_dma(t, buf);
    if constexpr (BF16) {
      if (dual && t + tstride < t1) tile_dma(t + tstride, buf + kOp / 2);
    }
  };

  __syncthreads();
  if (t0 < t1 && !(BF16 && dual)) step_dma(t0, lds);

  // fragment addresses: chunk (4 s + g) ^ i of quad row (base + i); with s = 4 m + t the lane
  // part only depends on t (4 address registers per operand, m goes to the immediate offset)
  unsigned za_l[4], ab_l[4];
  {
    const unsigned l0 = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)lds;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned lanepart = (((t ^ (i >> 2)) << 2) | (g ^ (i & 3))) << 4;
      za_l[t] = l0 + (16 * rg + i) * 512 + lanepart;
      ab_l[t] = l0 + kOp * 4 + (16 * cg + i) * 512 + lanepart;
    }
  }

  // bf16 variant, both operands PT16 (every MLP layer of a bf16 step): FOUR slots of one tile pair (16 + 16 KiB) each,
  // three tiles in flight while the fourth is multiplied.  A tile is retired by a COUNTED s_waitcnt vmcnt (each wave
  // issues `pw` <= 2 DMA instructions per tile: the two younger tiles may stay in flight) and a raw s_barrier; the DMA
  // of tile n + 3 goes into the slot tile n - 1 has just left.  With two tiles per barrier and a full drain at every
  // barrier (the loop below) the launch sat at ~4.3 TB/s: 64 KiB in flight per CU, none across the barrier.
  if constexpr (BF16) {
    if (dual) {
      const int pw = (wave < (Np >> zsh) ? 1 : 0) + (wave < (Kp >> ash) ? 1 : 0);
      auto slot_of = [&](long n) { return lds + ((n >> 1) & 1) * 2 * kOp + (n & 1) * (kOp / 2); };
      const long n_tiles = t0 < t1 ? (t1 - t0 + tstride - 1) / tstride : 0;
      for (long n = 0; n < 3 && n < n_tiles; ++n) tile_dma(t0 + n * tstride, slot_of(n));
      for (long n = 0; n < n_tiles; ++n) {
        const long younger = (n_tiles - 1 - n) < 2 ? (n_tiles - 1 - n) : 2;  // tiles issued after tile n
        const int allow = (int)younger * pw;
        if (allow >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (allow >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (allow >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // tile n has landed for everyone; everyone is done with tile n - 1
        if (n + 3 < n_tiles) tile_dma(t0 + (n + 3) * tstride, slot_of(n + 3));
        if (act) {
          const float* zimg = slot_of(n);
          const float* aimg = zimg + kOp;
          // the whole 32-point tile in ONE v_mfma_f32_16x16x32_bf16 per (m, n2): a lane's eight k-slots are the points
          // 4 g + j and 16 + 4 g + j (the two transposed reads of operand16), the same assignment for both operands
          bf16x4 am[2][4], bn[2][4];
          f32x4 none = zero4;
          int iv = i, gv = g;
          asm volatile("" : "+v"(iv), "+v"(gv));
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            operand16(zimg, true, rg, iv, gv, t2, am[t2], cg == 0, dbacc);
            operand16(aimg, true, cg, iv, gv, t2, bn[t2], false, none);
          }
          bf16x8w a8[4], b8[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            a8[m] = __builtin_shufflevector(am[0][m], am[1][m], 0, 1, 2, 3, 4, 5, 6, 7);
            b8[m] = __builtin_shufflevector(bn[0][m], bn[1][m], 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 4; ++n2)
              acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[m], b8[n2], acc[m][n2], 0, 0, 0);
        }
      }
      t0 = t1;  // (the loop below has nothing left)
    }
  }
  int cur = 1;
  for (long t = t0; t < t1; t += tstep) {
    cur ^= 1;
    __syncthreads();  // vmcnt(0): tile t has landed; everyone is done with the other buffer
    if (t + tstep < t1) step_dma(t + tstep, lds + (cur ^ 1) * 2 * kOp);
    [[maybe_unused]] const unsigned boff = cur * 2 * kOp * 4;
    if constexpr (BF16) {
      const int nsub = (dual && t + tstride < t1) ? 2 : 1;
#pragma unroll 1
      for (int u = 0; act && u < nsub; ++u) {
        const float* zimg = lds + cur * 2 * kOp + u * (kOp / 2);
        const float* aimg = zimg + kOp;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          // points 16 t2 + 4 g + j, j < 4: chunk c of row r lives at chunk position c ^ (r & 15)
          // (two 16-point v_mfma_f32_16x16x16_bf16 steps here: the 32-point form of the four-slot loop above costs this
          // mixed-format path 7 spilled registers inside the tile loop, 2.04 -> 2.30 ms on the config-3 decoder launch)
          bf16x4 am[4], bn[4];
          f32x4 none = zero4;
          // (the fragment addresses are recomputed per tile from an opaque copy of the lane ids: hoisted out of the
          // loop they spill, and a spill reload waits on vmcnt -- i.e. on the tile DMA that was just issued)
          int iv = i, gv = g;
          asm volatile("" : "+v"(iv), "+v"(gv));
          operand16(zimg, z16, rg, iv, gv, t2, am, cg == 0, dbacc);
          operand16(aimg, a16, cg, iv, gv, t2, bn, false, none);
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 4; ++n2)
              acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am[m], bn[n2], acc[m][n2], 0, 0, 0);
        }
      }
    } else if (act) {
      // 8 steps of 4 points; the fragments of step s+1 are read (inline asm, pinned) before the 16
      // MFMAs of step s issue: hipcc otherwise reads them right before use and exposes the LDS
      // latency 8 times per tile
      f32x4 fa[2], fb[2];
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3"
                   : "=&v"(fa[0]), "=&v"(fb[0])
                   : "v"(za_l[0] + boff), "v"(ab_l[0] + boff));
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int c = s & 1, n = c ^ 1;
        if (s + 1 < 8) {
          asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %5 offset:%6"
                       : "=&v"(fa[n]), "=&v"(fb[n]), "+v"(fa[c]), "+v"(fb[c])
                       : "v"(za_l[(s + 1) & 3] + boff), "v"(ab_l[(s + 1) & 3] + boff), "n"(((s + 1) >> 2) * 256));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[c]), "+v"(fb[c]));
        }
        if (cg == 0) dbacc += fa[c];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n2 = 0; n2 < 4; ++n2)
            acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][m], fb[c][n2], acc[m][n2], 0, 0, 0);
      }
    }
  }

  // ---- write out --------------------------------------------------------------------
  // acc[m][n][e] on lane (i, g) = D[row 4*(16rg + 4g + e) + m][col 4*(16cg + i) + n]
  // (fp32 tile format: the columns n = 0..3 of a lane are 4 consecutive features, one 16-byte store; PT16 format:
  // features 4 apart, scalar stores -- once per workgroup)
  if (act) {
    int col[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) col[n] = wg_feature(a16, cg, i, n);
    if (job.per_task) {
      // PT32 tensor, points = row index (n of dW), features = column index (k); ldo: features per tile of that tensor
      const int Ko = job.ldo > 0 ? job.ldo : Kp;
      float* out = job.dW + (size_t)split * ((size_t)(Np >> 5) * Ko * 32);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = wg_feature(z16, rg, 4 * g + e, m);
          if (row >= Np) continue;
          float* rbase = out + (size_t)(row >> 5) * (Ko >> 2) * 128 + (row & 31) * 4;
          if (!a16) {
            if (col[0] < Kp) {
              f32x4 v;
#pragma unroll
              for (int n = 0; n < 4; ++n) v[n] = acc[m][n][e];
              float* dst = rbase + (size_t)(col[0] >> 2) * 128;
              if (job.accumulate & NPF_WGRAD_ACCUMULATE) v += *(const f32x4*)dst;
              *(f32x4*)dst = v;
            }
          } else {
#pragma unroll
            for (int n = 0; n < 4; ++n)
              if (col[n] < Kp) {
                float* dst = rbase + (size_t)(col[n] >> 2) * 128 + (col[n] & 3);
                *dst = (job.accumulate & NPF_WGRAD_ACCUMULATE) ? *dst + acc[m][n][e] : acc[m][n][e];
              }
          }
        }
    } else {
      float* part = partials + J.part_off[j] + (size_t)split * ((size_t)Np * Kp + Np);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = wg_feature(z16, rg, 4 * g + e, m);
          if (row >= Np) continue;
          if (!a16) {
            if (col[0] < Kp) {
              f32x4 v;
#pragma unroll
              for (int n = 0; n < 4; ++n) v[n] = acc[m][n][e];
              *(f32x4*)(part + (size_t)row * Kp + col[0]) = v;
            }
          } else {
#pragma unroll
            for (int n = 0; n < 4; ++n)
              if (col[n] < Kp) part[(size_t)row * Kp + col[n]] = acc[m][n][e];
          }
        }
      if (cg == 0) {
        // dbacc[m] on lane (i, g): sum over the points = g (mod 4) of the lane's feature of operand m
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          dbacc[m] += __shfl_xor(dbacc[m], 16);
          dbacc[m] += __shfl_xor(dbacc[m], 32);
        }
        if (g == 0) {
          if (!z16) {
            if (wg_feature(false, rg, i, 0) < Np) *(f32x4*)(part + (size_t)Np * Kp + wg_feature(false, rg, i, 0)) = dbacc;
          } else {
#pragma unroll
            for (int m = 0; m < 4; ++m)
              if (wg_feature(true, rg, i, m) < Np) part[(size_t)Np * Kp + wg_feature(true, rg, i, m)] = dbacc[m];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 contraction on the bf16 matrix pipe (NPF_WGRAD_F32X6).  gfx950 multiplies fp32 at 1/16 of its bf16 rate
// (v_mfma_f32_16x16x4_f32: 32 cycles for 2 KFLOP; v_mfma_f32_16x16x32_bf16: 16 cycles for 16 KFLOP), so an fp32
// product is cheaper as SIX bf16 products: every fp32 operand is split EXACTLY into three bf16 terms
//   x = x0 + x1 + x2 (+ r, |r| <= 2^-27 |x|):  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)
// (round-to-nearest-even each time; the two subtractions are exact in fp32), and of the nine cross products of
// a * b the six with i + j <= 2 are accumulated in fp32 by the MFMA (a bf16 x bf16 product is exact in fp32); the
// dropped ones are below 2^-26 |a b|, under the fp32 rounding of the product itself.  Same contraction, same fp32
// PT32 operands in HBM, same result to fp32 summation-order noise (measured: closer to float64 than the fp32 MFMA
// kernel) -- at 6/16 of the matrix-pipe time.
//
// Front end = the fp32 kernel's: tiles by LDS-DMA into a double-buffered swizzled fp32 image, one barrier per tile.
// Workgroup = 8 waves (2 per SIMD, up to 256 registers each), wave (rg, cp) owns the 64 x 128 block of dW made of row
// block rg and column blocks 2 cp, 2 cp + 1 (128 accumulator registers).  The split happens in registers at the
// fragment read.  dZ side: the lane's quad row at its 8 points (4 g + j, 16 + 4 g + j = the k-slots of
// v_mfma_f32_16x16x32_bf16; 8 x ds_read_b128) once per tile, all four operands m split (48 registers).  A side: per
// step (column block, operand n2) one feature at the 8 points (8 x ds_read_b32), read two steps ahead, split one step
// ahead with its ~50 vector instructions dealt two per MFMA between the 24 MFMAs of the running step; the next
// tile's DMA pieces go out one per step.
// What was measured on the way (7 jobs of 256 x 256 over 262 144 points; fp32 kernel 1.85 ms, HBM floor 0.64 ms):
// a version that split each tile ONCE into three bf16 images in LDS (96 KiB, single-buffered: the 160 KiB do not hold
// two sets) needed two barriers and a vector-only phase per tile: 1.55 ms; splitting at the read with the compiler's
// lowering of __builtin_convertvector (16 instructions per pair) 1.58, hand-written (11) 1.47, interleaved with the
// MFMAs and with the DMA issue spread 1.38.  With zeros in LDS (no DMA) the same code runs 1.05: the rest is the
// clock the chip holds with real operands on the bf16 pipe.  Narrow jobs (one column block with data) still run all eight
// steps of a wave: a pipeline per column block (four steps each, no wasted MFMAs: 256 x 4 over 262 144 points 0.186 ->
// 0.150 ms) costs the square jobs a pipeline fill per tile (1.38 -> 1.46 ms), and a shortened second copy of the
// eight-step loop spilled 60 - 100 registers however it was written; not kept.
constexpr int kXThreads = 512;
constexpr int kXWaves = kXThreads / 64;

typedef float f32x8v __attribute__((ext_vector_type(8)));
struct X6Terms { bf16x8w t[3]; };

// Eleven vector instructions per pair of values (hipcc's lowering of the same arithmetic written with
// __builtin_convertvector takes sixteen: it converts element by element): RNE to bf16 (packed), both halves back to
// fp32 (shift / mask), exact remainders, twice.
__device__ __forceinline__ unsigned x6_cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ X6Terms x6_split(f32x8v v) {
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  u32x4v t0, t1, t2;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    const unsigned h = x6_cvt_pk(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = x6_cvt_pk(ra, rb);
    const float la = ra - __builtin_bit_cast(float, m << 16), lb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    t0[p] = h;
    t1[p] = m;
    t2[p] = x6_cvt_pk(la, lb);
  }
  X6Terms o;
  o.t[0] = __builtin_bit_cast(bf16x8w, t0);
  o.t[1] = __builtin_bit_cast(bf16x8w, t1);
  o.t[2] = __builtin_bit_cast(bf16x8w, t2);
  return o;
}

struct X6Raw { f32x4 q[8]; };  // quad row of the lane at its 8 points: q[4 h + j] = point 16 h + 4 g + j
__device__ __forceinline__ void x6_issue(const unsigned (&ad)[4], X6Raw& r) {
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
               "ds_read_b128 %4, %8 offset:256\n\tds_read_b128 %5, %9 offset:256\n\t"
               "ds_read_b128 %6, %10 offset:256\n\tds_read_b128 %7, %11 offset:256"
               : "=&v"(r.q[0]), "=&v"(r.q[1]), "=&v"(r.q[2]), "=&v"(r.q[3]), "=&v"(r.q[4]), "=&v"(r.q[5]), "=&v"(r.q[6]),
                 "=&v"(r.q[7])
               : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]));
}
// One feature of the lane's quad row at its 8 points (ds_read_b32 x 8; ad[j] already carries the feature's byte offset;
// SEL: column block + SEL).  The feature a lane takes at step n2 is n2 ^ 2 ((i >> 3) & 1): lanes i and i + 8 would hit
// the same banks with the same feature (4-way conflict with the g / g + 1 pair; this leaves 2-way).
struct X6Feat { float v[8]; };
template <int SEL>
__device__ __forceinline__ void x6_issue_feat(const unsigned (&ad)[4], X6Feat& r) {
  asm volatile("ds_read_b32 %0, %8 offset:%12\n\tds_read_b32 %1, %9 offset:%12\n\t"
               "ds_read_b32 %2, %10 offset:%12\n\tds_read_b32 %3, %11 offset:%12\n\t"
               "ds_read_b32 %4, %8 offset:%13\n\tds_read_b32 %5, %9 offset:%13\n\t"
               "ds_read_b32 %6, %10 offset:%13\n\tds_read_b32 %7, %11 offset:%13"
               : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]),
                 "=&v"(r.v[7])
               : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "n"(SEL * 8192), "n"(SEL * 8192 + 256));
}
// wait until at most N LDS operations of this wave are outstanding (they retire in order)
template <int N>
__device__ __forceinline__ void x6_wait_feat(X6Feat& r) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]), "+v"(r.v[7])
               : "n"(N));
}
__device__ __forceinline__ X6Terms x6_terms_of(const X6Feat& r) {
  return x6_split(f32x8v{r.v[0], r.v[1], r.v[2], r.v[3], r.v[4], r.v[5], r.v[6], r.v[7]});
}
__device__ __forceinline__ void x6_wait(X6Raw& r) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(r.q[0]), "+v"(r.q[1]), "+v"(r.q[2]), "+v"(r.q[3]), "+v"(r.q[4]), "+v"(r.q[5]), "+v"(r.q[6]), "+v"(r.q[7]));
}
__device__ __forceinline__ X6Terms x6_terms_of(const X6Raw& r, int f) {
  const f32x8v v = {r.q[0][f], r.q[1][f], r.q[2][f], r.q[3][f], r.q[4][f], r.q[5][f], r.q[6][f], r.q[7][f]};
  return x6_split(v);
}

__global__ __launch_bounds__(kXThreads, 1) void wgrad_x6_kernel(const WgradJobs J, float* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kQRows * 128];  // 2 buffers x (dZ, A), as wgrad_kernel
  constexpr int kOp = kQRows * 128;

  int j = 0;
  while (j + 1 < J.n_jobs && (int)blockIdx.x >= J.first_wg[j + 1]) ++j;
  const npf_wgrad_job_t& job = J.job[j];
  const int split = blockIdx.x - J.first_wg[j];
  const int n_split = J.first_wg[j + 1] - J.first_wg[j];
  const int Np = ((job.N + 31) >> 5) * 32, Kp = ((job.K + 31) >> 5) * 32;
  const long total_tiles = (long)J.n_tasks * J.tiles_per_task;
  long t0, t1;
  int tstride = 1;
  if (job.per_task) {
    t0 = (long)split * J.tiles_per_task;
    t1 = t0 + J.tiles_per_task;
  } else {
    t0 = split;
    t1 = total_tiles;
    tstride = n_split;
  }

  const int tid = threadIdx.x, lane = tid & 63;
  const int i = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave >> 1, cp = wave & 1;
  const bool act_r = 64 * rg < Np && 128 * cp < Kp;
  const int n_steps = !act_r ? 0 : (64 * (2 * cp + 1) < Kp ? 8 : 4);  // (column block, operand) steps with data

  for (int x = tid; x < 2 * 2 * kOp; x += kXThreads) lds[x] = 0.f;  // quad rows beyond Np / 4, Kp / 4 stay zero

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[2][4][4];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[c][m][n] = zero4;
  f32x4 dbacc = zero4;

  // DMA as in wgrad_kernel, pieces dealt to 8 waves (q = wave + 8 n: the swizzle term (2 q + h) & 15 = (2 wave + h) & 15)
  const unsigned dma_lane = 4u * (unsigned)((lane >> 5) * 128 + (((lane & 31) ^ ((2 * wave + (lane >> 5)) & 15)) << 2));
  const char* dz_base = (const char*)job.dZ;
  const char* a_base = (const char*)job.A;
  asm volatile("" : "+s"(dz_base), "+s"(a_base));
  const int Zf = job.ldz > 0 ? job.ldz : Np, Af = job.lda > 0 ? job.lda : Kp;
  auto tile_dma = [&](long t, float* buf) {
    const char* zsrc = dz_base + (size_t)t * Zf * 128;
    const char* asrc = a_base + (size_t)t * Af * 128;
    for (int q = wave; q < (Np >> 3); q += kXWaves)
      wg_dma16((const float*)(zsrc + (size_t)q * 1024 + (size_t)dma_lane), buf + q * 256);
    for (int q = wave; q < (Kp >> 3); q += kXWaves)
      wg_dma16((const float*)(asrc + (size_t)q * 1024 + (size_t)dma_lane), buf + kOp + q * 256);
  };

  __syncthreads();
  if (t0 < t1) tile_dma(t0, lds);

  int cur = 1;
  for (long t = t0; t < t1; t += tstride) {
    cur ^= 1;
    __syncthreads();  // vmcnt(0): tile t has landed; everyone is done with the other buffer
    // the next tile's DMA: a wave without work issues its (up to eight) pieces here, the others one per step below
    // (an LDS-DMA instruction costs its wave 60 - 180 cycles of issue: eight in a row ahead of the tile's first MFMA
    // showed up one for one in the tile time)
    const bool dma_next = t + tstride < t1;
    float* nbuf = lds + (cur ^ 1) * 2 * kOp;
    if (n_steps == 0) {
      if (dma_next) tile_dma(t + tstride, nbuf);
      continue;
    }
    const char* zsrc_n = dz_base + (size_t)(t + tstride) * Zf * 128 + (size_t)dma_lane;
    const char* asrc_n = a_base + (size_t)(t + tstride) * Af * 128 + (size_t)dma_lane;
    // fragment addresses (per tile, from an opaque copy of the lane ids: hoisted out of the loop they would be spilled,
    // and a spill reload waits on vmcnt, i.e. on the tile DMA just issued): chunk (4 g + j) ^ i of the lane's quad row;
    // the points 16 + .. are 256 bytes further, column block 2 cp + 1 is 8 KiB further
    int iv = i, gv = g;
    asm volatile("" : "+v"(iv), "+v"(gv));
    unsigned za[4], zb[4];
    {
      const unsigned l0 = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)lds + cur * 2 * kOp * 4;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const unsigned lanepart = (unsigned)(((4 * gv + jj) ^ iv) << 4);
        za[jj] = l0 + (16 * rg + iv) * 512 + lanepart;
        zb[jj] = l0 + kOp * 4 + (32 * cp + iv) * 512 + lanepart;
      }
    }
    X6Raw ra;
    x6_issue(za, ra);
    x6_wait(ra);
    if (cp == 0) dbacc += ((ra.q[0] + ra.q[1]) + (ra.q[2] + ra.q[3])) + ((ra.q[4] + ra.q[5]) + (ra.q[6] + ra.q[7]));
    X6Terms fa[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) fa[m] = x6_terms_of(ra, m);
    // A side: per step k = (column block c = k >> 2, operand n2 = k & 3) one feature of the lane's quad row, n2 ^ pi with
    // pi = 2 ((i >> 3) & 1), read TWO steps ahead (8 registers per step in flight) and split ONE step ahead, its ~50
    // vector instructions dealt between the 24 MFMAs of the running step (sched_group_barrier)
    const unsigned pi4 = (unsigned)((iv >> 3) & 1) << 3;  // 4 * pi
    auto feat_addr = [&](int n2, unsigned (&ad)[4]) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) ad[jj] = zb[jj] + ((4u * (unsigned)n2) ^ pi4);
    };
    X6Feat raw[2];
    {
      unsigned ad[4];
      feat_addr(0, ad);
      x6_issue_feat<0>(ad, raw[0]);
      feat_addr(1, ad);
      x6_issue_feat<0>(ad, raw[1]);
    }
    x6_wait_feat<8>(raw[0]);
    X6Terms fb = x6_terms_of(raw[0]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = k >> 2, n2 = k & 3;
      if (dma_next) {  // piece wave + 8 (k & 3) of dZ (steps 0..3) / of A (steps 4..7)
        const int q = wave + kXWaves * (k & 3);
        if (k < 4) {
          if (q < (Np >> 3)) wg_dma16((const float*)(zsrc_n + (size_t)q * 1024), nbuf + q * 256);
        } else {
          if (q < (Kp >> 3)) wg_dma16((const float*)(asrc_n + (size_t)q * 1024), nbuf + kOp + q * 256);
        }
      }
      if (k + 2 < 8) {
        unsigned ad[4];
        feat_addr((k + 2) & 3, ad);
        if (k + 2 < 4) x6_issue_feat<0>(ad, raw[k & 1]);
        else x6_issue_feat<1>(ad, raw[k & 1]);
      }
      X6Terms fb_next;
      if (k + 1 < 8) {
        if (k + 2 < 8) x6_wait_feat<8>(raw[(k + 1) & 1]);
        else x6_wait_feat<0>(raw[(k + 1) & 1]);
        fb_next = x6_terms_of(raw[(k + 1) & 1]);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        f32x4 a = acc[c][m][n2];
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m].t[2], fb.t[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m].t[0], fb.t[2], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m].t[1], fb.t[1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m].t[1], fb.t[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m].t[0], fb.t[1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m].t[0], fb.t[0], a, 0, 0, 0);
        acc[c][m][n2] = a;
      }
      if (k + 1 < 8) {
#pragma unroll
        for (int q = 0; q < 24; ++q) {  // one MFMA, two vector instructions, ...
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        fb = fb_next;
      }
    }
  }

  // ---- write out: acc[c][m][n][e] on lane (i, g) = D[row 4 (16 rg + 4 g + e) + m][col 4 (16 (2 cp + c) + i) + n] ----
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (4 * c >= n_steps) continue;
    const int cg = 2 * cp + c;
    const int col0 = 4 * (16 * cg + i);
    if (job.per_task) {
      const int Ko = job.ldo > 0 ? job.ldo : Kp;
      float* out = job.dW + (size_t)split * ((size_t)(Np >> 5) * Ko * 32);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 4 * (16 * rg + 4 * g + e) + m;
          if (row >= Np || col0 >= Kp) continue;
          f32x4 v;
#pragma unroll
          for (int n = 0; n < 4; ++n) v[n] = (i & 8) ? acc[c][m][n ^ 2][e] : acc[c][m][n][e];  // (lanes i >= 8: feature n2 ^ 2)
          float* dst = out + (size_t)(row >> 5) * (Ko >> 2) * 128 + (row & 31) * 4 + (size_t)(col0 >> 2) * 128;
          if (job.accumulate & NPF_WGRAD_ACCUMULATE) v += *(const f32x4*)dst;
          *(f32x4*)dst = v;
        }
    } else {
      float* part = partials + J.part_off[j] + (size_t)split * ((size_t)Np * Kp + Np);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 4 * (16 * rg + 4 * g + e) + m;
          if (row >= Np || col0 >= Kp) continue;
          f32x4 v;
#pragma unroll
          for (int n = 0; n < 4; ++n) v[n] = (i & 8) ? acc[c][m][n ^ 2][e] : acc[c][m][n][e];
          *(f32x4*)(part + (size_t)row * Kp + col0) = v;
        }
    }
  }
  if (!job.per_task && n_steps > 0 && cp == 0) {
    // dbacc[m] on lane (i, g): sum over the lane's points of feature 4 (16 rg + i) + m
    float* part = partials + J.part_off[j] + (size_t)split * ((size_t)Np * Kp + Np);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      dbacc[m] += __shfl_xor(dbacc[m], 16);
      dbacc[m] += __shfl_xor(dbacc[m], 32);
    }
    if (g == 0 && 4 * (16 * rg + i) < Np) *(f32x4*)(part + (size_t)Np * Kp + 4 * (16 * rg + i)) = dbacc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same contraction for 256 x 256 jobs with every operand value split ONCE per workgroup (round 3).  wgrad_x6_kernel is
// vector-issue-bound: 4.1 vector instructions per MFMA (PMC), because each wave splits the fragments it reads itself -- the A-side
// feature by the four row-block waves that use it, the dZ side by both column-pair waves: three times the unique work.  Here a
// tile goes in HALVES of 16 points: the workgroup's 512 lanes load the half tile's 2 x 4096 fp32 values from HBM into registers
// (four 16-byte loads per lane, issued a half tile ahead), each lane splits ITS 16 values (88 vector instructions) and writes the
// three terms as bf16 rows [point][feature] into LDS (12 x ds_write_b64); the matrix phase reads its operands with the
// hardware-transposing ds_read_b64_tr_b16 (4 points x 16 features per 16-lane group -> "lane = feature, elements = 4 points") and
// runs v_mfma_f32_32x32x16_bf16 (K = the half tile's 16 points): wave (rg, cp) owns rows 64 rg.. (dZ features) x columns
// 128 cp.. (A features) = 2 x 4 tiles of 32 x 32, 48 MFMAs of 32 cycles per half tile beside 88 + ~40 vector instructions.
// Term images are double-buffered (2 x 54 KiB): the split of half tile h + 1 rides inside the matrix phase of h; one barrier
// per half tile.  Measured (config 2, target-side launch: nine 256 x 256 jobs over 262 144 points + its narrow jobs and reduces):
// 1.79 ms with wgrad_x6_kernel -> 1.50 ms; a first version without the software pipeline (split, then reads, then MFMAs) 1.92.
// The transposed reads are the clang builtin (__builtin_amdgcn_ds_read_tr16_b64_v4i16: hipcc places the waits and builds the
// 8-element operands without register copies; as inline asm the two halves of an operand cost four v_mov each).  NPF_WGRAD_NO_H16 on a job
// keeps wgrad_x6_kernel.  LDS rows are 576 bytes apart: the transposed reads (rows q, q + 1 .. of two 16-feature blocks per 32-lane
// half) and the row writes (lanes dealt as 4 points x 4 feature quads per 16 lanes) are both bank-conflict free.
constexpr int kHRow = 576;                       // bytes between the rows (points) of a term image
constexpr int kHTerm = 16 * kHRow;               // one term of one operand
constexpr int kHBuf = 2 * 3 * kHTerm;            // (dZ, A) x three terms
typedef unsigned h16_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned h16_u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 1) void wgrad_h16_kernel(const WgradJobs J, float* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) char lds[2 * kHBuf];
  int j = 0;
  while (j + 1 < J.n_jobs && (int)blockIdx.x >= J.first_wg[j + 1]) ++j;
  const npf_wgrad_job_t& job = J.job[j];
  const int split = blockIdx.x - J.first_wg[j];
  const int n_split = J.first_wg[j + 1] - J.first_wg[j];
  constexpr int Np = 256, Kp = 256;
  const long total_tiles = (long)J.n_tasks * J.tiles_per_task;
  long t0, t1;
  int tstride = 1;
  if (job.per_task) {
    t0 = (long)split * J.tiles_per_task;
    t1 = t0 + J.tiles_per_task;
  } else {
    t0 = split;
    t1 = total_tiles;
    tstride = n_split;
  }
  const long n_half = t1 > t0 ? 2 * ((t1 - t0 + tstride - 1) / tstride) : 0;  // half tiles of this workgroup

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave >> 1, cp = wave & 1;
  // loads: lane = (p_lo, q, p_hi): point 4 p_hi + p_lo of the half tile, feature quad 8 wave + 4 jj + q (jj = 0, 1)
  const int p_lo = lane & 3, q_ld = (lane >> 2) & 3, p_hi = lane >> 4;
  const int pt_ld = 4 * p_hi + p_lo;
  const int Zf = job.ldz > 0 ? job.ldz : Np, Af = job.lda > 0 ? job.lda : Kp;
  const char* dz_base = (const char*)job.dZ;
  const char* a_base = (const char*)job.A;
  const unsigned ld_lane = (unsigned)(((8 * wave + q_ld) * 32 + pt_ld) * 16);  // byte offset inside the tile (jj = 0, half 0)
  auto tile_of = [&](long h) { return t0 + (h >> 1) * tstride; };
  // piece v of half tile h: v = 0, 1: dZ quads f4, f4 + 4; v = 2, 3: A
  auto load_piece = [&](long h, int v) -> f32x4 {
    const long t = tile_of(h);
    const unsigned off = ld_lane + (unsigned)(h & 1) * 256u + (unsigned)(v & 1) * 2048u;
    const char* src = (v < 2 ? dz_base + (size_t)t * Zf * 128 : a_base + (size_t)t * Af * 128) + off;
    return *(const f32x4*)src;
  };
  // term rows in LDS: point pt at pt * kHRow, feature quad f4 at + 8 f4
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned st_lane = (unsigned)(pt_ld * kHRow + (8 * wave + q_ld) * 8);
  // split one float4 (v = 0, 1: dZ quads f4, f4 + 4; v = 2, 3: A) into its three terms and write them as bf16 rows
  auto split_store1 = [&](const f32x4& x, int v, unsigned buf) {
    h16_u32x2 t[3];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const float a = x[2 * pr], b = x[2 * pr + 1];
      const unsigned h = x6_cvt_pk(a, b);
      const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
      const unsigned m = x6_cvt_pk(ra, rb);
      const float la = ra - __builtin_bit_cast(float, m << 16), lb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
      t[0][pr] = h;
      t[1][pr] = m;
      t[2][pr] = x6_cvt_pk(la, lb);
    }
    const unsigned ad = lds0 + buf * kHBuf + (v >> 1) * 3 * kHTerm + st_lane + (v & 1) * 32;
    asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:%4\n\tds_write_b64 %0, %3 offset:%5"
                 :
                 : "v"(ad), "v"(t[0]), "v"(t[1]), "v"(t[2]), "n"(kHTerm), "n"(2 * kHTerm)
                 : "memory");
  };
  // transposed operand reads: 16-lane group gq = lane >> 4: feature block (gq & 1), points 8 (gq >> 1) + 4 rd + q; lane 4 q + p
  // of the group supplies row q, columns 4 p .. 4 p + 3
  const int gq = lane >> 4, q_rd = (lane >> 2) & 3, p_rd = lane & 3;
  const unsigned rd_lane = (unsigned)((8 * (gq >> 1) + q_rd) * kHRow + (16 * (gq & 1) + 4 * p_rd) * 2);
  typedef short h16_s16x4 __attribute__((ext_vector_type(4)));
  typedef short h16_s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) h16_s16x4 h16_lds_s16x4;
  struct Frag { h16_s16x8 v; };
  auto frag = [&](unsigned buf, int operand, int term, int feat0, Frag& f) {  // 32 features from feat0, the 16 points
    const char* ad = lds + buf * kHBuf + (operand * 3 + term) * kHTerm + rd_lane + (unsigned)feat0 * 2;
    const h16_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((h16_lds_s16x4*)ad);
    const h16_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((h16_lds_s16x4*)(ad + 4 * kHRow));
    f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto op8 = [](const Frag& f) { return __builtin_bit_cast(bf16x8w, f.v); };

  f32x16v acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;
  f32x4 dbacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

  // One half tile: the matrix phase on buffer h & 1 with, riding between its MFMAs, the split of half tile h + 1 (in ``raw``,
  // loaded a half tile ago) into the other buffer -- one float4 per column tile, its registers reloaded at once with the same
  // piece of half tile h + 2 (a whole phase to land).  Past the end the pieces are stale: split and written all the same (the
  // other buffer is not read any more; no branch inside the phase: the scheduler interleaves one basic block).
  f32x4 raw[4];
  auto body = [&](long h) {
    const unsigned buf = (unsigned)(h & 1);
    const float live = h + 1 < n_half ? 1.f : 0.f;
    const long h2 = h + 2 < n_half ? h + 2 : (n_half > 0 ? n_half - 1 : 0);  // (clamped: a valid address either way)
    Frag fa[2][3], fb[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 3; ++t) frag(buf, 0, t, 64 * rg + 32 * i, fa[i][t]);
#pragma unroll
    for (int t = 0; t < 3; ++t) frag(buf, 1, t, 128 * cp, fb[0][t]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cb = c & 1;
      if (c + 1 < 4) {
#pragma unroll
        for (int t = 0; t < 3; ++t) frag(buf, 1, t, 128 * cp + 32 * (c + 1), fb[cb ^ 1][t]);
      }
#define HMM(I, TA, TB) \
  acc[I][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op8(fa[I][TA]), op8(fb[cb][TB]), acc[I][c], 0, 0, 0)
      HMM(0, 2, 0);
      HMM(1, 2, 0);
      HMM(0, 0, 2);
      HMM(1, 0, 2);
      HMM(0, 1, 1);
      HMM(1, 1, 1);
      HMM(0, 1, 0);
      HMM(1, 1, 0);
      HMM(0, 0, 1);
      HMM(1, 0, 1);
      HMM(0, 0, 0);
      HMM(1, 0, 0);
#undef HMM
      if (c < 2) dbacc[c] += raw[c] * live;
      split_store1(raw[c], c, buf ^ 1);
      raw[c] = load_piece(h2, c);
      // one MFMA, two vector instructions, ... (the split's 22 + the bias sum's 4 between the 12 MFMAs of this column tile)
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
    }
    __syncthreads();  // everyone is done with this buffer's reads and with the other buffer's writes
  };

  if (n_half > 0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) raw[v] = load_piece(0, v);
    dbacc[0] += raw[0];
    dbacc[1] += raw[1];
#pragma unroll
    for (int v = 0; v < 4; ++v) split_store1(raw[v], v, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) raw[v] = load_piece(n_half > 1 ? 1 : 0, v);
  }
  __syncthreads();
  for (long h = 0; h < n_half; ++h) body(h);

  // ---- write out: acc[i][c][r] on lane l = D[row 64 rg + 32 i + (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col 128 cp + 32 c + (l & 31)]
  const int col_l = lane & 31, row_l = 4 * (lane >> 5);
  if (job.per_task) {
    const int Ko = job.ldo > 0 ? job.ldo : Kp;
    float* out = job.dW + (size_t)split * ((size_t)(Np >> 5) * Ko * 32);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 64 * rg + 32 * i + (r & 3) + 8 * (r >> 2) + row_l, col = 128 * cp + 32 * c + col_l;
          float* dst = out + (size_t)(row >> 5) * (Ko >> 2) * 128 + (row & 31) * 4 + (size_t)(col >> 2) * 128 + (col & 3);
          *dst = (job.accumulate & NPF_WGRAD_ACCUMULATE) ? *dst + acc[i][c][r] : acc[i][c][r];
        }
  } else {
    float* part = partials + J.part_off[j] + (size_t)split * ((size_t)Np * Kp + Np);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 64 * rg + 32 * i + (r & 3) + 8 * (r >> 2) + row_l, col = 128 * cp + 32 * c + col_l;
          part[(size_t)row * Kp + col] = acc[i][c][r];
        }
    // bias gradient: this lane's dZ quads 8 wave + q_ld (+ 4) summed over its points; the quad's 16 lanes differ in p_lo, p_hi
#pragma unroll
    for (int v = 0; v < 2; ++v) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = dbacc[v][e];
        x += __shfl_xor(x, 1);
        x += __shfl_xor(x, 2);
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        dbacc[v][e] = x;
      }
      if (pt_ld == 0) *(f32x4*)(part + (size_t)Np * Kp + 4 * (8 * wave + 4 * v + q_ld)) = dbacc[v];
    }
  }
}

// dW[n][k] (+)= sum_s partial[s][n][k];  db[n] (+)= sum_s partial_db[s][n].
// One float4 of the slab per thread, four independent accumulators over the splits (the sum is
// latency-bound: each term is a separate 16-byte load from a different slab).
__global__ void wgrad_reduce_kernel(const WgradJobs J, const float* __restrict__ partials) {
  const int j = blockIdx.y;
  const npf_wgrad_job_t& job = J.job[j];
  if (job.per_task) return;
  const int n_split = J.first_wg[j + 1] - J.first_wg[j];
  const int Np = ((job.N + 31) >> 5) * 32, Kp = ((job.K + 31) >> 5) * 32;
  const size_t slab = (size_t)Np * Kp + Np;
  const float* part = partials + J.part_off[j];
  const int total4 = (Np * Kp + Np) >> 2;
  for (int i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += gridDim.x * blockDim.x) {
    const float* p = part + (size_t)i4 * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int sp = 0;
    for (; sp + 4 <= n_split; sp += 4) {
      s0 += *(const f32x4*)(p + (size_t)sp * slab);
      s1 += *(const f32x4*)(p + (size_t)(sp + 1) * slab);
      s2 += *(const f32x4*)(p + (size_t)(sp + 2) * slab);
      s3 += *(const f32x4*)(p + (size_t)(sp + 3) * slab);
    }
    for (; sp < n_split; ++sp) s0 += *(const f32x4*)(p + (size_t)sp * slab);
    const f32x4 s = (s0 + s1) + (s2 + s3);
    const int idx = i4 * 4;
    if (idx < Np * Kp) {
      const int n = idx / Kp, k = idx - n * Kp;  // Kp % 4 == 0: the float4 stays inside row n
      if (n < job.N) {
        float* d = job.dW + (size_t)n * job.ldw + k;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < job.K) d[e] = (job.accumulate & NPF_WGRAD_ACCUMULATE) ? d[e] + s[e] : s[e];
      }
    } else if (job.db) {
      const int n = idx - Np * Kp;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < job.N) job.db[n + e] = (job.accumulate & NPF_WGRAD_ACCUMULATE) ? job.db[n + e] + s[e] : s[e];
    }
  }
}

static int plan(const npf_wgrad_job_t* jobs, int n_jobs, int n_tasks, int tiles_per_task, WgradJobs* J) {
  if (!jobs || n_jobs <= 0 || n_jobs > kMaxJobs || n_tasks <= 0 || tiles_per_task <= 0) return NPF_EINVAL;
  const long total_tiles = (long)n_tasks * tiles_per_task;
  int n_shared = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const npf_wgrad_job_t& b = jobs[j];
    if (!b.dZ || !b.A || !b.dW || b.N <= 0 || b.K <= 0 || b.N > npf::kWgMaxF || b.K > npf::kWgMaxF) return NPF_EINVAL;
    if ((((uintptr_t)b.dZ) | ((uintptr_t)b.A)) & 15) return NPF_EINVAL;
    if (b.per_task && (((uintptr_t)b.dW) & 15)) return NPF_EINVAL;
    if (!b.per_task && b.ldw < b.K) return NPF_EINVAL;
    if (b.ldz < 0 || b.lda < 0 || b.ldo < 0 || (b.ldz & 31) || (b.lda & 31) || (b.ldo & 31)) return NPF_EINVAL;
    if ((b.ldz && b.ldz < npf::round_up(b.N, 32)) || (b.lda && b.lda < npf::round_up(b.K, 32)) ||
        (b.ldo && b.ldo < npf::round_up(b.K, 32)))
      return NPF_EINVAL;
    n_shared += b.per_task ? 0 : 1;
  }
  // workgroups per shared-weight job, proportional to its per-tile time: with the wave mapping
  // of wgrad_kernel a job with min(row blocks, column blocks) = b keeps b waves per SIMD busy
  // (fp32 variant: the MFMAs set the tile time.)  The bf16 variant has 1/8 of the MFMA cycles and its tile
  // time follows the bytes of the tile plus a latency floor (one tile in flight per workgroup): a skinny job
  // costs almost as much per tile as a square one, and priced by its MFMAs it becomes the launch's critical path.
  bool bf16 = true, x6 = true;
  for (int j = 0; j < n_jobs; ++j) {
    bf16 &= (jobs[j].accumulate & NPF_WGRAD_BF16) != 0;
    x6 &= (jobs[j].accumulate & (NPF_WGRAD_F32X6 | NPF_WGRAD_BF16)) == NPF_WGRAD_F32X6;
  }
  double cost[kMaxJobs], cost_sum = 0.0;
  for (int j = 0; j < n_jobs; ++j) {
    const int Np = npf::round_up(jobs[j].N, 32), Kp = npf::round_up(jobs[j].K, 32);
    const int ar = (Np + 63) / 64, ac = (Kp + 63) / 64;
    // cycles per tile: fp32 variant max(MFMA time of the busiest SIMD, tile bytes at ~6.5 B/clk per CU);
    // bf16 variant: tile bytes with a floor (measured: a 4 KiB tile costs ~0.56 of a 32 KiB one)
    const double zb = (jobs[j].accumulate & NPF_WGRAD_DZ16) ? 2.0 : 4.0, ab = (jobs[j].accumulate & NPF_WGRAD_A16) ? 2.0 : 4.0;
    const double bytes = 32.0 * (Np * zb + Kp * ab);
    if (jobs[j].per_task) cost[j] = 0.0;
    else if (bf16) cost[j] = bytes > 18432.0 ? bytes : 18432.0;
    else if (x6) {
      // split kernel: a wave owns a 64 x 128 block and runs 192 MFMAs of 16 cycles per tile beside ~800 vector
      // instructions (measured ~1.6 x the bare MFMA time on real data), two waves per SIMD when the job is square;
      // plus the per-tile prologue (dZ-side reads and split, barrier); or the tile's bytes
      const int waves = ar * ((ac + 1) / 2);
      const double mf = 3072.0 * 1.6 * ((waves + 3) / 4) + 2500.0, mem = bytes / 6.5;
      cost[j] = mf > mem ? mf : mem;
    } else {
      const double mf = 4096.0 * (ar < ac ? ar : ac), mem = bytes / 6.5;
      cost[j] = mf > mem ? mf : mem;
    }
    cost_sum += cost[j];
  }
  J->n_jobs = n_jobs;
  J->n_tasks = n_tasks;
  J->tiles_per_task = tiles_per_task;
  J->pad = 0;
  // floor: the shared-weight jobs of one launch must fit one wave of 256 workgroups (one per CU), a 257th
  // workgroup would wait for a second round; what the floors leave over goes to the jobs one by one
  long splits_of[kMaxJobs];
  long used = 0;
  for (int j = 0; j < n_jobs; ++j) {
    splits_of[j] = 0;
    if (jobs[j].per_task) continue;
    long sp = (long)(256.0 * cost[j] / cost_sum);
    if (sp < 1) sp = 1;
    if (sp > total_tiles) sp = total_tiles;
    splits_of[j] = sp;
    used += sp;
  }
  for (int j = 0; j < n_jobs && used < 256; ++j)
    if (!jobs[j].per_task && cost[j] * n_shared >= cost_sum && splits_of[j] < total_tiles) {  // (the costlier jobs)
      ++splits_of[j];
      ++used;
    }
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
      const long splits = splits_of[j];
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
  bool bf16 = true, x6 = true;
  for (int j = 0; j < n_jobs; ++j) {
    bf16 &= (jobs[j].accumulate & NPF_WGRAD_BF16) != 0;
    x6 &= (jobs[j].accumulate & (NPF_WGRAD_F32X6 | NPF_WGRAD_BF16)) == NPF_WGRAD_F32X6;
  }
  bool h16 = x6;
  for (int j = 0; j < n_jobs; ++j) h16 &= jobs[j].N == 256 && jobs[j].K == 256 && !(jobs[j].accumulate & NPF_WGRAD_NO_H16);
  if (h16)  // 256 x 256 jobs: every operand value split once per workgroup (wgrad_h16_kernel)
    hipLaunchKernelGGL(npf::wgrad_h16_kernel, dim3(n_wg), dim3(512), 0, (hipStream_t)stream, J, partials);
  else if (x6)
    hipLaunchKernelGGL(npf::wgrad_x6_kernel, dim3(n_wg), dim3(npf::kXThreads), 0, (hipStream_t)stream, J, partials);
  else if (bf16)
    hipLaunchKernelGGL(npf::wgrad_kernel<true>, dim3(n_wg), dim3(npf::kWgThreads), 0, (hipStream_t)stream, J, partials);
  else
    hipLaunchKernelGGL(npf::wgrad_kernel<false>, dim3(n_wg), dim3(npf::kWgThreads), 0, (hipStream_t)stream, J, partials);
  NPF_CHECK_LAUNCH();
  if (any_shared) {
    hipLaunchKernelGGL(npf::wgrad_reduce_kernel, dim3(65, n_jobs), dim3(256), 0, (hipStream_t)stream, J,
                       (const float*)partials);
    NPF_CHECK_LAUNCH();
  }
  return NPF_OK;
}
