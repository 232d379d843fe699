// x6 programs (npf_x6_run): whole sides of the model as one launch, every multiply an fp32 product on the bf16 matrix pipe.
//
// What it computes, in the reference's terms: the x-encoder (MLP(dx -> r), npf/architectures/mlp.py:95-109) from the raw
// frequencies, the scaled-dot cross attention over the task's context points (npf/architectures/attention.py:129-164,204-220:
// q K^T / sqrt(d), softmax, attn . V), the decoder (MergeFlatInputs: relu(x1 + resizer(R)) -> MLP, encoders.py:175-183) and
// its 256 -> 4 output layer -- forward in one launch; and the dgrad of all of that in one launch (softmax backward, ReLU masks
// as sign bits, dZ stored for the weight-gradient launch).  The same interpreter runs the context side (x-encoder,
// XY-encoder) and its dgrad.
//
// Arithmetic and layout = csrc/mlp_x6_kernel.hip (which this file generalises): every fp32 operand is split EXACTLY into three
// bf16 terms (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)), six of the nine cross products go through
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation (the dropped ones are below 2^-26 of a product): an fp32 result at 6/16 of the
// v_mfma_f32_16x16x4_f32 time.  A wave owns 16 points (half a PT32 tile) and keeps their F features in registers across the
// whole program (block b / element e of lane (p, g) = feature 16 b + 4 g + e of point p); the accumulators of one multiply ARE
// the B operands of the next.  Weights -- shared three-term images (npf_prepare_weights) or the task's keys / values
// (npf_x6_task_images) -- stream L2 -> LDS by LDS-DMA in slabs of 16 output rows x 3 terms through a three-slot ring that
// runs on ACROSS ops (slab S + 2 in flight while slab S multiplies, one counted s_waitcnt vmcnt + one barrier per slab).
// The softmax of a score row is in registers: a lane holds F / 4 keys of its point, the four lanes of a point meet by two
// wavefront shuffles (ds_bpermute), fp32 with max subtraction like torch.softmax.
#include "npf_common.hpp"

namespace npf {

typedef __bf16 xp_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned xp_u32x4 __attribute__((ext_vector_type(4)));

struct XpArgs {
  npf_x6_op_t op[NPF_X6_MAX_OPS];
  const char* mm_img[NPF_X6_MAX_OPS];  // the multiplies of the program in order (what the slab stream walks)
  int64_t mm_stride[NPF_X6_MAX_OPS];
  const float* out_w;
  const float* out_b;
  float* out_rows;
  int32_t n_ops, n_mm;
  int32_t total_tiles, tiles_per_task;
  int32_t wgs_per_task;  // 0: tiles dealt flat, two per workgroup
  int32_t xcd_remap;     // the workgroups of a task on one XCD (grid a multiple of 8)
};

template <int KF>
struct XpGeom {
  static constexpr int NB = KF / 16;              // 16-feature blocks of an activation = slabs per multiply
  static constexpr int KS = KF / 32;              // k-steps per slab
  static constexpr int RowB = KF * 2;             // bytes of an image row
  static constexpr int TermB = 16 * RowB;         // one term of a slab
  static constexpr int SlabB = 3 * TermB;
  static constexpr int Slots = 3;
  static constexpr int RPP = 1024 / RowB;         // rows per 1 KiB DMA piece
  static constexpr int PPT = 16 / RPP;            // pieces per term
  static constexpr int NP = 3 * PPT / 4;          // pieces per wave and slab
  static constexpr int BiasB = 2 * KF * 4;
};

__device__ __forceinline__ void xp_dma16(const void* src, void* lds_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_uniform, 16, 0, 0);
}

__device__ __forceinline__ unsigned xp_cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// (see x6m_split in mlp_x6_kernel.hip for the edge-value semantics)
__device__ __forceinline__ void xp_split(const f32x4& lo, const f32x4& hi, xp_u32x4& t0, xp_u32x4& t1, xp_u32x4& t2) {
  const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    const unsigned h = xp_cvt_pk(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = xp_cvt_pk(ra, rb);
    const float la = ra - __builtin_bit_cast(float, m << 16), lb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    t0[p] = h;
    t1[p] = m;
    t2[p] = xp_cvt_pk(la, lb);
  }
}

// sum / max over the four lanes (g = 0..3) that share a point
__device__ __forceinline__ float xp_sum4(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float xp_max4(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

template <int KF>
__global__ __launch_bounds__(256, 2) void x6_program_kernel(const XpArgs a) {
  using G = XpGeom<KF>;
  constexpr int NB = G::NB, KS = G::KS;
  __shared__ __attribute__((aligned(16))) char smem[G::Slots * G::SlabB + G::BiasB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  int bid = blockIdx.x;
  if (a.xcd_remap) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
  int task = 0;
  long tile;
  bool valid;
  if (a.wgs_per_task > 0) {
    task = bid / a.wgs_per_task;
    const int t_in = (bid - task * a.wgs_per_task) * 2 + (wave >> 1);
    valid = t_in < a.tiles_per_task;
    tile = (long)task * a.tiles_per_task + t_in;
  } else {
    tile = (long)bid * 2 + (wave >> 1);
    valid = tile < a.total_tiles;
  }
  if (!valid) tile = 0;  // (a wave without a tile still streams slabs and meets barriers; it loads tile 0 and stores nothing)
  // this lane's float4 column in its tile of a PT32 tensor with KF features: block b at + (4 b + g) * 128 floats
  const size_t lane_off = (size_t)tile * (KF * 32) + (size_t)(16 * (wave & 1) + p) * 4 + (size_t)g * 128;
  const size_t bits_off = ((size_t)tile * 2 + (wave & 1)) * 64 + lane;  // [tile][half][64 lanes] uint64
  const size_t row_idx = (size_t)tile * 32 + 16 * (wave & 1) + p;       // this lane's point in a rows tensor

  // DMA of a slab: 3 * PPT pieces of 1 KiB (term q / PPT, rows RPP (q % PPT) ..), NP per wave; the swizzle (chunk c of row r at
  // position c ^ (r & 15)) is applied to the source address: uniform base per piece + a lane offset
  constexpr int LPR = 64 / G::RPP;  // lanes per row of a piece
  unsigned dma_lane[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int r0 = G::RPP * ((wave + 4 * n) % G::PPT), row = r0 + lane / LPR, pos = lane % LPR;
    dma_lane[n] = (unsigned)((lane / LPR) * G::RowB + ((pos ^ (row & 15)) << 4));
  }
  const int n_slabs = a.n_mm * NB;
  auto dma_slab = [&](int S, char* slot) {
    const int j = S / NB;
    const char* base = a.mm_img[j] + (size_t)task * a.mm_stride[j] + (size_t)(S % NB) * 16 * G::RowB;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int n = 0; n < G::NP; ++n) {
      const int q = wave + 4 * n, term = q / G::PPT, r0 = G::RPP * (q % G::PPT);
      xp_dma16(base + (size_t)term * (KF * KF * 2) + r0 * G::RowB + dma_lane[(G::PPT == 8) ? (n & 1) : 0],
               slot + term * G::TermB + r0 * G::RowB);
    }
  };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lane_row = (unsigned)(p * G::RowB);
  float* bias_lds = (float*)(smem + G::Slots * G::SlabB);

  if (n_slabs > 0) dma_slab(0, smem);
  if (n_slabs > 1) dma_slab(1, smem + G::SlabB);
  int slot = 0, S0 = 0, jm = 0;  // ring slot of the next slab, its number, the multiply it belongs to
  f32x4 cur[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) cur[b] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int l = 0; l < a.n_ops; ++l) {
    const npf_x6_op_t& o = a.op[l];
    // ---------------------------------------------------------------- input side
    if (o.in_pt != nullptr) {
      const float* x = o.in_pt + lane_off;
#pragma unroll
      for (int b = 0; b < NB; ++b) cur[b] = *(const f32x4*)(x + b * 512);
    }
    if (o.in_rows != nullptr) {
      // cur <- [relu](in_w^T rows + in_b): the first layer of an MLP whose input has <= 4 features (mlp.py:96), or the dgrad
      // of an F -> 4 output layer (mlp.py:109): plain fp32 FMAs, the matrix from L1 / L2
      const f32x4 r = ((const f32x4*)o.in_rows)[row_idx];
      const int nb_in = o.in_n >> 4;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b < nb_in) {
          const int f = 16 * b + 4 * g;
          if (o.in_b != nullptr) v = *(const f32x4*)(o.in_b + f);
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const f32x4 w = *(const f32x4*)(o.in_w + (size_t)n * o.in_n + f);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(r[n], w[e], v[e]);
          }
          if (o.in_relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
        }
        cur[b] = v;
      }
    }
    if (o.pre_add != nullptr) {
      const float* x = o.pre_add + lane_off;
#pragma unroll
      for (int b = 0; b < NB; ++b) cur[b] += *(const f32x4*)(x + b * 512);
    }
    if (o.mask != nullptr) {
      const float* m = o.mask + lane_off;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const f32x4 v = *(const f32x4*)(m + b * 512);
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[b][e] = v[e] > 0.f ? cur[b][e] : 0.f;
      }
    }
    if (o.mask_bits != nullptr) {
      const unsigned long long w = o.mask_bits[bits_off];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[b][e] = ((w >> (4 * b + e)) & 1ull) ? cur[b][e] : 0.f;
    }
    if (o.sbwd_p != nullptr) {
      // softmax backward (the autograd of attention.py:161): dS = scale * P * (dP - sum_c dP_c P_c)
      const float* pp = o.sbwd_p + lane_off;
      f32x4 P[NB];
      float dot = 0.f;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        P[b] = *(const f32x4*)(pp + b * 512);
#pragma unroll
        for (int e = 0; e < 4; ++e) dot = fmaf(cur[b][e], P[b][e], dot);
      }
      dot = xp_sum4(dot);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[b][e] = o.sbwd_scale * P[b][e] * (cur[b][e] - dot);
    }
    if (o.store_in != nullptr && valid) {
      float* d = o.store_in + lane_off;
#pragma unroll
      for (int b = 0; b < NB; ++b) __builtin_nontemporal_store(cur[b], (f32x4*)(d + b * 512));
    }
    if (o.store_in_bits != nullptr && valid) {
      unsigned long long w = 0ull;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) w |= (unsigned long long)(cur[b][e] > 0.f) << (4 * b + e);
      o.store_in_bits[bits_off] = w;
    }
    if (o.w_img == nullptr) continue;

    // ---------------------------------------------------------------- the multiply
    // the bias into LDS (read back per slab; visible behind the barrier of the op's first slab)
    if (tid < KF) bias_lds[(jm & 1) * KF + tid] = o.bias != nullptr ? o.bias[(size_t)task * o.bias_task_stride + tid] : 0.f;
    // the input as three packed bf16 terms (the B operands), once per op
    xp_u32x4 tb[3][KS];
#pragma unroll
    for (int st = 0; st < KS; ++st) xp_split(cur[2 * st], cur[2 * st + 1], tb[0][st], tb[1][st], tb[2][st]);
    // an addend (MergeFlatInputs: relu(x1 + resizer(x2)), encoders.py:178-179; a gradient fan-in) waits in the registers of the
    // blocks it will be added to: the input is dead once it is split, and block s is only rewritten at slab s
    const bool has_add = o.addend != nullptr;
    if (has_add) {
      const float* ad = o.addend + lane_off;
#pragma unroll
      for (int b = 0; b < NB; ++b) cur[b] = *(const f32x4*)(ad + b * 512);
    }
    const bool post = o.softmax_n > 0;  // (the stores then follow the softmax)
    float* out = (o.store_out != nullptr && valid && !post) ? o.store_out + lane_off : nullptr;
    const unsigned bias_l = lds0 + G::Slots * G::SlabB + (jm & 1) * (KF * 4) + g * 16;
    const bool relu = o.relu != 0;
    unsigned long long pos_bits = 0ull;
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      const int S = S0 + s;
      // slab S has landed for everyone, everyone is done with slab S - 1 (whose slot slab S + 2 goes into).  Counted wait: the
      // NP pieces of slab S + 1 may stay in flight (vector-memory operations retire in order; loads and stores of this wave
      // issued since are older than them or make the wait stricter, never laxer)
      if (S + 1 < n_slabs) {
        if constexpr (G::NP == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if (S + 2 < n_slabs) dma_slab(S + 2, smem + ((slot + 2) % G::Slots) * G::SlabB);
      const unsigned sl = lds0 + slot * G::SlabB + lane_row;
      f32x4 acc, sm = {0.f, 0.f, 0.f, 0.f};
      xp_u32x4 fr[2][3];
      if constexpr (KF == 256) {
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %5 offset:8192\n\tds_read_b128 %3, %5 offset:16384"
                     : "=&v"(acc), "=&v"(fr[0][0]), "=&v"(fr[0][1]), "=&v"(fr[0][2])
                     : "v"(bias_l + 64 * s), "v"(sl + (((0 + g) ^ p) << 4)));
      } else {
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %5 offset:4096\n\tds_read_b128 %3, %5 offset:8192"
                     : "=&v"(acc), "=&v"(fr[0][0]), "=&v"(fr[0][1]), "=&v"(fr[0][2])
                     : "v"(bias_l + 64 * s), "v"(sl + (((0 + g) ^ p) << 4)));
      }
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        const int c = st & 1, n = c ^ 1;
        if (st + 1 < KS) {
          if constexpr (KF == 256) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:8192\n\tds_read_b128 %2, %7 offset:16384"
                         : "=&v"(fr[n][0]), "=&v"(fr[n][1]), "=&v"(fr[n][2]), "+v"(fr[c][0]), "+v"(fr[c][1]), "+v"(fr[c][2]), "+v"(acc)
                         : "v"(sl + (((4 * (st + 1) + g) ^ p) << 4)));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:4096\n\tds_read_b128 %2, %7 offset:8192"
                         : "=&v"(fr[n][0]), "=&v"(fr[n][1]), "=&v"(fr[n][2]), "+v"(fr[c][0]), "+v"(fr[c][1]), "+v"(fr[c][2]), "+v"(acc)
                         : "v"(sl + (((4 * (st + 1) + g) ^ p) << 4)));
          }
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[c][0]), "+v"(fr[c][1]), "+v"(fr[c][2]), "+v"(acc));
        }
#define XPMM(A, B, C) \
  C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xp_bf16x8, fr[c][A]), __builtin_bit_cast(xp_bf16x8, tb[B][st]), C, 0, 0, 0)
        XPMM(2, 0, sm);
        XPMM(0, 0, acc);
        XPMM(0, 2, sm);
        XPMM(1, 0, acc);
        XPMM(1, 1, sm);
        XPMM(0, 1, acc);
#undef XPMM
      }
      acc += sm;
      if (has_add) acc += cur[s];
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.f);
      }
      cur[s] = acc;  // (block s of the input is dead: its terms are in tb)
      if (out != nullptr) __builtin_nontemporal_store(acc, (f32x4*)(out + s * 512));
#pragma unroll
      for (int e = 0; e < 4; ++e) pos_bits |= (unsigned long long)(acc[e] > 0.f) << (4 * s + e);
      slot = (slot + 1) % G::Slots;
    }
    S0 += NB;
    ++jm;
    // ---------------------------------------------------------------- output side behind the last slab
    if (post) {
      // softmax over the first softmax_n features (keys) of scale * cur, fp32 with max subtraction (attention.py:158-164;
      // the scale is DotAttender's 1 / sqrt(kq_size), :217-218).  Feature 16 b + 4 g + e: a lane holds NB * 4 keys.
      const float sc = o.softmax_scale;
      const int n_valid = o.softmax_n;
      float mx = -INFINITY;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = 16 * b + 4 * g + e < n_valid;
          cur[b][e] = ok ? sc * cur[b][e] : -INFINITY;
          mx = fmaxf(mx, cur[b][e]);
        }
      mx = xp_max4(mx);
      float sum = 0.f;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          cur[b][e] = expf(cur[b][e] - mx);  // (exp(-inf) = 0 for the padding keys)
          sum += cur[b][e];
        }
      sum = xp_sum4(sum);
      const float inv = 1.f / sum;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[b][e] *= inv;
      if (o.store_out != nullptr && valid) {
        float* d = o.store_out + lane_off;
#pragma unroll
        for (int b = 0; b < NB; ++b) __builtin_nontemporal_store(cur[b], (f32x4*)(d + b * 512));
      }
    }
    if (o.store_bits != nullptr && valid) o.store_bits[bits_off] = pos_bits;
  }

  if (a.out_rows != nullptr) {
    // an F -> 4 layer on the registers the program leaves (the decoder's output layer, mlp.py:109): a lane holds KF / 4 of its
    // point's features, four fp32 dot products over them, summed over the point's four lanes.  W_out goes where the ring was.
    __syncthreads();  // (every wave is done with the last slabs)
    f32x4* wl = (f32x4*)smem;
    if (tid < KF) wl[tid] = ((const f32x4*)a.out_w)[tid];
    __syncthreads();
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const f32x4 w = wl[n * (KF / 4) + 4 * b + g];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[n] = fmaf(w[e], cur[b][e], r[n]);
      }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      r[n] = xp_sum4(r[n]);
      if (a.out_b != nullptr) r[n] += a.out_b[n];
    }
    if (valid && g == 0) ((f32x4*)a.out_rows)[row_idx] = r;
  }
}

// The three-term images of a PT32 tensor as per-task weights (npf_x6_task_images): one workgroup per (task, 32-point tile slot).
template <int KF>
__global__ __launch_bounds__(256) void x6_task_images_kernel(const float* __restrict__ src, int tiles_per_task, int pts,
                                                             unsigned short* __restrict__ row_img,
                                                             unsigned short* __restrict__ tr_img) {
  constexpr int TS = KF / 32;  // tile slots of a task (points <= KF)
  __shared__ f32x4 tile_s[(KF / 4) * 33];  // [f4][33]: a padded copy of the PT32 tile (f32x4 per (f4, point))
  const int task = blockIdx.x / TS, ts = blockIdx.x % TS, tid = threadIdx.x;
  const bool have = ts < tiles_per_task;
  if (have) {
    const f32x4* t = (const f32x4*)(src + ((size_t)task * tiles_per_task + ts) * (KF * 32));
    for (int i = tid; i < (KF / 4) * 32; i += 256) {
      const int f4 = i >> 5, pt = i & 31;
      f32x4 v = t[i];
      if (ts * 32 + pt >= pts) v = f32x4{0.f, 0.f, 0.f, 0.f};
      tile_s[f4 * 33 + pt] = v;
    }
  }
  __syncthreads();
  const size_t img = (size_t)KF * KF;  // elements of one term image
  auto split3 = [](float x, unsigned short& h0, unsigned short& h1, unsigned short& h2) {
    const unsigned a = xp_cvt_pk(x, 0.f) & 0xffffu;
    const float r1 = x - __builtin_bit_cast(float, a << 16);
    const unsigned b = xp_cvt_pk(r1, 0.f) & 0xffffu;
    const float r2 = r1 - __builtin_bit_cast(float, b << 16);
    h0 = (unsigned short)a;
    h1 = (unsigned short)b;
    h2 = (unsigned short)(xp_cvt_pk(r2, 0.f) & 0xffffu);
  };
  // row image: row = point (32 of them here), 16-byte chunk (st, gq) = features 32 st + 4 gq + i and 32 st + 16 + 4 gq + i
  if (row_img != nullptr) {
    for (int i = tid; i < 32 * (KF / 8); i += 256) {
      const int pt = i / (KF / 8), ch = i % (KF / 8), st = ch >> 2, gq = ch & 3;
      unsigned short h[3][8];
      if (have) {
        const f32x4 lo = tile_s[(8 * st + gq) * 33 + pt], hi = tile_s[(8 * st + 4 + gq) * 33 + pt];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          split3(lo[e], h[0][e], h[1][e], h[2][e]);
          split3(hi[e], h[0][4 + e], h[1][4 + e], h[2][4 + e]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int e = 0; e < 8; ++e) h[q][e] = 0;
      }
      unsigned short* d = row_img + (size_t)task * 3 * img + (size_t)(ts * 32 + pt) * KF + ch * 8;
#pragma unroll
      for (int q = 0; q < 3; ++q) *(uint4*)(d + q * img) = *(const uint4*)h[q];
    }
  }
  // transposed image: row = feature, the tile's 32 columns = points, chunk gq = points 4 gq + i and 16 + 4 gq + i
  if (tr_img != nullptr) {
    for (int i = tid; i < KF * 4; i += 256) {
      const int f = i >> 2, gq = i & 3;
      unsigned short h[3][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int pt = (e < 4) ? 4 * gq + e : 16 + 4 * gq + (e - 4);
        const float x = have ? tile_s[(f >> 2) * 33 + pt][f & 3] : 0.f;
        split3(x, h[0][e], h[1][e], h[2][e]);
      }
      unsigned short* d = tr_img + (size_t)task * 3 * img + (size_t)f * KF + ts * 32 + gq * 8;
#pragma unroll
      for (int q = 0; q < 3; ++q) *(uint4*)(d + q * img) = *(const uint4*)h[q];
    }
  }
}

}  // namespace npf

extern "C" int npf_x6_run(const npf_x6_op_t* ops, int32_t n_ops, const float* out_w, const float* out_b, float* out_rows,
                          int32_t n_tasks, int32_t tiles_per_task, int32_t per_task, int32_t width, void* stream) {
  if (!ops || n_ops <= 0 || n_ops > NPF_X6_MAX_OPS || n_tasks <= 0 || tiles_per_task <= 0) return NPF_EINVAL;
  if (width != 128 && width != 256) return NPF_EINVAL;
  if ((out_rows != nullptr) != (out_w != nullptr) || (out_b != nullptr && out_rows == nullptr)) return NPF_EINVAL;
  if ((((uintptr_t)out_w) | ((uintptr_t)out_rows)) & 15) return NPF_EINVAL;
  if (((uintptr_t)out_b) & 3) return NPF_EINVAL;
  npf::XpArgs a;
  a.n_mm = 0;
  bool have_cur = false;
  for (int l = 0; l < n_ops; ++l) {
    const npf_x6_op_t& o = ops[l];
    if ((o.in_pt != nullptr) && (o.in_rows != nullptr)) return NPF_EINVAL;
    if ((o.in_rows != nullptr) != (o.in_w != nullptr) || (o.in_b != nullptr && o.in_rows == nullptr)) return NPF_EINVAL;
    if (o.in_rows != nullptr && (o.in_n <= 0 || o.in_n > width || (o.in_n & 15))) return NPF_EINVAL;
    if (o.in_pt != nullptr || o.in_rows != nullptr) have_cur = true;
    if (!have_cur) return NPF_EINVAL;  // (the first op must bring an input)
    if ((((uintptr_t)o.in_pt) | ((uintptr_t)o.in_rows) | ((uintptr_t)o.in_w) | ((uintptr_t)o.in_b) | ((uintptr_t)o.pre_add) |
         ((uintptr_t)o.mask) | ((uintptr_t)o.sbwd_p) | ((uintptr_t)o.store_in) | ((uintptr_t)o.w_img) | ((uintptr_t)o.addend) |
         ((uintptr_t)o.store_out)) & 15)
      return NPF_EINVAL;
    if ((((uintptr_t)o.mask_bits) | ((uintptr_t)o.store_in_bits) | ((uintptr_t)o.store_bits)) & 7) return NPF_EINVAL;
    if (((uintptr_t)o.bias) & 3) return NPF_EINVAL;
    if ((o.w_task_stride & 15) || o.w_task_stride < 0 || o.bias_task_stride < 0) return NPF_EINVAL;
    if ((o.w_task_stride != 0 || o.bias_task_stride != 0) && !per_task) return NPF_EINVAL;
    if (o.softmax_n < 0 || o.softmax_n > width) return NPF_EINVAL;
    if (o.w_img == nullptr && (o.bias || o.addend || o.store_out || o.store_bits || o.relu || o.softmax_n)) return NPF_EINVAL;
    a.op[l] = o;
    if (o.w_img != nullptr) {
      a.mm_img[a.n_mm] = (const char*)o.w_img;
      a.mm_stride[a.n_mm] = o.w_task_stride;
      ++a.n_mm;
    }
  }
  for (int l = n_ops; l < NPF_X6_MAX_OPS; ++l) a.op[l] = ops[0];
  for (int j = a.n_mm; j < NPF_X6_MAX_OPS; ++j) {
    a.mm_img[j] = a.n_mm ? a.mm_img[0] : nullptr;
    a.mm_stride[j] = 0;
  }
  a.out_w = out_w;
  a.out_b = out_b;
  a.out_rows = out_rows;
  a.n_ops = n_ops;
  a.total_tiles = n_tasks * tiles_per_task;
  a.tiles_per_task = tiles_per_task;
  a.wgs_per_task = per_task ? (tiles_per_task + 1) / 2 : 0;
  const int n_wg = per_task ? n_tasks * a.wgs_per_task : (a.total_tiles + 1) / 2;
  // the workgroups of a task share its keys / values: on one XCD (one L2) when the grid allows the renumbering
  a.xcd_remap = (per_task && (n_wg % 8) == 0 && a.wgs_per_task > 1) ? 1 : 0;
  if (width == 256) hipLaunchKernelGGL(npf::x6_program_kernel<256>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(npf::x6_program_kernel<128>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, a);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_x6_task_images(const float* src, int32_t n_tasks, int32_t pts, int32_t width, void* row_img, void* tr_img,
                                  void* stream) {
  if (!src || n_tasks <= 0 || pts <= 0 || pts > width || (width != 128 && width != 256)) return NPF_EINVAL;
  if (!row_img && !tr_img) return NPF_EINVAL;
  if ((((uintptr_t)src) | ((uintptr_t)row_img) | ((uintptr_t)tr_img)) & 15) return NPF_EINVAL;
  const int tiles = (pts + 31) / 32;
  const int n_wg = n_tasks * (width / 32);
  if (width == 256)
    hipLaunchKernelGGL(npf::x6_task_images_kernel<256>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, src, tiles, pts,
                       (unsigned short*)row_img, (unsigned short*)tr_img);
  else
    hipLaunchKernelGGL(npf::x6_task_images_kernel<128>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, src, tiles, pts,
                       (unsigned short*)row_img, (unsigned short*)tr_img);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}
