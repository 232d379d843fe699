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
#include "x6_args.hpp"

// Timing-only diagnostic builds (tools/fastbuild.sh x6_kernel OUT.so -DXP_NO_...): each removes one ingredient of the slab
// loop, results are garbage, only the clock counts.  None is defined in the library build.
//   XP_NO_MFMA  no matrix instructions      XP_NO_FRAG  no weight-fragment reads from LDS     XP_NO_DMA  no slab DMA
//   XP_NO_STORE no global stores of the ops XP_NO_SPLIT the layer input is not split (bits reinterpreted)

namespace npf {

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

// (see x6m_split in mlp_x6_kernel.hip for the edge-value semantics)
__device__ __forceinline__ void xp_split(const f32x4& lo, const f32x4& hi, xp_u32x4& t0, xp_u32x4& t1, xp_u32x4& t2) {
#ifdef XP_NO_SPLIT
  t0 = __builtin_bit_cast(xp_u32x4, lo);
  t1 = __builtin_bit_cast(xp_u32x4, hi);
  t2 = t0 ^ t1;
  return;
#endif
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

// NPG = 16-point groups per wave.  <256, 1> / <128, 1>: a wave owns half a tile, 216-256 registers, two workgroups per CU.
// <256, 2>: a wave owns a whole tile -- every weight fragment read from LDS and every slab piece streamed from L2 feeds twice the
// matrix instructions -- and <512, 1>: 512 features of 16 points; both keep 320 registers of activation terms and run one wave
// per SIMD (512 registers).
// NW = waves per workgroup: 4, or 8 waves (<256, 1, 8>) that share ONE slab ring -- a CU then streams every slab once instead of
// once per workgroup of 64 points (half the L2 -> LDS traffic and half the pieces per wave).
// LA = slabs in flight ahead of the one being multiplied (ring of LA + 1 slots): 2, or 4 where one workgroup has the CU's LDS.
template <int KF, int NPG, int NW, int LA = 2>
__global__ __launch_bounds__(NW * 64, (NW == 4 && KF * NPG <= 256) ? 2 : 1) void x6_program_kernel(const XpArgs a) {
  using G = XpGeom<KF>;
  constexpr int NB = G::NB, KS = G::KS;
  constexpr int Slots = LA + 1;
  static_assert(LA >= 2 && LA <= 4 && LA <= NB, "slab look-ahead");
  constexpr int TPW = (NPG == 2 ? 4 : 2) * (NW / 4);  // tiles per workgroup
  constexpr int NPW = 3 * G::PPT / NW;                  // slab pieces per wave
  static_assert(NPG == 1 || NPG == 2, "a wave owns half a tile or a whole one");
  static_assert((NW == 4 || NW == 8) && (3 * G::PPT) % NW == 0 && NPW <= KS, "pieces dealt evenly, one per k-step");
  __shared__ __attribute__((aligned(16))) char smem[Slots * G::SlabB + G::BiasB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  int bid = blockIdx.x;
  if (a.xcd_remap) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
  const int w_tile = NPG == 2 ? wave : (wave >> 1);  // this wave's tile inside the workgroup
  const int half0 = NPG == 2 ? 0 : (wave & 1);       // its first half tile
  int task, t_in;
  bool valid;
  if (a.wgs_per_task > 0) {
    task = bid / a.wgs_per_task;
    t_in = (bid - task * a.wgs_per_task) * TPW + w_tile;
    valid = t_in < a.tiles_per_task;
  } else {
    const long t = (long)bid * TPW + w_tile;
    valid = t < a.total_tiles;
    task = valid ? (int)(t / a.tiles_per_task) : 0;
    t_in = valid ? (int)(t - (long)task * a.tiles_per_task) : 0;
  }
#ifdef XP_NO_STORE
  valid = valid && a.n_ops > 1000;  // (never true: the stores stay in the code, none executes)
#endif
  // (a wave without a tile still streams slabs and meets barriers; it loads tile 0 and stores nothing)
  const long tile = valid ? (long)task * a.tiles_per_task + t_in : 0;
  if (a.wgs_per_task == 0) task = 0;  // (flat launches share every weight: no per-task strides)
  // this lane's float4 column in its tile of a PT32 tensor with KF features: block b at + (4 b + g) * 128 floats; the second
  // point group of a wave (NPG == 2) 16 points = 64 floats further
  // Addresses = a wave-uniform part (scalar registers) + a 32-bit lane part: nothing 64-bit per lane stays live across the slab
  // loops (per-slab store addresses kept in vector registers were what hipcc spilled inside the loop).
  const size_t tile_off = (size_t)tile * (KF * 32);                       // floats, wave-uniform
  const unsigned lane_b = (unsigned)(((16 * half0 + p) * 4 + g * 128) * 4);  // bytes inside the tile
  const size_t bits_off = ((size_t)tile * 2 + half0) * 64;                // [tile][half][64 lanes] uint64, + lane
  const size_t row_off = (size_t)tile * 32 + 16 * half0;                  // a rows tensor: + p
  // PT32 operand ``base``: the float4 of block b of this lane's point in point group pg
  auto pt32 = [&](const float* base, int pg, int b) -> f32x4* {
    return (f32x4*)((char*)const_cast<float*>(base + tile_off + (size_t)(b * 512 + 64 * pg)) + lane_b);
  };
  // row-major operands [task][pts][KF] (NPF_X6_IN_RM / NPF_X6_ADD_RM): this lane's point, clamped into the task (padding points of
  // the last tile re-read its last point: loaded, never stored)
  auto rm32 = [&](const float* base, int pg, int b) -> const f32x4* {
    int pt = (valid ? t_in : 0) * 32 + 16 * (half0 + pg) + p;
    pt = pt < a.pts_per_task ? pt : a.pts_per_task - 1;
    const size_t tk = valid ? (size_t)(tile / a.tiles_per_task) : 0;
    return (const f32x4*)(base + (tk * a.pts_per_task + (size_t)pt) * KF + 4 * g + 16 * b);
  };

  // DMA of a slab: 3 * PPT pieces of 1 KiB (term q / PPT, rows RPP (q % PPT) ..), NP per wave; the swizzle (chunk c of row r at
  // position c ^ (r & 15)) is applied to the source address: uniform base per piece + a lane offset
  constexpr int LPR = 64 / G::RPP;  // lanes per row of a piece
  constexpr int NDL = G::PPT / NW > 0 ? G::PPT / NW : 1;  // distinct first rows of a wave's pieces
  unsigned dma_lane[NDL];
#pragma unroll
  for (int n = 0; n < NDL; ++n) {
    const int r0 = G::RPP * ((wave + NW * n) % G::PPT), row = r0 + lane / LPR, pos = lane % LPR;
    dma_lane[n] = (unsigned)((lane / LPR) * G::RowB + ((pos ^ (row & 15)) << 4));
  }
  const int n_slabs = a.n_mm * NB;
  // piece n of this wave's share of a slab (term q / PPT, rows RPP (q % PPT) .., q = wave + 4 n); ``rows`` = the slab's first row
  // inside its multiply's image ``img`` (wave-uniform)
  auto dma_piece = [&](const char* img, int rows, char* slot, int n) {
#ifdef XP_NO_DMA
    return;
#endif
    const char* base = img + (size_t)rows * G::RowB;
    asm volatile("" : "+s"(base));
    const int q = wave + NW * n, term = q / G::PPT, r0 = G::RPP * (q % G::PPT);
    xp_dma16(base + (size_t)term * ((size_t)KF * KF * 2) + r0 * G::RowB + dma_lane[n % NDL], slot + term * G::TermB + r0 * G::RowB);
  };
  auto mm_base = [&](int j) { return a.mm_img[j] + (size_t)task * a.mm_stride[j]; };
  auto dma_slab = [&](int S, char* slot) {
    const char* img = mm_base(S / NB);
#pragma unroll
    for (int n = 0; n < NPW; ++n) dma_piece(img, (S % NB) * 16, slot, n);
  };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lane_row = (unsigned)(p * G::RowB);
  float* bias_lds = (float*)(smem + Slots * G::SlabB);

#pragma unroll
  for (int i = 0; i < LA; ++i)
    if (n_slabs > i) dma_slab(i, smem + i * G::SlabB);
  int slot = 0, S0 = 0, jm = 0;  // ring slot of the next slab, its number, the multiply it belongs to
  f32x4 cur[NPG][NB];
#pragma unroll
  for (int pg = 0; pg < NPG; ++pg)
#pragma unroll
    for (int b = 0; b < NB; ++b) cur[pg][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int l = 0; l < a.n_ops; ++l) {
    const npf_x6_op_t& o = a.op[l];
    const int oflags = o.reserved[0];
    // ---------------------------------------------------------------- input side
#pragma unroll
    for (int pg = 0; pg < NPG; ++pg) {
      if (o.in_pt != nullptr) {
        if (oflags & NPF_X6_IN_RM) {
#pragma unroll
          for (int b = 0; b < NB; ++b) cur[pg][b] = *rm32(o.in_pt, pg, b);
        } else {
#pragma unroll
          for (int b = 0; b < NB; ++b) cur[pg][b] = *pt32(o.in_pt, pg, b);
        }
      }
      if (o.in_rows != nullptr) {
        // cur <- [relu](in_w^T rows + in_b): the first layer of an MLP whose input has <= 4 features (mlp.py:96), or the dgrad
        // of an F -> 4 output layer (mlp.py:109): plain fp32 FMAs, the matrix from L1 / L2
        const f32x4 r = ((const f32x4*)o.in_rows)[row_off + 16 * pg + p];
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
          cur[pg][b] = v;
        }
      }
      if (o.pre_add != nullptr) {
#pragma unroll
        for (int b = 0; b < NB; ++b) cur[pg][b] += *pt32(o.pre_add, pg, b);
      }
      if (o.mask != nullptr) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const f32x4 v = *pt32(o.mask, pg, b);
#pragma unroll
          for (int e = 0; e < 4; ++e) cur[pg][b][e] = v[e] > 0.f ? cur[pg][b][e] : 0.f;
        }
      }
      if constexpr (NB <= 16) {
        if (o.mask_bits != nullptr) {
          // one sign-extended bit-field extract + one AND per value (bit 4 b + e of the lane's 64-bit word)
          const unsigned long long w = o.mask_bits[bits_off + 64 * pg + lane];
          const int wl = (int)(unsigned)w, wh = (int)(unsigned)(w >> 32);
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // (through scalar temporaries: __builtin_bit_cast applied to an element of an ext_vector read element 0, hipcc 7.2)
              const int m = __builtin_amdgcn_sbfe(b < 8 ? wl : wh, 4 * (b & 7) + e, 1);
              const float x = cur[pg][b][e];
              cur[pg][b][e] = __int_as_float(__float_as_int(x) & m);
            }
        }
      }
      if constexpr (NB <= 16) if (o.sbwd_p != nullptr) {
        // softmax backward (the autograd of attention.py:161): dS = scale * P * (dP - sum_c dP_c P_c)
        f32x4 P[NB];
        float dot = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          P[b] = *pt32(o.sbwd_p, pg, b);
#pragma unroll
          for (int e = 0; e < 4; ++e) dot = fmaf(cur[pg][b][e], P[b][e], dot);
        }
        dot = xp_sum4(dot);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) cur[pg][b][e] = o.sbwd_scale * P[b][e] * (cur[pg][b][e] - dot);
      }
      if (o.store_in != nullptr && valid) {
#pragma unroll
        for (int b = 0; b < NB; ++b) __builtin_nontemporal_store(cur[pg][b], pt32(o.store_in, pg, b));
      }
      if constexpr (NB <= 16) {
        if (o.store_in_bits != nullptr && valid) {
          unsigned wlo = 0u, whi = 0u;
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) (b < 8 ? wlo : whi) |= (unsigned)(cur[pg][b][e] > 0.f) << (4 * (b & 7) + e);
          o.store_in_bits[bits_off + 64 * pg + lane] = ((unsigned long long)whi << 32) | wlo;
        }
      }
    }
    if (o.w_img == nullptr) continue;

    // ---------------------------------------------------------------- the multiply
    // the bias into LDS (read back per slab; visible behind the barrier of the op's first slab)
    for (int i = tid; i < KF; i += NW * 64)
      bias_lds[(jm & 1) * KF + i] = o.bias != nullptr ? o.bias[(size_t)task * o.bias_task_stride + i] : 0.f;
    // the input as three packed bf16 terms (the B operands), once per op
    xp_u32x4 tb[NPG][3][KS];
#pragma unroll
    for (int pg = 0; pg < NPG; ++pg)
#pragma unroll
      for (int st = 0; st < KS; ++st) xp_split(cur[pg][2 * st], cur[pg][2 * st + 1], tb[pg][0][st], tb[pg][1][st], tb[pg][2][st]);
    // an addend (MergeFlatInputs: relu(x1 + resizer(x2)), encoders.py:178-179; a gradient fan-in) waits in the registers of the
    // blocks it will be added to: the input is dead once it is split, and block s is only rewritten at slab s
    const bool has_add = o.addend != nullptr;
    if (has_add) {
#pragma unroll
      for (int pg = 0; pg < NPG; ++pg) {
        if (oflags & NPF_X6_ADD_RM) {
#pragma unroll
          for (int b = 0; b < NB; ++b) cur[pg][b] = *rm32(o.addend, pg, b);
        } else {
#pragma unroll
          for (int b = 0; b < NB; ++b) cur[pg][b] = *pt32(o.addend, pg, b);
        }
      }
    }
    // the images the slab stream reads during this multiply: its own, and -- for the last two slabs' prefetch -- the next one's
    const char* const img0 = mm_base(jm);
    const char* const img1 = mm_base(jm + 1 < a.n_mm ? jm + 1 : jm);
    const bool post = NB <= 16 && o.softmax_n > 0;  // (the stores then follow the softmax; 512-wide programs have none)
    const float* const out = (o.store_out != nullptr && valid && !post) ? o.store_out : nullptr;  // (wave-uniform)
    const unsigned bias_l = lds0 + Slots * G::SlabB + (jm & 1) * (KF * 4) + g * 16;
    const bool relu = o.relu != 0;
    unsigned pos_lo[NPG], pos_hi[NPG];  // the output's ReLU bits: blocks 0-7, 8-15
#pragma unroll
    for (int pg = 0; pg < NPG; ++pg) pos_lo[pg] = pos_hi[pg] = 0u;
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      const int S = S0 + s;
      // slab S has landed for everyone, everyone is done with slab S - 1 (whose slot slab S + 2 goes into).  Counted wait: the
      // NP pieces of slab S + 1 may stay in flight (vector-memory operations retire in order; loads and stores of this wave
      // issued since are older than them or make the wait stricter, never laxer)
      {
        // (wave-uniform: how many later slabs have pieces in flight -- LA - 1, fewer at the end of the program)
        const int rem = n_slabs - 1 - S;
#define XP_WAIT(N) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory")
        if (rem >= LA - 1) XP_WAIT((LA - 1) * NPW);
        else if (LA > 3 && rem == 2) XP_WAIT(2 * NPW);
        else if (LA > 2 && rem == 1) XP_WAIT(NPW);
        else XP_WAIT(0);
#undef XP_WAIT
      }
      // slab S + LA goes into the slot slab S - 1 has left.  Its pieces are issued one per k-step, each at a point where none of
      // this wave's LDS reads is outstanding.  (Measured, round 3: a vector-memory instruction -- slab piece or store -- issued
      // between the matrix instructions while fragment reads were in flight gave sporadic wrong results in single waves at full
      // grid sizes, counted or full waits alike; issued at these points never.  DESIGN.md 3.1.)
      const bool more = S + LA < n_slabs;
      char* const nslot = smem + ((slot + LA) % Slots) * G::SlabB;
      const unsigned sl = lds0 + slot * G::SlabB + lane_row;
      f32x4 acc[NPG], sm[NPG];
      xp_u32x4 fr[2][3];
#ifdef XP_NO_FRAG
      acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "=v"(fr[0][0]), "=v"(fr[0][1]), "=v"(fr[0][2]), "=v"(fr[1][0]), "=v"(fr[1][1]), "=v"(fr[1][2]));
#else
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %5 offset:%6\n\tds_read_b128 %3, %5 offset:%7"
                   : "=&v"(acc[0]), "=&v"(fr[0][0]), "=&v"(fr[0][1]), "=&v"(fr[0][2])
                   : "v"(bias_l + 64 * s), "v"(sl + (((0 + g) ^ p) << 4)), "n"(G::TermB), "n"(2 * G::TermB));
#endif
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        const int c = st & 1, n = c ^ 1;
        if (st < NPW && more) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          dma_piece(s + LA < NB ? img0 : img1, ((s + LA) % NB) * 16, nslot, st);
        }
#ifdef XP_NO_FRAG
        if (false) {
#else
        if (st + 1 < KS) {
#endif
          asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:%8\n\tds_read_b128 %2, %7 offset:%9"
                       : "=&v"(fr[n][0]), "=&v"(fr[n][1]), "=&v"(fr[n][2]), "+v"(fr[c][0]), "+v"(fr[c][1]), "+v"(fr[c][2]), "+v"(acc[0])
                       : "v"(sl + (((4 * (st + 1) + g) ^ p) << 4)), "n"(G::TermB), "n"(2 * G::TermB));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[c][0]), "+v"(fr[c][1]), "+v"(fr[c][2]), "+v"(acc[0]));
        }
        if (st == 0) {
#pragma unroll
          for (int pg = 0; pg < NPG; ++pg) {
            if (pg > 0) acc[pg] = acc[0];  // (the bias row is the same for every point group)
            sm[pg] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#ifdef XP_NO_MFMA
#define XPMM(A, B, C, PG) asm volatile("" : "+v"(C[PG]) : "v"(fr[c][A]), "v"(tb[PG][B][st]))
#else
#define XPMM(A, B, C, PG)                                                                                            \
  C[PG] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xp_bf16x8, fr[c][A]),                           \
                                                  __builtin_bit_cast(xp_bf16x8, tb[PG][B][st]), C[PG], 0, 0, 0)
#endif
        // six products per point group; consecutive instructions never share an accumulator
        if constexpr (NPG == 1) {
          XPMM(2, 0, sm, 0);
          XPMM(0, 0, acc, 0);
          XPMM(0, 2, sm, 0);
          XPMM(1, 0, acc, 0);
          XPMM(1, 1, sm, 0);
          XPMM(0, 1, acc, 0);
        } else {
          XPMM(2, 0, sm, 0);
          XPMM(2, 0, sm, 1);
          XPMM(0, 0, acc, 0);
          XPMM(0, 0, acc, 1);
          XPMM(0, 2, sm, 0);
          XPMM(0, 2, sm, 1);
          XPMM(1, 0, acc, 0);
          XPMM(1, 0, acc, 1);
          XPMM(1, 1, sm, 0);
          XPMM(1, 1, sm, 1);
          XPMM(0, 1, acc, 0);
          XPMM(0, 1, acc, 1);
        }
#undef XPMM
      }
#pragma unroll
      for (int pg = 0; pg < NPG; ++pg) {
        f32x4 r = acc[pg] + sm[pg];
        if (has_add) {  // (the empty asm keeps this a wave-uniform branch: as a select it costs every op four instructions per slab)
          asm volatile("" ::: "memory");
          r += cur[pg][s];
        }
        if (relu) {  // (v_max_f32 as is: fmaxf canonicalises its operand first, a second instruction per value)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = r[e];
            asm("v_max_f32 %0, 0, %0" : "+v"(x));
            r[e] = x;
          }
        }
        cur[pg][s] = r;  // (block s of the input is dead: its terms are in tb)
        if (out != nullptr) __builtin_nontemporal_store(r, pt32(out, pg, s));
        if constexpr (NB <= 16) {
#pragma unroll
          for (int e = 0; e < 4; ++e) (s < 8 ? pos_lo[pg] : pos_hi[pg]) |= (unsigned)(r[e] > 0.f) << (4 * (s & 7) + e);
        }
      }
      slot = (slot + 1) % Slots;
    }
    S0 += NB;
    ++jm;
    // ---------------------------------------------------------------- output side behind the last slab
    if constexpr (NB <= 16) if (post) {
      // softmax over the first softmax_n features (keys) of scale * cur, fp32 with max subtraction (attention.py:158-164;
      // the scale is DotAttender's 1 / sqrt(kq_size), :217-218).  Feature 16 b + 4 g + e: a lane holds NB * 4 keys.
      const float sc = o.softmax_scale;
      const int n_valid = o.softmax_n;
#pragma unroll
      for (int pg = 0; pg < NPG; ++pg) {
        float mx = -INFINITY;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = 16 * b + 4 * g + e < n_valid;
            cur[pg][b][e] = ok ? sc * cur[pg][b][e] : -INFINITY;
            mx = fmaxf(mx, cur[pg][b][e]);
          }
        mx = xp_max4(mx);
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            cur[pg][b][e] = expf(cur[pg][b][e] - mx);  // (exp(-inf) = 0 for the padding keys)
            sum += cur[pg][b][e];
          }
        sum = xp_sum4(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) cur[pg][b][e] *= inv;
        if (o.store_out != nullptr && valid) {
#pragma unroll
          for (int b = 0; b < NB; ++b) __builtin_nontemporal_store(cur[pg][b], pt32(o.store_out, pg, b));
        }
      }
    }
    if constexpr (NB <= 16) {
      if (o.store_bits != nullptr && valid) {
#pragma unroll
        for (int pg = 0; pg < NPG; ++pg) o.store_bits[bits_off + 64 * pg + lane] = ((unsigned long long)pos_hi[pg] << 32) | pos_lo[pg];
      }
    }
  }

  if (a.out_rows != nullptr) {
    // an F -> 4 layer on the registers the program leaves (the decoder's output layer, mlp.py:109): a lane holds KF / 4 of its
    // point's features, four fp32 dot products over them, summed over the point's four lanes.  W_out goes where the ring was.
    __syncthreads();  // (every wave is done with the last slabs)
    f32x4* wl = (f32x4*)smem;
    for (int i = tid; i < KF; i += NW * 64) wl[i] = ((const f32x4*)a.out_w)[i];
    __syncthreads();
#pragma unroll
    for (int pg = 0; pg < NPG; ++pg) {
      f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const f32x4 w = wl[n * (KF / 4) + 4 * b + g];
#pragma unroll
          for (int e = 0; e < 4; ++e) r[n] = fmaf(w[e], cur[pg][b][e], r[n]);
        }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        r[n] = xp_sum4(r[n]);
        if (a.out_b != nullptr) r[n] += a.out_b[n];
      }
      if (valid && g == 0) ((f32x4*)a.out_rows)[row_off + 16 * pg + p] = r;
    }
  }
}

// 512-feature programs with the contraction split over a PAIR of waves (npf_x6_run_ex, width 512, plain layers): wave (pg, kh)
// holds features 256 kh .. 256 kh + 255 of point group pg -- 160 registers of activation and terms instead of 320, so two waves
// per SIMD again -- multiplies its half of the inputs through the slab's 16 output rows and hands the partial sums of the rows
// it does not own to its partner through LDS (1 KiB per slab and pair, read behind the next slab's barrier).  Eight waves = four
// point groups x two halves share one slab ring: 6 pieces per wave and slab, as in the 256-wide instance.
// Ops: in_pt / addend (PT32 or row-major), bias, relu, store_out, the F -> 4 layer behind the program.
struct XwGeom {
  static constexpr int KF = 512, NB = 32, NBH = 16, KS = 8;  // output slabs per multiply, blocks and k-steps per wave
  static constexpr int RowB = KF * 2, TermB = 16 * RowB, SlabB = 3 * TermB, Slots = 3;
  static constexpr int NP = 6;                                // 48 one-row pieces per slab, 8 waves
  static constexpr int BiasB = 2 * KF * 4, XB = 2 * 4 * 1024;  // bias double buffer; exchange [slab parity][pair][64 lanes] float4
};

__global__ __launch_bounds__(512) void x6_wide512_kernel(const XpArgs a) {
  using G = XwGeom;
  constexpr int KF = G::KF, NB = G::NB, NBH = G::NBH, KS = G::KS;
  __shared__ __attribute__((aligned(16))) char smem[G::Slots * G::SlabB + G::BiasB + G::XB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave >> 1, kh = wave & 1;
  const int p = lane & 15, g = lane >> 4;
  const long t_flat = (long)blockIdx.x * 2 + (pg >> 1);
  const int half = pg & 1;
  const bool valid = t_flat < a.total_tiles;
  const long tile = valid ? t_flat : 0;
  const int t_in = (int)(tile % a.tiles_per_task);
  const size_t tk = (size_t)(tile / a.tiles_per_task);
  const size_t tile_off = (size_t)tile * (KF * 32);
  const unsigned lane_b = (unsigned)(((16 * half + p) * 4 + g * 128) * 4);
  const size_t row_off = (size_t)tile * 32 + 16 * half;
  auto pt32 = [&](const float* base, int b) -> f32x4* {  // block b of this wave's half (global block 16 kh + b)
    return (f32x4*)((char*)const_cast<float*>(base + tile_off + (size_t)((16 * kh + b) * 512)) + lane_b);
  };
  auto rm32 = [&](const float* base, int b) -> const f32x4* {
    int pt = t_in * 32 + 16 * half + p;
    pt = pt < a.pts_per_task ? pt : a.pts_per_task - 1;
    return (const f32x4*)(base + (tk * a.pts_per_task + (size_t)pt) * KF + 256 * kh + 4 * g + 16 * b);
  };
  // slab DMA: piece q = wave + 8 n (n < 6) = term q / 16, row q % 16 (one 1 KiB row per piece); chunk c of row r at c ^ (r & 15)
  unsigned dma_lane[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) dma_lane[n] = (unsigned)((lane ^ ((wave + 8 * n) & 15)) << 4);
  const int n_slabs = a.n_mm * NB;
  auto dma_piece = [&](const char* img, int rows, char* slot, int n) {
    const char* base = img + (size_t)rows * G::RowB;
    asm volatile("" : "+s"(base));
    const int q = wave + 8 * n, term = q >> 4, r0 = q & 15;
    xp_dma16(base + (size_t)term * ((size_t)KF * KF * 2) + r0 * G::RowB + dma_lane[n & 1], slot + term * G::TermB + r0 * G::RowB);
  };
  auto dma_slab = [&](int S, char* slot) {
    const char* img = a.mm_img[S / NB];
#pragma unroll
    for (int n = 0; n < G::NP; ++n) dma_piece(img, (S % NB) * 16, slot, n);
  };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lane_row = (unsigned)(p * G::RowB);
  float* bias_lds = (float*)(smem + G::Slots * G::SlabB);
  const unsigned xb0 = lds0 + G::Slots * G::SlabB + G::BiasB + (unsigned)(pg * 1024 + lane * 16);

  if (n_slabs > 0) dma_slab(0, smem);
  if (n_slabs > 1) dma_slab(1, smem + G::SlabB);
  int slot = 0, S0 = 0, jm = 0;
  f32x4 cur[NBH];
#pragma unroll
  for (int b = 0; b < NBH; ++b) cur[b] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int l = 0; l < a.n_ops; ++l) {
    const npf_x6_op_t& o = a.op[l];
    const int oflags = o.reserved[0];
    if (o.in_pt != nullptr) {
      if (oflags & NPF_X6_IN_RM) {
#pragma unroll
        for (int b = 0; b < NBH; ++b) cur[b] = *rm32(o.in_pt, b);
      } else {
#pragma unroll
        for (int b = 0; b < NBH; ++b) cur[b] = *pt32(o.in_pt, b);
      }
    }
    if (o.w_img == nullptr) continue;
    for (int i = tid; i < KF; i += 512) bias_lds[(jm & 1) * KF + i] = o.bias != nullptr ? o.bias[i] : 0.f;
    xp_u32x4 tb[3][KS];
#pragma unroll
    for (int st = 0; st < KS; ++st) xp_split(cur[2 * st], cur[2 * st + 1], tb[0][st], tb[1][st], tb[2][st]);
    const bool has_add = o.addend != nullptr;
    if (has_add) {
      if (oflags & NPF_X6_ADD_RM) {
#pragma unroll
        for (int b = 0; b < NBH; ++b) cur[b] = *rm32(o.addend, b);
      } else {
#pragma unroll
        for (int b = 0; b < NBH; ++b) cur[b] = *pt32(o.addend, b);
      }
    }
    const char* const img0 = a.mm_img[jm];
    const char* const img1 = a.mm_img[jm + 1 < a.n_mm ? jm + 1 : jm];
    const float* const out = (o.store_out != nullptr && valid) ? o.store_out : nullptr;
    const unsigned bias_l = lds0 + G::Slots * G::SlabB + (jm & 1) * (KF * 4) + g * 16;
    const bool relu = o.relu != 0;
    f32x4 hold = {0.f, 0.f, 0.f, 0.f};  // this wave's partial sums of the previous slab, when it owns that slab's rows
    // the rows of slab s belong to the wave whose half of the features they are (kh == s / 16): behind the next barrier it adds its
    // partner's partial sums and finishes the block
    auto finish = [&](int s) {
      if (kh == (s >> 4)) {
        f32x4 x;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(xb0 + (s & 1) * 4096) : "memory");
        f32x4 r = hold + x;
        if (has_add) {  // (a wave-uniform branch, see x6_program_kernel)
          asm volatile("" ::: "memory");
          r += cur[s & 15];
        }
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = r[e];
            asm("v_max_f32 %0, 0, %0" : "+v"(v));
            r[e] = v;
          }
        }
        cur[s & 15] = r;
        if (out != nullptr) __builtin_nontemporal_store(r, pt32(out, s & 15));
      }
    };
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      const int S = S0 + s;
      if (S + 1 < n_slabs) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (s > 0) finish(s - 1);
      const bool more = S + 2 < n_slabs;
      char* const nslot = smem + ((slot + 2) % G::Slots) * G::SlabB;
      const unsigned sl = lds0 + slot * G::SlabB + lane_row;
      const bool mine = kh == (s >> 4);
      f32x4 acc, sm;
      xp_u32x4 fr[2][3];
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %5 offset:%6\n\tds_read_b128 %3, %5 offset:%7"
                   : "=&v"(acc), "=&v"(fr[0][0]), "=&v"(fr[0][1]), "=&v"(fr[0][2])
                   : "v"(bias_l + 64 * s), "v"(sl + (((4 * (8 * kh + 0) + g) ^ p) << 4)), "n"(G::TermB), "n"(2 * G::TermB));
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        const int c = st & 1, n = c ^ 1;
        if (st < G::NP && more) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          dma_piece(s + 2 < NB ? img0 : img1, ((s + 2) % NB) * 16, nslot, st);
        }
        if (st + 1 < KS) {
          asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:%8\n\tds_read_b128 %2, %7 offset:%9"
                       : "=&v"(fr[n][0]), "=&v"(fr[n][1]), "=&v"(fr[n][2]), "+v"(fr[c][0]), "+v"(fr[c][1]), "+v"(fr[c][2]), "+v"(acc)
                       : "v"(sl + (((4 * (8 * kh + st + 1) + g) ^ p) << 4)), "n"(G::TermB), "n"(2 * G::TermB));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[c][0]), "+v"(fr[c][1]), "+v"(fr[c][2]), "+v"(acc));
        }
        if (st == 0) {
          if (!mine) acc = f32x4{0.f, 0.f, 0.f, 0.f};  // (the bias belongs to the owner's sum)
          sm = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#define XWMM(A, B, C) \
  C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xp_bf16x8, fr[c][A]), __builtin_bit_cast(xp_bf16x8, tb[B][st]), C, 0, 0, 0)
        XWMM(2, 0, sm);
        XWMM(0, 0, acc);
        XWMM(0, 2, sm);
        XWMM(1, 0, acc);
        XWMM(1, 1, sm);
        XWMM(0, 1, acc);
#undef XWMM
      }
      const f32x4 part = acc + sm;
      if (mine) {
        hold = part;
      } else {
        asm volatile("ds_write_b128 %0, %1" : : "v"(xb0 + (s & 1) * 4096), "v"(part) : "memory");
      }
      slot = (slot + 1) % G::Slots;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    finish(NB - 1);
    S0 += NB;
    ++jm;
  }

  if (a.out_rows != nullptr) {
    // the F -> 4 layer on the registers the program leaves: each wave of a pair over its 256 features, the partner's four partial
    // dot products through the exchange buffer
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    f32x4* wl = (f32x4*)smem;
    for (int i = tid; i < KF; i += 512) wl[i] = ((const f32x4*)a.out_w)[i];
    __syncthreads();
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NBH; ++b)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const f32x4 w = wl[n * (KF / 4) + 4 * (16 * kh + b) + g];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[n] = fmaf(w[e], cur[b][e], r[n]);
      }
#pragma unroll
    for (int n = 0; n < 4; ++n) r[n] = xp_sum4(r[n]);
    f32x4* xb = (f32x4*)(smem + G::Slots * G::SlabB + G::BiasB) + pg * 64 + lane;
    if (kh == 1) *xb = r;
    __syncthreads();
    if (kh == 0) {
      const f32x4 x = *xb;
#pragma unroll
      for (int n = 0; n < 4; ++n) r[n] += x[n] + (a.out_b != nullptr ? a.out_b[n] : 0.f);
      if (valid && g == 0) ((f32x4*)a.out_rows)[row_off + p] = r;
    }
  }
}

// The three-term images of a PT32 tensor as per-task weights (npf_x6_task_images): one workgroup per (task, 32-point tile slot).
// NT = terms written: 3 (the exact split), or 1 (the bf16 compute mode: the rounded value alone, npf_b16_task_images).
template <int KF, int NT>
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
      unsigned short* d = row_img + (size_t)task * NT * img + (size_t)(ts * 32 + pt) * KF + ch * 8;
#pragma unroll
      for (int q = 0; q < NT; ++q) *(uint4*)(d + q * img) = *(const uint4*)h[q];
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
      unsigned short* d = tr_img + (size_t)task * NT * img + (size_t)f * KF + ts * 32 + gq * 8;
#pragma unroll
      for (int q = 0; q < NT; ++q) *(uint4*)(d + q * img) = *(const uint4*)h[q];
    }
  }
}

}  // namespace npf

extern "C" int npf_x6_run_ex(const npf_x6_op_t* ops, int32_t n_ops, const float* out_w, const float* out_b, float* out_rows,
                             int32_t n_tasks, int32_t tiles_per_task, int32_t pts_per_task, int32_t per_task, int32_t width,
                             int32_t variant, void* stream) {
  if (width != 128 && width != 256 && width != 512) return NPF_EINVAL;
  if (variant < 0 || variant > 3 || (variant >= 2 && width == 128) || (variant == 3 && width != 256)) return NPF_EINVAL;
  npf::XpArgs a;
  const int rc = npf::xp_fill_args(ops, n_ops, out_w, out_b, out_rows, n_tasks, tiles_per_task, pts_per_task, per_task, width,
                                   NPF_X6_IN_RM | NPF_X6_ADD_RM, 256, a);
  if (rc != NPF_OK) return rc;
  // 256 features: 1 = a wave owns half a tile, four waves per workgroup, two workgroups per CU; 2 = a wave owns a whole tile (one
  // wave per SIMD); 3 = as 1 with eight waves per workgroup sharing one slab ring.  The library's choice is
  // NPF_X6_DEFAULT_VARIANT (measured on the config-2 train step: 3 -- 6.45 ms against 6.62 (1) and 7.85 (2))
  const int var = width == 256 ? (variant == 0 ? NPF_X6_DEFAULT_VARIANT : variant) : 1;
  const int npg = var == 2 ? 2 : 1;
  const int tpw = var == 1 ? 2 : 4;
  a.wgs_per_task = per_task ? (tiles_per_task + tpw - 1) / tpw : 0;
  const int n_wg = per_task ? n_tasks * a.wgs_per_task : (a.total_tiles + tpw - 1) / tpw;
  // the workgroups of a task share its keys / values: on one XCD (one L2) when the grid allows the renumbering
  a.xcd_remap = (per_task && (n_wg % 8) == 0 && a.wgs_per_task > 1) ? 1 : 0;
  const dim3 grid(n_wg), block(256);
  if (width == 512 && !per_task && variant != 1) {
    // plain layers (inputs, multiply, bias / addend / ReLU, stores): the contraction split over pairs of waves, two waves per SIMD
    bool plain = true;
    for (int l = 0; l < n_ops; ++l) {
      const npf_x6_op_t& o = ops[l];
      if (o.in_rows || o.pre_add || o.mask || o.store_in) plain = false;
    }
    if (plain) {
      hipLaunchKernelGGL(npf::x6_wide512_kernel, dim3((a.total_tiles + 1) / 2), dim3(512), 0, (hipStream_t)stream, a);
      NPF_CHECK_LAUNCH();
      return NPF_OK;
    }
  }
  if (width == 512) hipLaunchKernelGGL((npf::x6_program_kernel<512, 1, 4>), grid, block, 0, (hipStream_t)stream, a);
  else if (width == 128) hipLaunchKernelGGL((npf::x6_program_kernel<128, 1, 4>), grid, block, 0, (hipStream_t)stream, a);
  else if (var == 2) hipLaunchKernelGGL((npf::x6_program_kernel<256, 2, 4>), grid, block, 0, (hipStream_t)stream, a);
  else if (var == 3) hipLaunchKernelGGL((npf::x6_program_kernel<256, 1, 8, 2>), grid, dim3(512), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((npf::x6_program_kernel<256, 1, 4>), grid, block, 0, (hipStream_t)stream, a);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_x6_run(const npf_x6_op_t* ops, int32_t n_ops, const float* out_w, const float* out_b, float* out_rows,
                          int32_t n_tasks, int32_t tiles_per_task, int32_t per_task, int32_t width, void* stream) {
  return npf_x6_run_ex(ops, n_ops, out_w, out_b, out_rows, n_tasks, tiles_per_task, tiles_per_task * 32, per_task, width, 0,
                       stream);
}

static int xp_task_images(const float* src, int32_t n_tasks, int32_t pts, int32_t width, void* row_img, void* tr_img, bool one_term,
                          void* stream) {
  if (!src || n_tasks <= 0 || pts <= 0 || pts > width || (width != 128 && width != 256)) return NPF_EINVAL;
  if (!row_img && !tr_img) return NPF_EINVAL;
  if ((((uintptr_t)src) | ((uintptr_t)row_img) | ((uintptr_t)tr_img)) & 15) return NPF_EINVAL;
  const int tiles = (pts + 31) / 32;
  const dim3 grid(n_tasks * (width / 32)), block(256);
  unsigned short *ri = (unsigned short*)row_img, *ti = (unsigned short*)tr_img;
  hipStream_t st = (hipStream_t)stream;
  if (width == 256 && !one_term) hipLaunchKernelGGL((npf::x6_task_images_kernel<256, 3>), grid, block, 0, st, src, tiles, pts, ri, ti);
  else if (width == 256) hipLaunchKernelGGL((npf::x6_task_images_kernel<256, 1>), grid, block, 0, st, src, tiles, pts, ri, ti);
  else if (!one_term) hipLaunchKernelGGL((npf::x6_task_images_kernel<128, 3>), grid, block, 0, st, src, tiles, pts, ri, ti);
  else hipLaunchKernelGGL((npf::x6_task_images_kernel<128, 1>), grid, block, 0, st, src, tiles, pts, ri, ti);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_x6_task_images(const float* src, int32_t n_tasks, int32_t pts, int32_t width, void* row_img, void* tr_img,
                                  void* stream) {
  return xp_task_images(src, n_tasks, pts, width, row_img, tr_img, false, stream);
}

extern "C" int npf_b16_task_images(const float* src, int32_t n_tasks, int32_t pts, int32_t width, void* row_img, void* tr_img,
                                   void* stream) {
  return xp_task_images(src, n_tasks, pts, width, row_img, tr_img, true, stream);
}
