// b16 programs (npf_b16_run): the x6 programs' interpreter for the bf16 compute mode (BASELINE config 3) -- whole sides of the
// model as one launch, every multiply ONE bf16 product with fp32 accumulation.
//
// Same program format as csrc/x6_kernel.hip (npf_x6_op_t; the reference's terms: x-encoder npf/architectures/mlp.py:95-109,
// DotAttender attention.py:129-164,204-220, MergeFlatInputs + decoder encoders.py:175-183, base.py:327-367) and the same
// register-resident activation; the arithmetic is the bf16 mode's (oracle/npf_oracle.py _LinearBf16 / _ScaledotBf16):
// the input of every multiply is rounded to bf16 (nearest even, v_cvt_pk_bf16_f32), the weights are bf16 images, accumulation,
// bias, addend, ReLU and softmax are fp32.  Tensors that only feed the backward pass are PT16 (bf16 tiles, half the bytes):
//   store_in   <- bf16(cur) as a PT16 tensor   (NPF_X6_STORE_IN_F32: cur itself as a PT32 tensor -- a dZ that is also an fp32 addend)
//   store_out  <- bf16(cur) as a PT16 tensor   (NPF_X6_STORE_OUT_F32: cur itself, PT32 -- what a later addend / the model reads)
//   sbwd_p     =  PT16 (the saved probabilities ARE bf16(P))
// Against the three-term kernel a slab carries a third of the bytes per output row and a sixth of the matrix instructions: a slab
// holds 16 RB output rows (one barrier per RB 16-row blocks), a wave keeps nine LDS reads in flight ahead of its matrix
// instructions, the slab's output blocks are stored one slab later (a slab ahead of the barrier that waits for them), and every
// multiply's bias row is staged in LDS once per workgroup.  Eight waves of 16 points share one ring of three slabs.
#include "x6_args.hpp"

// Timing-only diagnostic builds (tools/fastbuild.sh b16_kernel OUT.so -DBP_NO_...): each removes one ingredient, results are
// garbage, only the clock counts.  None is defined in the library build.
//   BP_NO_MFMA no matrix instructions   BP_NO_FRAG no fragment / bias reads from LDS   BP_NO_DMA no slab DMA
//   BP_NO_STORE no global stores of the ops   BP_NO_BARRIER no slab barriers / counted waits   BP_NO_EPI no block epilogues
//   BP_STORE_LOCAL the PT16 stores of every workgroup land on the first 64 tiles (the instructions stay, HBM sees few of them)

namespace npf {

#ifdef NPF_STAMPS
// Diagnostic build only (tools/stamp_probe_b16.py): per-phase cycle sums of wave 0 of workgroup 0, in a buffer nothing else reads.
__device__ unsigned long long g_stamps_b16[16];
__device__ __forceinline__ unsigned long long bp_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define BP_STAMP(i) { const unsigned long long t__ = bp_stamp(); st_sum[i] += t__ - st_last; st_last = t__; }
#else
#define BP_STAMP(i)
#endif

template <int KF, int RB, int NW_, int SLOTS>
struct BpGeom {
  static constexpr int NW = NW_;
  static constexpr int NB = KF / 16;        // 16-feature blocks of an activation
  static constexpr int KS = KF / 32;        // k-steps per block row
  static constexpr int NS = NB / RB;        // slabs per multiply
  static constexpr int RowB = KF * 2;       // bytes of an image row
  static constexpr int SlabB = 16 * RB * RowB;
  static constexpr int Slots = SLOTS;       // ring slots: Slots - 1 slabs in flight ahead of the one being multiplied
  static constexpr int RPP = 1024 / RowB;   // rows per 1 KiB DMA piece
  static constexpr int NPW = 16 * RB / RPP / NW;  // pieces per wave and slab
  static constexpr int BiasB = NPF_X6_MAX_OPS * KF * 4;  // the bias rows of every multiply of the program
  static_assert(NB % RB == 0 && RB % 2 == 0 && NPW >= 1 && (16 * RB / RPP) % NW == 0, "slab geometry");
  static_assert(NS >= 2 && SLOTS >= 2 && SLOTS <= 4 && (NW_ == 4 || NW_ == 8), "ring geometry");
};

// round to bf16 and back (a value as the next multiply sees it)
__device__ __forceinline__ void bp_pack(const f32x4& lo, const f32x4& hi, xp_u32x4& t) {
  t[0] = xp_cvt_pk(lo[0], lo[1]);
  t[1] = xp_cvt_pk(lo[2], lo[3]);
  t[2] = xp_cvt_pk(hi[0], hi[1]);
  t[3] = xp_cvt_pk(hi[2], hi[3]);
}
__device__ __forceinline__ void bp_unpack(const xp_u32x4& t, f32x4& lo, f32x4& hi) {
  lo[0] = __uint_as_float(t[0] << 16);
  lo[1] = __uint_as_float(t[0] & 0xffff0000u);
  lo[2] = __uint_as_float(t[1] << 16);
  lo[3] = __uint_as_float(t[1] & 0xffff0000u);
  hi[0] = __uint_as_float(t[2] << 16);
  hi[1] = __uint_as_float(t[2] & 0xffff0000u);
  hi[2] = __uint_as_float(t[3] << 16);
  hi[3] = __uint_as_float(t[3] & 0xffff0000u);
}

// NPG = 16-point groups per wave (1 in every instance the library launches: a wave owns half a tile; 2 = a whole tile).  NW_ =
// waves per workgroup sharing one slab ring, SLOTS its slots: <.., RB = 4, 8, 3> one workgroup per CU; <.., RB = 2, 4, 3> two
// workgroups per CU, out of step with each other.
template <int KF, int NPG, int RB, int NW_, int SLOTS>
__global__ __launch_bounds__(NW_ * 64, (NW_ == 4 && NPG == 1) ? 2 : 1) void b16_program_kernel(const XpArgs a) {
  using G = BpGeom<KF, RB, NW_, SLOTS>;
  constexpr int NB = G::NB, KS = G::KS, NS = G::NS, NW = G::NW, Slots = G::Slots, NPW = G::NPW;
  constexpr int LA = Slots - 1;             // slabs in flight ahead
  constexpr int TPW = NPG * NW / 2;         // tiles per workgroup
  static_assert(NB <= 16, "a lane's ReLU bits are one 64-bit word");
  __shared__ __attribute__((aligned(16))) char smem[Slots * G::SlabB + G::BiasB + KF * 16 + kXpTableBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  int bid = blockIdx.x;
  if (a.xcd_remap) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
  const int w_tile = NPG == 2 ? wave : (wave >> 1);
  const int half0 = NPG == 2 ? 0 : (wave & 1);
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
#ifdef BP_NO_STORE
  valid = valid && a.n_ops > 1000;  // (never true: the stores stay in the code, none executes)
#endif
  // (a wave without a tile still streams slabs and meets barriers; it loads tile 0 and stores nothing)
  const long tile = valid ? (long)task * a.tiles_per_task + t_in : 0;
  if (a.wgs_per_task == 0) task = 0;  // (flat launches share every weight)
  // PT32: block b of this lane's point in point group pg at float tile * KF * 32 + b * 512 + 64 pg, + lane bytes
  // PT16: row 4 st + g (blocks 2 st, 2 st + 1), 16 bytes per point: byte tile * KF * 64 + st * 2048 + 256 pg, + lane bytes
  const size_t tile_off = (size_t)tile * (KF * 32);
  const unsigned lane_b = (unsigned)(((16 * half0 + p) * 4 + g * 128) * 4);
  const unsigned lane16 = (unsigned)((g * 32 + 16 * half0 + p) * 16);
  const size_t bits_off = ((size_t)tile * 2 + half0) * 64;
  const size_t row_off = (size_t)tile * 32 + 16 * half0;
  auto pt32 = [&](const float* base, int pg, int b) -> f32x4* {
    return (f32x4*)((char*)const_cast<float*>(base + tile_off + (size_t)(b * 512 + 64 * pg)) + lane_b);
  };
  auto pt16 = [&](const float* base, int pg, int st) -> xp_u32x4* {
#ifdef BP_STORE_LOCAL  // (timing only: every workgroup's PT16 tensors land on the first 64 tiles -- the stores stay, HBM sees few)
    return (xp_u32x4*)((char*)const_cast<float*>(base) + (tile_off & (size_t)(63 * KF * 32)) * 2 + (size_t)(st * 2048 + 256 * pg) + lane16);
#endif
    return (xp_u32x4*)((char*)const_cast<float*>(base) + tile_off * 2 + (size_t)(st * 2048 + 256 * pg) + lane16);
  };

  // slab DMA: 16 RB / RPP pieces of 1 KiB, piece q = rows RPP q ..; wave w takes q = w + NW n.  Chunk c of row r sits at position
  // c ^ (r & 15) of its row (the swizzle is applied to the SOURCE address)
  constexpr int LPR = 64 / G::RPP;  // lanes per row of a piece
  const int l_row = lane / LPR, l_pos = lane % LPR;
  const int n_slabs = a.n_mm * NS;
  // (the op table and the images of the multiplies, from LDS: x6_args.hpp)
  const unsigned table0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + Slots * G::SlabB + G::BiasB + KF * 16);
  auto mm_base = [&](int j) -> const char* {
    typedef const __attribute__((address_space(1))) char* gc_t;
    const unsigned at = table0 + NPF_X6_MAX_OPS * kXpOpDwords * 4 + (unsigned)j * 8;
    const unsigned long long img = xp_lds_u64(at), stride = xp_lds_u64(at + NPF_X6_MAX_OPS * 8);
    return (const char*)(gc_t)(img + (unsigned long long)task * stride);
  };
  // piece n of this wave's share of the slab whose first image row is ``rows``
  auto dma_piece = [&](const char* img, int rows, char* slot, int n) {
#ifdef BP_NO_DMA
    return;
#endif
    const int r0 = G::RPP * (wave + NW * n);
    const char* base = img + (size_t)(rows + r0) * G::RowB;
    asm volatile("" : "+s"(base));
    const unsigned dma_lane = (unsigned)(l_row * G::RowB + ((l_pos ^ ((r0 + l_row) & 15)) << 4));
    xp_dma16(base + dma_lane, slot + r0 * G::RowB);
  };
  auto dma_slab = [&](int S, char* slot) {
    const char* img = mm_base(S / NS);
#pragma unroll
    for (int n = 0; n < NPW; ++n) dma_piece(img, (S % NS) * 16 * RB, slot, n);
  };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lane_row = (unsigned)(p * G::RowB);
  unsigned fx[KS];  // this lane's chunk of row p per k-step, from the ring's first byte
#pragma unroll
  for (int st = 0; st < KS; ++st) fx[st] = lds0 + lane_row + (unsigned)((((4 * st + g) ^ p)) << 4);
  float* bias_lds = (float*)(smem + Slots * G::SlabB);
  f32x4* out_w_lds = (f32x4*)(smem + Slots * G::SlabB + G::BiasB);

  xp_stage_ops(a, (unsigned*)(smem + Slots * G::SlabB + G::BiasB + KF * 16), tid, NW * 64);
  __syncthreads();
  if (n_slabs > 0) dma_slab(0, smem);
  if (LA > 1 && n_slabs > 1) dma_slab(1, smem + G::SlabB);
  if (LA > 2 && n_slabs > 2) dma_slab(2, smem + 2 * G::SlabB);
  // every multiply's bias row goes to LDS once, here (read back per 16-row block behind the barriers of the slab loop), and so does
  // the matrix of the F -> 4 layer behind the program: no op starts by waiting for a global load.  All loads first, then the stores.
  {
    constexpr int NE = NPF_X6_MAX_OPS * KF / (NW * 64);
    float v[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = tid + k * (NW * 64), j = e / KF, i = e % KF;
      const float* b = j < a.n_mm ? a.mm_bias[j] : nullptr;
      v[k] = b != nullptr ? b[(size_t)task * a.mm_bias_stride[j] + i] : 0.f;
    }
    f32x4 w = {0.f, 0.f, 0.f, 0.f};
    if (a.out_rows != nullptr && tid < KF) w = ((const f32x4*)a.out_w)[tid];
#pragma unroll
    for (int k = 0; k < NE; ++k) bias_lds[tid + k * (NW * 64)] = v[k];
    if (tid < KF) out_w_lds[tid] = w;
  }
  if (n_slabs == 0) __syncthreads();  // (otherwise the first slab's barrier publishes them)
  int slot = 0, S0 = 0, jm = 0;
  f32x4 cur[NPG][NB];
#pragma unroll
  for (int pg = 0; pg < NPG; ++pg)
#pragma unroll
    for (int b = 0; b < NB; ++b) cur[pg][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef NPF_STAMPS
  unsigned long long st_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = bp_stamp();
  const unsigned long long st_first = st_last;
#endif
  for (int l = 0; l < a.n_ops; ++l) {
    const npf_x6_op_t o = xp_lds_op(table0, l);
    const int oflags = o.reserved[0];
    BP_STAMP(0)  // loop back edge, op fields
    // ---------------------------------------------------------------- input side
#pragma unroll
    for (int pg = 0; pg < NPG; ++pg) {
      if (o.in_pt != nullptr) {
#pragma unroll
        for (int b = 0; b < NB; ++b) cur[pg][b] = *pt32(o.in_pt, pg, b);
      }
      if (o.in_rows != nullptr) {
        // cur <- [relu](in_w^T rows + in_b), fp32 FMAs (the caller hands bf16-rounded rows and matrix: products of two bf16
        // values are exact in fp32)
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
      if (o.mask_bits != nullptr) {
        const unsigned long long w = o.mask_bits[bits_off + 64 * pg + lane];
        const int wl = (int)(unsigned)w, wh = (int)(unsigned)(w >> 32);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int m = __builtin_amdgcn_sbfe(b < 8 ? wl : wh, 4 * (b & 7) + e, 1);
            const float x = cur[pg][b][e];
            cur[pg][b][e] = __int_as_float(__float_as_int(x) & m);
          }
      }
      if (o.sbwd_p != nullptr) {
        // softmax backward with the SAVED probabilities, a PT16 tensor: dS = scale * P16 * (dP - sum_c dP_c P16_c)
        f32x4 P[NB];
        float dot = 0.f;
#pragma unroll
        for (int st = 0; st < KS; ++st) {
          const xp_u32x4 t = *pt16(o.sbwd_p, pg, st);
          bp_unpack(t, P[2 * st], P[2 * st + 1]);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) dot = fmaf(cur[pg][b][e], P[b][e], dot);
        dot = xp_sum4(dot);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) cur[pg][b][e] = o.sbwd_scale * P[b][e] * (cur[pg][b][e] - dot);
      }
      if (o.store_in != nullptr && valid) {
        if (oflags & NPF_X6_STORE_IN_F32) {
#pragma unroll
          for (int b = 0; b < NB; ++b) __builtin_nontemporal_store(cur[pg][b], pt32(o.store_in, pg, b));
        } else {
#pragma unroll
          for (int st = 0; st < KS; ++st) {
            xp_u32x4 t;
            bp_pack(cur[pg][2 * st], cur[pg][2 * st + 1], t);
            __builtin_nontemporal_store(t, pt16(o.store_in, pg, st));
          }
        }
      }
      if (o.store_in_bits != nullptr && valid) {
        unsigned wlo = 0u, whi = 0u;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) (b < 8 ? wlo : whi) |= (unsigned)(cur[pg][b][e] > 0.f) << (4 * (b & 7) + e);
        o.store_in_bits[bits_off + 64 * pg + lane] = ((unsigned long long)whi << 32) | wlo;
      }
    }
    BP_STAMP(1)  // input side (loads, prologue, mask, softmax backward, stores)
    if (o.w_img == nullptr) continue;

    // ---------------------------------------------------------------- the multiply
    // the input rounded to bf16 (the B operands), once per op
    xp_u32x4 tb[NPG][KS];
#pragma unroll
    for (int pg = 0; pg < NPG; ++pg)
#pragma unroll
      for (int st = 0; st < KS; ++st) bp_pack(cur[pg][2 * st], cur[pg][2 * st + 1], tb[pg][st]);
    // an addend waits in the registers of the blocks it will be added to (the input is dead once it is packed)
    const bool has_add = o.addend != nullptr;
    if (has_add) {
#pragma unroll
      for (int pg = 0; pg < NPG; ++pg)
#pragma unroll
        for (int b = 0; b < NB; ++b) cur[pg][b] = *pt32(o.addend, pg, b);
    }
    BP_STAMP(2)  // pack, addend loads
    const char* const img0 = mm_base(jm);
    const char* const img1 = mm_base(jm + 1 < a.n_mm ? jm + 1 : jm);
    const bool post = o.softmax_n > 0;  // (the stores then follow the softmax)
    const float* const out = (o.store_out != nullptr && valid && !post) ? o.store_out : nullptr;  // (wave-uniform)
    const bool out32 = (oflags & NPF_X6_STORE_OUT_F32) != 0;
    const unsigned bias_l = lds0 + Slots * G::SlabB + jm * (KF * 4) + g * 16;
    const bool relu = o.relu != 0;
    unsigned pos_lo[NPG], pos_hi[NPG];
#pragma unroll
    for (int pg = 0; pg < NPG; ++pg) pos_lo[pg] = pos_hi[pg] = 0u;
    // the RB output blocks of slab s leave for store_out (fp32 blocks, or pairs of blocks as PT16 rows)
    auto store_blocks = [&](int s) {
      if (out == nullptr) return;
#pragma unroll
      for (int pg = 0; pg < NPG; ++pg) {
        if (out32) {
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) __builtin_nontemporal_store(cur[pg][s * RB + rb], pt32(out, pg, s * RB + rb));
        } else {
#pragma unroll
          for (int h = 0; h < RB / 2; ++h) {
            const int st = (s * RB) / 2 + h;
            xp_u32x4 t;
            bp_pack(cur[pg][2 * st], cur[pg][2 * st + 1], t);
            __builtin_nontemporal_store(t, pt16(out, pg, st));
          }
        }
      }
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int S = S0 + s;
      // slab S has landed for everyone, everyone is done with slab S - 1 (whose slot slab S + LA goes into).  The pieces of the
      // LA - 1 later slabs may stay in flight (vector-memory operations retire in order; this wave's stores are issued BEFORE the
      // pieces at the same point, so by now they are a slab old)
      // (an op's first barrier: its addend -- NB NPG loads, the newest vector-memory operations of this wave -- may stay in flight
      // too; the blocks' epilogues wait for what they add)
#ifdef BP_NO_BARRIER
      if (false) {}
      else
#endif
      if (s == 0 && has_add && NB * NPG <= 32) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NB * NPG) : "memory");
      else if (LA > 2 && S + 2 < n_slabs) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((LA - 1) * NPW) : "memory");
      else if (LA > 1 && S + 1 < n_slabs) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NPW) : "memory");
#ifdef BP_NO_BARRIER
      else if (false) {}
#endif
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      BP_STAMP(3)  // counted wait + barrier
      const bool more = S + LA < n_slabs;
      char* const nslot = smem + ((slot + LA) % Slots) * G::SlabB;
      // HERE none of this wave's LDS reads is outstanding (csrc/x6_kernel.hip: vector-memory instructions issued while fragment
      // reads were in flight gave sporadic wrong results); from here to the end of the slab the wave keeps LDS reads in flight.
      // So the previous slab's output blocks are stored now -- a whole slab ahead of the barrier that waits for them -- and then
      // slab S + LA is sent into the slot slab S - 1 has left.
      if (s > 0) store_blocks(s - 1);
      if (more) {
#pragma unroll
        for (int n = 0; n < NPW; ++n) dma_piece(s + LA < NS ? img0 : img1, ((s + LA) % NS) * 16 * RB, nslot, n);
      }
      BP_STAMP(4)  // stores of the previous slab, slab pieces
      // The slab's LDS reads in order: per 16-row block its bias row, then its KS weight fragments.  W of them are in flight ahead
      // of the matrix instructions (a ring of W + 1 registers: a read is issued into the register consumed one step earlier);
      // LDS reads return in order, so read j has landed once at most min(W - 1, NR - 1 - j) later ones are outstanding.
      constexpr int W = 9, NR = RB * (KS + 1);
      xp_u32x4 ring[W + 1];
      // (addresses: one register per k-step for the slab -- the lane's swizzled chunk of row p -- and the 16-row block / the bias row as
      // the instruction's immediate offset: one vector add per k-step and slab instead of one per read)
      unsigned fb[KS];
#pragma unroll
      for (int st = 0; st < KS; ++st) fb[st] = fx[st] + (unsigned)(slot * G::SlabB);
      auto rd = [&](int j) {
        const int rb = j / (KS + 1), q = j % (KS + 1);
        xp_u32x4& dst = ring[j % (W + 1)];
#ifdef BP_NO_FRAG
        asm volatile("" : "=v"(dst) : "v"(fb[0]));
#else
#define BP_RD(N, BASE, OFF) case N: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(BASE), "n"(OFF)); break
        if (q == 0) {
          switch (s * RB + rb) {
            BP_RD(0, bias_l, 0); BP_RD(1, bias_l, 64); BP_RD(2, bias_l, 128); BP_RD(3, bias_l, 192); BP_RD(4, bias_l, 256);
            BP_RD(5, bias_l, 320); BP_RD(6, bias_l, 384); BP_RD(7, bias_l, 448); BP_RD(8, bias_l, 512); BP_RD(9, bias_l, 576);
            BP_RD(10, bias_l, 640); BP_RD(11, bias_l, 704); BP_RD(12, bias_l, 768); BP_RD(13, bias_l, 832); BP_RD(14, bias_l, 896);
            BP_RD(15, bias_l, 960);
          }
        } else {
          switch (rb) {
            BP_RD(0, fb[q - 1], 0); BP_RD(1, fb[q - 1], 16 * G::RowB); BP_RD(2, fb[q - 1], 32 * G::RowB); BP_RD(3, fb[q - 1], 48 * G::RowB);
          }
        }
#undef BP_RD
#endif
      };
#pragma unroll
      for (int j = 0; j < W; ++j) rd(j);
      // one accumulator per point group (a second one for the odd k-steps -- no back-to-back dependent matrix instructions --
      // measured 1 % slower: its sum is two more vector instructions per block, and the sibling wave fills the gaps)
      constexpr int NA = 1;
      f32x4 acc[NPG][NA];
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int rb = j / (KS + 1), q = j % (KS + 1), sb = s * RB + rb;
        xp_u32x4& f = ring[j % (W + 1)];
        switch (NR - 1 - j < W - 1 ? NR - 1 - j : W - 1) {
#define BP_WAIT(N) case N: asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f)); break
          BP_WAIT(0); BP_WAIT(1); BP_WAIT(2); BP_WAIT(3); BP_WAIT(4); BP_WAIT(5); BP_WAIT(6); BP_WAIT(7); BP_WAIT(8);
          BP_WAIT(9); BP_WAIT(10); BP_WAIT(11); BP_WAIT(12); BP_WAIT(13); BP_WAIT(14);
#undef BP_WAIT
        }
        if (q == 0) {
#pragma unroll
          for (int pg = 0; pg < NPG; ++pg) {
            acc[pg][0] = __builtin_bit_cast(f32x4, f);  // (the bias row is the same for every point group)
            if (NA == 2) acc[pg][NA - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        } else {
          const int st = q - 1;
#pragma unroll
          for (int pg = 0; pg < NPG; ++pg)
#ifdef BP_NO_MFMA
            asm volatile("" : "+v"(acc[pg][st & (NA - 1)]) : "v"(f), "v"(tb[pg][st]));
#else
            acc[pg][st & (NA - 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(xp_bf16x8, f), __builtin_bit_cast(xp_bf16x8, tb[pg][st]), acc[pg][st & (NA - 1)], 0, 0, 0);
#endif
        }
        if (j + W < NR) rd(j + W);
#ifdef BP_NO_EPI
        if (q == KS && a.n_ops > 1000) {
#else
        if (q == KS) {
#endif
#pragma unroll
          for (int pg = 0; pg < NPG; ++pg) {
            f32x4 r = NA == 2 ? acc[pg][0] + acc[pg][NA - 1] : acc[pg][0];
            if (has_add) {  // (the empty asm keeps this a wave-uniform branch: as a select it costs every op four instructions per block)
              asm volatile("" ::: "memory");
              r += cur[pg][sb];
            }
            if (relu) {  // (v_max_f32 as is: fmaxf would canonicalise its operand first, a second instruction per value)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float x = r[e];  // (through a scalar: see the note on ext_vector elements in csrc/x6_kernel.hip)
                asm("v_max_f32 %0, 0, %0" : "+v"(x));
                r[e] = x;
              }
            }
            cur[pg][sb] = r;  // (block sb of the input is dead: it is in tb)
#pragma unroll
            for (int e = 0; e < 4; ++e) (sb < 8 ? pos_lo[pg] : pos_hi[pg]) |= (unsigned)(r[e] > 0.f) << (4 * (sb & 7) + e);
          }
        }
      }
      BP_STAMP(5)  // reads, matrix instructions, epilogues
      slot = (slot + 1) % Slots;
    }
    store_blocks(NS - 1);
    BP_STAMP(6)  // the last slab's stores
    S0 += NS;
    ++jm;
    // ---------------------------------------------------------------- output side behind the last slab
    if (post) {
      // softmax over the first softmax_n features (keys) of scale * cur, fp32 with max subtraction (attention.py:158-164)
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
            cur[pg][b][e] = expf(cur[pg][b][e] - mx);
            sum += cur[pg][b][e];
          }
        sum = xp_sum4(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) cur[pg][b][e] *= inv;
        if (o.store_out != nullptr && valid) {
          if (out32) {
#pragma unroll
            for (int b = 0; b < NB; ++b) __builtin_nontemporal_store(cur[pg][b], pt32(o.store_out, pg, b));
          } else {
#pragma unroll
            for (int st = 0; st < KS; ++st) {
              xp_u32x4 t;
              bp_pack(cur[pg][2 * st], cur[pg][2 * st + 1], t);
              __builtin_nontemporal_store(t, pt16(o.store_out, pg, st));
            }
          }
        }
      }
    }
    if (o.store_bits != nullptr && valid) {
#pragma unroll
      for (int pg = 0; pg < NPG; ++pg) o.store_bits[bits_off + 64 * pg + lane] = ((unsigned long long)pos_hi[pg] << 32) | pos_lo[pg];
    }
  }

  BP_STAMP(7)  // softmax / ReLU bits behind the last op
  if (a.out_rows != nullptr) {
    // an F -> 4 layer on the registers the program leaves (the decoder's output layer, mlp.py:109), its input rounded to bf16
    // like every layer's (the caller hands the bf16-rounded matrix): fp32 dot products, summed over the point's four lanes
    const f32x4* wl = out_w_lds;  // (staged at the start of the workgroup, behind every barrier since)
#pragma unroll
    for (int pg = 0; pg < NPG; ++pg) {
      f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        xp_u32x4 t;
        f32x4 x[2];
        bp_pack(cur[pg][2 * st], cur[pg][2 * st + 1], t);
        bp_unpack(t, x[0], x[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const f32x4 w = wl[n * (KF / 4) + 4 * (2 * st + h) + g];
#pragma unroll
            for (int e = 0; e < 4; ++e) r[n] = fmaf(w[e], x[h][e], r[n]);
          }
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        r[n] = xp_sum4(r[n]);
        if (a.out_b != nullptr) r[n] += a.out_b[n];
      }
      if (valid && g == 0) ((f32x4*)a.out_rows)[row_off + 16 * pg + p] = r;
    }
  }
#ifdef NPF_STAMPS
  BP_STAMP(8)  // the F -> 4 layer
  if (blockIdx.x == 0 && tid == 0) {
    for (int i = 0; i < 10; ++i) g_stamps_b16[i] = st_sum[i];
    g_stamps_b16[10] = st_last - st_first;
  }
#endif
}

}  // namespace npf

#ifdef NPF_STAMPS
extern "C" int npf_debug_stamps_b16(unsigned long long* host16) {
  return hipMemcpyFromSymbol(host16, HIP_SYMBOL(npf::g_stamps_b16), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int npf_b16_run(const npf_x6_op_t* ops, int32_t n_ops, const float* out_w, const float* out_b, float* out_rows,
                           int32_t n_tasks, int32_t tiles_per_task, int32_t per_task, int32_t width, int32_t variant, void* stream) {
  if (width != 256 && width != 128) return NPF_EINVAL;
  if (variant < 0 || variant > 2) return NPF_EINVAL;
  npf::XpArgs a;
  const int rc = npf::xp_fill_args(ops, n_ops, out_w, out_b, out_rows, n_tasks, tiles_per_task, tiles_per_task * 32, per_task, width,
                                   NPF_X6_STORE_IN_F32 | NPF_X6_STORE_OUT_F32, 256, a);
  if (rc != NPF_OK) return rc;
  for (int l = 0; l < n_ops; ++l)
    if (ops[l].mask != nullptr) return NPF_EINVAL;  // (ReLU masks are bits here)
  // 1 = eight waves of 16 points share a ring of three 64-row slabs, one workgroup per CU (the library's choice); 2 = four waves
  // of 16 points and a ring of three 32-row slabs, two workgroups per CU, out of step with each other.  (Measured on config 3,
  // target side forward: 2.64 ms (1), 2.81 (2); 32 points per wave -- every weight fragment feeding two matrix instructions --
  // 4.17 at two waves per SIMD (256 registers do not hold it: spills in the slab loop) and 3.35 at one wave per SIMD with 404
  // registers; a ring of four slabs, or 3 / 5 / 13 LDS reads ahead instead of 9: no change.  DESIGN.md 8.1.)
  const int var = variant == 0 ? 1 : variant;
  const int tpw = var == 2 ? 2 : 4;
  a.wgs_per_task = per_task ? (tiles_per_task + tpw - 1) / tpw : 0;
  const int n_wg = per_task ? n_tasks * a.wgs_per_task : (a.total_tiles + tpw - 1) / tpw;
  a.xcd_remap = (per_task && (n_wg % 8) == 0 && a.wgs_per_task > 1) ? 1 : 0;
  const dim3 grid(n_wg);
  hipStream_t st = (hipStream_t)stream;
  if (width == 256) {
    if (var == 2) hipLaunchKernelGGL((npf::b16_program_kernel<256, 1, 2, 4, 3>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((npf::b16_program_kernel<256, 1, 4, 8, 3>), grid, dim3(512), 0, st, a);
  } else {
    if (var == 2) hipLaunchKernelGGL((npf::b16_program_kernel<128, 1, 2, 4, 3>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((npf::b16_program_kernel<128, 1, 4, 8, 3>), grid, dim3(512), 0, st, a);
  }
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}
