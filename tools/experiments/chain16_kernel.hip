// bf16 chain interpreter for gfx950 (MI355X): the bf16 compute mode of npf_chain_run (prog->reserved[2] == 1).
//
// Same program format and the same register-resident "transposed chain" idea as chain_kernel.hip -- every layer
// computes Y^T[n][p] = sum_k W[n][k] X^T[k][p] with the weights as the MFMA A operand (streamed through LDS) and the
// previous layer's accumulators as the B operand -- but laid out for v_mfma_f32_16x16x32_bf16, whose 16 cycles per
// instruction leave no room for the per-slab overheads the fp32 kernel hides behind 32-cycle fp32 MFMAs:
//
//   * one wavefront owns kH = 4 "halves" of 16 points (two PT tiles; column = lane & 15, the accumulator layout of
//     the fp32 kernel four times): every 16-byte weight fragment read from LDS feeds FOUR MFMAs, i.e. a quarter of
//     the LDS bytes, slab DMA and barriers per point of a 16-point wave (which needs one ds_read_b128 per 16-cycle
//     MFMA = 256 B/clk per CU, the whole LDS bandwidth);
//   * the layer runs IN PLACE: its input is packed to bf16 once (8 registers per 32 features and half), the fp32
//     accumulators of its output then overwrite the fp32 registers of its input: 256 accumulator + 128 packed
//     registers per lane.  That is a 512-register kernel: ONE wave per SIMD, one 256-point workgroup per CU.  The
//     two-waves-per-SIMD form of the same design (32 points per wave, 128 + 64 + fragments ~ 240 registers) was
//     built first: hipcc spilled 700-1200 registers there, and every spill reload waits on vmcnt, i.e. on the slab
//     DMA in flight.  With one wave per SIMD the latencies are hidden inside the wave instead: fragment reads run
//     two k-steps (16 MFMAs = 256 cycles) ahead, the slab DMA two to three slabs (2-3 x 1024 cycles) ahead;
//   * weight slabs (32 rows x K bf16 <= 16 KiB) stream by LDS-DMA through an eight-slot ring, retired by a counted
//     s_waitcnt vmcnt(n) -- never a drain -- in front of the one raw s_barrier per slab.
//
// Replaces, in the bf16 mode, the same reference functions as chain_kernel.hip: MLP.forward
// (npf/architectures/mlp.py:95-109), MergeFlatInputs.forward (encoders.py:175-183), BaseAttender.forward /
// DotAttender.score (attention.py:129-164,204-220), merge_r_z (neuralproc/base.py:554-575) and their autograd.
#include "chain16.hpp"

namespace npf16 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 4;                  // waves per workgroup (one per SIMD)
constexpr int kH = 2;                      // 16-point halves per wave
constexpr int kTW = kH / 2;                // PT tiles per wave
constexpr int kTilesPerWG = kWaves * kTW;
constexpr int kSlabRows = 32;
constexpr int kSlots = 8;                  // slab ring
constexpr int kAhead = 3;                  // slabs in flight ahead of the one being multiplied (<= kSlots - 1)
constexpr int kImageBytes = kSlabRows * 512;      // 32 rows x 256 bf16
constexpr int kSlotBytes = kImageBytes + 256;     // + the 32 biases of the slab (64 floats written)
constexpr int kMaxB = 16;                  // 16-feature blocks per point (<= 256 features)
constexpr int kMaxS = kMaxB / 2;           // 32-feature k-steps

__device__ __attribute__((aligned(256))) float g_zero[64] = {};  // source of zero chunks / "no bias"

struct Wv {
  int lane, p, g, wave;   // p = lane & 15 (point inside a half), g = lane >> 4 (k group); wave is uniform
  // uniform, per tile of the wave (half h lives in tile h >> 1): its task, its index inside the task, does it exist
  int task[kTW], tin[kTW];
  bool tv[kTW];
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }  // pin a wave-uniform value to an SGPR

__device__ __forceinline__ int eff_task(int task, int modulus) { return modulus > 0 ? (task % modulus) : task; }

__device__ __forceinline__ bf16x8 pack_bf16(const f32x4& lo, const f32x4& hi) {
  const bf16x2 p0 = __builtin_convertvector((f32x2{lo[0], lo[1]}), bf16x2), p1 = __builtin_convertvector((f32x2{lo[2], lo[3]}), bf16x2);
  const bf16x2 p2 = __builtin_convertvector((f32x2{hi[0], hi[1]}), bf16x2), p3 = __builtin_convertvector((f32x2{hi[2], hi[3]}), bf16x2);
  bf16x8 r;
  r[0] = p0[0]; r[1] = p0[1]; r[2] = p1[0]; r[3] = p1[1];
  r[4] = p2[0]; r[5] = p2[1]; r[6] = p3[0]; r[7] = p3[1];
  return r;
}
__device__ __forceinline__ f32x4 pt16_lo(const u32x4& r) {
  return f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xffff0000u),
               __builtin_bit_cast(float, r[1] << 16), __builtin_bit_cast(float, r[1] & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 pt16_hi(const u32x4& r) {
  return f32x4{__builtin_bit_cast(float, r[2] << 16), __builtin_bit_cast(float, r[2] & 0xffff0000u),
               __builtin_bit_cast(float, r[3] << 16), __builtin_bit_cast(float, r[3] & 0xffff0000u)};
}

// A PT operand of an elementwise op / epilogue -- an fp32 PT32 tensor ([F/4][32 points][4 features] per tile) or a PT16
// tensor ([F/8 rows][32 points][8 bf16], row 4 s + g = features {32 s + 4 g + i}, {32 s + 16 + 4 g + i}) -- read or
// written one block pair (32 features) of one half at a time.  Both formats hold F * 32 elements per tile.
struct PtRef {
  float* t32[kTW];           // lane pointer into each tile of the wave, half 0 (PT32)
  unsigned short* t16[kTW];  // the same for PT16
  bool is16;
};
__device__ __forceinline__ PtRef pt_ref(const void* base, bool is16, const npf_program_t& g, const Wv& w, int F, int modulus) {
  PtRef r;
  r.is16 = is16;
#pragma unroll
  for (int t = 0; t < kTW; ++t) {
    const size_t tile = (size_t)eff_task(w.task[t], modulus) * g.tiles_per_task + w.tin[t];
    r.t32[t] = (float*)base + tile * (size_t)(F * 32) + w.p * 4;
    r.t16[t] = (unsigned short*)base + (tile * (size_t)(F >> 3) * 32 + w.p) * 8 + w.g * 256;
  }
  return r;
}
__device__ __forceinline__ void pt_read_pair(const PtRef& r, const Wv& w, int st, int h, f32x4& lo, f32x4& hi) {
  if (r.is16) {
    const u32x4 v = *(const u32x4*)(r.t16[h >> 1] + 1024 * st + 128 * (h & 1));
    lo = pt16_lo(v);
    hi = pt16_hi(v);
  } else {
    const float* t = r.t32[h >> 1] + 64 * (h & 1);
    lo = *(const f32x4*)(t + (8 * st + w.g) * 128);
    hi = *(const f32x4*)(t + (8 * st + 4 + w.g) * 128);
  }
}
__device__ __forceinline__ void pt_write_pair(const PtRef& r, const Wv& w, int st, int h, const f32x4& lo, const f32x4& hi) {
  if (r.is16) {
    *(bf16x8*)(r.t16[h >> 1] + 1024 * st + 128 * (h & 1)) = pack_bf16(lo, hi);
  } else {
    float* t = r.t32[h >> 1] + 64 * (h & 1);
    *(f32x4*)(t + (8 * st + w.g) * 128) = lo;
    *(f32x4*)(t + (8 * st + 4 + w.g) * 128) = hi;
  }
}

// reductions over the 4 lane groups that share a point
__device__ __forceinline__ float xg_sum(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}
__device__ __forceinline__ float xg_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}

// ---- weight slab stream ------------------------------------------------------------------------------------------
// Every LINEAR of a bf16 program takes a bf16 image [rows][roundup(K, 32)] (row stride i3 floats, 16-byte aligned,
// optional per-task stride s0): npf_cast_bf16_weights for shared weights, STORE_WB / STORE_TRB images for the task's
// keys / values.  A slab = 32 consecutive rows, in LDS a dense [32][RB bytes] image, RB = 128 / 256 / 512 (the row
// bytes rounded up to a power of two), whose 16-byte chunks are XOR-swizzled inside each row (chunk c of row r at
// position c ^ (r & swz)); the swizzle is applied to the per-lane SOURCE address of the LDS-DMA (the LDS side of a DMA
// is linear), rows beyond N and chunks beyond the source row come from a zero buffer.  The slab's 32 biases follow at
// byte kImageBytes of the slot.
struct Stream {
  int op, nb;          // cursor: LINEAR op and slab of the next slab to issue (op == n_ops: none left)
  int issued;          // slabs issued so far (ring position)
  const char* W;       // image of the cursor's op, offset to the workgroup's task
  const float* bias;   // or nullptr
  int N, ldB, lrpp, n_slabs, n_pw;
  bool dense;          // every chunk of an LDS row exists in the source row (no zero-padding chunks)
  unsigned lo[2];      // per-lane source byte offsets of the even / odd pieces of a wave; 0xffffffff: a padding chunk
  int rl;              // the lane's row inside a piece
  int q[kAhead];       // DMA instructions per wave of the last kAhead slabs issued (q[0] = the newest)
};

__device__ __forceinline__ void dma16(const void* src, void* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const void* src, void* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 4, 0, 0);
}

// LDS row bytes of a layer with S k-steps of 32 features: 64 S rounded up to a power of two >= 128
__device__ __forceinline__ int lds_row_bytes(int S) { return S <= 2 ? 128 : (S <= 4 ? 256 : 512); }

__device__ __forceinline__ void stream_seek(Stream& s, const npf_program_t& g, const Wv& w, int wg_task) {
  while (s.op < g.n_ops && g.ops[s.op].op != NPF_OP_LINEAR) ++s.op;
  s.op = uni(s.op);
  s.nb = 0;
  if (s.op >= g.n_ops) return;
  const npf_op_t& o = g.ops[s.op];
  const int S = (o.i0 + 31) >> 5;
  const int srcB = S * 64, RB = lds_row_bytes(S);
  s.N = uni(o.i1);
  s.ldB = uni(o.i3 * 4);
  const int lcpr = RB == 512 ? 5 : (RB == 256 ? 4 : 3);   // log2 of the 16-byte chunks per row
  const int swz = RB == 128 ? 7 : 15;
  s.lrpp = uni(6 - lcpr);                                  // log2 of the rows per 1 KiB piece
  s.n_slabs = uni((s.N + kSlabRows - 1) / kSlabRows);
  s.n_pw = uni(RB >> 7);
  s.W = (const char*)o.p0 + (size_t)wg_task * o.s0 * 4;
  s.bias = o.p1 ? (const float*)o.p1 + (size_t)wg_task * o.s1 : nullptr;
  s.dense = srcB == RB;
  // piece q = wave + 4 i of a slab covers LDS bytes [1024 q, 1024 q + 1024) = rows (q << lrpp) + rl; position cpos of
  // row r holds source chunk cpos ^ (r & swz); the low 4 bits of the row repeat every 2 pieces of a wave
  s.rl = w.lane >> lcpr;
  const int cpos = w.lane & ((1 << lcpr) - 1);
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int row = ((w.wave + 4 * v) << s.lrpp) + s.rl;
    const int ch = cpos ^ (row & swz);
    s.lo[v] = ch * 16 < srcB ? (unsigned)(((w.wave << s.lrpp) + s.rl) * s.ldB + ch * 16) : 0xffffffffu;
  }
}

// Issue the cursor's slab into ring slot (issued % kSlots) and record how many DMA instructions this wave issued for
// it (n_pw + 1, the same for every wave; 0 once the stream is exhausted).
__device__ __forceinline__ void stream_issue(Stream& s, const npf_program_t& g, const Wv& w, char* smem, int wg_task) {
#pragma unroll
  for (int i = kAhead - 1; i > 0; --i) s.q[i] = s.q[i - 1];
  s.q[0] = 0;
  if (s.op >= g.n_ops) return;
  char* slot = smem + (s.issued & (kSlots - 1)) * kSlotBytes;
  const int row0 = s.nb * kSlabRows;
  const int n_pw = s.n_pw;
  const char* base = s.W + (size_t)row0 * s.ldB;
  char* dst = slot + w.wave * 1024;
  const int stepB = (4 << s.lrpp) * s.ldB;  // source bytes between a wave's consecutive pieces
  if (s.dense && row0 + kSlabRows <= s.N) {
    // full slab of a dense layer (all the heavy ones): one SALU add + one DMA instruction per piece
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < n_pw) dma16(base + (size_t)(i * stepB) + (size_t)s.lo[i & 1], dst + i * (kWaves * 1024));
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < n_pw) {
        const int n = row0 + ((w.wave + 4 * i) << s.lrpp) + s.rl;
        const unsigned off = s.lo[i & 1];
        const char* src = (n < s.N && off != 0xffffffffu) ? base + (size_t)(i * stepB) + (size_t)off : (const char*)g_zero;
        dma16(src, dst + i * (kWaves * 1024));
      }
  }
  {  // the slab's biases (every wave writes the same 64 floats: no wave-dependent branch, equal vmcnt on all waves)
    const int n = row0 + (w.lane & 31);
    const float* src = (s.bias != nullptr && n < s.N) ? s.bias + n : g_zero;
    dma4(src, slot + kImageBytes);
  }
  s.q[0] = n_pw + 1;
  s.issued = uni(s.issued + 1);
  s.nb = uni(s.nb + 1);
  if (s.nb >= s.n_slabs) {
    s.op = uni(s.op + 1);
    stream_seek(s, g, w, wg_task);
  }
}

// "All but the n youngest vector-memory operations of this wave are done", n = what was issued after the slab that has
// to have landed (<= 5 (kAhead - 1)).  Other vector-memory operations issued in between only make this wait for more.
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
#define NPF16_W(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
    NPF16_W(1) NPF16_W(2) NPF16_W(3) NPF16_W(4) NPF16_W(5) NPF16_W(6) NPF16_W(7) NPF16_W(8) NPF16_W(9) NPF16_W(10)
    NPF16_W(11) NPF16_W(12) NPF16_W(13) NPF16_W(14) NPF16_W(15)
#undef NPF16_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
// the slab about to be multiplied has landed: everything but the (kAhead - 1) slabs issued after it is done
__device__ __forceinline__ void wait_next_slab(const Stream& s) {
  int n = 0;
#pragma unroll
  for (int i = 0; i < kAhead - 1; ++i) n += s.q[i];
  wait_vm(n);
}

#ifdef NPF_STAMPS
// Diagnostic build only (tools/stamp_probe16.py): cycle sums per phase of wave 0 of workgroup 0, written to a buffer
// nothing else reads.
__device__ unsigned long long g_stamps16[16];
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define NPF16_STAMP(i) { const unsigned long long t__ = stamp(); st_sum[i] += t__ - st_last; st_last = t__; }
#define NPF16_STAMP_ARGS , unsigned long long (&st_sum)[8], unsigned long long& st_last
#define NPF16_STAMP_PASS , st_sum, st_last
#else
#define NPF16_STAMP(i)
#define NPF16_STAMP_ARGS
#define NPF16_STAMP_PASS
#endif

#define NPF16_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int I>
__device__ __forceinline__ void init_acc(const f32x4& bz0, const f32x4& bz1, f32x4 (&acc)[kH][kMaxB]) {
#pragma unroll
  for (int h = 0; h < kH; ++h) {
    acc[h][2 * I] = bz0;
    acc[h][2 * I + 1] = bz1;
  }
}
template <int I>
__device__ __forceinline__ void mfma_step(const f32x4& x0, const f32x4& x1, int st, const bf16x8 (&curb)[kH][kMaxS],
                                          f32x4 (&acc)[kH][kMaxB]) {
  const bf16x8 w0 = __builtin_bit_cast(bf16x8, x0), w1 = __builtin_bit_cast(bf16x8, x1);
#pragma unroll
  for (int h = 0; h < kH; ++h) acc[h][2 * I] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, curb[h][st], acc[h][2 * I], 0, 0, 0);
#pragma unroll
  for (int h = 0; h < kH; ++h)
    acc[h][2 * I + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, curb[h][st], acc[h][2 * I + 1], 0, 0, 0);
}

// One 32-row slab: acc[h][2 I + j] (j < 2: the slab's two 16-row blocks, h: the wave's halves) = bias + sum over the
// k-steps of 32 features.  Fragment (j, st) = chunk (4 st + g) ^ (p & swz) of slab row 16 j + p.
// Hot form: K = 256 (8 k-steps, 512-byte LDS rows).  Straight-line code: fragment reads run two k-steps ahead of the
// MFMAs (three register sets), every wait counts what may stay in flight, block j = 1 is an immediate offset.
template <int I>
__device__ __forceinline__ void slab_mfma_k256(unsigned sbase, const unsigned (&a0)[4], const unsigned bias_a,
                                               const bf16x8 (&curb)[kH][kMaxS], f32x4 (&acc)[kH][kMaxB]) {
  constexpr int kJ1 = 16 * 512;  // bytes between the slab's two 16-row blocks
  unsigned A[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) A[t] = a0[t] + sbase;
  f32x4 bz0, bz1, f0[2], f1[2], f2[2];
  const unsigned ba = bias_a + sbase;
  NPF16_READ(bz0, ba, 0);
  NPF16_READ(bz1, ba, 64);
  asm volatile("ds_read_b128 %0, %2 offset:0\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(f0[0]), "=&v"(f0[1]) : "v"(A[0]), "n"(kJ1));
  asm volatile("ds_read_b128 %0, %2 offset:0\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(f1[0]), "=&v"(f1[1]) : "v"(A[1]), "n"(kJ1));
  asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bz0), "+v"(bz1));
  init_acc<I>(bz0, bz1, acc);
#define NPF16_STEP(ST, CUR, NXT)                                                                                \
  {                                                                                                             \
    if constexpr ((ST) + 2 < 8)                                                                                 \
      asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\ts_waitcnt lgkmcnt(4)"    \
                   : "=&v"(NXT[0]), "=&v"(NXT[1]), "+v"(CUR[0]), "+v"(CUR[1])                                   \
                   : "v"(A[((ST) + 2) & 3]), "n"((((ST) + 2) >> 2) * 256), "n"((((ST) + 2) >> 2) * 256 + kJ1)); \
    else if constexpr ((ST) + 1 < 8)                                                                            \
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(CUR[0]), "+v"(CUR[1]));                                        \
    else                                                                                                        \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(CUR[0]), "+v"(CUR[1]));                                        \
    mfma_step<I>(CUR[0], CUR[1], ST, curb, acc);                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
  }
  NPF16_STEP(0, f0, f2)
  NPF16_STEP(1, f1, f0)
  NPF16_STEP(2, f2, f1)
  NPF16_STEP(3, f0, f2)
  NPF16_STEP(4, f1, f0)
  NPF16_STEP(5, f2, f1)
  NPF16_STEP(6, f0, f2)
  NPF16_STEP(7, f1, f0)
#undef NPF16_STEP
}

// Any other shape (S k-steps, RB-byte LDS rows: the skinny first / last layers, widths below 256): a guard per
// k-step, plain waits.
template <int I>
__device__ __forceinline__ void slab_mfma_any(unsigned sbase, int S, int RB, const unsigned (&a0)[4], const unsigned bias_a,
                                              const bf16x8 (&curb)[kH][kMaxS], f32x4 (&acc)[kH][kMaxB]) {
  f32x4 bz0, bz1;
  const unsigned ba = bias_a + sbase;
  NPF16_READ(bz0, ba, 0);
  NPF16_READ(bz1, ba, 64);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bz0), "+v"(bz1));
  init_acc<I>(bz0, bz1, acc);
  const unsigned j1 = (unsigned)(16 * RB);
#pragma unroll
  for (int st = 0; st < kMaxS; ++st) {
    if (st < S) {
      f32x4 x0, x1;
      const unsigned a = a0[st & 3] + sbase;
      const unsigned a1 = a + j1;
      asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(x0), "=&v"(x1)
                   : "v"(a), "v"(a1), "n"((st >> 2) * 256));
      mfma_step<I>(x0, x1, st, curb, acc);
    }
  }
}

// Stage I of a layer: put the slab kAhead further down the stream in flight, multiply slab I, wait (counted) for the
// next slab's pieces, barrier.
template <int I>
__device__ __forceinline__ void stage(Stream& s, const npf_program_t& g, const Wv& w, char* smem, unsigned smem_a, int wg_task,
                                      int& consumed, int S, int RB, const unsigned (&a0)[4], const unsigned bias_a,
                                      const bf16x8 (&curb)[kH][kMaxS], f32x4 (&acc)[kH][kMaxB] NPF16_STAMP_ARGS) {
  stream_issue(s, g, w, smem, wg_task);
  NPF16_STAMP(1)
  const unsigned sbase = smem_a + (unsigned)((consumed & (kSlots - 1)) * kSlotBytes);
  if (S == 8) slab_mfma_k256<I>(sbase, a0, bias_a, curb, acc);
  else slab_mfma_any<I>(sbase, S, RB, a0, bias_a, curb, acc);
  NPF16_STAMP(2)
  wait_next_slab(s);             // the NEXT slab has landed (this wave's pieces) ...
  NPF16_STAMP(3)
  __builtin_amdgcn_s_barrier();  // ... for everyone, and everyone is done reading this one
  consumed = uni(consumed + 1);
  NPF16_STAMP(4)
}

__global__ __launch_bounds__(64 * kWaves, 1) void chain16_kernel(const npf_program_t g) {
  __shared__ __attribute__((aligned(1024))) char smem[kSlots * kSlotBytes];

  Wv w0;
  w0.lane = threadIdx.x & 63;
  w0.p = w0.lane & 15;
  w0.g = w0.lane >> 4;
  w0.wave = uni(threadIdx.x >> 6);
  int wg_task = 0;  // per-task weights / biases: every tile of the workgroup belongs to this task
  {
    // tiles of the workgroup: inside one task (wg_per_task, required by per-task weights) or dealt flat over the batch
    int first, task0 = 0;
    if (g.wg_per_task) {
      const int wgs = (g.tiles_per_task + kTilesPerWG - 1) / kTilesPerWG;
      task0 = blockIdx.x / wgs;
      first = (blockIdx.x - task0 * wgs) * kTilesPerWG + w0.wave * kTW;
      wg_task = task0;
    } else {
      first = (blockIdx.x * kWaves + w0.wave) * kTW;
    }
#pragma unroll
    for (int t = 0; t < kTW; ++t) {
      const int ft = first + t;
      if (g.wg_per_task) {
        w0.tv[t] = ft < g.tiles_per_task;
        w0.task[t] = task0;
        w0.tin[t] = w0.tv[t] ? ft : 0;  // (keep addresses in range; loads are discarded, stores skipped)
      } else {
        w0.tv[t] = ft < g.n_tasks * g.tiles_per_task;
        w0.task[t] = w0.tv[t] ? uni(ft / g.tiles_per_task) : 0;
        w0.tin[t] = w0.tv[t] ? uni(ft - w0.task[t] * g.tiles_per_task) : 0;
      }
    }
  }

  f32x4 acc[kH][kMaxB];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < kH; ++h)
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) acc[h][b] = zero4;
  float acc_dot[kH];
#pragma unroll
  for (int h = 0; h < kH; ++h) acc_dot[h] = 0.f;

  // ---- slab stream prologue: kAhead slabs in flight, the first one landed
  Stream s;
  s.op = 0;
  s.issued = 0;
#pragma unroll
  for (int i = 0; i < kAhead; ++i) s.q[i] = 0;
  stream_seek(s, g, w0, wg_task);
  int consumed = 0;  // slabs multiplied so far
#pragma unroll
  for (int i = 0; i < kAhead; ++i) stream_issue(s, g, w0, smem, wg_task);
  wait_next_slab(s);
  __builtin_amdgcn_s_barrier();
  const unsigned smem_a = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
#ifdef NPF_STAMPS
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = stamp();
#endif

  for (int ip = 0; ip < g.n_ops; ++ip) {
    const npf_op_t& o = g.ops[ip];
    const int opc = o.op;
    // Per-op copies of the lane ids behind an empty asm: everything an op derives from them (tile addresses, feature
    // indices, swizzled LDS addresses) is then computed inside the op.  Without it hipcc hoists these loop-invariant
    // values of ALL ops out of the interpreter loop and spills hundreds of registers into the MFMA stages.
    Wv w = w0;
    asm volatile("" : "+v"(w.p), "+v"(w.g), "+v"(w.lane));
    // the task / tile-in-task / point of every half (flat dealing: the tile's own task)
    int h_task[kH], h_pt[kH];
    bool h_ok[kH], h_valid[kH];
#pragma unroll
    for (int h = 0; h < kH; ++h) {
      h_task[h] = w.task[h >> 1];
      h_valid[h] = w.tv[h >> 1];
      h_pt[h] = w.tin[h >> 1] * 32 + 16 * (h & 1) + w.p;
      h_ok[h] = h_valid[h] && h_pt[h] < g.pts_per_task;
    }
    NPF16_STAMP(7)  // ops other than LINEAR (+ the interpreter's per-op setup)
    if (opc == NPF_OP_LINEAR) {
      const int S = uni((o.i0 + 31) >> 5), N = o.i1;
      const int NB = uni((N + kSlabRows - 1) / kSlabRows);
      const int RB = uni(lds_row_bytes(S));
      const int swz = RB == 128 ? 7 : 15;
      bf16x8 curb[kH][kMaxS];
#pragma unroll
      for (int h = 0; h < kH; ++h)
#pragma unroll
        for (int st = 0; st < kMaxS; ++st)
          if (st < S) curb[h][st] = pack_bf16(acc[h][2 * st], acc[h][2 * st + 1]);
      // fragment addresses of the layer (relative to the slot): chunk (4 st + g) ^ ps of row p, with st = 4 m + t
      // the lane part only depends on t (m goes to the immediate offset)
      const int ps = w.p & swz;
      unsigned a0[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a0[t] = (unsigned)(w.p * RB) + (unsigned)(((t ^ (ps >> 2)) << 6) | ((w.g ^ (ps & 3)) << 4));
      const unsigned bias_a = (unsigned)(kImageBytes + 16 * w.g);
      NPF16_STAMP(0)  // layer setup: pack the input, fragment addresses

#define NPF16_STAGE(I) if (I < NB) stage<I>(s, g, w, smem, smem_a, wg_task, consumed, S, RB, a0, bias_a, curb, acc NPF16_STAMP_PASS);
      NPF16_STAGE(0) NPF16_STAGE(1) NPF16_STAGE(2) NPF16_STAGE(3) NPF16_STAGE(4) NPF16_STAGE(5) NPF16_STAGE(6) NPF16_STAGE(7)
#undef NPF16_STAGE

      // ---- epilogue of the layer, in place: addend / relu / relu-backward mask
      const bool relu = (o.flags & NPF_F_RELU) != 0;
      const bool mask = (o.flags & NPF_F_MASK_PT) != 0;
      const bool add = (o.flags & NPF_F_ADD_PT) != 0;
      if (add || mask) {
        const PtRef src = pt_ref(o.p2, (o.flags & NPF_F_P16) != 0, g, w, NB * 32, o.i4);
#pragma unroll
        for (int st = 0; st < kMaxS; ++st)
          if (st < NB) {
#pragma unroll
            for (int h = 0; h < kH; ++h) {
              f32x4 lo = zero4, hi = zero4;
              if (h_valid[h]) pt_read_pair(src, w, st, h, lo, hi);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (mask) {
                  acc[h][2 * st][e] = lo[e] > 0.f ? acc[h][2 * st][e] : 0.f;
                  acc[h][2 * st + 1][e] = hi[e] > 0.f ? acc[h][2 * st + 1][e] : 0.f;
                } else {
                  acc[h][2 * st][e] += lo[e];
                  acc[h][2 * st + 1][e] += hi[e];
                }
              }
            }
          }
      }
      if (relu) {
#pragma unroll
        for (int h = 0; h < kH; ++h)
#pragma unroll
          for (int b = 0; b < kMaxB; ++b)
            if (b < 2 * NB) {
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[h][b][e] = fmaxf(acc[h][b][e], 0.f);
            }
      }
      NPF16_STAMP(5)  // layer epilogue
    } else if (opc == NPF_OP_LOAD_PT || opc == NPF_OP_ADD_PT || opc == NPF_OP_MASK_POS || opc == NPF_OP_ROWDOT_PT ||
               opc == NPF_OP_SOFTMAX_BWD) {
      const int FS = o.i0 >> 5;
      const PtRef src = pt_ref(o.p0, (o.flags & NPF_F_P16) != 0, g, w, o.i0, o.i4);
      float dot[kH];
#pragma unroll
      for (int h = 0; h < kH; ++h) dot[h] = 0.f;
#pragma unroll
      for (int st = 0; st < kMaxS; ++st)
        if (st < FS) {
#pragma unroll
          for (int h = 0; h < kH; ++h) {
            f32x4 v[2] = {zero4, zero4};
            if (h_valid[h]) pt_read_pair(src, w, st, h, v[0], v[1]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              f32x4 c = acc[h][2 * st + j];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (opc == NPF_OP_LOAD_PT) c[e] = v[j][e];
                else if (opc == NPF_OP_ADD_PT) c[e] = o.i1 ? fmaxf(c[e] + v[j][e], 0.f) : c[e] + v[j][e];
                else if (opc == NPF_OP_MASK_POS) c[e] = v[j][e] > 0.f ? c[e] : 0.f;
                else if (opc == NPF_OP_ROWDOT_PT) dot[h] += c[e] * v[j][e];
                else c[e] = o.f0 * v[j][e] * (c[e] - acc_dot[h]);
              }
              acc[h][2 * st + j] = c;
            }
          }
        }
      if (opc == NPF_OP_ROWDOT_PT) {
#pragma unroll
        for (int h = 0; h < kH; ++h) acc_dot[h] = xg_sum(dot[h]);
      }
    } else if (opc == NPF_OP_STORE_PT) {
      const int FS = o.i0 >> 5;
      const PtRef dst = pt_ref(o.p0, (o.flags & NPF_F_P16) != 0, g, w, o.i0, o.i4);
#pragma unroll
      for (int st = 0; st < kMaxS; ++st)
        if (st < FS) {
#pragma unroll
          for (int h = 0; h < kH; ++h)
            if (h_valid[h]) pt_write_pair(dst, w, st, h, acc[h][2 * st], acc[h][2 * st + 1]);
        }
    } else if (opc == NPF_OP_STORE_TR) {
      // feature-major fp32 copy [task][feature][point]
      const int F = o.i0, ld = o.i1;
#pragma unroll
      for (int h = 0; h < kH; ++h)
        if (h_valid[h]) {
          float* dst = (float*)o.p0 + (size_t)h_task[h] * F * ld + (size_t)(4 * w.g) * ld + h_pt[h];
#pragma unroll
          for (int b = 0; b < kMaxB; ++b) {
            if (16 * b + 16 <= F) {  // (uniform) whole block valid
#pragma unroll
              for (int e = 0; e < 4; ++e) dst[(size_t)(16 * b + e) * ld] = acc[h][b][e];
            } else if (16 * b < F) {  // (uniform) the one partial block
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (16 * b + 4 * w.g + e < F) dst[(size_t)(16 * b + e) * ld] = acc[h][b][e];
            }
          }
        }
    } else if (opc == NPF_OP_STORE_WB) {
      // bf16 row image [task][ld rows = points][Fp]: the lane's 8 values of feature group st are 16 contiguous bytes
      const int F = o.i0, ld = o.i1, Fp = ((F + 31) >> 5) * 32;
#pragma unroll
      for (int h = 0; h < kH; ++h)
        if (h_valid[h]) {
          unsigned short* dst = (unsigned short*)o.p0 + ((size_t)h_task[h] * ld + h_pt[h]) * Fp + 8 * w.g;
#pragma unroll
          for (int st = 0; st < kMaxS; ++st)
            if (32 * st < Fp) *(bf16x8*)(dst + 32 * st) = pack_bf16(acc[h][2 * st], acc[h][2 * st + 1]);
        }
    } else if (opc == NPF_OP_STORE_TRB) {
      // bf16 transposed image [task][F][ld columns]: point 32 t + r at column 32 t + 8 ((r & 15) >> 2) + 4 (r >> 4) + (r & 3)
      const int F = o.i0, ld = o.i1;
#pragma unroll
      for (int h = 0; h < kH; ++h)
        if (h_valid[h]) {
          const int col = (h_pt[h] & ~31) + 8 * (w.p >> 2) + 4 * (h & 1) + (w.p & 3);
          unsigned short* dst = (unsigned short*)o.p0 + (size_t)h_task[h] * F * ld + (size_t)(4 * w.g) * ld + col;
#pragma unroll
          for (int b = 0; b < kMaxB; ++b) {
            if (16 * b + 16 <= F) {  // (uniform) whole block valid
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const __bf16 v = (__bf16)acc[h][b][e];
                dst[(size_t)(16 * b + e) * ld] = __builtin_bit_cast(unsigned short, v);
              }
            } else if (16 * b < F) {  // (uniform) the one partial block
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const __bf16 v = (__bf16)acc[h][b][e];
                if (16 * b + 4 * w.g + e < F) dst[(size_t)(16 * b + e) * ld] = __builtin_bit_cast(unsigned short, v);
              }
            }
          }
        }
    } else if (opc == NPF_OP_LOAD_ROWS) {
      const int kd = o.i0;
#pragma unroll
      for (int h = 0; h < kH; ++h) {
        const int task = o.i4 > 0 ? h_task[h] % o.i4 : h_task[h];
        const float* src = (const float*)o.p0 + ((size_t)task * g.pts_per_task + h_pt[h]) * kd;
#pragma unroll
        for (int b = 0; b < kMaxB; ++b) acc[h][b] = zero4;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int f = 16 * b + 4 * w.g + e;
            if (h_ok[h] && f < kd) acc[h][b][e] = src[f];
          }
      }
    } else if (opc == NPF_OP_STORE_ROWS) {
      const int nd = o.i0;
#pragma unroll
      for (int h = 0; h < kH; ++h) {
        const int task = o.i4 > 0 ? h_task[h] % o.i4 : h_task[h];
        float* dst = (float*)o.p0 + ((size_t)task * g.pts_per_task + h_pt[h]) * nd;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int f = 16 * b + 4 * w.g + e;
            if (h_ok[h] && f < nd) dst[f] = acc[h][b][e];
          }
      }
    } else if (opc == NPF_OP_SOFTMAX) {
      // i1 = 0: softmax of the row; 1: also store (row max, row sum) to p0[task][pt][2]; 2: take them from p0
      const int nvalid = o.i0, smode = o.i1;
      const int FB = ((nvalid + 31) >> 5) * 2;
      const float scale = o.f0;
#pragma unroll
      for (int h = 0; h < kH; ++h) {
        float* stats = (float*)o.p0 + ((size_t)h_task[h] * g.pts_per_task + h_pt[h]) * 2;
        float m = -INFINITY, sum = 0.f;
        if (smode == 2) {
          if (h_ok[h]) {
            m = stats[0];
            sum = stats[1];
          } else {
            m = 0.f;
            sum = 1.f;
          }
        } else {
#pragma unroll
          for (int b = 0; b < kMaxB; ++b)
            if (b < FB)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (16 * b + 4 * w.g + e < nvalid) m = fmaxf(m, acc[h][b][e]);
          m = xg_max(m);
        }
        float part = 0.f;
#pragma unroll
        for (int b = 0; b < kMaxB; ++b)
          if (b < FB)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool ok = 16 * b + 4 * w.g + e < nvalid;
              const float ex = ok ? expf((acc[h][b][e] - m) * scale) : 0.f;
              acc[h][b][e] = ex;
              part += ex;
            }
        if (smode != 2) sum = xg_sum(part);
        if (smode == 1 && h_ok[h] && w.g == 0) {
          stats[0] = m;
          stats[1] = sum;
        }
        const float inv = 1.f / sum;
#pragma unroll
        for (int b = 0; b < kMaxB; ++b)
          if (b < FB) acc[h][b] *= inv;
      }
    } else if (opc == NPF_OP_ADD_TASKVEC) {
      const int FB = o.i0 >> 4;
#pragma unroll
      for (int h = 0; h < kH; ++h) {
        const int task = o.i4 > 0 ? h_task[h] % o.i4 : h_task[h];
        const float* v = (const float*)o.p0 + (size_t)task * o.i0;
#pragma unroll
        for (int b = 0; b < kMaxB; ++b)
          if (b < FB) {
            f32x4 c = acc[h][b] + *(const f32x4*)(v + 16 * b + 4 * w.g);
            if (o.i1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) c[e] = fmaxf(c[e], 0.f);
            }
            acc[h][b] = c;
          }
      }
    } else if (opc == NPF_OP_RELU || opc == NPF_OP_SCALE) {
#pragma unroll
      for (int h = 0; h < kH; ++h)
#pragma unroll
        for (int b = 0; b < kMaxB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[h][b][e] = (opc == NPF_OP_RELU) ? fmaxf(acc[h][b][e], 0.f) : o.f0 * acc[h][b][e];
    }
  }
  // no LDS-DMA may be in flight when the workgroup's LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef NPF_STAMPS
  NPF16_STAMP(7)
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int i = 0; i < 8; ++i) g_stamps16[i] = st_sum[i];
#endif
}

#ifdef NPF_STAMPS
extern "C" int npf_debug_stamps16(unsigned long long* out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps16), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace npf16

// Launch of a validated bf16 program (called by npf_chain_run).
int npf16_chain_launch(const npf_program_t& g, void* stream) {
  for (int i = 0; i < g.n_ops; ++i) {
    const npf_op_t& o = g.ops[i];
    if (o.op == NPF_OP_LINEAR) {
      if (o.i0 > 256 || o.i1 > 256 || o.i2 != NPF_W_ROWMAJOR || (((uintptr_t)o.p0) & 15) || o.i3 < ((o.i0 + 31) >> 5) * 16 ||
          (o.i3 & 3) || (o.s0 & 3))
        return NPF_EINVAL;
      if (o.flags & NPF_F_ADD_RM) return NPF_EINVAL;
    } else if (o.op == NPF_OP_LAYERNORM || o.op == NPF_OP_LAYERNORM_BWD || o.op == NPF_OP_LOAD_RM) {
      return NPF_EINVAL;
    } else if (o.op == NPF_OP_LOAD_PT || o.op == NPF_OP_STORE_PT || o.op == NPF_OP_ADD_PT || o.op == NPF_OP_MASK_POS ||
               o.op == NPF_OP_ROWDOT_PT || o.op == NPF_OP_SOFTMAX_BWD || o.op == NPF_OP_ADD_TASKVEC ||
               o.op == NPF_OP_SOFTMAX || o.op == NPF_OP_STORE_TR) {
      if (o.i0 > 256) return NPF_EINVAL;
    }
  }
  const int per_wg = npf16::kTilesPerWG;
  const long wgs = g.wg_per_task ? (long)g.n_tasks * ((g.tiles_per_task + per_wg - 1) / per_wg)
                                 : ((long)g.n_tasks * g.tiles_per_task + per_wg - 1) / per_wg;
  if (wgs <= 0 || wgs > 0x7fffffffL) return NPF_EINVAL;
  hipLaunchKernelGGL(npf16::chain16_kernel, dim3((unsigned)wgs), dim3(64 * npf16::kWaves), 0, (hipStream_t)stream, g);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}
