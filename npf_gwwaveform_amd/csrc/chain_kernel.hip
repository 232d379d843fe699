// Register-resident "transposed chain" interpreter for gfx950 (MI355X).
//
// One wavefront owns 16 points (half a PT32 tile) and keeps their activations -- up to 256
// features -- in registers for the whole chain, feature-major:  every layer computes
//     Y^T[n][p] = sum_k W[n][k] X^T[k][p]
// with v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD), A operand = rows of a 64-row
// slab of W read from LDS, B operand = the previous layer's accumulator registers.  The
// 16x16 accumulator layout (column = lane & 15 = point, row = 4*(lane>>4) + reg = feature)
// is *already* the B-operand layout of the next layer once the weight slab is read with
// the same k permutation (one ds_read_b128 = the 4 k-steps of a 16-feature block), so
// activations never leave the register file between layers: no LDS round trip, no
// transposes, and HBM only sees what the program explicitly stores.
// Scaled-dot attention is two such layers whose weights are the task's keys / values with a
// feature-axis softmax in between (the feature axis is in-lane plus two cross-lane swaps).
//
// Weights stream global -> LDS by LDS-DMA (global_load_lds) in 32-row slabs, one slab ahead
// of the MFMAs, through a 2-slot ring, one barrier per slab.  Workgroup = 4 waves (one per
// SIMD) = 2 tiles = 64 points, TWO workgroups per CU (2 x 64 KB of LDS, 2 x 256 registers per
// SIMD): the waves of one workgroup run in phase (they meet at every slab barrier), so the
// per-slab scalar work, DMA issue and barrier bubbles of one workgroup are hidden by the
// MFMAs of the other, independent one -- an 8-wave workgroup idles the matrix pipe there.
//
// Replaces the torch op sequences of MLP.forward (npf/architectures/mlp.py:95-109),
// MergeFlatInputs.forward (encoders.py:175-183), BaseAttender.forward / DotAttender.score
// (attention.py:129-164,204-220), merge_r_z (neuralproc/base.py:554-575) and their autograd.
#include <type_traits>


#include "npf_common.hpp"

#ifndef NPF_BF16_PREFETCH
#define NPF_BF16_PREFETCH 3  // bf16 instance: k-steps of fragment reads in flight ahead of the MFMAs
#endif

namespace npf {

constexpr int kWaves = 4;  // waves that issue the slab DMA (the first four of a workgroup)
// Workgroup shapes (template parameter WAVES of the kernel):
//   4 waves = 64 points, 2-slot slab ring, two workgroups per CU that drift against each other;
//   8 waves = 128 points = two *phase groups* of 4 waves (one wave of each group per SIMD), one
//     workgroup per CU, 3-slot ring.  Group A (waves 0-3) streams the slabs and is one slab
//     ahead; group B (waves 4-7) only consumes.  Their barriers sit at different places of the
//     slab loop (A: after the MFMAs, B: before them), so inside one barrier interval A runs
//     epilogue -> DMA issue -> MFMAs while B runs MFMAs -> epilogue: one wave's scalar/VALU/DMA
//     phase always meets the MFMAs of the other wave of its SIMD, and every weight slab is
//     fetched once per 128 points instead of once per 64.
// The kernel exists in two widths (template parameter MAXB = 16-feature blocks a wave keeps in
// registers): 16 (<= 256 features, 2 workgroups per CU) and 32 (<= 512 features, e.g. the
// r = 512 decode-only configuration: 128 + 128 activation registers, one workgroup per CU).
constexpr int kSlabRows = 32;
constexpr int kBlk = kSlabRows / 16;            // 16-row output blocks (accumulators) per slab
constexpr int slab_floats(int maxb) { return kSlabRows * 16 * maxb + 64; }  // rows, then the biases
// The slab ring of an instance: fp32 instances 2 slots (3 in the paired variant) of slab_floats(MAXB) floats.  The bf16
// instance streams half-size slabs (32 rows x 256 bf16 = 4096 floats + biases): FOUR slots fit the same LDS, and its
// 256 -> 256 layers keep two slabs in flight behind a counted s_waitcnt (fast_layer, RING) -- with one slab in flight
// and a vmcnt(0) per slab every stage lasts as long as one DMA round trip (~1800 cycles against 256 cycles of MFMA).
constexpr int kRingFloats = kSlabRows * 128 + 64;
constexpr bool ring_instance(bool bf16, bool paired) { return bf16 && !paired; }
constexpr int slot_stride(int maxb, bool bf16, bool paired) { return ring_instance(bf16, paired) ? kRingFloats : slab_floats(maxb); }
template <bool BF16, bool PAIRED>
__device__ __forceinline__ int next_slot(int slot) {
  if constexpr (ring_instance(BF16, PAIRED)) return (slot + 1) & 3;
  else if constexpr (PAIRED) return slot == 2 ? 0 : slot + 1;
  else return slot ^ 1;
}

// ReLU masks as bits ("PTM" tensors, NPF_OP_STORE_MASK / NPF_OP_MASK_BITS / NPF_F_MASK_BITS; bf16 instance): one 32-bit
// word per point, lane group g and group of 128 features -- [tile][word][g][32 points] -- holding the lane's own 32 values
// of blocks 8 w .. 8 w + 7: element e of block b = 8 w + bb at bit 31 - (4 bb + e) (the order an add-with-carry chain
// shifts them in).  32 bytes per point and 256 features instead of the 512 B of the PT16 activation.
__device__ __forceinline__ bool mask_bit(unsigned word, int bb, int e) { return (word & (0x80000000u >> (4 * bb + e))) != 0u; }

struct Wave {
  int tid, lane, wave;     // wave is wave-uniform (readfirstlane)
  int p, g, half;          // p = lane & 15 (point / slab row), g = lane >> 4 (k group), half of the tile
  int task, tile_in_task;  // wave-uniform
  bool valid;              // wave-uniform: this wave has a real tile
};

__device__ __forceinline__ int eff_task(const Wave& w, int modulus) {
  return modulus > 0 ? (w.task % modulus) : w.task;
}

// Pointer to this lane's float4 column inside its tile of a PT32 tensor with F features:
// feature quad f4 of the lane's point is at ptr[f4 * 128].
__device__ __forceinline__ const float* pt_lane(const void* base, const npf_program_t& g, const Wave& w, int F,
                                                int modulus) {
  const size_t tile = (size_t)eff_task(w, modulus) * g.tiles_per_task + w.tile_in_task;
  return (const float*)base + tile * (size_t)(F * 32) + (16 * w.half + w.p) * 4;
}

// Pointer to the row of this lane's point in a row-major [task][pt][F] tensor (the lane's features
// are at + 16 b + 4 g); padding points re-read the task's last point (loaded, never stored).
__device__ __forceinline__ const float* rm_lane(const void* base, const npf_program_t& g, const Wave& w, int pt, int F,
                                                int modulus) {
  const int p = pt < g.pts_per_task ? pt : g.pts_per_task - 1;
  return (const float*)base + ((size_t)eff_task(w, modulus) * g.pts_per_task + p) * (size_t)F;
}

// ---- weight slabs -------------------------------------------------------------------
// A slab = rows [nb*32, nb*32+32) of a layer's weight matrix, all Kp = roundup(K, 32)
// columns, as a dense [32][Kp] fp32 image in LDS whose 16-byte chunks are XOR-swizzled
// inside each row (chunk c of row r lives at chunk position c ^ (r & swz), swz = 15, or 7
// when Kp/4 is not a multiple of 16), followed by the 32 biases of those rows.  The MFMA
// A-operand read (16 rows x 4 k-groups per ds_read_b128) is then bank-conflict free, and --
// because the image has no padding -- one wavefront-wide LDS-DMA (global_load_lds_dwordx4:
// 64 lanes x 16 B = 1 KiB, linear in LDS, per-lane source address) fills 1 KiB of it straight
// from global memory: the swizzle is applied to the *source* address (cdna guide 5.4 rule
// 21).  No staging registers, no ds_write, and no VGPR result for hipcc to wait on: the DMA
// stays in flight across the MFMA loop and is retired by the vmcnt(0) of the slab barrier.
// Every load of the slab path is an LDS-DMA on purpose: a VGPR-destination load next to
// LDS-DMA makes hipcc drain vmcnt(0) at unrelated places (cdna guide 5, trap (b)).
// Out-of-range rows / columns are fetched from a 16-byte zero buffer.
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __attribute__((aligned(16))) float g_zero128[32] = {};  // "no bias" source of the fast path

// Per-LINEAR constants of the slab stream (all wave-uniform).
struct SlabOp {
  const float* W;     // weights, already offset to this workgroup's task
  const float* bias;  // nullptr if none, offset to the task
  int K, N, Kp, cpr, swz, lcpr;  // cpr = 16-byte chunks per row, lcpr = log2(cpr) or -1
  int mode, ldw, Fq, n_slabs;
  bool vec16;
  // fast path (16-byte pieces, power-of-two row length, no column padding): piece number i
  // of a wave reads  base(slab) + i * step + lo[i & 7]
  bool fast;
  unsigned lo[8];   // per-lane source offsets (bytes, >= 0), one per piece phase
  int step;         // scalar source advance between a wave's consecutive pieces (floats)
  int slab_stride;  // scalar source advance between consecutive slabs (floats)
  int n_pw;         // pieces per wave per slab
};

// BF16: the weights are a bf16 image ([N][roundup(K, 32)] bf16, k-permuted inside each group of 32, see
// npf_cast_bf16_weights): for the slab stream it is simply a matrix of K/2 floats per row.
template <bool BF16>
__device__ __forceinline__ SlabOp make_slab_op(const npf_op_t& o, int task) {
  SlabOp s;
  s.K = BF16 ? (((o.i0 + 31) >> 5) * 16) : o.i0;
  s.N = o.i1;
  s.mode = o.i2;
  s.Kp = ((s.K + 31) >> 5) * 32;
  s.cpr = s.Kp >> 2;
  s.swz = (s.cpr & 15) ? 7 : 15;
  s.lcpr = (s.cpr & (s.cpr - 1)) ? -1 : (31 - __builtin_clz(s.cpr));
  s.ldw = o.i3;
  s.Fq = ((s.N + 31) >> 5) * 8;
  s.n_slabs = (s.N + kSlabRows - 1) / kSlabRows;
  const float* W = (const float*)o.p0;
  if (s.mode == NPF_W_ROWMAJOR) {
    s.W = W + (size_t)task * o.s0;
    s.vec16 = ((o.i3 & 3) == 0) && ((s.K & 3) == 0) && ((o.s0 & 3) == 0) && ((((uintptr_t)o.p0) & 15) == 0);
  } else if (s.mode == NPF_W_PT_ROWS) {
    s.W = W + (size_t)task * o.i3 * (size_t)(s.Kp * 32);  // + tile * Kp * 32 per 32 rows
    s.vec16 = true;
  } else {
    s.W = W + (size_t)task * o.i3 * s.Fq * 128;
    s.vec16 = false;
  }
  s.bias = o.p1 ? (const float*)o.p1 + (size_t)task * o.s1 : nullptr;
  return s;
}

// Lane offsets of the fast DMA path.  Piece q = wave + 4 i covers linear chunks [64 q, 64 q + 64)
// of the slab image: rows q*rpp .. (rpp = 64 / cpr rows per piece) or, for rows longer than a
// piece (cpr = 128), half of row q / 2.  The swizzle only depends on the low 4 bits of the row,
// which repeat every 4 (8 for cpr = 128) pieces of a wave.
__device__ __forceinline__ void slab_fast_setup(SlabOp& s, const Wave& w) {
  s.fast = s.vec16 && s.lcpr >= 0 && s.K == s.Kp && s.mode != NPF_W_PT_COLS;
  s.n_pw = s.Kp >> 5;  // (32 rows * Kp * 4 B / 1 KiB) / 4 waves
  if (!s.fast) return;
  const int rstride = (s.mode == NPF_W_ROWMAJOR) ? s.ldw : 4;      // floats per matrix row step
  const int cstride = (s.mode == NPF_W_ROWMAJOR) ? 4 : 128;        // floats per chunk step
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const int lin = (w.wave + 4 * v) * 64 + w.lane;   // linear chunk of piece v of this wave
    const int row = lin >> s.lcpr, cpos = lin & (s.cpr - 1);
    const int ch = cpos ^ (row & s.swz);
    // offset relative to the source of piece v = slab base + v * step (step: see below)
    // (row advances by exactly (256 >> lcpr) per piece phase, so the difference is row(0) * rstride >= 0)
    s.lo[v] = 4u * (unsigned)(row * rstride + ch * cstride - v * (((4 * 64) >> s.lcpr) * rstride));
  }
  s.step = ((4 * 64) >> s.lcpr) * rstride;  // 4 pieces further = this many rows further
  s.slab_stride = (s.mode == NPF_W_ROWMAJOR) ? kSlabRows * s.ldw : s.Kp * 32;
}

struct SlabCursor {
  int op;  // index of the LINEAR op the next slab belongs to (n_ops if none)
  int nb;  // slab index inside that op
};

__device__ __forceinline__ void dma16(const float* src, float* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}
// uniform 64-bit base + 32-bit unsigned lane offset: selects the SGPR-base form of the
// instruction (global_load_lds_dwordx4 v_off, s[base:base+1]) -- no VALU address arithmetic
__device__ __forceinline__ void dma16_so(const char* uniform_base, unsigned lane_off, float* lds_dst_wave_uniform) {
  dma16((const float*)(uniform_base + (size_t)lane_off), lds_dst_wave_uniform);
}
__device__ __forceinline__ void dma4(const float* src, float* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 4, 0, 0);
}

// LDS-DMA fill of slab nb of op s into `slot`, one piece (wave-instruction) at a time.
// 16-byte pieces (1 KiB per wave-instruction) for row-major weights with 16-byte aligned rows
// and for the task's keys; 4-byte pieces (256 B per wave-instruction) for odd-shaped first
// layers and the transposed values.  Piece q of the slab is issued by wave q % 8.
struct SlabDma {
  float* slot;
  int row0;     // first matrix row of the slab
  int q;        // next piece of this wave
  int n_instr;  // pieces in the slab
  bool on;
};

__device__ __forceinline__ SlabDma dma_begin(const SlabOp& s, int nb, const Wave& w, float* slot, bool on) {
  SlabDma d;
  d.slot = slot;
  d.row0 = nb * kSlabRows;
  d.q = w.wave;
  d.n_instr = on ? (s.vec16 ? (s.Kp >> 3) : (s.Kp >> 1)) : 0;
  d.on = on;
  return d;
}

__device__ __forceinline__ void dma_piece(const SlabOp& s, SlabDma& d, const Wave& w) {
  const float* Z = g_zero16;
  const int lin = d.q * 64 + w.lane;
  if (s.vec16) {
    int row, cpos;
    if (s.lcpr >= 0) {
      row = lin >> s.lcpr;
      cpos = lin & (s.cpr - 1);
    } else {
      row = lin / s.cpr;
      cpos = lin - row * s.cpr;
    }
    const int ch = cpos ^ (row & s.swz);  // the matrix chunk that lives at this LDS position
    const int n = d.row0 + row;
    const float* src;
    if (s.mode == NPF_W_PT_ROWS)
      src = (n < s.N) ? s.W + (size_t)(n >> 5) * (s.Kp * 32) + ((size_t)ch * 32 + (n & 31)) * 4 : Z;
    else
      src = (n < s.N && ch * 4 < s.K) ? s.W + (size_t)n * s.ldw + ch * 4 : Z;
    dma16(src, d.slot + d.q * 256);
  } else {
    const int row = lin / s.Kp, colpos = lin - row * s.Kp;
    const int col = (((colpos >> 2) ^ (row & s.swz)) << 2) | (colpos & 3);
    const int n = d.row0 + row;
    const float* src;
    if (s.mode == NPF_W_PT_COLS)
      src = (col < s.K) ? s.W + ((size_t)(col >> 5) * s.Fq + (n >> 2)) * 128 + (col & 31) * 4 + (n & 3) : Z;
    else
      src = (n < s.N && col < s.K) ? s.W + (size_t)n * s.ldw + col : Z;
    dma4(src, d.slot + d.q * 64);
  }
  d.q += kWaves;
}

__device__ __forceinline__ void dma4_so(const char* uniform_base, unsigned lane_off, float* lds_dst_wave_uniform) {
  // (the empty asm keeps the 32-bit offset in this basic block: hoisted out of a loop, its
  // zero-extension becomes a 64-bit VGPR pair and instruction selection falls back to a VALU add)
  asm volatile("" : "+v"(lane_off));
  dma4((const float*)(uniform_base + (size_t)lane_off), lds_dst_wave_uniform);
}
// A full slab on the fast path: one SALU add + one DMA instruction per 1 KiB piece.  The
// lane-offset registers lo[] live for the whole layer: hipcc waits for vmcnt(0) before it
// overwrites a VGPR that an in-flight LDS-DMA instruction used as its address, so any
// per-piece address arithmetic in VGPRs serialises the wave behind its own DMA.
// NPW > 0: compile-time piece count.
template <int NPW>
__device__ __forceinline__ void dma_fast_slab(const SlabOp& s, int nb, float* slot, const Wave& w) {
  const char* base = (const char*)(s.W + (size_t)nb * s.slab_stride);
  float* dst = slot + w.wave * 256;
#pragma unroll
  for (int i = 0; i < (NPW > 0 ? NPW : 16); ++i)
    if (NPW > 0 || i < s.n_pw) dma16_so(base + (size_t)(i * s.step) * 4, s.lo[i & 7], dst + i * (kWaves * 256));
}
// the 32 biases of a full slab (fast path): wave 0, lanes 32..63 re-read the first 32
__device__ __forceinline__ void dma_fast_bias(const SlabOp& s, int row0, float* slot, const Wave& w) {
  if (w.wave == 0) {
    const char* base = s.bias != nullptr ? (const char*)(s.bias + row0) : (const char*)g_zero128;
    dma4_so(base, (unsigned)(w.lane & 31) * 4u, slot + kSlabRows * s.Kp);
  }
}
__device__ __forceinline__ void dma_bias(const SlabOp& s, int row0, float* slot, const Wave& w) {
  if (w.wave == 0) {
    const int n = row0 + w.lane;
    dma4((s.bias != nullptr && w.lane < kSlabRows && n < s.N) ? s.bias + n : g_zero16, slot + kSlabRows * s.Kp);
  }
}

// the rest of the slab's pieces + the 64 biases of its rows
__device__ __forceinline__ void dma_finish(const SlabOp& s, SlabDma& d, const Wave& w) {
  if (d.on && s.fast && d.row0 + kSlabRows <= s.N) {
    // full slab on the fast path: ~6 instructions per 1 KiB piece
    dma_fast_slab<0>(s, d.row0 / kSlabRows, d.slot, w);
    d.q = d.n_instr;
  }
  while (d.q < d.n_instr) dma_piece(s, d, w);
  if (d.on) dma_bias(s, d.row0, d.slot, w);
}

// One 32-row slab = two 16-row output blocks (independent accumulator chains):
//   acc[j][n][p] = bias[16j + n] + sum_k W[16j + n][k] cur[k][p],  j < NBLK.
// MFMA step s of input block kb contracts features kb*16 + 4g + s (g = lane >> 4): exactly
// what accumulator register s of block kb holds on this lane, and what the A lane (n, g)
// reads as element s of chunk 4*kb + g of row n.
// KB16S > 0: the number of 16-feature input blocks is a compile-time constant (straight-line
// code: hipcc hoists the LDS reads of later blocks above the MFMAs of earlier ones, which
// hides the LDS latency inside one wave); KB16S == 0: runtime count with a guard per block.
template <int NBLK, int KB16S, int MAXB>
__device__ __forceinline__ void slab_mfma(const float* slot, int KB16, const Wave& w, const f32x4 (&cur)[MAXB],
                                          f32x4 (&acc)[kBlk]) {
  if (KB16S > 0) KB16 = KB16S;
  const int Kp = KB16 * 16;
  const int cpr = Kp >> 2;
  const int swz = (cpr & 15) ? 7 : 15;
  const int ps = w.p & swz;  // rows 16j + p: the low 4 bits are p for every block
  const float* a = slot + w.p * Kp;
  const float* bias = slot + kSlabRows * Kp + 4 * w.g;
#pragma unroll
  for (int j = 0; j < NBLK; ++j) acc[j] = *(const f32x4*)(bias + 16 * j);
  if constexpr (KB16S > 0) {
    // Software pipeline, pinned by hand: the A fragments of block kb+1 are read (inline-asm
    // ds_read_b128, invisible to hipcc's scheduler and waitcnt pass) BEFORE the MFMAs of block
    // kb issue, into the other of two register sets; one s_waitcnt lgkmcnt(0) per block.
    // hipcc itself sinks every fragment read behind the previous MFMAs to shorten live ranges
    // and then waits for it immediately, which exposes the LDS latency in every block.
    // Chunk (4 kb + g) ^ ps of row p: with kb = 4 m + t the lane part only depends on t.
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)a;
    unsigned addr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) addr[t] = lds0 + (((t ^ (ps >> 2)) << 6) | ((w.g ^ (ps & 3)) << 4));
    constexpr int kRowBlk = 16 * KB16S * 16 * 4;  // bytes between output blocks j (16 rows)
    f32x4 fr[2][NBLK] = {};
    // one statement = wait for the current fragments + issue the next ones: the "+v" operands
    // make the MFMAs of the block depend on it, so nothing can be scheduled around the wait
#ifdef NPF_EXP_NOLDS  // diagnostic build: same MFMA stream without the LDS fragment reads (results are garbage)
#define NPF_STEP(curf, nxtf, kbn) asm volatile("s_nop 0" : "+v"(nxtf[0]), "+v"(nxtf[NBLK - 1]), "+v"(curf[0]), "+v"(curf[NBLK - 1]));
#define NPF_LAST(curf) asm volatile("s_nop 0" : "+v"(curf[0]), "+v"(curf[NBLK - 1]));
#else
#define NPF_STEP(curf, nxtf, kbn)                                                                          \
  if constexpr (NBLK == 2)                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6" \
                 : "=&v"(nxtf[0]), "=&v"(nxtf[1]), "+v"(curf[0]), "+v"(curf[1])                            \
                 : "v"(addr[(kbn)&3]), "n"(((kbn) >> 2) * 256), "n"(((kbn) >> 2) * 256 + kRowBlk));        \
  else                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2 offset:%3"                                   \
                 : "=&v"(nxtf[0]), "+v"(curf[0])                                                           \
                 : "v"(addr[(kbn)&3]), "n"(((kbn) >> 2) * 256));
#define NPF_LAST(curf)                                                                    \
  if constexpr (NBLK == 2)                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(curf[0]), "+v"(curf[1]));                  \
  else                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(curf[0]));
#endif
    // prologue: fragments of block 0
    if constexpr (NBLK == 2)
      asm volatile("ds_read_b128 %0, %2 offset:0\n\tds_read_b128 %1, %2 offset:%3"
                   : "=&v"(fr[0][0]), "=&v"(fr[0][1])
                   : "v"(addr[0]), "n"(kRowBlk));
    else
      asm volatile("ds_read_b128 %0, %1 offset:0" : "=&v"(fr[0][0]) : "v"(addr[0]));
#pragma unroll
    for (int kb = 0; kb < KB16S; ++kb) {
      if (kb + 1 < KB16S) {
        if ((kb & 1) == 0) { NPF_STEP(fr[0], fr[1], kb + 1) } else { NPF_STEP(fr[1], fr[0], kb + 1) }
      } else {
        if ((kb & 1) == 0) { NPF_LAST(fr[0]) } else { NPF_LAST(fr[1]) }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NBLK; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[kb & 1][j][s], cur[kb][s], acc[j], 0, 0, 0);
    }
#undef NPF_STEP
#undef NPF_LAST
  } else {
#pragma unroll
    for (int kb = 0; kb < MAXB; ++kb) {
      if (kb < KB16) {
        const int off = (((4 * kb + w.g) ^ ps) << 2);
        f32x4 x[NBLK];
#pragma unroll
        for (int j = 0; j < NBLK; ++j) x[j] = *(const f32x4*)(a + j * 16 * Kp + off);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < NBLK; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j][s], cur[kb][s], acc[j], 0, 0, 0);
      }
    }
  }
}

template <int NBLK, int MAXB>
__device__ __forceinline__ void slab_mfma_any(const float* slot, int KB16, const Wave& w, const f32x4 (&cur)[MAXB],
                                              f32x4 (&acc)[kBlk]) {
  if constexpr (MAXB >= 32) {
    if (KB16 == 32) {
      slab_mfma<NBLK, 32, MAXB>(slot, KB16, w, cur, acc);
      return;
    }
  }
  switch (KB16) {
    case 16: slab_mfma<NBLK, 16, MAXB>(slot, KB16, w, cur, acc); break;
    case 8: slab_mfma<NBLK, 8, MAXB>(slot, KB16, w, cur, acc); break;
    case 4: slab_mfma<NBLK, 4, MAXB>(slot, KB16, w, cur, acc); break;
    case 2: slab_mfma<NBLK, 2, MAXB>(slot, KB16, w, cur, acc); break;
    default: slab_mfma<NBLK, 0, MAXB>(slot, KB16, w, cur, acc); break;
  }
}

// bf16 instance: one k-step = 32 features = the lane's own values of two adjacent 16-row blocks of the
// previous layer (packed to bf16 once per layer, `curb`), against one 16-byte fragment of the k-permuted
// bf16 weight image per output block: v_mfma_f32_16x16x32_bf16, fp32 accumulation.  The LDS image is
// the fp32 one with rows of KpF = roundup(K/2, 32) floats (same swizzle, same fragment addresses).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 pack_bf16(const f32x4& lo, const f32x4& hi) {
  const bf16x2 p0 = __builtin_convertvector((f32x2{lo[0], lo[1]}), bf16x2), p1 = __builtin_convertvector((f32x2{lo[2], lo[3]}), bf16x2);
  const bf16x2 p2 = __builtin_convertvector((f32x2{hi[0], hi[1]}), bf16x2), p3 = __builtin_convertvector((f32x2{hi[2], hi[3]}), bf16x2);
  bf16x8 r;
  r[0] = p0[0]; r[1] = p0[1]; r[2] = p1[0]; r[3] = p1[1];
  r[4] = p2[0]; r[5] = p2[1]; r[6] = p3[0]; r[7] = p3[1];
  return r;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// the two f32x4 (blocks 2s, 2s+1 of the lane) held by one 16-byte PT16 chunk
__device__ __forceinline__ f32x4 pt16_lo(const u32x4& r) {
  return f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xffff0000u),
               __builtin_bit_cast(float, r[1] << 16), __builtin_bit_cast(float, r[1] & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 pt16_hi(const u32x4& r) {
  return f32x4{__builtin_bit_cast(float, r[2] << 16), __builtin_bit_cast(float, r[2] & 0xffff0000u),
               __builtin_bit_cast(float, r[3] << 16), __builtin_bit_cast(float, r[3] & 0xffff0000u)};
}
// Pointer (bf16 units) to this lane's chunk of feature group 0 of its tile in a PT16 tensor with F features;
// group s is at + 1024 s.
__device__ __forceinline__ const unsigned short* pt16_lane(const void* base, const npf_program_t& g, const Wave& w, int F,
                                                           int modulus) {
  const size_t tile = (size_t)eff_task(w, modulus) * g.tiles_per_task + w.tile_in_task;
  return (const unsigned short*)base + (tile * (size_t)(F >> 3) * 32 + (16 * w.half + w.p)) * 8 + w.g * 256;
}

template <int NBLK, int MAXB>
__device__ __forceinline__ void slab_mfma_bf16(const float* slot, int KpF, int S, const Wave& w,
                                               const bf16x8 (&curb)[MAXB / 2], f32x4 (&acc)[kBlk]) {
  const int cpr = KpF >> 2;
  const int swz = (cpr & 15) ? 7 : 15;
  const int ps = w.p & swz;
  const float* a = slot + w.p * KpF;
  const float* bias = slot + kSlabRows * KpF + 4 * w.g;
#pragma unroll
  for (int j = 0; j < NBLK; ++j) acc[j] = *(const f32x4*)(bias + 16 * j);
#pragma unroll
  for (int st = 0; st < MAXB / 2; ++st) {
    if (st < S) {
      const int off = (((4 * st + w.g) ^ ps) << 2);
#pragma unroll
      for (int j = 0; j < NBLK; ++j) {
        const f32x4 x = *(const f32x4*)(a + j * 16 * KpF + off);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), curb[st], acc[j], 0, 0, 0);
      }
    }
  }
}

// The 256-feature slab loop of the fast path: the pinned fragment pipeline of slab_mfma<2, 16>
// accumulating straight into the slab's two output blocks, plus `side(kb)` after the MFMAs of
// every 16-feature block.  A wave's own VMEM / VALU instructions issue in the shadow of its own
// MFMAs for free (the matrix pipe is busy for 32 cycles per MFMA, an issue takes 4-16), while
// the same instructions issued by the *other* wave of the SIMD during this loop get one issue
// slot per MFMA (~25 cycles per VALU/LDS instruction, ~85 per VMEM instruction;
// tools/issue_probe.hip).  So the slab DMA, the addend loads and the previous slab's epilogue
// all ride inside this loop.
// BF16: KB16S counts 32-feature k-steps (16 floats of a bf16 weight row each), the B operands are the
// packed `curb`, one v_mfma_f32_16x16x32_bf16 per output block and step.
template <int KB16S, int MAXB, bool BF16, class Side>
__device__ __forceinline__ void slab_mfma_side(const float* slot, const Wave& w, const f32x4 (&cur)[MAXB],
                                               const bf16x8 (&curb)[MAXB / 2], f32x4& acc0, f32x4& acc1, Side side) {
  constexpr int Kp = 16 * KB16S;
  constexpr int swz = ((Kp >> 2) & 15) ? 7 : 15;
  const int ps = w.p & swz;
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)(slot + w.p * Kp);
  unsigned addr[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) addr[t] = lds0 + (((t ^ (ps >> 2)) << 6) | ((w.g ^ (ps & 3)) << 4));
  constexpr int kRowBlk = 16 * Kp * 4;
  if constexpr (BF16) {
    // bf16: a k-step is only two 16-cycle MFMAs, far less than the LDS latency under load (8 waves per CU reading): the
    // fragment reads run kPf k-steps ahead of the MFMAs (kPf + 1 register sets), every wait counts what may stay in flight
    constexpr int kPf = NPF_BF16_PREFETCH;
    static_assert(kPf >= 1 && kPf <= 3 && KB16S > kPf, "prefetch depth");
    f32x4 fq[kPf + 1][2];
#define NPF_RD(set, kbn)                                                                \
  asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"         \
               : "=&v"(fq[set][0]), "=&v"(fq[set][1])                                   \
               : "v"(addr[(kbn)&3]), "n"(((kbn) >> 2) * 256), "n"(((kbn) >> 2) * 256 + kRowBlk));
#pragma unroll
    for (int i = 0; i < kPf; ++i) { NPF_RD(i, i) }
#pragma unroll
    for (int kb = 0; kb < KB16S; ++kb) {
      const int c = kb % (kPf + 1);
      if (kb + kPf < KB16S) {
        NPF_RD((kb + kPf) % (kPf + 1), kb + kPf)
        if constexpr (kPf == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fq[c][0]), "+v"(fq[c][1]));
        else if constexpr (kPf == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fq[c][0]), "+v"(fq[c][1]));
        else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fq[c][0]), "+v"(fq[c][1]));
      } else {
        const int left = KB16S - 1 - kb;  // k-steps still in flight behind this one
        if (left == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fq[c][0]), "+v"(fq[c][1]));
        else if (left == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fq[c][0]), "+v"(fq[c][1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fq[c][0]), "+v"(fq[c][1]));
      }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fq[c][0]), curb[kb], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fq[c][1]), curb[kb], acc1, 0, 0, 0);
      side(kb);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef NPF_RD
    return;
  }
  f32x4 fr[2][2] = {};
  asm volatile("ds_read_b128 %0, %2 offset:0\n\tds_read_b128 %1, %2 offset:%3"
               : "=&v"(fr[0][0]), "=&v"(fr[0][1])
               : "v"(addr[0]), "n"(kRowBlk));
#define NPF_STEP2(curf, nxtf, kbn)                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6" \
               : "=&v"(nxtf[0]), "=&v"(nxtf[1]), "+v"(curf[0]), "+v"(curf[1])                            \
               : "v"(addr[(kbn)&3]), "n"(((kbn) >> 2) * 256), "n"(((kbn) >> 2) * 256 + kRowBlk));
#pragma unroll
  for (int kb = 0; kb < KB16S; ++kb) {
    if (kb + 1 < KB16S) {
      if ((kb & 1) == 0) { NPF_STEP2(fr[0], fr[1], kb + 1) } else { NPF_STEP2(fr[1], fr[0], kb + 1) }
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[(KB16S - 1) & 1][0]), "+v"(fr[(KB16S - 1) & 1][1]));
    }
    if constexpr (BF16) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[kb & 1][0]), curb[kb], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[kb & 1][1]), curb[kb], acc1, 0, 0, 0);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[kb & 1][0][s], cur[kb][s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[kb & 1][1][s], cur[kb][s], acc1, 0, 0, 0);
      }
    }
    side(kb);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef NPF_STEP2
}

// One K -> N layer on the fast path (K = 16 KB16S in {32, 64, 128, 256} inputs, N = 32 NB outputs, full
// slabs), software-pipelined over its NB slabs.  Stage I:
//   MFMA loop of slab I accumulating into out[2I], out[2I+1] (initialised with the biases), with
//   inside it, in this order over the loop's KB16S blocks:
//     the epilogue of slab I-1, in place (addend / relu / relu-backward mask), over the first E blocks
//     the addend loads of slab I (HBM; consumed in stage I+1)
//     the K/32 DMA pieces of slab I+1 (one SALU add + one instruction each), one per block
//     the bias piece of slab I+1
//   barrier (slab I consumed by the workgroup, slab I+1 landed).
// EPI: 0 = out + addend, 1 = relu(out + addend), 2 = addend > 0 ? out : 0.
#ifndef NPF_DMA_KB0
#define NPF_DMA_KB0 0  // k-block of the stage's first DMA piece
#endif
#ifdef NPF_STAMPS
// Diagnostic build only (tools/stamp_probe.py): per-phase cycle sums of the slab loop of wave 0
// of workgroup 0, written to a buffer nothing else reads.
__device__ unsigned long long g_stamps[16];  // wave 0 (group A), wave 4 (group B, paired variant)
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define NPF_STAMP(i) { const unsigned long long t__ = stamp(); st_sum[i] += t__ - st_last; st_last = t__; }
#define NPF_STAMP_ARGS , unsigned long long (&st_sum)[8], unsigned long long& st_last
#define NPF_STAMP_PASS , st_sum, st_last
#else
#define NPF_STAMP(i)
#define NPF_STAMP_ARGS
#define NPF_STAMP_PASS
#endif

// The program's op table in LDS (bf16 instance).  The descriptors are kernel arguments: every `g.ops[i].field` is a scalar
// load whose first touch of a descriptor misses the scalar cache and costs a memory round trip (measured with
// tools/stamp_probe_bf16.py: ~4800 cycles at the top of EVERY layer, a quarter of the bare bf16 chain) -- and a scalar
// load cannot be issued ahead, it shares lgkmcnt with the stream of LDS fragment waits.  So the workgroup copies the table
// to LDS once (vector loads, one round trip), and an op is fetched by nine broadcast ds_read_b64 + readfirstlane.
constexpr int kOpDwords = sizeof(npf_op_t) / 4;
static_assert(sizeof(npf_op_t) == 80, "npf_op_t layout");
__device__ __forceinline__ npf_op_t lds_op(const float* ops_lds, int i) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)(ops_lds + i * kOpDwords);
  u32x2_t r[10];
  asm volatile(
      "ds_read_b64 %0, %10\n\tds_read_b64 %1, %10 offset:8\n\tds_read_b64 %2, %10 offset:16\n\tds_read_b64 %3, %10 offset:24\n\t"
      "ds_read_b64 %4, %10 offset:32\n\tds_read_b64 %5, %10 offset:40\n\tds_read_b64 %6, %10 offset:48\n\tds_read_b64 %7, %10 offset:56\n\t"
      "ds_read_b64 %8, %10 offset:64\n\tds_read_b64 %9, %10 offset:72\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
        "=&v"(r[9])
      : "v"(a));
  unsigned d[kOpDwords];
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    d[2 * k] = (unsigned)__builtin_amdgcn_readfirstlane((int)r[k][0]);
    d[2 * k + 1] = (unsigned)__builtin_amdgcn_readfirstlane((int)r[k][1]);
  }
  npf_op_t o;
  __builtin_memcpy(&o, d, sizeof(o));
  // (a generic pointer rebuilt from integers makes every access through it a flat_load / flat_store -- slower to issue,
  // counted on vmcnt AND lgkmcnt, out of order; built as a global pointer first, the accesses are global_*)
  static_assert(offsetof(npf_op_t, p0) == 32 && offsetof(npf_op_t, p1) == 40 && offsetof(npf_op_t, p2) == 48 &&
                offsetof(npf_op_t, p3) == 72, "npf_op_t layout");
  typedef const __attribute__((address_space(1))) void* gvoid_t;
  typedef __attribute__((address_space(1))) void* gvoid_w_t;
  o.p0 = (const void*)(gvoid_t)(((unsigned long long)d[9] << 32) | d[8]);
  o.p1 = (const void*)(gvoid_t)(((unsigned long long)d[11] << 32) | d[10]);
  o.p2 = (const void*)(gvoid_t)(((unsigned long long)d[13] << 32) | d[12]);
  o.p3 = (void*)(gvoid_w_t)(((unsigned long long)d[19] << 32) | d[18]);
  return o;
}

// RING (bf16 instance, layers without a per-point PT addend: EPI 1 / 0 on the bias alone, or EPI 3 = mask bits).
// The layer is ONE software pipeline over its 8 NB k-steps; the slabs stream through a four-slot LDS ring:
//   * fragment reads run three k-steps ahead of the MFMAs and cross the slab boundary (k-steps 5..7 of stage I read slab
//     I + 1, whose biases are fetched at k-step 5 straight into the next accumulators): no LDS round trip at a stage start;
//   * ONE raw s_barrier per stage, in its middle (after k-step 3), preceded by a counted s_waitcnt vmcnt: "slab I + 1 has
//     landed for every wave, every wave is done with slab I - 1".  Nothing at the stage boundary itself;
//   * the second half of stage I issues the DMA of stream slab I + 3 (into the slot slab I - 1 has just left): two stages
//     of flight time before the wait of stage I + 2.  A layer that starts with only its slab 0 in LDS issues slabs 1 and
//     2 in the first half of stage 0 (and pays one DMA latency there);
//   * CHAINED hand-over: when the next LINEAR is a ring layer on the same weight geometry (`peek`, at the top of stage
//     NB - 3), its slabs 0..2 are simply stream slabs NB..NB + 2, issued by the last three stages; that layer then starts
//     with `pre3` and this one ends without any drain.  Otherwise the last stage hands over through the generic DMA code
//     (slab 0 of whatever comes next, vmcnt(0), full barrier);
//   * NPF_F_STORE_IN (`st16`): the PT16 copy of the layer's input IS the packed `curb`: stage I stores chunk I at k-step 4,
//     in front of that stage's DMA, and the counted waits of the stages >= 1 allow for it.
// No register-destination load is issued inside the stages (a wait on one would drag every older DMA with it: vmcnt
// retires in order); all LDS reads are inline asm (a C++ LDS load behind an LDS-DMA in flight makes hipcc insert
// vmcnt(0)).  The accumulators are `cur` itself (the input lives on as the packed `curb`): no copy back.
template <int EPI, int KB16S, int NB, int MAXB, bool PAIRED, bool BF16, bool P16, class Peek, class NextLayer>
__device__ __forceinline__ void fast_layer_ring(const Wave& w, float* smem, int& slot, f32x4 (&cur)[MAXB], f32x4 (&)[MAXB],
                                                const SlabOp& op, const unsigned* mbits, const unsigned short* st16,
                                                unsigned* mst, bool pre3, Peek peek, NextLayer next_layer NPF_STAMP_ARGS) {
  static_assert(BF16 && !PAIRED && NB >= 4 && KB16S == 8 && (EPI == 0 || EPI == 1 || EPI == 3), "ring variant");
  constexpr int Kp = 16 * KB16S;          // floats per LDS row
  constexpr int NPW = Kp / 32;            // weight DMA pieces per wave and slab (+ 1 for the biases)
  constexpr int kRowBlk = 16 * Kp * 4;    // bytes between the two 16-row output blocks of a slab
  static_assert(NPW == 4, "five DMA instructions per slab over k-steps 4..7");
  unsigned mw[2] = {0u, 0u};
  if constexpr (EPI == 3) {  // the whole layer's mask: two words per lane, first used in stage 1
    mw[0] = mbits[0];
    if constexpr (NB > 4) mw[1] = mbits[128];
  }
  bf16x8 curb[MAXB / 2] = {};
#pragma unroll
  for (int st = 0; st < KB16S; ++st) curb[st] = pack_bf16(cur[2 * st], cur[2 * st + 1]);
  unsigned mo[2] = {0u, 0u};  // NPF_F_STORE_BITS (`mst`): the output's ReLU bits, shifted in as the epilogue goes (PTM order)
  auto epi_part = [&](int I, int part) __attribute__((always_inline)) {
    const int j = part >> 1, e0 = (part & 1) * 2;
    f32x4 o = cur[2 * I + j];
#pragma unroll
    for (int e = e0; e < e0 + 2; ++e) {
      if constexpr (EPI == 3) o[e] = mask_bit(mw[(2 * I + j) >> 3], (2 * I + j) & 7, e) ? o[e] : 0.f;
      else if constexpr (EPI == 1) o[e] = fmaxf(o[e], 0.f);
    }
    if constexpr (EPI == 1) {
      if (mst != nullptr) {
#pragma unroll
        for (int e = e0; e < e0 + 2; ++e)
          asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mo[(2 * I + j) >> 3]) : "v"(o[e]) : "vcc");
      }
    }
    cur[2 * I + j] = o;
  };
  const char* wbase = (const char*)op.W;
  const float* bias_src = op.bias;
  const int slab_stride = op.slab_stride, step = op.step;
  unsigned lo[2] = {op.lo[0], op.lo[1]};  // (bf16 256-wide rows: 2 rows per piece, the swizzle repeats every 2 pieces)
  const int s0 = slot;
  // DMA piece i (i < NPW: weights, i == NPW: the biases) of slab `sb_src` of the layer (wb, bs) into ring slot (s0 + sb) & 3
  auto issue_of = [&](const char* wb, const float* bs, int sb_src, int sb, int i) __attribute__((always_inline)) {
    float* dst = smem + ((s0 + sb) & 3) * kRingFloats;
    if (i < NPW) {
      dma16_so(wb + (size_t)sb_src * slab_stride * 4 + (size_t)(i * step) * 4, lo[i & 1], dst + w.wave * 256 + i * (kWaves * 256));
    } else {
      // (every wave writes the same 32 biases.  One wave per slab in turn -- 4.25 instead of 5 DMA instructions per wave
      // and stage, with the waits counted per wave -- was measured on one box against this form: config 3 12.87 -> 13.00 ms)
      const char* bsrc = bs != nullptr ? (const char*)(bs + sb_src * kSlabRows) : (const char*)g_zero128;
      dma4_so(bsrc, (unsigned)(w.lane & 31) * 4u, dst + kSlabRows * Kp);
    }
  };
  bool chain = false;
  const char* nwbase = wbase;
  const float* nbias = bias_src;
  // LDS addresses: lane part (row p, chunk (4 kb + g) ^ (p & 15): the lane part only depends on kb & 3) + slot base
  const unsigned lds_ring = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)smem;
  const int ps = w.p & 15;
  unsigned lane_off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) lane_off[t] = (unsigned)(w.p * Kp * 4) + (unsigned)(((t ^ (ps >> 2)) << 6) | ((w.g ^ (ps & 3)) << 4));
  const unsigned bias_off = (unsigned)((kSlabRows * Kp + 4 * w.g) * 4);
  auto slot_lds = [&](int J) { return lds_ring + (unsigned)(((s0 + J) & 3) * (kRingFloats * 4)); };
  f32x4 fq[4][2];
  unsigned a_cur[4], a_nxt[4] = {0u, 0u, 0u, 0u}, b_nxt = 0u;
#define NPF_RD(set, A, kk)                                                              \
  asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"         \
               : "=&v"(fq[set][0]), "=&v"(fq[set][1])                                   \
               : "v"(A[(kk)&3]), "n"(((kk) >> 2) * 256), "n"(((kk) >> 2) * 256 + kRowBlk));
#define NPF_BIAS(J, A)                                                                  \
  asm volatile("ds_read_b128 %0, %2 offset:0\n\tds_read_b128 %1, %2 offset:64"          \
               : "=&v"(cur[2 * (J)]), "=&v"(cur[2 * (J) + 1])                           \
               : "v"(A));
  {
    const unsigned base = slot_lds(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) a_cur[t] = base + lane_off[t];
    const unsigned b0 = base + bias_off;
    NPF_BIAS(0, b0)
    NPF_RD(0, a_cur, 0)
    NPF_RD(1, a_cur, 1)
    NPF_RD(2, a_cur, 2)
  }
  NPF_STAMP(5)  // layer setup (pack, mask words, first reads)
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    if (I == NB - 3) {  // `op` changes here
      chain = peek();
      nwbase = (const char*)op.W;
      nbias = op.bias;
      NPF_STAMP(7)  // peek
    }
    if (I == NB - 1 && !chain) next_layer(smem + ((s0 + NB) & 3) * kRingFloats);  // slab 0 of the next LINEAR
    if (I + 1 < NB) {
      const unsigned base = slot_lds(I + 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) a_nxt[t] = base + lane_off[t];
      b_nxt = base + bias_off;
    }
#pragma unroll
    for (int kb = 0; kb < KB16S; ++kb) {
      // reads three k-steps ahead; the next slab's biases go out just before its first fragments
      if (kb == 5 && I + 1 < NB) { NPF_BIAS(I + 1, b_nxt) }
      if (kb + 3 < KB16S) { NPF_RD((kb + 3) & 3, a_cur, kb + 3) }
      else if (I + 1 < NB) { NPF_RD((kb + 3) & 3, a_nxt, kb + 3 - KB16S) }
      // wait for this k-step's fragments (and, at k-step 0, the biases, which are older): what may stay in flight are the
      // reads issued since -- up to three fragment pairs, plus the bias pair of the next slab from k-step 5 on
      {
        const int G = KB16S * I + kb, last = KB16S * NB - 1;
        const int ahead = (last - G) < 3 ? (last - G) : 3;
        const int cnt = 2 * ahead + ((kb >= 5 && I + 1 < NB) ? 2 : 0);
        const int c = kb & 3;
        if (kb == 0) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fq[c][0]), "+v"(fq[c][1]), "+v"(cur[2 * I]), "+v"(cur[2 * I + 1]) : "n"(cnt));
        else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fq[c][0]), "+v"(fq[c][1]) : "n"(cnt));
        cur[2 * I] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fq[c][0]), curb[kb], cur[2 * I], 0, 0, 0);
        cur[2 * I + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fq[c][1]), curb[kb], cur[2 * I + 1], 0, 0, 0);
      }
      // beside the MFMAs, in the second half: the previous slab's epilogue (the mask words, a register load issued at the
      // top of the layer, are then first needed after a wait that is due anyway) and the DMA of stream slab I + 3
      if (I > 0 && kb >= 4) epi_part(I - 1, kb - 4);
      if (I == 0 && kb < 4) {
        if (!pre3) {  // slabs 1 and 2 of a layer that starts with slab 0 alone: 10 instructions over 4 k-steps
          constexpr int n0[5] = {0, 3, 6, 8, 10};
#pragma unroll
          for (int n = n0[kb]; n < n0[kb + 1]; ++n) issue_of(wbase, bias_src, 1 + n / (NPW + 1), 1 + n / (NPW + 1), n % (NPW + 1));
        }
      }
      if (kb == 4 && st16 != nullptr) __builtin_nontemporal_store(curb[I], (bf16x8*)(st16 + 1024 * I));  // NPF_F_STORE_IN: chunk I of the PT16 copy of the input
      if (kb >= 4) {
#pragma unroll
        for (int i = kb - 4; i < (kb == 7 ? NPW + 1 : kb - 3); ++i) {
          if (I + 3 < NB) issue_of(wbase, bias_src, I + 3, I + 3, i);
          else if (chain) issue_of(nwbase, nbias, I + 3 - NB, I + 3, i);
        }
      }
      if (kb == 3) {
        NPF_STAMP(1)  // first half of the stage
        if (I + 2 < NB || chain) {  // slab I + 2 may stay in flight -- and the input store of stage I - 1 issued in front of it
          if (I >= 1 && st16 != nullptr) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + 2) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + 1) : "memory");
        } else if (I + 1 < NB) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        NPF_STAMP(0)  // counted DMA wait
        __builtin_amdgcn_s_barrier();
        NPF_STAMP(2)  // barrier
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    NPF_STAMP(1)  // second half of the stage
    if (I == NB - 1 && !chain) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // (vmcnt(0): the next layer's slab 0 has landed)
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) a_cur[t] = a_nxt[t];
  }
#undef NPF_RD
#undef NPF_BIAS
  slot = (s0 + NB) & 3;
#pragma unroll
  for (int part = 0; part < 4; ++part) epi_part(NB - 1, part);
  if constexpr (EPI == 1) {
    if (mst != nullptr) {
      mst[0] = mo[0];
      if constexpr (NB > 4) mst[128] = mo[1];
    }
  }
  NPF_STAMP(6)  // last epilogue
}

template <int EPI, int KB16S, int NB, int MAXB, bool PAIRED, bool BF16, bool P16, class NextLayer>
__device__ __forceinline__ void fast_layer(const Wave& w, float* smem, int& slot, f32x4 (&cur)[MAXB],
                                           f32x4 (&out)[MAXB], const SlabOp& op, bool issuer, bool grp_b,
                                           const float* addt, int astep, NextLayer next_layer) {
  constexpr int kSlabFloats = slot_stride(MAXB, BF16, PAIRED);
  constexpr int Kp = 16 * KB16S;
  constexpr int NPW = Kp / 32;                                  // DMA pieces per wave per slab
  constexpr int E = KB16S >= 8 ? 4 : (KB16S >= 4 ? 2 : 1);      // blocks that carry the previous epilogue
  constexpr int PPB = 4 / E;                                    // epilogue parts per block
  static_assert(2 * NB <= MAXB && KB16S <= MAXB, "layer does not fit the register file");
  f32x4 ad[2] = {};
  // bf16: the layer's input lives on as the packed `curb`, so the accumulators ARE `cur` (no copy back: at the interpreter's
  // back edge that copy is 32 VALU moves issued against the sibling wave's MFMA stream, ~4000 cycles per layer)
  f32x4 (&acc)[MAXB] = BF16 ? cur : out;
  bf16x8 curb[MAXB / 2] = {};
  if constexpr (BF16) {
#pragma unroll
    for (int st = 0; st < KB16S; ++st) curb[st] = pack_bf16(cur[2 * st], cur[2 * st + 1]);
  }
  auto epi_part = [&](int I, int part) __attribute__((always_inline)) {
    const int j = part >> 1, e0 = (part & 1) * 2;
    f32x4 o = acc[2 * I + j];
#pragma unroll
    for (int e = e0; e < e0 + 2; ++e) {
      float a;
      if constexpr (P16) {  // element e of block j = 16-bit half (e & 1) of raw word 2 j + (e >> 1)
        const unsigned wd = __builtin_bit_cast(u32x4, ad[0])[2 * j + (e >> 1)];
        a = __builtin_bit_cast(float, (e & 1) ? (wd & 0xffff0000u) : (wd << 16));
      } else {
        a = ad[j][e];
      }
      if (EPI == 2) o[e] = a > 0.f ? o[e] : 0.f;
      else if (EPI == 1) o[e] = fmaxf(o[e] + a, 0.f);
      else o[e] = o[e] + a;
    }
    acc[2 * I + j] = o;
  };
  const char* wbase = (const char*)op.W;
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    const int nxt = next_slot<BF16, PAIRED>(slot);
    float* nslot = smem + nxt * kSlabFloats;
    const float* sl = smem + slot * kSlabFloats;
    if (I == NB - 1 && issuer) next_layer(nslot);  // slab 0 of the next LINEAR (generic DMA): `op` changes here
    if (PAIRED && grp_b) __syncthreads();
    const float* bias = sl + kSlabRows * Kp + 4 * w.g;
    acc[2 * I] = *(const f32x4*)bias;
    acc[2 * I + 1] = *(const f32x4*)(bias + 16);
    const char* src = wbase + (size_t)(I + 1) * op.slab_stride * 4;
    const char* bsrc = op.bias != nullptr ? (const char*)(op.bias + (I + 1) * kSlabRows) : (const char*)g_zero128;
    slab_mfma_side<KB16S, MAXB, BF16>(sl, w, cur, curb, acc[2 * I], acc[2 * I + 1], [&](int kb) __attribute__((always_inline)) {
      if (I > 0 && kb < E) {
#pragma unroll
        for (int q = 0; q < PPB; ++q) epi_part(I - 1, kb * PPB + q);
      }
      if (kb == E) {  // (HBM latency: these must be long gone before the barrier drains vmcnt)
        if constexpr (P16) {  // PT16 mask / addend: one 16-byte chunk per slab, kept raw and unpacked at use
          ad[0] = *(const f32x4*)(addt + 512 * I);
        } else {
          ad[0] = *(const f32x4*)(addt + (8 * I + w.g) * astep);
          ad[1] = *(const f32x4*)(addt + (8 * I + 4 + w.g) * astep);
        }
      }
      if (I < NB - 1 && (!PAIRED || issuer)) {
#pragma unroll
        for (int i = 0; i < NPW; ++i)
          if (kb == (NPF_DMA_KB0 + i < KB16S - 1 ? NPF_DMA_KB0 + i : KB16S - 1))
            dma16_so(src + (size_t)(i * op.step) * 4, op.lo[i & 7], nslot + w.wave * 256 + i * (kWaves * 256));
        // (every issuing wave writes the same 32 biases: no wave-dependent branch in this loop)
        if (kb == (NPF_DMA_KB0 + NPW < KB16S - 1 ? NPF_DMA_KB0 + NPW : KB16S - 1))
          dma4_so(bsrc, (unsigned)(w.lane & 31) * 4u, nslot + kSlabRows * Kp);
      }
    });
    if (!PAIRED || !grp_b) __syncthreads();
    slot = nxt;
  }
#pragma unroll
  for (int part = 0; part < 4; ++part) epi_part(NB - 1, part);
  if constexpr (!BF16) {
#pragma unroll
    for (int b = 0; b < 2 * NB; ++b) cur[b] = out[b];
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


// EXTRA: the instance that also carries the rarely used LayerNorm ops (transformer attention).  They
// are kept out of the main instances on purpose: their per-feature index values are loop-invariant,
// hipcc hoists them out of the op loop and the main kernel then spills hundreds of registers.
// FKB, FNB: the one layer shape (K = 16 FKB, N = 32 FNB) this instance runs software-pipelined
// (0: none).  One shape per instance: two pipelined shapes in one kernel make hipcc spill inside the
// MFMA loops; the launcher picks the instance by the program's most frequent square layer.
// BF16: the instance whose LINEAR ops multiply in bf16 (weights = bf16 images, activations rounded to
// bf16 at the MFMA input, fp32 accumulation, fp32 registers / epilogue / HBM tensors); shared row-major
// weights only, generic slab loop.
template <int MAXB, int WAVES, bool EXTRA, int FKB, int FNB, bool BF16>
__global__ __launch_bounds__(64 * WAVES, (MAXB <= 16 && WAVES == 4) ? 2 : 1) void chain_kernel(const npf_program_t g) {
  constexpr int kMaxB16 = MAXB;
  constexpr int kTilesPerWG = WAVES / 2;
  constexpr bool kPaired = WAVES == 8;
  constexpr int kSlabFloats = slot_stride(MAXB, BF16, kPaired);  // floats between ring slots
  constexpr int kSlots = ring_instance(BF16, kPaired) ? 4 : (kPaired ? 3 : 2);
  constexpr bool kOpsInLds = ring_instance(BF16, kPaired);
  __shared__ __attribute__((aligned(16))) float smem[kSlots * kSlabFloats + (kOpsInLds ? NPF_MAX_OPS * kOpDwords : 0)];
  [[maybe_unused]] const float* ops_lds = smem + kSlots * kSlabFloats;

  Wave w;
  w.tid = threadIdx.x;
  w.lane = w.tid & 63;
  w.p = w.lane & 15;
  w.g = w.lane >> 4;
  w.wave = __builtin_amdgcn_readfirstlane(w.tid >> 6);
  w.half = w.wave & 1;
  const int wtile = w.wave >> 1;
  const bool grp_b = kPaired && w.wave >= kWaves;  // wave-uniform: the consuming phase group
  const bool issuer = !grp_b;
  if (g.wg_per_task) {
    const int wgs = (g.tiles_per_task + kTilesPerWG - 1) / kTilesPerWG;
    w.task = blockIdx.x / wgs;
    w.tile_in_task = (blockIdx.x - w.task * wgs) * kTilesPerWG + wtile;
    w.valid = w.tile_in_task < g.tiles_per_task;
  } else {
    const int gt = blockIdx.x * kTilesPerWG + wtile;
    w.valid = gt < g.n_tasks * g.tiles_per_task;
    w.task = w.valid ? gt / g.tiles_per_task : 0;
    w.tile_in_task = w.valid ? gt - w.task * g.tiles_per_task : 0;
  }
  if (!w.valid) w.tile_in_task = 0;  // keep addresses in range; loads are zeroed, stores skipped
  const int pt = w.tile_in_task * 32 + 16 * w.half + w.p;  // point index inside the task
  const bool pt_ok = w.valid && pt < g.pts_per_task;        // real (non padding) point
  // slab weights are per workgroup: with per-task weights every wave of the WG shares w.task
  const int wg_task = g.wg_per_task ? w.task : 0;

  // Two workgroups share each SIMD and execute the same instruction stream: started together
  // they would stay in lockstep (equal MFMA arbitration keeps them aligned) and idle the
  // matrix pipe during every per-slab scalar phase.  Delay the one in the odd wave slot by
  // about half a slab period so that one workgroup's scalar phase meets the other's MFMAs.
  if (!kPaired && !(g.reserved[0] & 16)) {
    const unsigned wave_slot = __builtin_amdgcn_s_getreg(6148);  // HW_REG_HW_ID[3:0] = WAVE_ID
    if (wave_slot & 1) __builtin_amdgcn_s_sleep(40);  // (bf16 instance: 0 / 10 / 20 / 40 / 80 measured, within 1 % of each other)
  }
  // (Staggering the first round of workgroups by XCD -- block b runs on XCD b % 8 -- so that the 256 CUs do not reach the
  // stores between two layers together was measured: config 2 10.05 -> 10.17 ms, config 3 12.88 -> 12.93: not kept.)

  f32x4 cur[kMaxB16], out[kMaxB16];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < kMaxB16; ++b) {
    cur[b] = zero4;
    out[b] = zero4;
  }
  float acc_dot = 0.f;

  // slab prefetch cursor: (op, slab) of the next slab to load, and that op's constants
  SlabCursor pf;
  pf.op = 0;
  pf.nb = 0;
  SlabOp pfs;
  pfs.n_slabs = 0;
  pfs.fast = false;
  if constexpr (kOpsInLds) {
    for (int t = w.tid; t < g.n_ops * kOpDwords; t += 64 * WAVES) ((unsigned*)ops_lds)[t] = ((const unsigned*)g.ops)[t];
    __syncthreads();  // (published before the first seek below; later reads are inline-asm ds_read)
  }
  auto seek = [&]() {
    if constexpr (kOpsInLds) {
      while (pf.op < g.n_ops) {
        const npf_op_t cand = lds_op(ops_lds, pf.op);
        if (cand.op == NPF_OP_LINEAR) {
          pf.nb = 0;
          pfs = make_slab_op<BF16>(cand, wg_task);
          slab_fast_setup(pfs, w);
          return;
        }
        ++pf.op;
      }
      pf.nb = 0;
      return;
    }
    while (pf.op < g.n_ops && g.ops[pf.op].op != NPF_OP_LINEAR) ++pf.op;
    pf.nb = 0;
    if (pf.op < g.n_ops) {
      pfs = make_slab_op<BF16>(g.ops[pf.op], wg_task);
      slab_fast_setup(pfs, w);
    }
  };
  auto advance = [&]() {
    if (++pf.nb >= pfs.n_slabs) {
      ++pf.op;
      seek();
    }
  };
  if (issuer) seek();
  else pf.op = g.n_ops;
  int slot = 0;
  if (pf.op < g.n_ops) {  // prologue: slab 0 of the first LINEAR
    SlabDma d0 = dma_begin(pfs, pf.nb, w, smem, true);
    dma_finish(pfs, d0, w);
    advance();
  }
  __syncthreads();  // (its vmcnt(0) retires the DMA)

#ifdef NPF_STAMPS
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = stamp();
#endif
  const float* Z = g_zero16;
  for (int ip = 0; ip < g.n_ops; ++ip) {
    [[maybe_unused]] npf_op_t o_lds;
    if constexpr (kOpsInLds) o_lds = lds_op(ops_lds, ip);
    const npf_op_t& o = kOpsInLds ? o_lds : g.ops[ip];
    const int opc = o.op;
    NPF_STAMP(3)  // back edge of the interpreter loop + fetch of the opcode
    if (opc == NPF_OP_LINEAR) {
      const int KB16 = ((o.i0 + 31) >> 5) * 2, N = o.i1;
      // bf16 instance: LDS row length in floats and number of 32-feature k-steps; the layer's input packed
      [[maybe_unused]] const int bf_kpf = ((((o.i0 + 31) >> 5) * 16 + 31) >> 5) * 32, bf_steps = (o.i0 + 31) >> 5;
      [[maybe_unused]] bf16x8 curb[kMaxB16 / 2];
      if constexpr (BF16) {
#pragma unroll
        for (int st = 0; st < kMaxB16 / 2; ++st) curb[st] = pack_bf16(cur[2 * st], cur[2 * st + 1]);
      }
      const int NB = (N + kSlabRows - 1) / kSlabRows;
      const bool relu = (o.flags & NPF_F_RELU) != 0;
      const bool mask = (o.flags & NPF_F_MASK_PT) != 0;  // out = (tile > 0) ? acc : 0  (relu backward)
      // bf16 instance: the same with the mask as bits (PTM tensor p2, two words per lane and 256 features)
      [[maybe_unused]] const bool maskb = BF16 && (o.flags & NPF_F_MASK_BITS) != 0;
      [[maybe_unused]] const unsigned* mbits = (const unsigned*)Z;
      if constexpr (BF16) {
        if (maskb) {
          const size_t tile = (size_t)eff_task(w, o.i4) * g.tiles_per_task + w.tile_in_task;
          mbits = (const unsigned*)o.p2 + (tile * ((N + 127) >> 7) * 4 + w.g) * 32 + 16 * w.half + w.p;  // word q at + 128 q
        }
      }
      const bool add = ((o.flags & (NPF_F_ADD_PT | NPF_F_MASK_PT)) != 0) & w.valid & !(BF16 && (o.flags & NPF_F_P16));
      [[maybe_unused]] const bool add16 = BF16 && ((o.flags & (NPF_F_ADD_PT | NPF_F_MASK_PT)) != 0) & w.valid & ((o.flags & NPF_F_P16) != 0);
      const bool add_rm = ((o.flags & NPF_F_ADD_RM) != 0) & w.valid;  // row-major addend: feature quad stride 4 floats
      const float* addt = add ? pt_lane(o.p2, g, w, ((N + 31) >> 5) * 32, o.i4)
                              : (add_rm ? rm_lane(o.p2, g, w, pt, N, o.i4) : Z);
      const int astep = add ? 128 : (add_rm ? 4 : 0);  // (no addend: every load reads the zero buffer)
      // bf16 instance: the addend / mask may be a PT16 tensor: one chunk per slab at addt16 + 1024 nb (bf16 units)
      [[maybe_unused]] const bool p16 = add16;
      [[maybe_unused]] const unsigned short* addt16 =
          p16 ? pt16_lane(o.p2, g, w, ((N + 31) >> 5) * 32, o.i4) : (const unsigned short*)Z;
      NPF_STAMP(5)  // everything between LINEAR slab loops (other ops, layer setup)
      // Fast path for the 256 -> 256 layers (all the heavy ones at r = 256, attention included):
      // the 8 slab steps are unrolled, so the epilogue writes the slab's own two output blocks
      // (static registers), the slab DMA is one SALU add + one instruction per piece, and the
      // cursor logic runs once per layer.  While the sibling wave of the SIMD is inside its
      // MFMA loop every VALU instruction of this wave waits for an fp32 MFMA to drain (~32
      // cycles), so what counts outside the MFMA loop is the number of VALU instructions.
      bool fast_shape = false;
      if constexpr (FKB > 0)  // (bf16 instance: FKB counts 32-feature steps)
        fast_shape = (BF16 ? o.i0 == 32 * FKB : (KB16 == FKB && o.i0 == 16 * FKB)) && N == 32 * FNB && g.reserved[0] == 0 &&
                     (grp_b || (pf.op == ip && (pf.nb == 1 || (ring_instance(BF16, kPaired) && pf.nb == 3)) && pfs.fast)) &&
                     !(p16 && !mask);
      // NPF_F_STORE_IN (bf16 programs): a ring layer stores its packed input itself, chunk by chunk; everything else
      // stores it here, before the layer, exactly like NPF_OP_STORE_PT
      [[maybe_unused]] const bool ring_plain = (o.flags & (NPF_F_ADD_PT | NPF_F_MASK_PT | NPF_F_ADD_RM)) == 0;
      [[maybe_unused]] const unsigned short* st16 = nullptr;
      if constexpr (BF16) {
        if ((o.flags & NPF_F_STORE_IN) && w.valid) {
          const int Fin = ((o.i0 + 31) >> 5) * 32;
          if (o.flags & NPF_F_STORE_P16) {
            unsigned short* t16 = (unsigned short*)pt16_lane(o.p3, g, w, Fin, 0);
            if (ring_instance(BF16, kPaired) && fast_shape && (maskb || ring_plain)) {
              st16 = t16;
            } else {
#pragma unroll
              for (int st = 0; st < kMaxB16 / 2; ++st)
                if (32 * st < Fin) *(bf16x8*)(t16 + 1024 * st) = curb[st];
            }
          } else {
            float* t = (float*)pt_lane(o.p3, g, w, Fin, 0);
#pragma unroll
            for (int b = 0; b < kMaxB16; ++b)
              if (16 * b < Fin) *(f32x4*)(t + (4 * b + w.g) * 128) = cur[b];
          }
        }
      }
      // NPF_F_STORE_BITS (bf16 programs): the ReLU ring layer shifts the bits in inside its epilogue; everything else
      // runs the NPF_OP_STORE_MASK code behind the layer
      [[maybe_unused]] unsigned* mst = nullptr;
      [[maybe_unused]] bool mst_after = false;
      if constexpr (BF16) {
        if ((o.flags & NPF_F_STORE_BITS) && w.valid) {
          const size_t tile = (size_t)w.task * g.tiles_per_task + w.tile_in_task;
          mst = (unsigned*)o.p2 + (tile * ((N + 127) >> 7) * 4 + w.g) * 32 + 16 * w.half + w.p;
          mst_after = !(ring_instance(BF16, kPaired) && fast_shape && ring_plain && relu && !maskb);
        }
      }
      if (fast_shape) {
        if constexpr (FKB > 0) {
          // this layer's slabs 1.. stream inside the pipeline; then the cursor jumps to the next
          // LINEAR and its slab 0 goes out through the generic DMA code before the last stage
          auto next_layer = [&](float* nslot) __attribute__((always_inline)) {
            pf.nb = FNB - 1;
            advance();
            if (pf.op < g.n_ops) {
              SlabDma d = dma_begin(pfs, pf.nb, w, nslot, true);
              dma_finish(pfs, d, w);
              advance();
            }
          };
          bool done = false;
          NPF_STAMP(4)  // LINEAR prologue up to the dispatch of the pipelined layer
          if constexpr (ring_instance(BF16, kPaired)) {
            // two slabs in flight: every layer that needs no per-point tensor inside its stages
            const bool plain = ring_plain;
            const bool pre3 = pf.nb == 3;  // (the previous ring layer issued this layer's slabs 0..2)
            // The cursor leaves this layer three stages before its end.  Next LINEAR = a ring layer on the same weight
            // geometry (everything `lo`, `step`, `slab_stride` depend on): only the two base pointers change, its first
            // three slabs are issued by this layer's last three stages (returns true).  Anything else: the generic setup,
            // and slab 0 through `ring_tail` in the last stage.
            const int tK = o.i0, tN = o.i1, tmode = o.i2, tld = o.i3;
            auto peek = [&]() __attribute__((always_inline)) -> bool {
              ++pf.op;
              pf.nb = 0;
              while (pf.op < g.n_ops) {
                const npf_op_t cand = lds_op(ops_lds, pf.op);
                if (cand.op == NPF_OP_LINEAR) {
                  const bool same = cand.i0 == tK && cand.i1 == tN && tmode == NPF_W_ROWMAJOR && cand.i2 == NPF_W_ROWMAJOR &&
                                    cand.i3 == tld && (cand.flags & (NPF_F_ADD_PT | NPF_F_MASK_PT | NPF_F_ADD_RM)) == 0 &&
                                    (((uintptr_t)cand.p0) & 15) == 0 && (cand.s0 & 3) == 0;
                  if (same) {
                    pfs.W = (const float*)cand.p0 + (size_t)wg_task * cand.s0;
                    pfs.bias = cand.p1 ? (const float*)cand.p1 + (size_t)wg_task * cand.s1 : nullptr;
                    pf.nb = 3;
                    return true;
                  }
                  pfs = make_slab_op<BF16>(cand, wg_task);
                  slab_fast_setup(pfs, w);
                  return false;
                }
                ++pf.op;
              }
              return false;
            };
            auto ring_tail = [&](float* nslot) __attribute__((always_inline)) {
              if (pf.op < g.n_ops) {
                SlabDma d = dma_begin(pfs, pf.nb, w, nslot, true);
                dma_finish(pfs, d, w);
                advance();
              }
            };
            if (maskb) {
              fast_layer_ring<3, FKB, FNB, MAXB, kPaired, BF16, false>(w, smem, slot, cur, out, pfs, mbits, st16, nullptr, pre3, peek, ring_tail NPF_STAMP_PASS);
              done = true;
            } else if (plain && relu) {
              fast_layer_ring<1, FKB, FNB, MAXB, kPaired, BF16, false>(w, smem, slot, cur, out, pfs, mbits, st16, mst, pre3, peek, ring_tail NPF_STAMP_PASS);
              done = true;
            } else if (plain) {
              fast_layer_ring<0, FKB, FNB, MAXB, kPaired, BF16, false>(w, smem, slot, cur, out, pfs, mbits, st16, nullptr, pre3, peek, ring_tail NPF_STAMP_PASS);
              done = true;
            }
          }
          if (done) {
          } else if (mask && p16) {
            if constexpr (BF16)
              fast_layer<2, FKB, FNB, MAXB, kPaired, BF16, true>(w, smem, slot, cur, out, pfs, issuer, grp_b, (const float*)addt16, 0, next_layer);
          } else if (mask) fast_layer<2, FKB, FNB, MAXB, kPaired, BF16, false>(w, smem, slot, cur, out, pfs, issuer, grp_b, addt, astep, next_layer);
          else if (relu) fast_layer<1, FKB, FNB, MAXB, kPaired, BF16, false>(w, smem, slot, cur, out, pfs, issuer, grp_b, addt, astep, next_layer);
          else fast_layer<0, FKB, FNB, MAXB, kPaired, BF16, false>(w, smem, slot, cur, out, pfs, issuer, grp_b, addt, astep, next_layer);
        }
      } else {
      // Generic path: runtime slab loop with *static* register indices: finished blocks enter a register
      // queue (out[] shifts down by one slab per iteration), so the loop body exists once
      // (~12 KB of code instead of 8 unrolled copies that overflow the 64 KB instruction
      // cache) and no dynamically indexed register array is needed.
      [[maybe_unused]] unsigned gmw[2] = {0u, 0u};
      if constexpr (BF16) {
        // (the queue's old content is never read: defining it here keeps 64 registers from being carried -- and spilled --
        // around the interpreter loop on behalf of the ring layers, which do not use `out` at all)
#pragma unroll
        for (int b = 0; b < kMaxB16; ++b) out[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (maskb && w.valid) {
          gmw[0] = mbits[0];
          if (N > 128) gmw[1] = mbits[128];
        }
      }
      for (int nb = 0; nb < NB; ++nb) {
        // 1. addend tiles (consumed after the barrier).  Before the slab DMA: hipcc guards the reuse of the
        // addend registers with a full vmcnt drain, which must not include the DMA issued for the next slab
        f32x4 ad[kBlk];
#pragma unroll
        for (int j = 0; j < kBlk; ++j) ad[j] = *(const f32x4*)(addt + (4 * kBlk * nb + 4 * j + w.g) * astep);
        if constexpr (BF16) {
          if (p16) {  // (the fp32 loads above read the zero buffer in this case: astep applies to an fp32 tensor)
            const u32x4 r = *(const u32x4*)(addt16 + 1024 * nb);
            ad[0] = pt16_lo(r);
            ad[1] = pt16_hi(r);
          }
        }
        // 2. start filling the other slot with the next slab (possibly the next layer's)
        if (!grp_b) NPF_STAMP(4)  // (group A) loop back-edge; (without stamps: the statement below is its body)
        if (pf.op < g.n_ops) {
          const int nslot = next_slot<BF16, kPaired>(slot);
          SlabDma d = dma_begin(pfs, pf.nb, w, smem + nslot * kSlabFloats, !(g.reserved[0] & 1));
          NPF_STAMP(6)  // loop top
          dma_finish(pfs, d, w);
          NPF_STAMP(0)  // DMA pieces
          advance();
          NPF_STAMP(7)  // cursor advance (+ next layer's slab constants)
        }
        // 3. the slab's MFMAs
        f32x4 acc[kBlk];
#pragma unroll
        for (int j = 0; j < kBlk; ++j) acc[j] = zero4;
        const float* sl = smem + slot * kSlabFloats;
        if (grp_b) {
          NPF_STAMP(6)
          __syncthreads();  // group B's barrier: the slab landed (A passed its vmcnt(0))
          NPF_STAMP(4)      // (group B) barrier wait
        }
        if (!(g.reserved[0] & 8)) {
          if constexpr (BF16) {
            if (N - nb * kSlabRows > 16) slab_mfma_bf16<2, MAXB>(sl, bf_kpf, bf_steps, w, curb, acc);
            else slab_mfma_bf16<1, MAXB>(sl, bf_kpf, bf_steps, w, curb, acc);
          } else {
            if (N - nb * kSlabRows > 16) slab_mfma_any<2, MAXB>(sl, KB16, w, cur, acc);
            else slab_mfma_any<1, MAXB>(sl, KB16, w, cur, acc);
          }
        }
        NPF_STAMP(1)  // addend loads + MFMA loop
        if (!grp_b && !(g.reserved[0] & 4)) __syncthreads();  // slab consumed; vmcnt(0) lands the next one
        NPF_STAMP(2)  // barrier (+ wait for the DMA)
        slot = next_slot<BF16, kPaired>(slot);
#pragma unroll
        for (int b = 0; b < kMaxB16 - kBlk; ++b) out[b] = out[b + kBlk];
#pragma unroll
        for (int j = 0; j < kBlk; ++j) {
          f32x4 v;
          if (maskb) {
            if constexpr (BF16) {
              const int blk = kBlk * nb + j;  // (uniform) the block: word blk >> 3, bits 31 - 4 (blk & 7) - e
              const unsigned word = (blk >= 8 ? gmw[1] : gmw[0]) << (4 * (blk & 7));
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (word & (0x80000000u >> e)) ? acc[j][e] : 0.f;
            }
          } else if (mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ad[j][e] > 0.f ? acc[j][e] : 0.f;
          } else {
            v = acc[j] + ad[j];
          }
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          out[kMaxB16 - kBlk + j] = v;
        }
        NPF_STAMP(3)  // epilogue
      }
      // the layer's blocks now sit at out[16 - 2 NB .. 15]
      // (one unrolled copy per possible NB keeps every register index static)
#pragma unroll
      for (int nbv = 1; nbv <= kMaxB16 / kBlk; ++nbv)
        if (NB == nbv) {
#pragma unroll
          for (int b = 0; b < kBlk * nbv; ++b) cur[b] = out[b + kMaxB16 - kBlk * nbv];
        }
      }  // generic slab loop
      if constexpr (BF16) {
        if (mst_after) {  // NPF_F_STORE_BITS on a layer that did not go through the ReLU ring
#pragma unroll
          for (int q = 0; q < kMaxB16 / 8; ++q) {
            if (128 * q < N) {
              unsigned word = 0u;
#pragma unroll
              for (int bb = 0; bb < 8; ++bb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float x = (16 * (8 * q + bb) < N) ? cur[8 * q + bb][e] : 0.f;
                  asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(word) : "v"(x) : "vcc");
                }
              mst[128 * q] = word;
            }
          }
        }
      }
    } else if (opc == NPF_OP_LOAD_PT || opc == NPF_OP_ADD_PT || opc == NPF_OP_MASK_POS || opc == NPF_OP_ROWDOT_PT ||
               opc == NPF_OP_SOFTMAX_BWD) {
      const int FB = o.i0 >> 4;
      const float* t = pt_lane(o.p0, g, w, o.i0, o.i4);
      float dot = 0.f;
      // bf16 instance: the operand may be a PT16 tensor (bf16 tiles): one 16-byte chunk per block pair
      [[maybe_unused]] const bool is16 = BF16 && (o.flags & NPF_F_P16);
      [[maybe_unused]] const unsigned short* t16 = is16 ? pt16_lane(o.p0, g, w, o.i0, o.i4) : (const unsigned short*)Z;
      [[maybe_unused]] u32x4 r16 = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int b = 0; b < kMaxB16; ++b) {
        if (b < FB) {
          f32x4 v = zero4;
          if constexpr (BF16) {
            if (is16) {
              if (!(b & 1) && w.valid) r16 = *(const u32x4*)(t16 + 1024 * (b >> 1));
              v = (b & 1) ? pt16_hi(r16) : pt16_lo(r16);
            } else if (w.valid) {
              v = *(const f32x4*)(t + (4 * b + w.g) * 128);
            }
          } else
          if (w.valid) v = *(const f32x4*)(t + (4 * b + w.g) * 128);
          f32x4 c = cur[b];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (opc == NPF_OP_LOAD_PT) c[s] = v[s];
            else if (opc == NPF_OP_ADD_PT) c[s] = o.i1 ? fmaxf(c[s] + v[s], 0.f) : c[s] + v[s];
            else if (opc == NPF_OP_MASK_POS) c[s] = v[s] > 0.f ? c[s] : 0.f;
            else if (opc == NPF_OP_ROWDOT_PT) dot += c[s] * v[s];
            else c[s] = o.f0 * v[s] * (c[s] - acc_dot);
          }
          cur[b] = c;
        }
      }
      if (opc == NPF_OP_ROWDOT_PT) acc_dot = xg_sum(dot);
    } else if (BF16 && (opc == NPF_OP_STORE_MASK || opc == NPF_OP_MASK_BITS)) {
      if constexpr (BF16) {
        // STORE_MASK: PTM tensor p0 <- (cur > 0), i0 = F;  MASK_BITS: cur <- bit ? cur : 0
        const int FB = o.i0 >> 4;
        const size_t tile = (size_t)eff_task(w, o.i4) * g.tiles_per_task + w.tile_in_task;
        unsigned* t = (unsigned*)o.p0 + (tile * ((o.i0 + 127) >> 7) * 4 + w.g) * 32 + 16 * w.half + w.p;
#pragma unroll
        for (int q = 0; q < kMaxB16 / 8; ++q) {
          if (8 * q < FB) {
            if (opc == NPF_OP_STORE_MASK) {
              unsigned word = 0u;
#pragma unroll
              for (int bb = 0; bb < 8; ++bb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float x = (8 * q + bb < FB) ? cur[8 * q + bb][e] : 0.f;
                  // word = 2 word + (x > 0): one compare and one add-with-carry per value
                  asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(word) : "v"(x) : "vcc");
                }
              if (w.valid) t[128 * q] = word;
            } else {
              const unsigned word = w.valid ? t[128 * q] : 0u;
#pragma unroll
              for (int bb = 0; bb < 8; ++bb)
                if (8 * q + bb < FB) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) cur[8 * q + bb][e] = mask_bit(word, bb, e) ? cur[8 * q + bb][e] : 0.f;
                }
            }
          }
        }
      }
    } else if (opc == NPF_OP_LOAD_RM) {
      const int FB = o.i0 >> 4;
      const float* t = rm_lane(o.p0, g, w, pt, o.i0, o.i4) + 4 * w.g;
#pragma unroll
      for (int b = 0; b < kMaxB16; ++b) {
        if (b < FB) cur[b] = w.valid ? *(const f32x4*)(t + 16 * b) : zero4;
      }
    } else if (opc == NPF_OP_STORE_PT) {
      const int FB = o.i0 >> 4;
      float* t = (float*)pt_lane(o.p0, g, w, o.i0, o.i4);
      if constexpr (BF16) {
        if (o.flags & NPF_F_P16) {  // PT16 destination
          unsigned short* t16 = (unsigned short*)pt16_lane(o.p0, g, w, o.i0, o.i4);
          if (w.valid) {
#pragma unroll
            for (int st = 0; st < kMaxB16 / 2; ++st)
              if (2 * st < FB) __builtin_nontemporal_store(pack_bf16(cur[2 * st], cur[2 * st + 1]), (bf16x8*)(t16 + 1024 * st));
          }
          continue;
        }
      }
      if (w.valid) {
#pragma unroll
        for (int b = 0; b < kMaxB16; ++b)
          if (b < FB) __builtin_nontemporal_store(cur[b], (f32x4*)(t + (4 * b + w.g) * 128));
      }
    } else if (opc == NPF_OP_STORE_TR) {
      // feature-major copy [task][feature][point]: the layout the slab DMA wants when these
      // activations are used as per-task weights with the points as the contraction index
      const int F = o.i0, ld = o.i1;
      float* dst = (float*)o.p0 + (size_t)w.task * F * ld + pt;
      if (w.valid) {
#pragma unroll
        for (int b = 0; b < kMaxB16; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int f = 16 * b + 4 * w.g + e;
            if (f < F) dst[(size_t)f * ld] = cur[b][e];
          }
      }
    } else if (opc == NPF_OP_STORE_WB || opc == NPF_OP_STORE_TRB) {
      if constexpr (BF16) {
        const int F = o.i0, ld = o.i1, Fp = ((F + 31) >> 5) * 32;
        if (opc == NPF_OP_STORE_WB) {
          // row image: this lane's 8 values of the feature group s (blocks 2s, 2s+1) are 16 contiguous bytes
          unsigned short* dst = (unsigned short*)o.p0 + ((size_t)w.task * ld + pt) * Fp + 8 * w.g;
          if (w.valid) {
#pragma unroll
            for (int st = 0; st < kMaxB16 / 2; ++st)
              if (32 * st < Fp) *(bf16x8*)(dst + 32 * st) = pack_bf16(cur[2 * st], cur[2 * st + 1]);
          }
        } else {
          // transposed image: point p = 32 t + r sits at column 32 t + 8 ((r & 15) >> 2) + 4 (r >> 4) + (r & 3)
          const int r = pt & 31;
          const int col = (pt & ~31) + 8 * ((r & 15) >> 2) + 4 * (r >> 4) + (r & 3);
          unsigned short* dst = (unsigned short*)o.p0 + (size_t)w.task * F * ld + col;
          if (w.valid) {
#pragma unroll
            for (int b = 0; b < kMaxB16; ++b)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int f = 16 * b + 4 * w.g + e;
                const __bf16 v = (__bf16)cur[b][e];
                if (f < F) dst[(size_t)f * ld] = __builtin_bit_cast(unsigned short, v);
              }
          }
        }
      }
    } else if (opc == NPF_OP_LOAD_ROWS) {
      const int kd = o.i0;
      const float* src = (const float*)o.p0 + ((size_t)eff_task(w, o.i4) * g.pts_per_task + pt) * kd;
#pragma unroll
      for (int b = 0; b < kMaxB16; ++b) cur[b] = zero4;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int f = 16 * b + 4 * w.g + s;
          if (pt_ok && f < kd) cur[b][s] = src[f];
        }
    } else if (opc == NPF_OP_STORE_ROWS) {
      const int nd = o.i0;
      float* dst = (float*)o.p0 + ((size_t)eff_task(w, o.i4) * g.pts_per_task + pt) * nd;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int f = 16 * b + 4 * w.g + s;
          if (pt_ok && f < nd) dst[f] = cur[b][s];
        }
    } else if (opc == NPF_OP_SOFTMAX) {
      // i1 = 0: softmax of the row; 1: the same, and (max, sum) go to p0[task][pt][2]; 2: max and
      // sum come from p0 -- a block of a longer row whose statistics were combined on the way
      const int nvalid = o.i0, smode = o.i1;
      const int FB = ((nvalid + 31) >> 5) * 2;
      const float scale = o.f0;
      float* stats = (float*)o.p0 + ((size_t)w.task * g.pts_per_task + pt) * 2;
      float m = -INFINITY, sum = 0.f;
      if (smode == 2) {
        if (pt_ok) {
          m = stats[0];
          sum = stats[1];
        } else {
          m = 0.f;
          sum = 1.f;
        }
      } else {
#pragma unroll
        for (int b = 0; b < kMaxB16; ++b)
          if (b < FB)
#pragma unroll
            for (int s = 0; s < 4; ++s)
              if (16 * b + 4 * w.g + s < nvalid) m = fmaxf(m, cur[b][s]);
        m = xg_max(m);
      }
      float part = 0.f;
#pragma unroll
      for (int b = 0; b < kMaxB16; ++b)
        if (b < FB)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bool ok = 16 * b + 4 * w.g + s < nvalid;
            const float e = ok ? expf((cur[b][s] - m) * scale) : 0.f;
            cur[b][s] = e;
            part += e;
          }
      if (smode != 2) sum = xg_sum(part);
      if (smode == 1 && pt_ok && w.g == 0) {
        stats[0] = m;
        stats[1] = sum;
      }
      const float inv = 1.f / sum;
#pragma unroll
      for (int b = 0; b < kMaxB16; ++b)
        if (b < FB) cur[b] *= inv;
    } else if (opc == NPF_OP_LAYERNORM || opc == NPF_OP_LAYERNORM_BWD) {
      if constexpr (EXTRA) {
        // nn.LayerNorm over the F valid features of the point (biased variance, eps inside the root).
        // Forward: cur = x.  Backward: cur = dy, x reloaded from the saved forward input.
        // gamma / beta are padded with zeros to a multiple of 32 floats (host side) and the padding
        // features of x / dy are zero, so no per-feature masks are needed: the padding's share of
        // the variance sum, (Fp - F) mean^2, is subtracted after the reduction.
        const int F = o.i0, FB = ((F + 31) >> 5) * 2;
        const float invF = 1.f / (float)F, eps = o.f0, npad = (float)(16 * FB - F);
        const bool bwd = opc == NPF_OP_LAYERNORM_BWD;
        const float* gam = (const float*)(bwd ? o.p1 : o.p0) + 4 * w.g;
        const float* bet = (const float*)o.p1 + 4 * w.g;  // forward only
        const int lim = F - 4 * w.g;                      // feature 16 b + 4 g + e is valid iff 16 b + e < lim
        const float* xt = bwd ? pt_lane(o.p0, g, w, 16 * FB, 0) : Z;
        f32x4 x[kMaxB16];
        float s1 = 0.f;
#pragma unroll
        for (int b = 0; b < kMaxB16; ++b) {
          x[b] = zero4;
          if (b < FB) {
            if (bwd) {
              if (w.valid) x[b] = *(const f32x4*)(xt + (4 * b + w.g) * 128);
            } else {
              x[b] = cur[b];
            }
            s1 += (x[b][0] + x[b][1]) + (x[b][2] + x[b][3]);
          }
        }
        const float mean = xg_sum(s1) * invF;
        float s2 = 0.f;
#pragma unroll
        for (int b = 0; b < kMaxB16; ++b)
          if (b < FB) {
            x[b] = x[b] - mean;
            s2 += (x[b][0] * x[b][0] + x[b][1] * x[b][1]) + (x[b][2] * x[b][2] + x[b][3] * x[b][3]);
          }
        const float var = fmaxf((xg_sum(s2) - npad * mean * mean) * invF, 0.f);
        const float rstd = 1.f / sqrtf(var + eps);
        if (!bwd) {
#pragma unroll
          for (int b = 0; b < kMaxB16; ++b)
            if (b < FB) cur[b] = x[b] * rstd * *(const f32x4*)(gam + 16 * b) + *(const f32x4*)(bet + 16 * b);
        } else {
          float* xo = o.p2 ? (float*)pt_lane(o.p2, g, w, 16 * FB, 0) : nullptr;
          float m1 = 0.f, m2 = 0.f;
#pragma unroll
          for (int b = 0; b < kMaxB16; ++b)
            if (b < FB) {
              const f32x4 xh = x[b] * rstd;
              const f32x4 dyx = cur[b] * xh;  // (dy is zero on padding features)
              const f32x4 gg = cur[b] * *(const f32x4*)(gam + 16 * b);
              x[b] = xh;
              cur[b] = gg;
              m1 += (gg[0] + gg[1]) + (gg[2] + gg[3]);
              m2 += (gg[0] * xh[0] + gg[1] * xh[1]) + (gg[2] * xh[2] + gg[3] * xh[3]);
              if (xo && w.valid) *(f32x4*)(xo + (4 * b + w.g) * 128) = dyx;
            }
          m1 = xg_sum(m1) * invF;
          m2 = xg_sum(m2) * invF;
#pragma unroll
          for (int b = 0; b < kMaxB16; ++b)
            if (b < FB) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                cur[b][e] = (16 * b + e < lim) ? rstd * (cur[b][e] - m1 - x[b][e] * m2) : 0.f;
            }
        }
      }
    } else if (opc == NPF_OP_ADD_TASKVEC) {
      const int FB = o.i0 >> 4;
      const float* v = (const float*)o.p0 + (size_t)eff_task(w, o.i4) * o.i0;
#pragma unroll
      for (int b = 0; b < kMaxB16; ++b) {
        if (b < FB) {
          f32x4 c = cur[b] + *(const f32x4*)(v + 16 * b + 4 * w.g);
          if (o.i1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) c[s] = fmaxf(c[s], 0.f);
          }
          cur[b] = c;
        }
      }
    } else if (opc == NPF_OP_RELU || opc == NPF_OP_SCALE) {
#pragma unroll
      for (int b = 0; b < kMaxB16; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) cur[b][s] = (opc == NPF_OP_RELU) ? fmaxf(cur[b][s], 0.f) : o.f0 * cur[b][s];
    }
  }
#ifdef NPF_STAMPS
  NPF_STAMP(5)
  if (blockIdx.x == 0 && (w.tid == 0 || w.tid == 256))
    for (int i = 0; i < 8; ++i) g_stamps[(w.tid >> 5) + i] = st_sum[i];
#endif
}

#ifdef NPF_STAMPS
extern "C" int npf_debug_stamps(unsigned long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) == hipSuccess ? 0 : -1;
}
#endif

static int validate(const npf_program_t* g) {
  if (!g || g->n_ops < 0 || g->n_ops > NPF_MAX_OPS) return NPF_EINVAL;
  if (g->n_tasks <= 0 || g->pts_per_task <= 0) return NPF_EINVAL;
  if (g->tiles_per_task != (g->pts_per_task + 31) / 32) return NPF_EINVAL;
  for (int i = 0; i < g->n_ops; ++i) {
    const npf_op_t& o = g->ops[i];
    if ((o.flags & NPF_F_P16) && g->reserved[2] != 1) return NPF_EINVAL;  // PT16 operands: bf16 instance only
    switch (o.op) {
      case NPF_OP_LINEAR:
        if (o.i0 <= 0 || o.i0 > NPF_MAX_FEATURES || o.i1 <= 0 || o.i1 > NPF_MAX_FEATURES || !o.p0) return NPF_EINVAL;
        if (o.i2 < 0 || o.i2 > 2) return NPF_EINVAL;
        if (o.i2 != NPF_W_ROWMAJOR && !g->wg_per_task) return NPF_EINVAL;  // per-task weights
        if (o.i2 == NPF_W_ROWMAJOR && o.s0 != 0 && !g->wg_per_task) return NPF_EINVAL;
        if (o.i2 == NPF_W_ROWMAJOR && o.i3 < (g->reserved[2] == 1 ? ((o.i0 + 31) >> 5) * 16 : o.i0)) return NPF_EINVAL;
        if (o.i2 == NPF_W_PT_ROWS && o.i3 * 32 < o.i1) return NPF_EINVAL;
        if (o.i2 == NPF_W_PT_COLS && o.i3 * 32 < o.i0) return NPF_EINVAL;
        if (o.i2 != NPF_W_ROWMAJOR && (((uintptr_t)o.p0) & 15)) return NPF_EINVAL;
        if ((o.flags & (NPF_F_ADD_PT | NPF_F_MASK_PT)) && (!o.p2 || (((uintptr_t)o.p2) & 15))) return NPF_EINVAL;
        if ((o.flags & NPF_F_ADD_PT) && (o.flags & NPF_F_MASK_PT)) return NPF_EINVAL;
        if (o.flags & NPF_F_MASK_BITS) {  // (bf16 programs only)
          if (g->reserved[2] != 1 || (o.flags & (NPF_F_ADD_PT | NPF_F_MASK_PT | NPF_F_ADD_RM)) || !o.p2 || (((uintptr_t)o.p2) & 3))
            return NPF_EINVAL;
        }
        if (o.flags & NPF_F_ADD_RM) {
          if ((o.flags & (NPF_F_ADD_PT | NPF_F_MASK_PT)) || !o.p2 || (((uintptr_t)o.p2) & 15) || (o.i1 & 31)) return NPF_EINVAL;
        }
        if (o.s1 != 0 && !g->wg_per_task) return NPF_EINVAL;
        if (o.flags & NPF_F_STORE_BITS) {  // (bf16 programs only)
          if (g->reserved[2] != 1 || !(o.flags & NPF_F_RELU) || o.i1 > 256 || !o.p2 || (((uintptr_t)o.p2) & 3) ||
              (o.flags & (NPF_F_ADD_PT | NPF_F_MASK_PT | NPF_F_ADD_RM | NPF_F_MASK_BITS)))
            return NPF_EINVAL;
        }
        if (o.flags & (NPF_F_STORE_IN | NPF_F_STORE_P16)) {  // (bf16 programs only: see DESIGN.md 9 for the fp32 attempt)
          if (g->reserved[2] != 1 || !(o.flags & NPF_F_STORE_IN) || !o.p3 || (((uintptr_t)o.p3) & 15)) return NPF_EINVAL;
        }
        break;
      case NPF_OP_LOAD_PT:
      case NPF_OP_STORE_PT:
      case NPF_OP_ADD_PT:
      case NPF_OP_MASK_POS:
      case NPF_OP_ROWDOT_PT:
      case NPF_OP_SOFTMAX_BWD:
        if (o.i0 <= 0 || o.i0 > NPF_MAX_FEATURES || (o.i0 & 31) || !o.p0 || (((uintptr_t)o.p0) & 15)) return NPF_EINVAL;
        break;
      case NPF_OP_STORE_MASK:
      case NPF_OP_MASK_BITS:
        if (g->reserved[2] != 1 || o.i0 <= 0 || o.i0 > 256 || (o.i0 & 31) || !o.p0 || (((uintptr_t)o.p0) & 3)) return NPF_EINVAL;
        break;
      case NPF_OP_ADD_TASKVEC:
      case NPF_OP_LOAD_RM:
        if (o.i0 <= 0 || o.i0 > NPF_MAX_FEATURES || (o.i0 & 31) || !o.p0 || (((uintptr_t)o.p0) & 15)) return NPF_EINVAL;
        break;
      case NPF_OP_LOAD_ROWS:
      case NPF_OP_STORE_ROWS:
        if (o.i0 <= 0 || o.i0 > 32 || !o.p0) return NPF_EINVAL;
        break;
      case NPF_OP_SOFTMAX:
        if (o.i0 <= 0 || o.i0 > NPF_MAX_FEATURES || o.i1 < 0 || o.i1 > 2) return NPF_EINVAL;
        if (o.i1 != 0 && (!o.p0 || (((uintptr_t)o.p0) & 7))) return NPF_EINVAL;
        break;
      case NPF_OP_STORE_TR:
        if (o.i0 <= 0 || o.i0 > NPF_MAX_FEATURES || !o.p0 || o.i1 < g->tiles_per_task * 32) return NPF_EINVAL;
        break;
      case NPF_OP_STORE_WB:
      case NPF_OP_STORE_TRB:
        if (g->reserved[2] != 1 || o.i0 <= 0 || o.i0 > 256 || !o.p0 || (((uintptr_t)o.p0) & 15) ||
            o.i1 < g->tiles_per_task * 32 || (o.i1 & 31))
          return NPF_EINVAL;
        break;
      case NPF_OP_LAYERNORM:
        if (o.i0 <= 0 || o.i0 > 256 || !o.p0 || !o.p1 || ((((uintptr_t)o.p0) | ((uintptr_t)o.p1)) & 15) || !(o.f0 > 0.f))
          return NPF_EINVAL;
        break;
      case NPF_OP_LAYERNORM_BWD:
        if (o.i0 <= 0 || o.i0 > 256 || !o.p0 || !o.p1 ||
            ((((uintptr_t)o.p0) | ((uintptr_t)o.p1) | ((uintptr_t)o.p2)) & 15) || !(o.f0 > 0.f))
          return NPF_EINVAL;
        break;
      case NPF_OP_RELU:
      case NPF_OP_SCALE:
      case NPF_OP_END:
        break;
      default:
        return NPF_EINVAL;
    }
  }
  return NPF_OK;
}

}  // namespace npf

extern "C" int npf_chain_run(const npf_program_t* prog, void* stream) {
  const int rc = npf::validate(prog);
  if (rc != NPF_OK) return rc;
  int n_ops = prog->n_ops;
  for (int i = 0; i < prog->n_ops; ++i)
    if (prog->ops[i].op == NPF_OP_END) {
      n_ops = i;
      break;
    }
  npf_program_t g = *prog;
  g.n_ops = n_ops;
  auto grid_for = [&](int tiles_per_wg) -> long {
    if (g.wg_per_task) return (long)g.n_tasks * ((g.tiles_per_task + tiles_per_wg - 1) / tiles_per_wg);
    return ((long)g.n_tasks * g.tiles_per_task + tiles_per_wg - 1) / tiles_per_wg;
  };
  bool wide = false;
  for (int i = 0; i < g.n_ops; ++i) {
    const npf_op_t& o = g.ops[i];
    const bool feat_op = o.op == NPF_OP_LOAD_PT || o.op == NPF_OP_STORE_PT || o.op == NPF_OP_ADD_PT ||
                         o.op == NPF_OP_MASK_POS || o.op == NPF_OP_ROWDOT_PT || o.op == NPF_OP_SOFTMAX_BWD ||
                         o.op == NPF_OP_ADD_TASKVEC || o.op == NPF_OP_SOFTMAX || o.op == NPF_OP_STORE_TR ||
                         o.op == NPF_OP_LOAD_RM;
    if (o.op == NPF_OP_LINEAR && (o.i0 > 256 || o.i1 > 256)) wide = true;
    if (feat_op && o.i0 > 256) wide = true;
  }
  // 64-point workgroups (two per CU) are the default; the 128-point paired variant is kept as a
  // tested option (reserved[1] == 2, NPF_FORCE_WG=2): it halves the slab traffic per point but
  // its lock-step phases overlap slightly worse on MI355X (23.5 vs 23.9 M points/s on config 2).
  const bool paired = !wide && g.reserved[1] == 2;
  const long grid = grid_for(paired ? 4 : 2);
  if (grid <= 0 || grid > 0x7fffffffL) return NPF_EINVAL;
  bool extra = false;
  int n256 = 0, n128 = 0;
  for (int i = 0; i < g.n_ops; ++i) {
    const npf_op_t& o = g.ops[i];
    extra |= o.op == NPF_OP_LAYERNORM || o.op == NPF_OP_LAYERNORM_BWD;
    if (o.op == NPF_OP_LINEAR) {
      n256 += o.i0 == 256 && o.i1 == 256;
      n128 += o.i0 == 128 && o.i1 == 128;
    }
  }
  if (extra && wide) return NPF_EINVAL;
  // reserved[2] == 1: every LINEAR of the program takes a bf16 weight image (npf_cast_bf16_weights)
  const bool bf16 = g.reserved[2] == 1;
  if (bf16) {
    if (extra || wide) return NPF_EINVAL;
    for (int i = 0; i < g.n_ops; ++i) {
      const npf_op_t& o = g.ops[i];
      if (o.op == NPF_OP_LINEAR && (o.i2 != NPF_W_ROWMAJOR || (((uintptr_t)o.p0) & 15) || o.i3 < ((o.i0 + 31) >> 5) * 16 ||
                                    (o.s0 & 3)))
        return NPF_EINVAL;
    }
    // (a 128-point paired bf16 instance <16, 8, ..., true> -- half the slab DMA per MFMA -- was measured: bare 8-layer
    // chain 2.27 vs 2.08 ms, config-3 step 15.7 vs 15.9 ms: no gain, not kept)
    hipLaunchKernelGGL((npf::chain_kernel<16, 4, false, 8, 8, true>), dim3((unsigned)grid_for(2)), dim3(256), 0,
                       (hipStream_t)stream, g);
    NPF_CHECK_LAUNCH();
    return NPF_OK;
  }
  const dim3 b4(256), b8(512);
  const hipStream_t st = (hipStream_t)stream;
  if (extra)
    hipLaunchKernelGGL((npf::chain_kernel<16, 4, true, 0, 0, false>), dim3((unsigned)grid_for(2)), b4, 0, st, g);
  else if (wide)
    hipLaunchKernelGGL((npf::chain_kernel<32, 4, false, 0, 0, false>), dim3((unsigned)grid), b4, 0, st, g);
  else if (paired)
    hipLaunchKernelGGL((npf::chain_kernel<16, 8, false, 16, 8, false>), dim3((unsigned)grid), b8, 0, st, g);
  else if (n128 > n256)
    hipLaunchKernelGGL((npf::chain_kernel<16, 4, false, 8, 4, false>), dim3((unsigned)grid), b4, 0, st, g);
  else
    hipLaunchKernelGGL((npf::chain_kernel<16, 4, false, 16, 8, false>), dim3((unsigned)grid), b4, 0, st, g);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}
