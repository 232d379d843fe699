// Register-resident "transposed chain" interpreter for gfx950 (MI355X).
//
// One wavefront owns 16 points (half a PT32 tile) and keeps their activations -- up to 256
// features -- in registers for the whole chain, feature-major:  every layer computes
//     Y^T[n][p] = sum_k W[n][k] X^T[k][p]
// with v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD), A operand = rows of a 32-row
// slab of W read from LDS, B operand = the previous layer's accumulator registers.  The
// 16x16 accumulator layout (column = lane & 15 = point, row = 4*(lane>>4) + reg = feature)
// is *already* the B-operand layout of the next layer once the weight slab is read with
// the same k permutation (one ds_read_b128 = the 4 k-steps of a 16-feature block), so
// activations never leave the register file between layers: no LDS round trip, no
// transposes, and HBM only sees what the program explicitly stores.
// Scaled-dot attention is two such layers whose weights are the task's keys / values with a
// feature-axis softmax in between (the feature axis is in-lane plus two cross-lane swaps).
//
// Weights stream global -> registers -> LDS in 32-row slabs, one slab ahead of the MFMAs,
// through a 2-slot ring; one barrier per slab (8192 MFMA cycles per SIMD at K = 256).
// Workgroup = 8 waves (two per SIMD, <= 256 registers each) = 4 tiles = 128 points.
//
// Replaces the torch op sequences of MLP.forward (npf/architectures/mlp.py:95-109),
// MergeFlatInputs.forward (encoders.py:175-183), BaseAttender.forward / DotAttender.score
// (attention.py:129-164,204-220), merge_r_z (neuralproc/base.py:554-575) and their autograd.
#include "npf_common.hpp"

namespace npf {

constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kTilesPerWG = 4;
constexpr int kMaxB16 = NPF_MAX_FEATURES / 16;   // 16-feature blocks a wave keeps in registers
constexpr int kMaxStride = NPF_MAX_FEATURES + 8;
constexpr int kBiasOff = 32 * kMaxStride;        // the slab's 32 biases live behind its rows
constexpr int kSlabFloats = kBiasOff + 32;
constexpr int kSlots = 2;
constexpr int kStage = 4;                        // float4 per thread per slab (32 x 256 / 512 / 4)

// Row stride (floats) of a slab with Kp columns: ds_read_b128 of 16 rows x 4 k-groups is
// conflict free when the stride is 8 mod 64 floats.
__device__ __forceinline__ int slab_stride(int Kp) { return Kp + ((Kp & 32) ? 40 : 8); }

struct Wave {
  int tid, p, g, half;     // p = lane & 15 (point / slab row), g = lane >> 4 (k group), half of the tile
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

// ---- slab staging ------------------------------------------------------------------
struct SlabCursor {
  int op;  // index of the LINEAR op the next slab belongs to (n_ops if none)
  int nb;  // slab index inside that op
};

__device__ __forceinline__ void cursor_seek(const npf_program_t& g, SlabCursor& c) {
  while (c.op < g.n_ops && g.ops[c.op].op != NPF_OP_LINEAR) ++c.op;
  c.nb = 0;
}

__device__ __forceinline__ void cursor_next(const npf_program_t& g, SlabCursor& c) {
  const int NB = (g.ops[c.op].i1 + 31) >> 5;
  if (++c.nb >= NB) {
    ++c.op;
    cursor_seek(g, c);
  }
}

// global -> registers.  stage[i] holds float4 number tid + 512*i of the slab image.
__device__ __forceinline__ void stage_load(const npf_program_t& g, const SlabCursor& c, const Wave& w,
                                           f32x4 (&stage)[kStage], float& bias_stage) {
  const npf_op_t& o = g.ops[c.op];
  const int K = o.i0, N = o.i1, mode = o.i2;
  const int Kp = ((K + 31) >> 5) * 32;
  const int total = 8 * Kp;  // float4 in the slab
  const float* W = (const float*)o.p0;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  if (mode == NPF_W_ROWMAJOR) {
    const int ldw = o.i3;
    W += (size_t)w.task * o.s0;
    const bool vec_ok = ((ldw & 3) == 0) && ((o.s0 & 3) == 0) && ((((uintptr_t)o.p0) & 15) == 0);
    const int kq = Kp >> 2;  // float4 per slab row
#pragma unroll
    for (int i = 0; i < kStage; ++i) {
      const int idx = w.tid + i * kThreads;
      f32x4 v = zero4;
      if (idx < total) {
        const int row = idx / kq, c4 = idx - row * kq;
        const int n = c.nb * 32 + row, k = c4 * 4;
        if (n < N) {
          const float* src = W + (size_t)n * ldw + k;
          if (vec_ok && k + 4 <= K) {
            v = *(const f32x4*)src;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (k + j < K) v[j] = src[j];
          }
        }
      }
      stage[i] = v;
    }
  } else if (mode == NPF_W_PT_ROWS) {
    // rows = points of the task's PT32 tensor (its tile nb), columns = its features
    const float* src = W + ((size_t)w.task * o.i3 + c.nb) * (size_t)(Kp * 32);
#pragma unroll
    for (int i = 0; i < kStage; ++i) {
      const int idx = w.tid + i * kThreads;  // = d4 * 32 + point
      stage[i] = (idx < total && c.nb * 32 + (idx & 31) < N) ? *(const f32x4*)(src + (size_t)idx * 4) : zero4;
    }
  } else {
    // NPF_W_PT_COLS: rows = features of the task's PT32 tensor (block nb), columns = its points
    const int Fq = ((N + 31) >> 5) * 8;  // float4 rows per tile of the source tensor
#pragma unroll
    for (int i = 0; i < kStage; ++i) {
      const int idx = w.tid + i * kThreads;
      const int n4 = idx / Kp, cpt = idx - n4 * Kp;
      const float* src = W + (((size_t)w.task * o.i3 + (cpt >> 5)) * Fq + (c.nb * 8 + n4)) * 128 + (cpt & 31) * 4;
      stage[i] = (idx < total && cpt < K) ? *(const f32x4*)src : zero4;
    }
  }
  bias_stage = 0.f;
  if (w.tid < 32 && o.p1 != nullptr) {
    const int n = c.nb * 32 + w.tid;
    if (n < N) bias_stage = ((const float*)o.p1)[(size_t)w.task * o.s1 + n];
  }
}

// registers -> LDS slot
__device__ __forceinline__ void stage_write(const npf_program_t& g, const SlabCursor& c, const Wave& w,
                                            const f32x4 (&stage)[kStage], float bias_stage, float* slot) {
  const npf_op_t& o = g.ops[c.op];
  const int mode = o.i2;
  const int Kp = ((o.i0 + 31) >> 5) * 32;
  const int total = 8 * Kp;
  const int stride = slab_stride(Kp);
  if (mode == NPF_W_ROWMAJOR) {
    const int kq = Kp >> 2;
#pragma unroll
    for (int i = 0; i < kStage; ++i) {
      const int idx = w.tid + i * kThreads;
      if (idx < total) {
        const int row = idx / kq, c4 = idx - row * kq;
        *(f32x4*)(slot + row * stride + c4 * 4) = stage[i];
      }
    }
  } else if (mode == NPF_W_PT_ROWS) {
#pragma unroll
    for (int i = 0; i < kStage; ++i) {
      const int idx = w.tid + i * kThreads;
      if (idx < total) *(f32x4*)(slot + (idx & 31) * stride + (idx >> 5) * 4) = stage[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < kStage; ++i) {
      const int idx = w.tid + i * kThreads;
      if (idx < total) {
        const int n4 = idx / Kp, cpt = idx - n4 * Kp;
#pragma unroll
        for (int j = 0; j < 4; ++j) slot[(n4 * 4 + j) * stride + cpt] = stage[i][j];
      }
    }
  }
  if (w.tid < 32) slot[kBiasOff + w.tid] = bias_stage;
}

// One 32-row slab = two 16-row output blocks (two independent accumulator chains):
//   acc_j[n][p] = bias[n] + sum_k W[16j + n][k] cur[k][p].
// MFMA step s of input block kb contracts features kb*16 + 4g + s (g = lane >> 4): exactly
// what accumulator register s of block kb holds on this lane, and what the A lane (n, g)
// reads as element s of the float4 at W[n][kb*16 + 4g].
__device__ __forceinline__ void slab_mfma(const float* slot, int KB16, const Wave& w, const f32x4 (&cur)[kMaxB16],
                                          f32x4& acc0, f32x4& acc1) {
  const int stride = slab_stride(KB16 * 16);
  const float* a0 = slot + w.p * stride + 4 * w.g;
  const float* a1 = a0 + 16 * stride;
  acc0 = *(const f32x4*)(slot + kBiasOff + 4 * w.g);
  acc1 = *(const f32x4*)(slot + kBiasOff + 16 + 4 * w.g);
#pragma unroll
  for (int kb = 0; kb < kMaxB16; ++kb) {
    if (kb < KB16) {
      const f32x4 x0 = *(const f32x4*)(a0 + kb * 16);
      const f32x4 x1 = *(const f32x4*)(a1 + kb * 16);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[s], cur[kb][s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[s], cur[kb][s], acc1, 0, 0, 0);
      }
    }
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

#define NPF_SET_BLOCK(arr, idx, val) \
  switch (idx) {                     \
    case 0: arr[0] = val; break;     \
    case 1: arr[1] = val; break;     \
    case 2: arr[2] = val; break;     \
    case 3: arr[3] = val; break;     \
    case 4: arr[4] = val; break;     \
    case 5: arr[5] = val; break;     \
    case 6: arr[6] = val; break;     \
    case 7: arr[7] = val; break;     \
    case 8: arr[8] = val; break;     \
    case 9: arr[9] = val; break;     \
    case 10: arr[10] = val; break;   \
    case 11: arr[11] = val; break;   \
    case 12: arr[12] = val; break;   \
    case 13: arr[13] = val; break;   \
    case 14: arr[14] = val; break;   \
    default: arr[15] = val; break;   \
  }

__global__ __launch_bounds__(kThreads, 2) void chain_kernel(const npf_program_t g) {
  __shared__ __attribute__((aligned(16))) float smem[kSlots * kSlabFloats];

  Wave w;
  w.tid = threadIdx.x;
  const int lane = w.tid & 63;
  w.p = lane & 15;
  w.g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(w.tid >> 6);
  w.half = wave & 1;
  const int wtile = wave >> 1;
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

  f32x4 cur[kMaxB16], out[kMaxB16];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < kMaxB16; ++b) {
    cur[b] = zero4;
    out[b] = zero4;
  }
  float acc_dot = 0.f;

  f32x4 stage[kStage];
  float bias_stage = 0.f;
  SlabCursor pf;
  pf.op = 0;
  cursor_seek(g, pf);
  int slot = 0;
  if (pf.op < g.n_ops) {  // prologue: slab 0 of the first LINEAR
    stage_load(g, pf, w, stage, bias_stage);
    stage_write(g, pf, w, stage, bias_stage, smem);
    cursor_next(g, pf);
  }
  __syncthreads();

  for (int ip = 0; ip < g.n_ops; ++ip) {
    const npf_op_t& o = g.ops[ip];
    const int opc = o.op;
    if (opc == NPF_OP_LINEAR) {
      const int KB16 = ((o.i0 + 31) >> 5) * 2, NB = (o.i1 + 31) >> 5;
      const bool relu = (o.flags & NPF_F_RELU) != 0;
      const bool add = (o.flags & NPF_F_ADD_PT) != 0;
      const float* addt = add ? pt_lane(o.p2, g, w, NB * 32, o.i4) : nullptr;
      for (int nb = 0; nb < NB; ++nb) {
        const bool has_next = pf.op < g.n_ops;
        if (has_next) stage_load(g, pf, w, stage, bias_stage);
        f32x4 ad0 = zero4, ad1 = zero4;
        if (add && w.valid) {
          ad0 = *(const f32x4*)(addt + (8 * nb + w.g) * 128);
          ad1 = *(const f32x4*)(addt + (8 * nb + 4 + w.g) * 128);
        }
        f32x4 acc0, acc1;
        slab_mfma(smem + slot * kSlabFloats, KB16, w, cur, acc0, acc1);
        if (has_next) {
          stage_write(g, pf, w, stage, bias_stage, smem + (slot ^ 1) * kSlabFloats);
          cursor_next(g, pf);
        }
        __syncthreads();
        slot ^= 1;
        acc0 += ad0;
        acc1 += ad1;
        if (relu) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc0[s] = fmaxf(acc0[s], 0.f);
            acc1[s] = fmaxf(acc1[s], 0.f);
          }
        }
        NPF_SET_BLOCK(out, 2 * nb, acc0);
        NPF_SET_BLOCK(out, 2 * nb + 1, acc1);
      }
#pragma unroll
      for (int b = 0; b < kMaxB16; ++b)
        if (b < 2 * NB) cur[b] = out[b];
    } else if (opc == NPF_OP_LOAD_PT || opc == NPF_OP_ADD_PT || opc == NPF_OP_MASK_POS || opc == NPF_OP_ROWDOT_PT ||
               opc == NPF_OP_SOFTMAX_BWD) {
      const int FB = o.i0 >> 4;
      const float* t = pt_lane(o.p0, g, w, o.i0, o.i4);
      float dot = 0.f;
#pragma unroll
      for (int b = 0; b < kMaxB16; ++b) {
        if (b < FB) {
          f32x4 v = zero4;
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
    } else if (opc == NPF_OP_STORE_PT) {
      const int FB = o.i0 >> 4;
      float* t = (float*)pt_lane(o.p0, g, w, o.i0, o.i4);
      if (w.valid) {
#pragma unroll
        for (int b = 0; b < kMaxB16; ++b)
          if (b < FB) *(f32x4*)(t + (4 * b + w.g) * 128) = cur[b];
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
      const int nvalid = o.i0;
      const int FB = ((nvalid + 31) >> 5) * 2;
      const float scale = o.f0;
      float m = -INFINITY;
#pragma unroll
      for (int b = 0; b < kMaxB16; ++b)
        if (b < FB)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            if (16 * b + 4 * w.g + s < nvalid) m = fmaxf(m, cur[b][s]);
      m = xg_max(m);
      float sum = 0.f;
#pragma unroll
      for (int b = 0; b < kMaxB16; ++b)
        if (b < FB)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bool ok = 16 * b + 4 * w.g + s < nvalid;
            const float e = ok ? expf((cur[b][s] - m) * scale) : 0.f;
            cur[b][s] = e;
            sum += e;
          }
      sum = xg_sum(sum);
      const float inv = 1.f / sum;
#pragma unroll
      for (int b = 0; b < kMaxB16; ++b)
        if (b < FB) cur[b] *= inv;
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
}

static int validate(const npf_program_t* g) {
  if (!g || g->n_ops < 0 || g->n_ops > NPF_MAX_OPS) return NPF_EINVAL;
  if (g->n_tasks <= 0 || g->pts_per_task <= 0) return NPF_EINVAL;
  if (g->tiles_per_task != (g->pts_per_task + 31) / 32) return NPF_EINVAL;
  for (int i = 0; i < g->n_ops; ++i) {
    const npf_op_t& o = g->ops[i];
    switch (o.op) {
      case NPF_OP_LINEAR:
        if (o.i0 <= 0 || o.i0 > NPF_MAX_FEATURES || o.i1 <= 0 || o.i1 > NPF_MAX_FEATURES || !o.p0) return NPF_EINVAL;
        if (o.i2 < 0 || o.i2 > 2) return NPF_EINVAL;
        if (o.i2 != NPF_W_ROWMAJOR && !g->wg_per_task) return NPF_EINVAL;  // per-task weights
        if (o.i2 == NPF_W_ROWMAJOR && o.s0 != 0 && !g->wg_per_task) return NPF_EINVAL;
        if (o.i2 == NPF_W_ROWMAJOR && o.i3 < o.i0) return NPF_EINVAL;
        if (o.i2 == NPF_W_PT_ROWS && o.i3 * 32 < o.i1) return NPF_EINVAL;
        if (o.i2 == NPF_W_PT_COLS && o.i3 * 32 < o.i0) return NPF_EINVAL;
        if (o.i2 != NPF_W_ROWMAJOR && (((uintptr_t)o.p0) & 15)) return NPF_EINVAL;
        if ((o.flags & NPF_F_ADD_PT) && (!o.p2 || (((uintptr_t)o.p2) & 15))) return NPF_EINVAL;
        if (o.s1 != 0 && !g->wg_per_task) return NPF_EINVAL;
        break;
      case NPF_OP_LOAD_PT:
      case NPF_OP_STORE_PT:
      case NPF_OP_ADD_PT:
      case NPF_OP_MASK_POS:
      case NPF_OP_ROWDOT_PT:
      case NPF_OP_SOFTMAX_BWD:
        if (o.i0 <= 0 || o.i0 > NPF_MAX_FEATURES || (o.i0 & 31) || !o.p0 || (((uintptr_t)o.p0) & 15)) return NPF_EINVAL;
        break;
      case NPF_OP_ADD_TASKVEC:
        if (o.i0 <= 0 || o.i0 > NPF_MAX_FEATURES || (o.i0 & 31) || !o.p0 || (((uintptr_t)o.p0) & 15)) return NPF_EINVAL;
        break;
      case NPF_OP_LOAD_ROWS:
      case NPF_OP_STORE_ROWS:
        if (o.i0 <= 0 || o.i0 > 32 || !o.p0) return NPF_EINVAL;
        break;
      case NPF_OP_SOFTMAX:
        if (o.i0 <= 0 || o.i0 > NPF_MAX_FEATURES) return NPF_EINVAL;
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
  long grid;
  if (g.wg_per_task)
    grid = (long)g.n_tasks * ((g.tiles_per_task + npf::kTilesPerWG - 1) / npf::kTilesPerWG);
  else
    grid = ((long)g.n_tasks * g.tiles_per_task + npf::kTilesPerWG - 1) / npf::kTilesPerWG;
  if (grid <= 0 || grid > 0x7fffffffL) return NPF_EINVAL;
  hipLaunchKernelGGL(npf::chain_kernel, dim3((unsigned)grid), dim3(npf::kThreads), 0, (hipStream_t)stream, g);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}
