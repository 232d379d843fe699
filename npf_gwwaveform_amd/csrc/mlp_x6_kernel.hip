// A stack of 256 -> 256 Linear layers on register-resident activations with the fp32 products on the bf16 matrix pipe
// (npf_mlp_x6_run): the hidden layers of the reference's flat MLPs (npf/architectures/mlp.py:95-109: Linear, ReLU, Linear, ...)
// forward, and their dgrad (mask by the saved activation, store dZ for the weight gradient, multiply by W^T) backward.
//
// Arithmetic = wgrad_x6_kernel's (csrc/wgrad_kernel.hip): every fp32 operand is split EXACTLY into three bf16 terms
// (x = x0 + x1 + x2, x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1); the remainder is below 2^-27 |x|) and the six
// cross products with i + j <= 2 go through v_mfma_f32_16x16x32_bf16 with fp32 accumulation: an fp32 result (the dropped
// products are below 2^-26 of a product), at 6/16 of the v_mfma_f32_16x16x4_f32 time.  The weights arrive already split
// (three k-permuted bf16 images per layer, npf_prepare_weights); the layer input is split in registers once per layer.
//
// Layout = the chain kernel's: a wave owns 16 points (half a PT32 tile), block b / element e of lane (p, g) = feature
// 16 b + 4 g + e of point p; a workgroup = 4 waves = 2 tiles, two workgroups per CU.  Weights stream L2 -> LDS by LDS-DMA in
// slabs of 16 output rows x 3 terms (24 KiB) through a three-slot ring: slab S + 2 is in flight while slab S multiplies, one
// counted s_waitcnt vmcnt + one barrier per slab.  Measured standalone (a probe of round 2, see tools/experiments/README.md): 8 layers over
// 1 M points 4.75 ms = 231 TF/s fp32-equivalent, against 131 TF/s of the fp32 chain kernel on the same stack.
//
// Tried, not kept: the compiler waits for vmcnt(0) -- draining the slab ring -- once per layer for the bias load and once at
// the first use of an activation block that may come from a load (the input, the addend).  Fetching bias and sign bits one
// layer ahead with hand-written loads and consuming the tracked loads in front of the ring removed every in-loop vmcnt(0)
// (asm checked) and changed nothing measurable (config 2: 2.573 -> 2.565 ms per step in this kernel, three A/B pairs on
// one box): two drains in sixteen slabs are hidden by the second workgroup of the CU.
#include "npf_common.hpp"

namespace npf {

typedef __bf16 x6_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned x6_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kXF = 256;                          // layer width
constexpr int kXRows = 16;                        // output rows per slab
constexpr int kXSlabs = kXF / kXRows;             // 16 slabs per layer
constexpr int kXTermBytes = kXRows * kXF * 2;     // one term of a slab: 8 KiB
constexpr int kXSlabBytes = 3 * kXTermBytes;      // 24 KiB
constexpr int kXSlots = 3;
constexpr int kXBiasBytes = 2 * kXF * 4;          // the running and the next layer's bias

struct X6Args {
  npf_x6_layer_t layer[NPF_X6_MAX_LAYERS];
  const float* x;
  const float* in_rows;  // (x == nullptr) the stack's input = in_w^T in_rows: [points][4] rows through a [4][256] matrix
  const float* in_w;
  float* y;
  const float* out_w;  // (out_rows != nullptr) a 256 -> 4 layer behind the stack: out_rows [points][4] = out_w [4][256] cur + out_b
  const float* out_b;
  float* out_rows;
  int32_t n_layers;
  int32_t total_tiles;
};

__device__ __forceinline__ void x6_dma16(const void* src, void* lds_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_uniform, 16, 0, 0);
}

__device__ __forceinline__ unsigned x6m_cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// the 8 values of a lane's two adjacent blocks (the B operand of one 32-feature k-step) as three packed bf16 terms.
// Edge values (tests/test_hip_kernels.py::test_mlp_x6_split_edge_values): exact for every finite |x| <= 3.3895e38 (the largest
// bf16), subnormals included; a non-finite x gives non-finite terms (inf - inf), i.e. a non-finite output like fp32 arithmetic;
// a finite |x| above the largest bf16 (the last 0.4 % of the fp32 range) rounds to infinity in the first term: non-finite too.
__device__ __forceinline__ void x6m_split(const f32x4& lo, const f32x4& hi, x6_u32x4& t0, x6_u32x4& t1, x6_u32x4& t2) {
  const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    const unsigned h = x6m_cvt_pk(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = x6m_cvt_pk(ra, rb);
    const float la = ra - __builtin_bit_cast(float, m << 16), lb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    t0[p] = h;
    t1[p] = m;
    t2[p] = x6m_cvt_pk(la, lb);
  }
}

__global__ __launch_bounds__(256, 2) void mlp_x6_kernel(const X6Args a) {
  __shared__ __attribute__((aligned(16))) char smem[kXSlots * kXSlabBytes + kXBiasBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  const long tile = (long)blockIdx.x * 2 + (wave >> 1);
  const bool valid = tile < a.total_tiles;  // (wave-uniform; a wave without a tile still streams slabs and meets barriers)
  // this lane's float4 column in its tile of a PT32 tensor with 256 features: block b at + (4 b + g) * 128 floats
  const size_t lane_off = (size_t)(valid ? tile : 0) * (kXF * 32) + (size_t)(16 * (wave & 1) + p) * 4 + (size_t)g * 128;

  // sign-bit tensors: one 64-bit word per lane and half tile, [tile][half][64 lanes]
  const size_t bits_off = ((size_t)(valid ? tile : 0) * 2 + (wave & 1)) * 64 + lane;
  f32x4 cur[16];
  if (a.in_rows == nullptr) {
    const float* x = a.x + lane_off;
#pragma unroll
    for (int b = 0; b < 16; ++b) cur[b] = *(const f32x4*)(x + b * 512);
  } else {
    // the dgrad of a 256 -> 4 layer in front of the stack (the decoder's output layer, mlp.py:109): this lane's point has four
    // dOut values, its 64 features of the gradient = those through W_out [4][256].  W_out waits in the ring's third slot,
    // which no slab enters before every wave has passed the first stage's barrier.
    f32x4* wl = (f32x4*)(smem + 2 * kXSlabBytes);
    wl[tid] = ((const f32x4*)a.in_w)[tid];
    __syncthreads();
    const f32x4 dz = ((const f32x4*)a.in_rows)[(size_t)(valid ? tile : 0) * 32 + 16 * (wave & 1) + p];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      f32x4 v = dz[0] * wl[4 * b + g];
#pragma unroll
      for (int n = 1; n < 4; ++n) {
        const f32x4 w = wl[n * 64 + 4 * b + g];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(dz[n], w[e], v[e]);
      }
      cur[b] = v;
    }
  }

  // DMA of slab s of a layer: 24 pieces of 1 KiB (term q / 8, rows 2 (q % 8), + 1), six per wave; the swizzle (chunk c of
  // row r at position c ^ (r & 15)) is applied to the source address: uniform base per piece + one of two lane offsets
  unsigned dma_lane[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int r0 = 2 * ((wave + 4 * n) & 7), row = r0 + (lane >> 5), pos = lane & 31;
    dma_lane[n] = (unsigned)((lane >> 5) * 512 + ((pos ^ (row & 15)) << 4));
  }
  const int n_slabs = a.n_layers * kXSlabs;
  auto dma_slab = [&](int S, char* slot) {
    const char* base = (const char*)a.layer[S / kXSlabs].w_img + (size_t)(S % kXSlabs) * kXRows * kXF * 2;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const int q = wave + 4 * n, term = q >> 3, r0 = 2 * (q & 7);
      x6_dma16(base + (size_t)term * (kXF * kXF * 2) + r0 * 512 + dma_lane[n & 1], slot + term * kXTermBytes + r0 * 512);
    }
  };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lane_row = (unsigned)(p * 512);
  float* bias_lds = (float*)(smem + kXSlots * kXSlabBytes);

  dma_slab(0, smem);
  if (n_slabs > 1) dma_slab(1, smem + kXSlabBytes);
  int slot = 0;
  for (int l = 0; l < a.n_layers; ++l) {
    const npf_x6_layer_t& ly = a.layer[l];
    // dgrad: the gradient stops where the forward activation was not positive; dZ goes out for the weight gradient
    if (ly.mask != nullptr && valid) {
      const float* m = ly.mask + lane_off;
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const f32x4 v = *(const f32x4*)(m + b * 512);
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[b][e] = v[e] > 0.f ? cur[b][e] : 0.f;
      }
    }
    // the same from the 64 sign bits per lane the forward pass left (bit 4 b + e = block b, element e): 8 bytes per lane
    // and layer instead of 256
    if (ly.mask_bits != nullptr && valid) {
      const unsigned long long w = ly.mask_bits[bits_off];
#pragma unroll
      for (int b = 0; b < 16; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) cur[b][e] = ((w >> (4 * b + e)) & 1ull) ? cur[b][e] : 0.f;
    }
    if (ly.store_in != nullptr && valid) {
      float* d = ly.store_in + lane_off;
#pragma unroll
      for (int b = 0; b < 16; ++b) __builtin_nontemporal_store(cur[b], (f32x4*)(d + b * 512));
    }
    // the layer's bias into LDS (read back per slab; visible behind the barrier of the layer's first slab)
    bias_lds[(l & 1) * kXF + tid] = ly.bias != nullptr ? ly.bias[tid] : 0.f;
    // the layer's input as three packed bf16 terms (the B operands), once per layer
    x6_u32x4 tb[3][8];
#pragma unroll
    for (int st = 0; st < 8; ++st) x6m_split(cur[2 * st], cur[2 * st + 1], tb[0][st], tb[1][st], tb[2][st]);
    // an addend (the decoder's relu(x1 + resizer(R)), encoders.py:178-179) waits in the registers of the blocks it will
    // be added to: the layer's input is dead once it is split, and block s is only rewritten at slab s
    const bool has_add = ly.addend != nullptr;
    if (has_add) {
      const float* ad = ly.addend + lane_off;
#pragma unroll
      for (int b = 0; b < 16; ++b) cur[b] = valid ? *(const f32x4*)(ad + b * 512) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float* out = (ly.store_out != nullptr && valid) ? ly.store_out + lane_off : nullptr;
    const unsigned bias_l = lds0 + kXSlots * kXSlabBytes + (l & 1) * (kXF * 4) + g * 16;
    const bool relu = ly.relu != 0;
    unsigned long long pos_bits = 0ull;  // where this layer's output is positive (store_bits)
#pragma unroll
    for (int s = 0; s < kXSlabs; ++s) {
      const int S = l * kXSlabs + s;
      // slab S has landed for everyone, everyone is done with slab S - 1 (whose slot slab S + 2 goes into).  Counted wait:
      // the six pieces of slab S + 1 may stay in flight (vector-memory operations retire in order; the stores and mask
      // loads of this wave issued since are older than them or make the wait stricter, never laxer)
      if (S + 1 < n_slabs) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (S + 2 < n_slabs) dma_slab(S + 2, smem + ((slot + 2) % kXSlots) * kXSlabBytes);
      const unsigned sl = lds0 + slot * kXSlabBytes + lane_row;
      f32x4 acc, sm = {0.f, 0.f, 0.f, 0.f};
      // fragments of k-step st + 1 are read while step st multiplies
      x6_u32x4 fr[2][3];
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %5 offset:8192\n\tds_read_b128 %3, %5 offset:16384"
                   : "=&v"(acc), "=&v"(fr[0][0]), "=&v"(fr[0][1]), "=&v"(fr[0][2])
                   : "v"(bias_l + 64 * s), "v"(sl + (((0 + g) ^ p) << 4)));
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int c = st & 1, n = c ^ 1;
        if (st + 1 < 8) {
          asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:8192\n\tds_read_b128 %2, %7 offset:16384"
                       : "=&v"(fr[n][0]), "=&v"(fr[n][1]), "=&v"(fr[n][2]), "+v"(fr[c][0]), "+v"(fr[c][1]), "+v"(fr[c][2]), "+v"(acc)
                       : "v"(sl + (((4 * (st + 1) + g) ^ p) << 4)));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[c][0]), "+v"(fr[c][1]), "+v"(fr[c][2]));
        }
#define X6MM(A, B, C) \
  C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(x6_bf16x8, fr[c][A]), __builtin_bit_cast(x6_bf16x8, tb[B][st]), C, 0, 0, 0)
        X6MM(2, 0, sm);
        X6MM(0, 0, acc);
        X6MM(0, 2, sm);
        X6MM(1, 0, acc);
        X6MM(1, 1, sm);
        X6MM(0, 1, acc);
#undef X6MM
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
      slot = (slot + 1) % kXSlots;
    }
    if (ly.store_bits != nullptr && valid) ly.store_bits[bits_off] = pos_bits;
  }
  if (valid && a.y != nullptr) {
    float* y = a.y + lane_off;
#pragma unroll
    for (int b = 0; b < 16; ++b) *(f32x4*)(y + b * 512) = cur[b];
  }
  if (a.out_rows != nullptr) {
    // the decoder's output layer (mlp.py:109) on the registers the stack leaves: a lane holds 64 of its point's 256 features,
    // four fp32 dot products over them, summed over the point's four lanes.  W_out goes where the ring was.
    __syncthreads();  // (every wave is done with the last slabs)
    f32x4* wl = (f32x4*)smem;
    wl[tid] = ((const f32x4*)a.out_w)[tid];
    __syncthreads();
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 16; ++b)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const f32x4 w = wl[n * 64 + 4 * b + g];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[n] = fmaf(w[e], cur[b][e], r[n]);
      }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      r[n] += __shfl_xor(r[n], 16);
      r[n] += __shfl_xor(r[n], 32);
      if (a.out_b != nullptr) r[n] += a.out_b[n];
    }
    if (valid && g == 0) ((f32x4*)a.out_rows)[(size_t)tile * 32 + 16 * (wave & 1) + p] = r;
  }
}

}  // namespace npf

extern "C" int npf_mlp_x6_run_rows(const npf_x6_layer_t* layers, int32_t n_layers, const float* x, const float* in_rows,
                                   const float* in_w, float* y, const float* out_w, const float* out_b, float* out_rows,
                                   int32_t n_tasks, int32_t tiles_per_task, void* stream) {
  if (!layers || n_layers <= 0 || n_layers > NPF_X6_MAX_LAYERS || n_tasks <= 0 || tiles_per_task <= 0) return NPF_EINVAL;
  if ((x == nullptr) == (in_rows == nullptr) || (in_rows != nullptr) != (in_w != nullptr)) return NPF_EINVAL;
  if ((out_rows != nullptr) != (out_w != nullptr) || (out_b != nullptr && out_rows == nullptr)) return NPF_EINVAL;
  if (y == nullptr && out_rows == nullptr) return NPF_EINVAL;
  if ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)in_rows) | ((uintptr_t)in_w) | ((uintptr_t)out_w) | ((uintptr_t)out_rows)) & 15)
    return NPF_EINVAL;
  if (((uintptr_t)out_b) & 3) return NPF_EINVAL;
  npf::X6Args a;
  for (int l = 0; l < n_layers; ++l) {
    const npf_x6_layer_t& ly = layers[l];
    if (!ly.w_img || (((uintptr_t)ly.w_img) & 15)) return NPF_EINVAL;
    if ((((uintptr_t)ly.mask) | ((uintptr_t)ly.store_in) | ((uintptr_t)ly.store_out) | ((uintptr_t)ly.addend)) & 15) return NPF_EINVAL;
    if ((((uintptr_t)ly.mask_bits) | ((uintptr_t)ly.store_bits)) & 7) return NPF_EINVAL;
    if (ly.bias && (((uintptr_t)ly.bias) & 3)) return NPF_EINVAL;
    a.layer[l] = ly;
  }
  for (int l = n_layers; l < NPF_X6_MAX_LAYERS; ++l) a.layer[l] = layers[0];
  a.x = x;
  a.in_rows = in_rows;
  a.in_w = in_w;
  a.y = y;
  a.out_w = out_w;
  a.out_b = out_b;
  a.out_rows = out_rows;
  a.n_layers = n_layers;
  a.total_tiles = n_tasks * tiles_per_task;
  const int n_wg = (a.total_tiles + 1) / 2;
  hipLaunchKernelGGL(npf::mlp_x6_kernel, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, a);
  NPF_CHECK_LAUNCH();
  return NPF_OK;
}

extern "C" int npf_mlp_x6_run(const npf_x6_layer_t* layers, int32_t n_layers, const float* x, float* y, int32_t n_tasks,
                              int32_t tiles_per_task, void* stream) {
  if (!x || !y) return NPF_EINVAL;
  return npf_mlp_x6_run_rows(layers, n_layers, x, nullptr, nullptr, y, nullptr, nullptr, nullptr, n_tasks, tiles_per_task, stream);
}
