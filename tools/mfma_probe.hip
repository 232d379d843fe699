// Development probe: what MFMA issue rate do the building blocks of chain_kernel reach?
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void dma16(const float* src, float* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// 4-wave workgroups, 2 per CU, 32-row x 256 slabs through a 2-slot LDS ring, like chain_kernel.
// DMODE 0: no DMA; 1: 8 pieces/wave before the MFMA loop; 2: one piece every other MFMA block;
// 3: pieces before the loop but only 2 per wave (1/4 of the traffic)
template <int DMODE>
__global__ __launch_bounds__(256, 2) void probe_dma(const float* W, float* out, int slabs, int KB) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 8192 + 64];
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * 8192; i += 256) smem[i] = 1e-3f * (i & 7);
  __syncthreads();
  f32x4 cur[16];
  for (int b = 0; b < 16; ++b) cur[b] = f32x4{1.f + b, 0.5f, 0.25f, lane * 1e-3f};
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  int slot = 0;
  for (int sl = 0; sl < slabs; ++sl) {
    const float* a0 = smem + slot * 8192 + p * 256;
    const float* a1 = a0 + 16 * 256;
    float* nxt = smem + (slot ^ 1) * 8192;
    const float* src = W + (size_t)(sl & 7) * 8192;  // 8 slabs = one 256x256 layer
    if (DMODE == 1 || DMODE == 3 || DMODE == 4) {
      const int np = DMODE == 3 ? 2 : 8;
      for (int i = 0; i < np; ++i) {
        const int q = wave + 4 * i, row = q;
        dma16(src + row * 256 + ((lane ^ (row & 15)) << 2), nxt + q * 256);
      }
    }
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      if (DMODE >= 4 && kb >= KB) continue;  // runtime guard per block, like chain_kernel
      const int off = (((4 * kb + g) ^ p) << 2);
      const f32x4 x0 = *(const f32x4*)(a0 + off);
      const f32x4 x1 = *(const f32x4*)(a1 + off);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[s], cur[kb][s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[s], cur[kb][s], acc1, 0, 0, 0);
      }
      if (DMODE == 2 && (kb & 1)) {
        const int q = wave + 4 * (kb >> 1), row = q;
        dma16(src + row * 256 + ((lane ^ (row & 15)) << 2), nxt + q * 256);
      }
    }
    __syncthreads();
    slot ^= 1;
  }
  out[blockIdx.x * 256 + tid] = acc0[0] + acc1[1] + acc0[2] + acc1[3];
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(float* out, int slabs) {
  __shared__ __attribute__((aligned(16))) float smem[16384];
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  for (int i = tid; i < 16384; i += 512) smem[i] = 1e-3f * (i & 7);
  __syncthreads();
  f32x4 cur[16];
  for (int b = 0; b < 16; ++b) cur[b] = f32x4{1.f + b, 0.5f, 0.25f, lane * 1e-3f};
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  const float* a0 = smem + p * 256;
  const float* a1 = a0 + 16 * 256;
  for (int sl = 0; sl < slabs; ++sl) {
    if (MODE == 0) {  // registers only
#pragma unroll
      for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[kb][s], cur[(kb + 1) & 15][s], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[kb][3 - s], cur[(kb + 1) & 15][s], acc1, 0, 0, 0);
        }
    } else {  // LDS operands, swizzled like chain_kernel
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const int off = (((4 * kb + g) ^ p) << 2);
        const f32x4 x0 = *(const f32x4*)(a0 + off);
        const f32x4 x1 = *(const f32x4*)(a1 + off);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[s], cur[kb][s], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[s], cur[kb][s], acc1, 0, 0, 0);
        }
      }
    }
    if (MODE == 2) __syncthreads();
    if (MODE == 3) {  // 4 accumulators: more independent chains
    }
  }
  out[blockIdx.x * 512 + tid] = acc0[0] + acc1[1] + acc0[2] + acc1[3];
}

// 32x32x2 variant, 256 threads (1 wave / SIMD), 4 accumulators, registers only
__global__ __launch_bounds__(256, 1) void probe32(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = lane * 1e-3f, b = 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][lane & 15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
double time_ms(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float* out;
  hipMalloc(&out, 2048 * 512 * 4);
  const int slabs = 64, grid = 2048;
  const double flop = (double)grid * 8 /*waves*/ * slabs * 128 * 2048.0;
  const char* names[] = {"16x16x4 regs only", "16x16x4 + ds_read per 8 mfma", "16x16x4 + ds_read + barrier/slab"};
  double t;
  t = time_ms([&] { hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 0, 0, out, slabs); });
  printf("%-40s %7.3f ms %7.1f TF/s\n", names[0], t, flop / t * 1e-9);
  t = time_ms([&] { hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 0, 0, out, slabs); });
  printf("%-40s %7.3f ms %7.1f TF/s\n", names[1], t, flop / t * 1e-9);
  t = time_ms([&] { hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(512), 0, 0, out, slabs); });
  printf("%-40s %7.3f ms %7.1f TF/s\n", names[2], t, flop / t * 1e-9);
  {
    float* W;
    hipMalloc(&W, 256 * 256 * 4);
    hipMemset(W, 0, 256 * 256 * 4);
    const int g4 = 4096, sl4 = 64;
    const double fl4 = (double)g4 * 4 * sl4 * 128 * 2048.0;
    const char* dn[] = {"4w WG: no DMA", "4w WG: 8 DMA pieces/wave before MFMAs", "4w WG: DMA pieces inside MFMA loop",
                        "4w WG: 2 DMA pieces/wave before MFMAs"};
    t = time_ms([&] { hipLaunchKernelGGL(probe_dma<0>, dim3(g4), dim3(256), 0, 0, W, out, sl4, 16); });
    printf("%-40s %7.3f ms %7.1f TF/s\n", dn[0], t, fl4 / t * 1e-9);
    t = time_ms([&] { hipLaunchKernelGGL(probe_dma<1>, dim3(g4), dim3(256), 0, 0, W, out, sl4, 16); });
    printf("%-40s %7.3f ms %7.1f TF/s\n", dn[1], t, fl4 / t * 1e-9);
    t = time_ms([&] { hipLaunchKernelGGL(probe_dma<2>, dim3(g4), dim3(256), 0, 0, W, out, sl4, 16); });
    printf("%-40s %7.3f ms %7.1f TF/s\n", dn[2], t, fl4 / t * 1e-9);
    t = time_ms([&] { hipLaunchKernelGGL(probe_dma<3>, dim3(g4), dim3(256), 0, 0, W, out, sl4, 16); });
    printf("%-40s %7.3f ms %7.1f TF/s\n", dn[3], t, fl4 / t * 1e-9);
  }
  {
    float* W;
    hipMalloc(&W, 256 * 256 * 4);
    hipMemset(W, 0, 256 * 256 * 4);
    const int g4 = 4096, sl4 = 64;
    const double fl4 = (double)g4 * 4 * sl4 * 128 * 2048.0;
    t = time_ms([&] { hipLaunchKernelGGL(probe_dma<4>, dim3(g4), dim3(256), 0, 0, W, out, sl4, 16); });
    printf("%-40s %7.3f ms %7.1f TF/s\n", "4w WG: 8 DMA + runtime kb guards", t, fl4 / t * 1e-9);
  }
  const int iters = 128;
  const double flop32 = (double)grid * 4 * iters * 64 * 4096.0;
  t = time_ms([&] { hipLaunchKernelGGL(probe32, dim3(grid), dim3(256), 0, 0, out, iters); });
  printf("%-40s %7.3f ms %7.1f TF/s\n", "32x32x2 regs only, 1 wave/SIMD", t, flop32 / t * 1e-9);
  return 0;
}
