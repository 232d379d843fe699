// Development probe: how fast does a wave get through a block of instructions of one kind while
// the other wave of its SIMD runs a dense fp32 MFMA loop?  One 8-wave workgroup (wave i and
// wave i + 4 share SIMD i); waves 4-7 run MFMAs (or idle), wave 0 times the block with
// s_memtime.  Build: hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang diagnostic ignored "-Wunused-result"
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// KIND: 0 salu chain, 1 v_mov independent, 2 v_add chain, 3 lds-dma x16 (saddr), 4 global_load x16,
//       5 ds_read_b128 x64, 6 v_readlane x64, 7 s_nop x64
// MODE: 0 sibling idle, 1 sibling MFMA 2 chains, 2 sibling MFMA 4 chains, 3 sibling MFMA 2 chains at prio 0 / us at prio 3
template <int KIND>
__global__ __launch_bounds__(512, 1) void probe(const float* src, float* out, unsigned long long* cyc, int mode, int reps) {
  __shared__ __attribute__((aligned(16))) float smem[16384];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 16384; i += 512) smem[i] = 1e-3f * (i & 7);
  __syncthreads();
  if (wave >= 4) {
    if (mode == 0) return;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = lane * 1e-3f, y = 1.f + lane;
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem + lane * 16u;
    f32x4 fr = {0, 0, 0, 0};
    unsigned long long m0 = now();
#define MM2 a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
#define MM8 MM2 MM2 MM2 MM2
    for (int i = 0; i < reps; ++i) {
      if (mode == 2) {  // 4 chains
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          MM2
          a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        }
      } else if (mode == 4) {  // s_nop 0 every 8 MFMAs
#pragma unroll
        for (int u = 0; u < 8; ++u) { MM8 asm volatile("s_nop 0" : "+v"(a0), "+v"(a1)); }
      } else if (mode == 5) {  // s_nop 7 every 8 MFMAs
#pragma unroll
        for (int u = 0; u < 8; ++u) { MM8 asm volatile("s_nop 7" : "+v"(a0), "+v"(a1)); }
      } else if (mode == 6) {  // single dependent chain
#pragma unroll
        for (int u = 0; u < 64; ++u) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      } else if (mode == 7) {  // ds_read + waitcnt every 8 MFMAs (like the kernel's fragment pipeline)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2" : "=&v"(fr), "+v"(a0) : "v"(lds));
          MM8
        }
      } else if (mode == 8) {  // s_sleep 0 every 8 MFMAs
#pragma unroll
        for (int u = 0; u < 8; ++u) { MM8 asm volatile("s_sleep 0" : "+v"(a0), "+v"(a1)); }
      } else if (mode == 9) {  // s_setprio 0 / own priority drop every 8 MFMAs
#pragma unroll
        for (int u = 0; u < 8; ++u) { MM8 asm volatile("s_setprio 0" : "+v"(a0), "+v"(a1)); }
      } else if (mode == 10) {  // a VALU instruction of its own every 8 MFMAs
#pragma unroll
        for (int u = 0; u < 8; ++u) { MM8 asm volatile("v_mov_b32 %0, %0" : "+v"(x)); }
      } else {  // 2 chains
#pragma unroll
        for (int u = 0; u < 8; ++u) { MM8 }
      }
    }
    unsigned long long m1 = now();
    if (tid == 256) cyc[2] = m1 - m0;
    a0 += fr;
    out[tid] = a0[0] + a1[1] + a2[2] + a3[3];
    return;
  }
  if (wave != 0) return;
  if (mode == 3) __builtin_amdgcn_s_setprio(3);
  __builtin_amdgcn_s_sleep(100);  // let the sibling get going
  float v0 = lane, v1 = 1.f, v2 = 2.f, v3 = 3.f;
  unsigned s0 = 1;
  f32x4 r0, r1, r2, r3;
  const unsigned lo = lane * 16u;
  unsigned long long t0 = now();
  if (KIND == 0) {
    asm volatile(".rept 64\n\ts_add_u32 %0, %0, 3\n\t.endr" : "+s"(s0));
  } else if (KIND == 1) {
    asm volatile(".rept 16\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4\n\t.endr"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3)
                 : "v"(lo));
  } else if (KIND == 2) {
    asm volatile(".rept 64\n\tv_add_f32 %0, %0, %1\n\t.endr" : "+v"(v0) : "v"(v1));
  } else if (KIND == 3) {
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    asm volatile("s_mov_b32 m0, %2\n\t.rept 16\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 1024\n\t.endr" ::"v"(lo), "s"(src), "s"(lds) : "memory");
  } else if (KIND == 4) {
    asm volatile(".rept 4\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\tglobal_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072\n\t.endr"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                 : "v"(lo), "s"(src)
                 : "memory");
  } else if (KIND == 5) {
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem + lo;
    asm volatile(".rept 16\n\tds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t.endr"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                 : "v"(lds)
                 : "memory");
  } else if (KIND == 6) {
    asm volatile(".rept 64\n\tv_readlane_b32 %0, %1, 3\n\t.endr" : "=s"(s0) : "v"(v1));
  } else {
    asm volatile(".rept 64\n\ts_nop 0\n\t.endr");
  }
  unsigned long long t1 = now();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t2 = now();
  if (lane == 0) {
    cyc[0] = t1 - t0;
    cyc[1] = t2 - t0;
  }
  out[512 + lane] = v0 + v1 + v2 + v3 + s0 + r0[0] + r1[1] + r2[2] + r3[3];
}

// every wave of the chip in the 2-chain MFMA loop: the fp32 MFMA rate the chip sustains
// (clock under load included), the ceiling for everything else in this repo
__global__ __launch_bounds__(512, 1) void peak(float* out, int reps, int chains) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const float x = threadIdx.x * 1e-3f, y = 1.f + threadIdx.x;
  unsigned long long t0 = now();
  if (chains == 2) {
    for (int i = 0; i < reps; ++i) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      }
    }
  } else {
    for (int i = 0; i < reps; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = now();
  out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + (float)(t1 - t0);
}

// the same with operands that change from MFMA to MFMA (16 random registers each side): the
// data-dependent power draw decides the clock the chip holds
__global__ __launch_bounds__(512, 1) void peak_random(const float* rnd, float* out, int reps) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
  float xs[16], ys[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    xs[i] = rnd[(threadIdx.x * 16 + i) & 65535];
    ys[i] = rnd[(threadIdx.x * 16 + i + 7777) & 65535];
  }
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[u & 15], ys[(u * 5) & 15], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[(u + 3) & 15], ys[(u * 7 + 1) & 15], a1, 0, 0, 0);
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a0[2] + a1[3];
}

template <int KIND>
void run(const char* name, int n_instr, const float* src, float* out, unsigned long long* cyc) {
  printf("%-28s\n", name);
  const char* modes[] = {"sibling idle", "2 chains", "4 chains", "2 chains, probe prio 3", "s_nop 0 / 8 MFMA", "s_nop 7 / 8 MFMA",
                         "1 dependent chain", "ds_read+waitcnt / 8 MFMA", "s_sleep 0 / 8 MFMA", "s_setprio 0 / 8 MFMA", "v_mov / 8 MFMA"};
  for (int mode = 0; mode < 11; ++mode) {
    printf("   %-26s probe ticks/instr:", modes[mode]);
    double mf = 0;
    for (int it = 0; it < 5; ++it) {
      hipMemset(cyc, 0, 64);
      hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(512), 0, 0, src, out, cyc, mode, 400);
      hipDeviceSynchronize();
      unsigned long long h[3];
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      printf(" %9.1f", (double)h[0] / n_instr);
      mf = (double)h[2] / (400.0 * 64);
    }
    printf("   | sibling ticks per MFMA %.2f\n", mf);
  }
}

int main() {
  float *src, *out;
  unsigned long long* cyc;
  hipMalloc(&src, 1 << 20);
  hipMemset(src, 0, 1 << 20);
  hipMalloc(&out, 1 << 16);
  hipMalloc(&cyc, 64);
  {
    float* big;
    hipMalloc(&big, 256 * 8 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves : {4, 8})
      for (int chains : {2, 4})
        for (int wgs : {256, 2048}) {
          const int reps = 4000;
          hipLaunchKernelGGL(peak, dim3(wgs), dim3(64 * waves), 0, 0, big, 100, chains);
          hipEventRecord(e0);
          hipLaunchKernelGGL(peak, dim3(wgs), dim3(64 * waves), 0, 0, big, reps, chains);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          const double flop = (double)wgs * waves * reps * 64 * 2048.0;
          printf("pure MFMA f32 16x16x4: %d WGs x %d waves, %d chains: %.3f ms  %.1f TFLOP/s\n", wgs, waves, chains, ms, flop / ms * 1e-9);
        }
  }
  {
    float *big, *rnd;
    hipMalloc(&big, 2048 * 512 * 4);
    hipMalloc(&rnd, 65536 * 4);
    float* h = (float*)malloc(65536 * 4);
    for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(rnd, h, 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024, 2048, 2048, 2048, 256}) {
      const int reps = 4000;
      hipLaunchKernelGGL(peak_random, dim3(wgs), dim3(512), 0, 0, rnd, big, 100);
      hipEventRecord(e0);
      hipLaunchKernelGGL(peak_random, dim3(wgs), dim3(512), 0, 0, rnd, big, reps);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)wgs * 8 * reps * 64 * 2048.0;
      printf("random-operand MFMA f32 16x16x4: %d WGs x 8 waves, 2 chains: %.3f ms  %.1f TFLOP/s\n", wgs, ms, flop / ms * 1e-9);
    }
  }
  if (getenv("NPF_PEAK_ONLY")) return 0;
  run<0>("64 x s_add_u32 chain", 64, src, out, cyc);
  run<1>("64 x v_mov_b32 indep", 64, src, out, cyc);
  run<3>("16 x lds-dma 1 KiB (saddr)", 16, src, out, cyc);
  run<4>("16 x global_load_dwordx4", 16, src, out, cyc);
  run<5>("64 x ds_read_b128", 64, src, out, cyc);
  return 0;
}
