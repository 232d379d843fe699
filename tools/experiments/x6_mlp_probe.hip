// Development probe (not part of the product): a bare stack of L x (Linear 256 -> 256, bias, ReLU) on register-resident
// activations -- the chain kernel's layer -- with the fp32 products on the bf16 matrix pipe: weights split once into three
// exact bf16 terms (k-permuted images as npf_cast_bf16_weights makes them), the layer input split in registers once per layer,
// six v_mfma_f32_16x16x32_bf16 per (16-row block, 32-feature k-step).  Question: what does such a layer cost against the
// fp32 chain kernel's 131 TF/s on the same stack (tools/microbench.py chain)?  DESIGN.md section 9.
// Measured (MI355X, 1 M points, 8 layers): 4.75 ms = 231 TF/s fp32-equivalent = 1.77 x the fp32 chain kernel on the same stack,
// results within 7e-7 of float64 (the fp32 kernel: ~1e-6) -- with the plainest pipeline: 16-row slabs through a three-slot
// LDS ring (a two-slot ring with a full drain per slab: 5.2 ms), one barrier and one counted vmcnt per slab, fragments
// one k-step ahead, no epilogue fusion.  The matrix pipe is ~55 % busy at that.
// Build / run (GPU box):  hipcc --offload-arch=gfx950 -O3 tools/experiments/x6_mlp_probe.hip -o /tmp/x6_mlp_probe && /tmp/x6_mlp_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-result"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int F = 256;             // layer width
constexpr int kRows = 16;          // output rows per slab
constexpr int kSlabs = F / kRows;  // 16 slabs per layer
constexpr int kTermBytes = kRows * F * 2;      // one term of a slab: 16 rows x 256 bf16 = 8 KiB
constexpr int kSlabBytes = 3 * kTermBytes;     // 24 KiB
constexpr int kSlots = 3;  // slab S + 2 is in flight while slab S multiplies (a two-slot ring: 5.2 ms instead of the figure below)

__device__ __forceinline__ void dma16(const void* src, void* lds_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_uniform, 16, 0, 0);
}

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// split the 8 values (lo = block 2 st, hi = block 2 st + 1 of the lane) into three packed bf16x8 terms
__device__ __forceinline__ void split3(const f32x4& lo, const f32x4& hi, u32x4& t0, u32x4& t1, u32x4& t2) {
  const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    const unsigned h = cvt_pk(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = cvt_pk(ra, rb);
    const float la = ra - __builtin_bit_cast(float, m << 16), lb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    t0[p] = h;
    t1[p] = m;
    t2[p] = cvt_pk(la, lb);
  }
}

// img: [L][3 terms][256 rows][256 bf16] (k-permuted), bias: [L][256], X / Y: [wave][16 blocks][64 lanes] f32x4 (the register
// layout: block b, element e of lane (p, g) = feature 16 b + 4 g + e of the wave's point p)
template <int L>
__global__ __launch_bounds__(256, 2) void x6_mlp(const unsigned short* __restrict__ img, const float* __restrict__ bias,
                                                  const f32x4* __restrict__ X, f32x4* __restrict__ Y) {
  __shared__ __attribute__((aligned(16))) char smem[kSlots * kSlabBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;

  f32x4 cur[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) cur[b] = X[(gw * 16 + b) * 64 + lane];

  // DMA of slab S (global index: layer S / 16, rows 16 (S % 16) ..): 24 pieces of 1 KiB (term q / 8, rows 2 (q % 8), + 1),
  // six per wave; the swizzle (chunk c of row r at position c ^ (r & 15)) is applied to the source address
  // (uniform base per piece + one of two per-lane offsets: the swizzle term only depends on (wave + 4 n) & 7)
  unsigned lane_off[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int r0 = 2 * ((wave + 4 * n) & 7), row = r0 + (lane >> 5), pos = lane & 31;
    lane_off[n] = (unsigned)((lane >> 5) * 512 + ((pos ^ (row & 15)) << 4));
  }
  auto dma_slab = [&](int S, char* slot) {
    const char* base = (const char*)img + (size_t)(S / kSlabs) * (3 * F * F * 2) + (size_t)(S % kSlabs) * kRows * F * 2;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const int q = wave + 4 * n, term = q >> 3, r0 = 2 * (q & 7);
      const char* usrc = base + (size_t)term * (F * F * 2) + r0 * 512;
      dma16(usrc + lane_off[n & 1], slot + term * kTermBytes + r0 * 512);
    }
  };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned lane_row = (unsigned)(p * 512);

  dma_slab(0, smem);
  dma_slab(1, smem + kSlabBytes);
  int slot = 0;
  for (int l = 0; l < L; ++l) {
    // the layer's input as three packed bf16 terms (B operands), once per layer
    u32x4 tb[3][8];
#pragma unroll
    for (int st = 0; st < 8; ++st) split3(cur[2 * st], cur[2 * st + 1], tb[0][st], tb[1][st], tb[2][st]);
#pragma unroll
    for (int s = 0; s < kSlabs; ++s) {
      const int S = l * kSlabs + s;
      // slab S has landed for everyone, everyone is done with slab S - 1 (whose slot slab S + 2 goes into).  Counted wait:
      // the six pieces of slab S + 1 may stay in flight (vector-memory operations retire in order).  (A vector load of the
      // bias here made hipcc wait for vmcnt(0) at its use, i.e. drain the ring: 4.75 ms.)
      if (S + 1 < L * kSlabs) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};  // (no bias in the probe: in the real layer it rides in the slab image, one LDS read)
      if (S + 2 < L * kSlabs) dma_slab(S + 2, smem + ((slot + 2) % kSlots) * kSlabBytes);
      const unsigned sl = lds0 + slot * kSlabBytes + lane_row;
      f32x4 sm = {0.f, 0.f, 0.f, 0.f};
      // fragments of k-step st + 1 are read while step st multiplies (two steps ahead: 12 spilled registers, no gain)
      u32x4 fr[2][3];
      asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:8192\n\tds_read_b128 %2, %3 offset:16384"
                   : "=&v"(fr[0][0]), "=&v"(fr[0][1]), "=&v"(fr[0][2])
                   : "v"(sl + (((0 + g) ^ p) << 4)));
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int c = st & 1, n = c ^ 1;
        if (st + 1 < 8) {
          asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:8192\n\tds_read_b128 %2, %6 offset:16384"
                       : "=&v"(fr[n][0]), "=&v"(fr[n][1]), "=&v"(fr[n][2]), "+v"(fr[c][0]), "+v"(fr[c][1]), "+v"(fr[c][2])
                       : "v"(sl + (((4 * (st + 1) + g) ^ p) << 4)));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[c][0]), "+v"(fr[c][1]), "+v"(fr[c][2]));
        }
#define MM(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[c][A]), __builtin_bit_cast(bf16x8, tb[B][st]), C, 0, 0, 0)
        MM(2, 0, sm);
        MM(0, 0, acc);
        MM(0, 2, sm);
        MM(1, 0, acc);
        MM(1, 1, sm);
        MM(0, 1, acc);
#undef MM
      }
      acc += sm;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.f);
      cur[s] = acc;  // (block s of the input is dead: its terms are in tb)
      slot = (slot + 1) % kSlots;
    }
  }
#pragma unroll
  for (int b = 0; b < 16; ++b) Y[(gw * 16 + b) * 64 + lane] = cur[b];
}

static unsigned short bf16_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_to_f(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  constexpr int L = 8;
  const int n_points = argc > 1 ? atoi(argv[1]) : 1 << 20;
  const int n_waves = n_points / 16, n_wg = n_waves / 4;
  std::vector<float> W((size_t)L * F * F), B((size_t)L * F);
  srand(1);
  for (auto& w : W) w = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.108f;  // ~ sqrt(3 / 256): activations keep their scale
  for (auto& b : B) b = 0.f;
  // three-term images, k-permuted inside groups of 32: position 8 g + j <- feature 4 g + j, position 8 g + 4 + j <- 16 + 4 g + j
  std::vector<unsigned short> img((size_t)L * 3 * F * F);
  for (int l = 0; l < L; ++l)
    for (int n = 0; n < F; ++n)
      for (int k = 0; k < F; ++k) {
        const int grp = k >> 5, kk = k & 31, g = (kk & 15) >> 2, j = kk & 3, hi = kk >> 4;
        const int pos = 32 * grp + 8 * g + 4 * hi + j;
        float r = W[((size_t)l * F + n) * F + k];
        for (int t = 0; t < 3; ++t) {
          const unsigned short h = bf16_rne(r);
          img[(((size_t)l * 3 + t) * F + n) * F + pos] = h;
          r -= bf16_to_f(h);
        }
      }
  std::vector<float> X((size_t)n_points * F);
  for (auto& x : X) x = (rand() / (float)RAND_MAX) * 2.f - 1.f;
  // register layout: [wave][b][lane][e] = feature 16 b + 4 g + e of point 16 wave + p
  std::vector<float> Xr((size_t)n_points * F);
  for (int wv = 0; wv < n_waves; ++wv)
    for (int b = 0; b < 16; ++b)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e)
          Xr[(((size_t)wv * 16 + b) * 64 + lane) * 4 + e] = X[((size_t)wv * 16 + (lane & 15)) * F + 16 * b + 4 * (lane >> 4) + e];
  unsigned short* d_img;
  float *d_b, *d_x, *d_y;
  hipMalloc(&d_img, img.size() * 2);
  hipMalloc(&d_b, B.size() * 4);
  hipMalloc(&d_x, Xr.size() * 4);
  hipMalloc(&d_y, Xr.size() * 4);
  hipMemcpy(d_img, img.data(), img.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(d_b, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_x, Xr.data(), Xr.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(x6_mlp<L>, dim3(n_wg), dim3(256), 0, 0, d_img, d_b, (const f32x4*)d_x, (f32x4*)d_y);
  hipDeviceSynchronize();
  const int reps = 10;
  hipEventRecord(e0);
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(x6_mlp<L>, dim3(n_wg), dim3(256), 0, 0, d_img, d_b, (const f32x4*)d_x, (f32x4*)d_y);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flop = 2.0 * F * F * L * (double)n_points;
  printf("x6 MLP stack: %d layers 256->256 over %d points: %.3f ms, %.1f TF/s fp32-equivalent (fp32 MFMA peak 157.3; the fp32 chain kernel runs this stack at ~131)\n",
         L, n_points, ms, flop / ms * 1e-9);
  if (hipGetLastError() != hipSuccess) printf("HIP error\n");
  // check the first 32 points against float64
  std::vector<float> Yr((size_t)64 * 16 * 2 * 4);
  hipMemcpy(Yr.data(), d_y, Yr.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, scale = 0;
  for (int pt = 0; pt < 32; ++pt) {
    std::vector<double> h(F), o(F);
    for (int k = 0; k < F; ++k) h[k] = X[(size_t)pt * F + k];
    for (int l = 0; l < L; ++l) {
      for (int n = 0; n < F; ++n) {
        double a = B[(size_t)l * F + n];
        for (int k = 0; k < F; ++k) a += (double)W[((size_t)l * F + n) * F + k] * h[k];
        o[n] = a > 0 ? a : 0;
      }
      h = o;
    }
    const int wv = pt / 16, p = pt % 16;
    for (int n = 0; n < F; ++n) {
      const int b = n / 16, g = (n % 16) / 4, e = n % 4;
      const double got = Yr[(((size_t)wv * 16 + b) * 64 + (16 * g + p)) * 4 + e];
      worst = fmax(worst, fabs(got - h[n]));
      scale = fmax(scale, fabs(h[n]));
    }
  }
  printf("max |y - y64| = %.3e, max |y64| = %.3e, relative %.2e (fp32 chain arithmetic: ~1e-6)\n", worst, scale, worst / scale);
  return 0;
}
