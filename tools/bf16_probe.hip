// Development probe: operand layout of v_mfma_f32_16x16x32_bf16 on gfx950, as the bf16 chain path assumes it:
//   A: lane (m = lane & 15, kg = lane >> 4) holds A[m][8 kg .. 8 kg + 7]
//   B: lane (n = lane & 15, kg)             holds B[8 kg .. 8 kg + 7][n]
//   C: lane (n = lane & 15, rg = lane >> 4) holds C[4 rg + i][n], i < 4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#pragma clang diagnostic ignored "-Wunused-result"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const float* A, const float* B, float* C) {  // A [16][32], B [32][16], C [16][16] row-major
  const int lane = threadIdx.x, m = lane & 15, kg = lane >> 4;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)A[m * 32 + 8 * kg + i];
    b[i] = (__bf16)B[(8 * kg + i) * 16 + m];
  }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) C[(4 * kg + i) * 16 + m] = acc[i];
}

int main() {
  float hA[512], hB[512], hC[256], ref[256];
  for (int i = 0; i < 512; ++i) {
    hA[i] = (float)((i * 7) % 13 - 6) / 8.f;   // exactly representable in bf16
    hB[i] = (float)((i * 5) % 11 - 5) / 4.f;
  }
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      float s = 0;
      for (int kk = 0; kk < 32; ++kk) s += hA[m * 32 + kk] * hB[kk * 16 + n];
      ref[m * 16 + n] = s;
    }
  float *dA, *dB, *dC;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < 256; ++i) err = fmax(err, fabs(hC[i] - ref[i]));
  printf("bf16 16x16x32 layout check: max |C - ref| = %g (%s)\n", err, err < 1e-5 ? "layout as assumed" : "LAYOUT MISMATCH");
  return err < 1e-5 ? 0 : 1;
}
