// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 operand and scale lane maps on gfx950 (exact small-integer data).
// Assumed map: lane l holds A[row l&15][k = 32*(l>>4) + j], j = 0..31 (8 VGPRs, byte order), B[k][col l&15] likewise;
// scale VGPR byte `opsel` of lane l applies to that lane's 32 k-elements (E8M0: 2^(byte-127)).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(const uint8_t* A, const uint8_t* Bt, const uint8_t* sa, const uint8_t* sb, float* C) {
  const int l = threadIdx.x;
  v8i a = *(const v8i*)(A + (l & 15) * 128 + 32 * (l >> 4));
  v8i b = *(const v8i*)(Bt + (l & 15) * 128 + 32 * (l >> 4));
  const int scale_a = sa[(l & 15) * 4 + (l >> 4)], scale_b = sb[(l & 15) * 4 + (l >> 4)];
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
  for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];     // C[row][col], row = 4*(l>>4)+r, col = l&15
}

static float e4m3_to_f(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0: random scales; 1: all scales 1; 2: only scale_a varies; 3: only scale_b varies
  static const uint8_t codes[9] = {0x00, 0x38, 0x40, 0x44, 0x30, 0xB8, 0xC0, 0xC4, 0x3C};   // 0 1 2 3 .5 -1 -2 -3 1.5
  uint8_t hA[16 * 128], hB[16 * 128], hsa[64], hsb[64];
  uint32_t st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 16; };
  for (int i = 0; i < 16 * 128; ++i) { hA[i] = codes[rnd() % 9]; hB[i] = codes[rnd() % 9]; }
  for (int i = 0; i < 64; ++i) { hsa[i] = 127 + (int)(rnd() % 4) - 1; hsb[i] = 127 + (int)(rnd() % 3); }
  if (mode == 1 || mode == 3) for (int i = 0; i < 64; ++i) hsa[i] = 127;
  if (mode == 1 || mode == 2) for (int i = 0; i < 64; ++i) hsb[i] = 127;
  float ref[256];
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int k = 0; k < 128; ++k)
        s += (double)e4m3_to_f(hA[i * 128 + k]) * ldexp(1.0, hsa[i * 4 + k / 32] - 127) * e4m3_to_f(hB[j * 128 + k]) * ldexp(1.0, hsb[j * 4 + k / 32] - 127);
      ref[i * 16 + j] = (float)s;
    }
  uint8_t *dA, *dB, *dsa, *dsb; float* dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dC, 1024);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa, 64, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
  float hC[256];
  hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
  // the kernel's MFMA "A" rows index C rows?  try both orientations
  double e1 = 0, e2 = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { e1 = fmax(e1, fabs(hC[i * 16 + j] - ref[i * 16 + j])); e2 = fmax(e2, fabs(hC[j * 16 + i] - ref[i * 16 + j])); }
  printf("max |C - A.B^T| = %g   max |C^T - A.B^T| = %g   (ref[0][1]=%g C[0][1]=%g C[1][0]=%g)\n", e1, e2, ref[1], hC[1], hC[16]);
  return 0;
}
