// Which (row, k-block) does lane l's scale byte address in v_mfma_scale_f32_16x16x128_f8f6f4?  A = B = all 1.0; one lane's
// scale is raised to 2.0 (A side, then B side), k-block contributions are made distinguishable by zeroing B outside one block.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void probe(int hot_lane, int side, int live_block, int opsel, float* C) {
  const int l = threadIdx.x;
  v8i one; for (int i = 0; i < 8; ++i) one[i] = 0x38383838;
  v8i zero; for (int i = 0; i < 8; ++i) zero[i] = 0;
  v8i a = one, b = (l >> 4) == live_block ? one : zero;         // only k-block `live_block` contributes
  int sc_hot = 127 | (127 << 8) | (127 << 16) | (127 << 24);
  sc_hot = (sc_hot & ~(0xFF << (8 * opsel))) | (128 << (8 * opsel));
  const int sc_one = 127 | (127 << 8) | (127 << 16) | (127 << 24);
  const int sa = (side == 0 && l == hot_lane) ? sc_hot : sc_one, sb = (side == 1 && l == hot_lane) ? sc_hot : sc_one;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, sa, 1, sb);
  for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
__global__ void probe1(int hot_lane, int side, int kk, float* C) {
  const int l = threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0; b[i] = 0; }
  // element kk lives in lanes with (l>>4) == kk/32, byte kk%32
  if ((l >> 4) == kk / 32) { a[(kk % 32) / 4] = 0x38 << (8 * (kk % 4)); b[(kk % 32) / 4] = 0x38 << (8 * (kk % 4)); }
  const int sc_one = 127, sc_hot = 128;
  const int sa = (side == 0 && l == hot_lane) ? sc_hot : sc_one, sb = (side == 1 && l == hot_lane) ? sc_hot : sc_one;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
int main() {
  float* dC; hipMalloc(&dC, 1024); float h[256];
  // finer: only ONE k element of B is live (byte kk of the row), so C[i][j] = scale applied to element kk of row i
  for (int side = 0; side < 2; ++side)
    for (int lane = 0; lane < 64; lane += 16) {
      printf("side %c hot lane %2d: k elements whose product is doubled (row/col 0): ", side ? 'B' : 'A', lane);
      for (int kk = 0; kk < 128; ++kk) {
        hipLaunchKernelGGL(probe1, dim3(1), dim3(64), 0, 0, lane, side, kk, dC);
        hipMemcpy(h, dC, 1024, hipMemcpyDeviceToHost);
        if (h[0] != 1.0f) printf("%d(%g) ", kk, h[0]);
      }
      printf("\n");
    }
  return 0;
}
