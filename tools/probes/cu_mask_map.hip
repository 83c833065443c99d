// Which physical compute units does bit b of a hipExtStreamCreateWithCUMask mask enable on this device?
// For every bit b: a stream whose mask has only that bit set runs a kernel of 64 workgroups, each recording its XCC_ID and HW_ID
// (SE, CU).  Prints, per bit, the set of (xcc, se, cu) seen.  Build: hipcc --offload-arch=gfx950 -O2 cu_mask_map.hip -o cu_mask_map.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>
__global__ void where(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
    unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((32 - 1) << 11));
    out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hw;
  }
  // keep the CU busy for a moment so that workgroups spread over every enabled CU
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 20000) {}
}
int main(int argc, char** argv) {
  int nb = argc > 1 ? atoi(argv[1]) : 256, words = 8;
  unsigned* d; hipMalloc(&d, 4096 * 8);
  std::vector<unsigned> h(4096 * 2);
  for (int b = 0; b < nb; ++b) {
    unsigned mask[8] = {0};
    mask[b / 32] = 1u << (b % 32);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, words, mask) != hipSuccess) { printf("bit %d: stream creation failed\n", b); continue; }
    hipLaunchKernelGGL(where, dim3(64), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
    std::set<unsigned> seen;
    for (int i = 0; i < 64; ++i) {
      const unsigned xcc = h[2 * i] & 15, hw = h[2 * i + 1];
      const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;     // gfx9 HW_ID: [11:8] CU_ID, [12] SH_ID, [15:13] SE_ID
      seen.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
    }
    printf("bit %3d:", b);
    for (unsigned v : seen) printf(" (xcc %u se %u sh %u cu %u)", v >> 16, (v >> 8) & 255, (v >> 4) & 15, v & 15);
    printf("\n");
    hipStreamDestroy(s);
  }
  return 0;
}
