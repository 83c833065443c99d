// C[M,N] = epilogue(X[M,K] . W[N,K]^T) on MFMA, in two flavours sharing one kernel body:
//  * hm_gemm          : plain row-major X -- the ViT-H patch-embed / QKV / proj / MLP GEMMs and the
//                       decoder's to_kv projection (nn.Linear calls at vit.py:82-87,:110-126,
//                       pose_transformer.py:114; >97 % of the FLOPs of HAMER.forward_step).
//  * hm_conv2d_nhwc   : implicit GEMM for the YOLOv7 convolutions (Conv.fuseforward common.py:114,
//                       RepConv common.py:502-504, Detect yolo.py:151): X rows are output pixels, the K
//                       axis walks (ky, kx, ci) of an NHWC tensor, out-of-image taps read a zero line.
//
// gfx950 design.  Block tile (WM*MI*16) x (WN*NI*16) x 64 with WM x WN waves, each wave MI x NI MFMA
// 16x16x32 accumulators.  Both operands are K-contiguous and staged global->LDS with 16-byte LDS-DMA
// (global_load_lds_dwordx4) into a ring of STAGES buffers; the ring is advanced with a COUNTED
// s_waitcnt vmcnt(N) and a raw s_barrier, so STAGES-1 K-tiles of loads stay in flight across the
// barrier (a __syncthreads would drain them).  The LDS image is lane-linear, so the bank-conflict XOR
// swizzle (16-B chunk ^= row&7 inside each 128-B row) is applied to the per-lane SOURCE address and
// again on the ds_read_b128 fragment reads.  The MFMA "A" operand is the W tile and "B" the X tile,
// so a lane's 4 accumulator registers are 4 consecutive output columns of one row: bias / residual /
// outputs move as 8- or 16-byte vectors.  Tiles are walked in an XCD-aware order.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "hamer_hip_internal.h"

namespace {

constexpr int BK_DEFAULT = 64;
int g_px_grid = -2;    // workgroups of gemm_px_kernel: -2 read HM_PX_GRID on first use, -1 default

struct KArgs {                                    // kernel-side view of either entry point
  const void* X; const void* W; void* C; const float* bias; const float* resid;
  int M, N, K, ldx, ldw, ldc, ldr, resid_mod;
  int group_m;                                    // tile order: runs of group_m M-tiles per N-tile (L2 reuse of both panels)
  // deferred LayerNorm (HM_EPI_RESID_LN produces, HM_EPI_LN_STORE / HM_EPI_LN_GELU consume)
  const float* ln_gamma; void* ln_xg; float* ln_stats; const float* ln_colsum;
  int ln_P;
  const unsigned char* xs; const float* wscale; unsigned char* out_scales;   // hm_gemm_fp8
  const void* resid16; int ldr16;                 // HM_EPI_ADD_RELU: 16-bit residual (conv)
  int ksplit;                                     // HM_EPI_F32 only: K is cut into ksplit ranges, one workgroup and one [M][ldc] slab of C each
  int general_loader;                             // CONV: 1 = the general implicit-GEMM loader even where the lean one applies (HM_OPT_CONV_GENERAL_LOADER: A/B runs, tests)
  float out_scale;                                // HM_EPI_GELU (16-bit out): C = gelu(acc + bias) * out_scale (a power of two; 1 = the plain epilogue)
  int kser;                                       // KSER kernels: the same ranges summed one after the other by ONE workgroup (0 / 1: one range)
  // convolution geometry (CONV only)
  const void* zeros;
  int H, Wd, Hout, Wout, ksz, stride, pad, cin_log2, taps;
};

// GELU with the erf of Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below a 16-bit output ulp);
// 1 + erf(-|z|) is formed without cancellation.  The fp32 decoder keeps the libm erff (common.h).
__device__ __forceinline__ float gelu_fast(float v) {
  const float z = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));   // v_rcp_f32 (1 ulp); __frcp_rn expands to a 10-instruction IEEE divide
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float q = p * t * __expf(-z * z);             // 1 - erf(z)
  return 0.5f * v * (v >= 0.f ? 2.0f - q : q);
}

// The same GELU on two values at once: the polynomial, the products and the final combination become v_pk_fma_f32 /
// v_pk_mul_f32 (two results per issue slot; nothing competes for the VALU in the epilogue), ~10 issue slots per element
// instead of ~17.  gelu(v) = max(v, 0) - 0.5 |v| (1 - erf(|v| / sqrt 2)), exp(-z^2) as exp2(v^2 * (-0.5 log2 e)).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_fast2(f32x2_t v) {
  const f32x2_t av = __builtin_elementwise_abs(v);
  const f32x2_t d = __builtin_elementwise_fma(av, f32x2_t{0.3275911f * 0.70710678118654752440f, 0.3275911f * 0.70710678118654752440f}, f32x2_t{1.0f, 1.0f});
  f32x2_t t;
  t[0] = __builtin_amdgcn_rcpf(d[0]); t[1] = __builtin_amdgcn_rcpf(d[1]);
  f32x2_t p = __builtin_elementwise_fma(t, f32x2_t{1.061405429f, 1.061405429f}, f32x2_t{-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(t, p, f32x2_t{1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(t, p, f32x2_t{-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(t, p, f32x2_t{0.254829592f, 0.254829592f});
  const f32x2_t x2 = v * v * f32x2_t{-0.5f * 1.44269504088896340736f, -0.5f * 1.44269504088896340736f};
  f32x2_t e;
  e[0] = __builtin_amdgcn_exp2f(x2[0]); e[1] = __builtin_amdgcn_exp2f(x2[1]);
  const f32x2_t q = p * t * e;                          // 1 - erf(|v| / sqrt 2)
  return __builtin_elementwise_fma(av * q, f32x2_t{-0.5f, -0.5f}, __builtin_elementwise_max(v, f32x2_t{0.f, 0.f}));
}

// Tile walk.  Blocks b, b+8, .. share an XCD (and its 4 MB L2); xcd_remap gives every XCD a contiguous run of
// tile ids, and inside that run tiles are ordered in super-rows of `group_m` M-tiles, M fastest, so the ~32
// tiles an XCD works on at once form a group_m x (32 / group_m) rectangle: group_m A panels and 32/group_m W
// panels stream through its L2 once per round instead of 2 + tiles_n (PMC: FETCH_SIZE of the N = 3840..6144
// GEMMs was 9-13x their algorithmic bytes with the plain row-major walk).
__device__ __forceinline__ void tile_coords(int wgid, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
  const int per_super = group_m * tiles_n;
  const int super = wgid / per_super, r = wgid - super * per_super;
  const int rows = min(group_m, tiles_m - super * group_m);     // the last super-row may be short
  tn = r / rows;
  tm = super * group_m + (r - tn * rows);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N < 0, "add this vmcnt immediate");
}

// n-subtiles per epilogue column group: the largest divisor of NI that keeps a wave's staging within 16 KB (<= 16 / MI)
constexpr int epi_cgn(int MI, int NI, int KB = 16) {
  int c = NI < KB / MI ? NI : KB / MI;
  while (NI % c != 0) --c;
  return c;
}
constexpr int epi_stage_bytes(int MI, int NI, int KB = 16) { return MI * 16 * epi_cgn(MI, NI, KB) * 64; }

// Epilogue of one wave through LDS.  acc[ni][mi] holds C[mb + mi*16 + (lane&15)][nb + ni*16 + 4*(lane>>4) .. +3]
// (the MFMA accumulator layout: 16 rows x 16-byte pieces per store).  Written straight from that layout every
// store instruction touches 16 rows with 32-64 B each (PMC: WRITE_SIZE 1.4-1.7x the output bytes) and the
// residual loads form 32 dependent load->add->store chains per wave; measured, that epilogue was 30-40 % of the
// GEMM time.  Instead each wave transposes its tile through a private 16 KB LDS region (the K-loop ring is free
// by then), in column groups of 64: accumulator pieces are written with a 16-byte-chunk XOR swizzle
// (chunk ^= row & 15: conflict-free for the 16 rows of a piece), read back row-major, and every global
// instruction then covers 4 full 256-byte row segments (128 B for 16-bit outputs).  Residual rows are fetched
// row-major too, all issued before the LDS round trip so their latency overlaps it.
//
// Deferred LayerNorm.  LN(x) . W^T = rstd * (x*gamma . W^T - mean * colsum) + (b + W.beta), colsum[n] = sum_k W[n][k] gamma[k]:
// the GEMM that produces the fp32 residual stream x (HM_EPI_RESID_LN) also stores x * gamma as the next GEMM's
// 16-bit operand and, per 64-column group and row ([D/64][M][2]), the partial (sum, sum of squares) of x; the consuming GEMM
// (HM_EPI_LN_STORE / HM_EPI_LN_GELU) reduces the partials to (mean, rstd) per tile row before its K loop
// (`rowstat`, in LDS) and applies them here.  That removes the LayerNorm launch: 94 MB of HBM traffic per call
// become 31 MB of extra stores in an epilogue that is already streaming the same rows.
constexpr bool epi_ln_in(int EPI) { return EPI == HM_EPI_LN_STORE || EPI == HM_EPI_LN_GELU; }

// sum over each aligned group of 8 lanes, on the VALU's DPP path (quad swaps, then the half-row mirror)
__device__ __forceinline__ float row8_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return v;
}

template <class T, int EPI, int MI, int NI, bool RESID_REGS = true, int STAGE_KB = 16>
__device__ __forceinline__ void epilogue(const KArgs& g, f32x4_t (&acc)[NI][MI], int mb, int nb, int lane, char* wlds,
                                         const float2* rowstat, const float* cbias, const float* cvec2, int split) {
  using elem = typename T::elem;
  // n-subtiles per column group: <= 16 KB of LDS per wave; the residual epilogue splits that between the
  // transposed accumulators and the residual rows, which arrive by LDS-DMA (no registers, all in flight at once)
  constexpr int CGN0 = epi_cgn(MI, NI, STAGE_KB);
  constexpr bool LNOUT = EPI == HM_EPI_RESID_LN, LNIN = epi_ln_in(EPI);
  constexpr bool RESREG = EPI == HM_EPI_RESID_F32 && RESID_REGS;   // residual rows prefetched into registers, one column group ahead
  constexpr bool RES = LNOUT || (EPI == HM_EPI_RESID_F32 && !RESID_REGS);   // residual rows through LDS (deferred-LN producer; fp8 kernel: no registers to spare)
  constexpr bool MX8 = EPI == HM_EPI_GELU_MX8;
  constexpr bool ACT_GELU = EPI == HM_EPI_GELU || EPI == HM_EPI_LN_GELU || MX8;
  // (RESREG keeps the half-width column group too: two groups of residual rows in registers are 2 x ITS x 4 VGPRs)
  constexpr int CGN = ((RES || RESREG) && CGN0 > 1) ? CGN0 / 2 : CGN0;
  constexpr int NCH = CGN * 4;                     // 16-byte chunks per staged row (4, 8 or 16)
  constexpr int RS = NCH * 16;                     // staged row stride in bytes
  constexpr int RPI = 64 / NCH;                    // rows per row-major instruction (1 KiB)
  constexpr int ITS = MI * 16 / RPI;               // row-major instructions per column group
  constexpr int NCG = NI / CGN;                    // column groups of the wave's strip
  const int arow = lane & 15, apiece = lane >> 4;  // accumulator layout
  const int rrow = lane / NCH, rslot = lane % NCH; // row-major layout
  // cbias / cvec2: this wave's columns of the bias (zeros when there is none) and of ln_gamma / ln_colsum, staged in
  // LDS by the kernel prologue -- a global load here sits on the epilogue's critical path once per column group
  char* rlds = wlds + MI * 16 * RS;                // residual rows (RES only)
  // LNOUT: per-row partial (sum, sum of squares) of this lane's columns, one pair per row-major instruction
  constexpr int CPG = LNOUT ? 64 / (CGN * 16) : 1;  // column groups per 64-column statistics group
  static_assert(!LNOUT || ((CGN * 16 <= 64) && (64 % (CGN * 16) == 0) && (NI * 16) % 64 == 0), "LN statistics groups are 64 columns");
  static_assert(!LNIN || NCH >= 8, "deferred-LN consumers use the 16-byte store path");
  float s1[LNOUT ? ITS : 1], s2[LNOUT ? ITS : 1];
  // RESREG.  The fp32 residual epilogue moves 2 x 4 bytes per output element through HBM and sits at the HBM bound; what
  // can be saved is latency: the rows of column group cg + 1 are requested (row-major, 16 bytes per lane, straight into
  // registers -- the K loop's fragment registers are free by now) before group cg is processed, so each wave pays ONE
  // exposed round trip instead of one per column group.
  f32x4_t rr[RESREG ? 2 : 1][RESREG ? ITS : 1];
  auto resid_fetch = [&](int cg, f32x4_t (&dst)[RESREG ? ITS : 1]) {
    if constexpr (RESREG) {
      const int ncol0 = nb + cg * CGN * 16;
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int row = it * RPI + rrow;
        int m = mb + row, n = ncol0 + ((rslot ^ (row & (NCH - 1))) << 2);
        m = m < g.M ? m : g.M - 1;                 // out-of-range lanes fetch a valid address; their result is never stored
        n = n < g.N ? n : 0;
        const int rm = g.resid_mod > 0 ? m % g.resid_mod : m;
        dst[it] = *(const f32x4_t*)(g.resid + (size_t)rm * g.ldr + n);
      }
    }
  };
  if constexpr (RESREG) resid_fetch(0, rr[0]);
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) {
    const int ncol0 = nb + cg * CGN * 16;
    if constexpr (RESREG) {
      if (cg + 1 < NCG) resid_fetch(cg + 1, rr[(cg + 1) & 1]);
    }
    if (RES) {                                     // residual rows -> LDS, row-major, same (row, slot) image as the reads below
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int row = it * RPI + rrow;
        int m = mb + row, n = ncol0 + ((rslot ^ (row & (NCH - 1))) << 2);
        m = m < g.M ? m : g.M - 1;                 // out-of-range lanes fetch a valid address; their result is never stored
        n = n < g.N ? n : 0;
        const int rm = g.resid_mod > 0 ? m % g.resid_mod : m;
        glds16(g.resid + (size_t)rm * g.ldr + n, rlds + it * 1024);
      }
    }
    // accumulator layout -> LDS (16-byte chunk XOR swizzle)
#pragma unroll
    for (int nl = 0; nl < CGN; ++nl)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = mi * 16 + arow;
        const int slot = (nl * 4 + apiece) ^ (row & (NCH - 1));
        *(f32x4_t*)(wlds + row * RS + slot * 16) = acc[cg * CGN + nl][mi];
      }
    if (RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // LDS -> row-major, bias / activation / residual, coalesced stores
    constexpr bool RELU = EPI == HM_EPI_RELU || EPI == HM_EPI_ADD_RELU;
    if constexpr ((EPI == HM_EPI_STORE || EPI == HM_EPI_GELU || EPI == HM_EPI_SILU || LNIN || MX8 || RELU) && NCH >= 8) {
      static_assert(!MX8 || NCH == 16, "MXFP8 output: 8 lanes per row, 4 lanes per 32-column scale block");
      // 16-bit outputs: 8 columns (two staged chunks) per lane -> one 16-byte store, NCH/2 lanes per row
      constexpr int LPR = NCH / 2, RPI2 = 64 / LPR, ITS2 = MI * 16 / RPI2;
      const int row0 = lane / LPR, j = lane % LPR;
      f32x4_t cs0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, cs1 = cs0;
      const int nw = cg * CGN * 16 + 8 * j;         // column inside the wave's strip
      if (LNIN) { cs0 = *(const f32x4_t*)(cvec2 + nw); cs1 = *(const f32x4_t*)(cvec2 + nw + 4); }
      const f32x4_t bi0 = *(const f32x4_t*)(cbias + nw), bi1 = *(const f32x4_t*)(cbias + nw + 4);
#pragma unroll 2
      for (int it = 0; it < ITS2; ++it) {
        const int row = it * RPI2 + row0, m = mb + row, n = ncol0 + 8 * j;
        const int sw = row & (NCH - 1);
        f32x4_t v0 = *(const f32x4_t*)(wlds + row * RS + ((2 * j) ^ sw) * 16);
        f32x4_t v1 = *(const f32x4_t*)(wlds + row * RS + ((2 * j + 1) ^ sw) * 16);
        float2 st = float2{0.f, 1.f};
        if (LNIN) st = rowstat[row];
        if (m >= g.M || n >= g.N) continue;
        if (LNIN) { v0 = (v0 - st.x * cs0) * st.y; v1 = (v1 - st.x * cs1) * st.y; }
        // (__fadd_rn: never contracted into the activation's first multiply, so every tile shape rounds alike)
#pragma unroll
        for (int q = 0; q < 4; ++q) { v0[q] = __fadd_rn(v0[q], bi0[q]); v1[q] = __fadd_rn(v1[q], bi1[q]); }
        if constexpr (MX8) {
          // MXFP8 out: this lane's 8 columns and its 3 quad neighbours' form one 32-column block of row m (N % 64 == 0,
          // so a row's lanes are all in or all out: the quad reduction sees no inactive lane)
          float amax = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2_t gq = gelu_fast2(f32x2_t{v0[q], v1[q]});
            v0[q] = gq[0]; v1[q] = gq[1];
            amax = fmaxf(amax, fmaxf(fabsf(v0[q]), fabsf(v1[q])));
          }
          const unsigned sb = mx8_scale_byte(quad_max(amax));
          const float inv = mx8_inv_scale(sb);
          int2 o8;
          o8.x = mx8_pack4(v0[0] * inv, v0[1] * inv, v0[2] * inv, v0[3] * inv);
          o8.y = mx8_pack4(v1[0] * inv, v1[1] * inv, v1[2] * inv, v1[3] * inv);
          *(int2*)((char*)g.C + (size_t)m * g.ldc + n) = o8;
          if ((j & 3) == 0) g.out_scales[(size_t)(n >> 5) * g.M + m] = (unsigned char)sb;
          continue;
        }
        typename T::vec8 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a = v0[q], b = v1[q];
          if (ACT_GELU) { const f32x2_t gq = gelu_fast2(f32x2_t{a, b}); a = gq[0]; b = gq[1]; }
          else if (EPI == HM_EPI_SILU) { const f32x2_t sq = silu2(f32x2_t{a, b}); a = sq[0]; b = sq[1]; }
          if constexpr (EPI == HM_EPI_GELU) { a *= g.out_scale; b *= g.out_scale; }      // (x 1.0f is exact: the same bytes without a prescale)
          o[q] = (elem)a; o[4 + q] = (elem)b;
        }
        if constexpr (RELU) {                        // (+ identity), ReLU; host guarantees N % 8 == 0 and 16-byte rows
          typename T::vec8 idn;
#pragma unroll
          for (int q = 0; q < 8; ++q) idn[q] = (elem)0.0f;
          if (EPI == HM_EPI_ADD_RELU) idn = *(const typename T::vec8*)((const elem*)g.resid16 + (size_t)m * g.ldr16 + n);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[q] = (elem)fmaxf(__fadd_rn(v0[q], (float)idn[q]), 0.f);
            o[4 + q] = (elem)fmaxf(__fadd_rn(v1[q], (float)idn[4 + q]), 0.f);
          }
        }
        if (n + 8 <= g.N && (g.ldc & 7) == 0) {
          *(typename T::vec8*)((elem*)g.C + (size_t)m * g.ldc + n) = o;
        } else if (n + 8 <= g.N) {                   // rows not 16-byte aligned: two 8-byte stores
          typename T::vec4 h0, h1;
#pragma unroll
          for (int q = 0; q < 4; ++q) { h0[q] = o[q]; h1[q] = o[4 + q]; }
          *(typename T::vec4*)((elem*)g.C + (size_t)m * g.ldc + n) = h0;
          *(typename T::vec4*)((elem*)g.C + (size_t)m * g.ldc + n + 4) = h1;
        } else {                                     // ragged N (N % 8 == 4): first half only
          typename T::vec4 h;
#pragma unroll
          for (int q = 0; q < 4; ++q) h[q] = o[q];
          *(typename T::vec4*)((elem*)g.C + (size_t)m * g.ldc + n) = h;
        }
      }
    } else if constexpr (LNOUT) {
      static_assert(RPI % NCH == 0 && NCH == 8, "the column of a lane must not depend on `it`; the row reduction is over 8 lanes");
      if (cg % CPG == 0) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) s1[it] = s2[it] = 0.f;
      }
      const int n = ncol0 + ((rslot ^ (rrow & (NCH - 1))) << 2);     // same for every `it` (RPI % NCH == 0)
      const bool ncol_ok = n < g.N;
      const f32x4_t ga = *(const f32x4_t*)(cvec2 + (n - nb)), bi = *(const f32x4_t*)(cbias + (n - nb));
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int row = it * RPI + rrow, m = mb + row;
        f32x4_t v = *(const f32x4_t*)(wlds + row * RS + rslot * 16);
        const f32x4_t r = *(const f32x4_t*)(rlds + row * RS + rslot * 16);
        if (m >= g.M || !ncol_ok) continue;
        v += bi + r;
        *(f32x4_t*)((float*)g.C + (size_t)m * g.ldc + n) = v;
        s1[it] += (v[0] + v[1]) + (v[2] + v[3]);
        s2[it] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        typename T::vec4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (elem)(v[q] * ga[q]);
        *(typename T::vec4*)((elem*)g.ln_xg + (size_t)m * g.N + n) = o;
      }
      if (cg % CPG == CPG - 1) {                     // a 64-column group is complete: reduce over the 8 lanes of each row (DPP)
        const int pidx = (nb + (cg / CPG) * 64) >> 6;
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const float a = row8_sum(s1[it]), b = row8_sum(s2[it]);
          const int m = mb + it * RPI + rrow;
          if (rslot == 0 && m < g.M && pidx < g.ln_P) *(float2*)(g.ln_stats + ((size_t)pidx * g.M + m) * 2) = float2{a, b};
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int row = it * RPI + rrow, m = mb + row;
        const int n = ncol0 + ((rslot ^ (row & (NCH - 1))) << 2);
        f32x4_t v = *(const f32x4_t*)(wlds + row * RS + rslot * 16);
        f32x4_t r = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (RES) r = *(const f32x4_t*)(rlds + row * RS + rslot * 16);
        if constexpr (RESREG) r = rr[cg & 1][it];
        if (m >= g.M || n >= g.N) continue;
        {
          const f32x4_t bi = *(const f32x4_t*)(cbias + (n - nb));
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = __fadd_rn(v[q], bi[q]);
        }
        if (EPI == HM_EPI_GELU) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = gelu_fast(v[q]) * g.out_scale;
        } else if (EPI == HM_EPI_SILU) {
          v = silu4(v);
        } else if (RELU) {
          if (EPI == HM_EPI_ADD_RELU) {
            const typename T::vec4 idn = *(const typename T::vec4*)((const elem*)g.resid16 + (size_t)m * g.ldr16 + n);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = __fadd_rn(v[q], (float)idn[q]);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        if (RES || RESREG) {
          *(f32x4_t*)((float*)g.C + (size_t)m * g.ldc + n) = v + r;
        } else if (EPI == HM_EPI_F32) {
          *(f32x4_t*)((float*)g.C + ((size_t)split * g.M + m) * g.ldc + n) = v;      // split-K: slab `split` of C
        } else {
          typename T::vec4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (elem)v[q];
          *(typename T::vec4*)((elem*)g.C + (size_t)m * g.ldc + n) = o;
        }
      }
    }
  }
}

// WK > 1 (round 3, convolutions on the small maps): the workgroup carries WK copies of the WM x WN wave grid, each with its own
// LDS ring; at every step group kg works on K tile kt * WK + kg, so the serial chain of K-steps -- what a layer with 30-120 row
// tiles is bound by -- is WK times shorter with WK times the bytes in flight, and no partial sums leave the CU: after the loop
// the groups kg > 0 hand their accumulators to group 0 through LDS (added in the order kg = 1, 2, ..: deterministic).
// KDUAL (WK = 1): the SAME summation order in one group of waves -- even K tiles into one accumulator set, odd ones into a
// second, added at the end -- so that a layer may run on either form, by launch size, with bit-identical results.
template <class T, int EPI, int WM, int WN, int MI, int NI, int STAGES, bool CONV, int BK = 64, int SCHED = 0, int WK = 1, bool KDUAL = false, bool KSER = false>
__global__ __launch_bounds__(64 * WM * WN * WK, WK > 1 ? 1 : 2) void gemm_tn_kernel(const KArgs g) {
  constexpr int NWG = WM * WN;                        // waves of one K group
  constexpr int NW = NWG * WK;
  constexpr int BM = WM * MI * 16, BN = WN * NI * 16;
  constexpr int ROWB = BK * 2;                        // bytes per tile row in LDS (128 or 64)
  constexpr int CH = BK / 8;                          // 16-byte chunks per row
  constexpr int RPI = 1024 / ROWB;                    // rows covered by one 1-KiB LDS-DMA instruction
  constexpr int XI = BM / NWG / RPI, WI = BN / NWG / RPI;   // staging instructions per wave and K-tile
  static_assert(BK == 64 || BK == 32, "BK is 64 or 32");
  static_assert(!CONV || BK == 64, "the implicit-GEMM loader is written for BK = 64");
  static_assert(BM % (NWG * RPI) == 0 && BN % (NWG * RPI) == 0, "tile rows must split into whole DMA pieces per wave");
  static_assert(WK == 1 || (CONV && SCHED == 0), "K groups: written for the convolution flavour");
  static_assert(!KDUAL || (WK == 1 && CONV && SCHED == 0), "the two-accumulator form is the one-group twin of WK = 2");
  static_assert(!KSER || (WK == 1 && CONV && SCHED == 0 && EPI != HM_EPI_F32), "serial K ranges: the one-workgroup twin of split-K");
  constexpr int XTILE_BYTES = BM * BK * 2, WTILE_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = XTILE_BYTES + WTILE_BYTES;
  constexpr int LOADS = XI + WI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using vec8 = typename T::vec8;
  using elem = typename T::elem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = WK > 1 ? wave_all / NWG : 0, wave = WK > 1 ? wave_all % NWG : wave_all;      // K group, wave within the group
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  int tm, tn;
  const int wgid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = EPI == HM_EPI_F32 ? wgid % g.ksplit : 0;   // which K range (split-K: ksplit > 1)
  tile_coords(EPI == HM_EPI_F32 ? wgid / g.ksplit : wgid, tiles_m, tiles_n, g.group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wave / WN, wc = wave % WN;

  const elem* __restrict__ X = (const elem*)g.X;
  const elem* __restrict__ W = (const elem*)g.W;

  // staging geometry: wave w loads X rows [XI*RPI*w, +XI*RPI) and W rows [WI*RPI*w, +WI*RPI); lane -> (row =
  // lane / CH, physical 16-B chunk = lane % CH); logical chunk = physical ^ key(row) with key = row & 7 for
  // 128-B rows and (row >> 2) & 3 for 64-B rows (4 rows share a 256-B bank row)
  const int srow = lane / CH;
  const int chunk = (lane % CH) ^ (BK == 64 ? (srow & 7) : ((srow >> 2) & 3));
  const elem* xsrc[XI];
  int pix_y[XI], pix_x[XI];
  const elem* wsrc[WI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    int gm = (SCHED == 97 ? 0 : m0) + wave * XI * RPI + i * RPI + srow;     // (97: experiment, every tile reads panel 0)
    gm = gm < g.M ? gm : g.M - 1;                          // edge rows: valid memory, never stored
    if (CONV) {
      const int hw = g.Hout * g.Wout;
      const int n = gm / hw, rem = gm - n * hw;
      const int oy = rem / g.Wout, ox = rem - oy * g.Wout;
      pix_y[i] = oy * g.stride - g.pad;
      pix_x[i] = ox * g.stride - g.pad;
      xsrc[i] = X + (size_t)n * g.H * g.Wd * g.ldx;        // image base
    } else {
      pix_y[i] = pix_x[i] = 0;
      xsrc[i] = X + (size_t)gm * g.ldx + chunk * 8;
    }
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    int gn = (SCHED == 97 ? 0 : n0) + wave * WI * RPI + i * RPI + srow;
    gn = gn < g.N ? gn : g.N - 1;
    wsrc[i] = W + (size_t)gn * g.ldw + chunk * 8;
  }

  const int nk_all = EPI == HM_EPI_F32 ? g.K / BK / g.ksplit : g.K / BK;
  const int kt0 = split * nk_all;                       // split-K: this workgroup's K range starts kt0 tiles in
  const int nk = nk_all / WK;                           // steps of the loop (host: nk_all % WK == 0); group kg takes tile kt * WK + kg
  // Round 4: the LEAN implicit-GEMM loader.  The general loader below derives (tap, ky, kx, ci), the bounds test and a 64-bit
  // source address per lane, piece and K-step: ~100 vector / scalar instructions per wave and step (10 v_mul_lo_u32, 8
  // v_mad_u64_u32, a division, five divergent branches) beside 32 MFMAs -- with two waves per SIMD the vector issue port, not
  // the MFMA pipe, paced the convolutions (550-620 TFLOP/s on layers whose tiles run at 1000 as plain GEMMs).  With
  // Cin % 64 == 0 (every layer but the first two, which have direct kernels) a 64-deep K-step lies inside ONE tap, so the K
  // position is wave-uniform: (tap, ky, kx, ci) advance in scalar registers from one stage() call to the next (the calls walk the
  // K tiles in order, WK apart), the pixel part of the address is a per-lane pointer formed once (pcen), and "is this tap's pixel
  // inside the image" is one bit of a per-lane mask formed once (vmask): per piece one 64-bit add, a bit test and a select.
  const bool lean = CONV && g.cin_log2 >= 6 && g.taps <= 32 && !g.general_loader;
  const elem* pcen[XI];
  unsigned vmask[XI];
  int s_ci = 0, s_kx = 0, s_ky = 0, s_tap = 0;
  if (CONV) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      pcen[i] = xsrc[i] + ((ptrdiff_t)pix_y[i] * g.Wd + pix_x[i]) * g.ldx + chunk * 8;     // (only dereferenced under a valid tap)
      unsigned cm = 0, m = 0;                              // in-image test is separable: column bits, then one copy per valid row
      for (int kx = 0; kx < g.ksz; ++kx) cm |= ((unsigned)(pix_x[i] + kx) < (unsigned)g.Wd) ? (1u << kx) : 0u;
      for (int ky = 0; ky < g.ksz; ++ky) m |= ((unsigned)(pix_y[i] + ky) < (unsigned)g.H) ? (cm << (ky * g.ksz)) : 0u;
      vmask[i] = m;
    }
    const int k_first = (kt0 + (WK > 1 ? kg : 0)) * BK;
    s_tap = k_first >> g.cin_log2;
    s_ci = k_first & ((1 << g.cin_log2) - 1);
    s_ky = s_tap / g.ksz;
    s_kx = s_tap - s_ky * g.ksz;
  }
  auto stage = [&](int buf, int kt) {
    if (SCHED == 93) return;                            // ablation: no global loads at all
    char* lx = smem + (kg * STAGES + buf) * STAGE_BYTES + wave * XI * 1024;
    char* lw = smem + (kg * STAGES + buf) * STAGE_BYTES + XTILE_BYTES + wave * WI * 1024;
    if (WK > 1) kt = kt * WK + kg;
    if (CONV && lean) {
      const ptrdiff_t delta = ((ptrdiff_t)s_ky * g.Wd + s_kx) * g.ldx + s_ci;        // wave-uniform
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const elem* src = ((vmask[i] >> s_tap) & 1u) ? pcen[i] + delta : (const elem*)g.zeros;
        glds16(src, lx + i * 1024);
      }
      s_ci += BK * WK;                                   // the next call stages the K tile WK further on
      while (s_ci >= (1 << g.cin_log2)) {
        s_ci -= 1 << g.cin_log2;
        ++s_tap;
        if (++s_kx == g.ksz) { s_kx = 0; ++s_ky; }
      }
    } else if (CONV) {
      const int k = (kt0 + kt) * BK + chunk * 8;
      const int tap = k >> g.cin_log2, ci = k & ((1 << g.cin_log2) - 1);
      const int ky = tap / g.ksz, kx = tap - ky * g.ksz;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int iy = pix_y[i] + ky, ix = pix_x[i] + kx;
        const bool ok = tap < g.taps && iy >= 0 && iy < g.H && ix >= 0 && ix < g.Wd;
        const elem* src = ok ? xsrc[i] + ((size_t)iy * g.Wd + ix) * g.ldx + ci : (const elem*)g.zeros;
        glds16(src, lx + i * 1024);
      }
    } else {
#pragma unroll
      for (int i = 0; i < XI; ++i) glds16(xsrc[i] + (size_t)kt * BK, lx + i * 1024);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) glds16(wsrc[i] + (size_t)kt * BK, lw + i * 1024);
  };

  f32x4_t acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment reads: row = tile_row + (lane&15); 16-B chunk = ks*4 + (lane>>4), swizzled with the staging key
  const int frow = lane & 15, fsw = BK == 64 ? (lane & 7) : ((lane >> 2) & 3), fch = lane >> 4;
  // one K sub-step (32 deep): fragment reads + NI x MI MFMAs
  auto substep = [&](int buf, int ks, f32x4_t (&acc)[NI][MI]) {
    if (SCHED == 92) return;                            // ablation: LDS-DMA fill only
    const char* lx = smem + (kg * STAGES + buf) * STAGE_BYTES;
    const char* lw = lx + XTILE_BYTES;
    const int coff = ((ks * 4 + fch) ^ fsw) * 16;
    vec8 wf[NI], xf[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) xf[i] = *(const vec8*)(lx + (wr * 16 * MI + i * 16 + frow) * ROWB + coff);
    __builtin_amdgcn_sched_barrier(0);                 // X fragments first (round 4): the first MFMAs need W0 and all of X
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[i] = *(const vec8*)(lw + (wc * 16 * NI + i * 16 + frow) * ROWB + coff);
    if (SCHED >= 1) __builtin_amdgcn_s_setprio(1);
#ifdef HM_ABLATIONS
    if constexpr (SCHED == 98) {
      // timing ablation (WRONG results): the same fragment reads feeding HALF as many, twice as long MFMAs (32x32x16 instead of
      // 16x16x32): is the K loop bound by vector-instruction ISSUE (an MFMA holds the issue port 8 cycles whatever its shape)?
      typedef __attribute__((ext_vector_type(16))) float f32x16_t;
      static_assert(NI == 8 && MI == 4, "written for the 64 x 128 wave tile");
      f32x16_t* a32 = (f32x16_t*)&acc[0][0];                // 8 accumulators of 16 registers over the same 128 registers
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            if constexpr (std::is_same<T, TF16>::value) a32[nb * 2 + mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[nb * 2 + kk], xf[mb * 2 + kk], a32[nb * 2 + mb], 0, 0, 0);
            else a32[nb * 2 + mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb * 2 + kk], xf[mb * 2 + kk], a32[nb * 2 + mb], 0, 0, 0);
          }
      __builtin_amdgcn_s_setprio(0);
      return;
    }
#endif
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = T::mfma(wf[ni], xf[mi], acc[ni][mi]);
    if (SCHED >= 1) __builtin_amdgcn_s_setprio(0);
  };

  // ---- K loop: ring of STAGES buffers, STAGES-1 tiles of LDS-DMA in flight across the barrier
  if (EPI == HM_EPI_F32) {                              // (the implicit-GEMM X loader takes kt0 itself: its address is per tap)
    if (!CONV) {
#pragma unroll
      for (int i = 0; i < XI; ++i) xsrc[i] += (size_t)kt0 * BK;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) wsrc[i] += (size_t)kt0 * BK;
  }
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) stage(s, s);
  // deferred-LN consumer: (mean, rstd) of this tile's rows from the producer's 64-column partials, kept in LDS past
  // the ring; the K loop's barriers order these writes before the epilogue reads
  constexpr int SKB = NWG > 8 ? 8 : 16;                // sixteen waves: 8 KB of epilogue staging each
  constexpr int RING_BYTES = WK * STAGES * STAGE_BYTES, EPI_BYTES = NWG * epi_stage_bytes(MI, NI, SKB);
  static_assert(WK == 1 || EPI_BYTES + (WK - 1) * NWG * MI * NI * 1024 <= RING_BYTES, "epilogue staging + the K groups' accumulators fit in the rings");
  float2* rowstat = (float2*)(smem + (RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES));
  float* colvec = (float*)(rowstat + BM);              // [2][BN]: bias (or zeros) | ln_gamma or ln_colsum
  for (int c = tid; c < BN; c += 64 * NW) {
    const int n = min(n0 + c, g.N - 1);
    colvec[c] = (g.bias && split == 0) ? g.bias[n] : 0.f;
    if constexpr (EPI == HM_EPI_RESID_LN) colvec[BN + c] = g.ln_gamma[n];
    if constexpr (epi_ln_in(EPI)) colvec[BN + c] = g.ln_colsum[n];
  }
  if constexpr (epi_ln_in(EPI)) {
    for (int r = tid; r < BM; r += 64 * NW) rowstat[r] = ((const float2*)g.ln_stats)[min(m0 + r, g.M - 1)];   // (mean, rstd) per row
  }
  int rd = 0, wrb = STAGES - 1;                        // ring slots: read tile kt, write tile kt+STAGES-1
  auto kstep = [&](int kt, f32x4_t (&acc)[NI][MI]) {
    // tile kt must have landed: in steady state the STAGES-2 newer tiles may still be in flight
    if (SCHED != 94 && SCHED != 95) {                  // (ablations 94 / 95: nobody waits for the copies; 95: no barrier either)
      if (kt + STAGES - 2 < nk) wait_vmcnt<LOADS * (STAGES - 2)>();
      else wait_vmcnt<0>();
    }
    if (SCHED != 95) __builtin_amdgcn_s_barrier();     // everyone's tile kt landed; everyone is done reading slot wrb
    const bool more = kt + STAGES - 1 < nk;
    if (SCHED <= 1) {                                  // loads first, then the whole tile
      if (more) stage(wrb, kt + STAGES - 1);
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) substep(rd, ks, acc);
    } else {                                           // first sub-step, then the loads, then the rest
      substep(rd, 0, acc);
      if (more) stage(wrb, kt + STAGES - 1);
#pragma unroll
      for (int ks = 1; ks < BK / 32; ++ks) substep(rd, ks, acc);
    }
    rd = rd + 1 == STAGES ? 0 : rd + 1;
    wrb = wrb + 1 == STAGES ? 0 : wrb + 1;
  };
  if constexpr (KSER) {
    // Round 4: the split-K summation order WITHOUT the split.  Split-K (conv_split_rule: a rule on one image, so that a frame
    // gets the same bytes alone and in a batch) gives every K range its own workgroup, an fp32 slab in HBM and a reduce kernel
    // that adds the slabs in order -- right for a few frames, where a layer has too few tiles to fill the chip, and 15-30 %
    // slower than no split at all once a pass carries 48 frames.  Here ONE workgroup walks the ranges one after the other with the
    // ring running through: a range accumulates from zero exactly as its split-K workgroup does (two sets for even / odd K tiles
    // under KDUAL, joined as that kernel joins them), and the running total takes the ranges in the reduce kernel's order
    // ((0 + p0) + p1) + ...; the ordinary epilogue then rounds acc + bias -> activation as that kernel does.  Same bytes, no
    // slabs, no second launch.
    f32x4_t tot[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) tot[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nkr = nk / g.kser;                        // (host: divisible, and even under KDUAL)
    int kt = 0;
    for (int r = 0; r < g.kser; ++r) {
      const int end = kt + nkr;
      if constexpr (KDUAL) {
        f32x4_t acc2[NI][MI];
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
          for (int b = 0; b < MI; ++b) acc2[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (; kt < end; kt += 2) {
          kstep(kt, acc);
          kstep(kt + 1, acc2);
        }
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
          for (int b = 0; b < MI; ++b) acc[a][b] += acc2[a][b];
      } else {
        for (; kt < end; ++kt) kstep(kt, acc);
      }
#pragma unroll
      for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) {
          tot[a][b] += acc[a][b];
          acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = tot[a][b];
  } else if constexpr (KDUAL) {                        // (host: nk even)
    f32x4_t acc2[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc2[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; kt += 2) {
      kstep(kt, acc);
      kstep(kt + 1, acc2);
    }
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] += acc2[a][b];
  } else {
    for (int kt = 0; kt < nk; ++kt) kstep(kt, acc);
  }

  __builtin_amdgcn_s_barrier();                        // every wave is done reading the ring: reuse it for the epilogue
  if constexpr (WK > 1) {
    // accumulators of the K groups 1.. -> group 0, through LDS behind the epilogue staging: [kg - 1][wave][ni][mi][lane] x 16 B
    f32x4_t* red = (f32x4_t*)(smem + EPI_BYTES);
    if (kg > 0) {
#pragma unroll
      for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) red[((((kg - 1) * NWG + wave) * NI + a) * MI + b) * 64 + lane] = acc[a][b];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int q = 1; q < WK; ++q)
#pragma unroll
      for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] += red[((((q - 1) * NWG + wave) * NI + a) * MI + b) * 64 + lane];
  }
  if (SCHED == 96 || SCHED == 98) {                     // ablation: no epilogue (keep the accumulators alive)
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) asm volatile("" :: "v"(acc[a][b]));
    return;
  }
  epilogue<T, EPI, MI, NI, true, SKB>(g, acc, m0 + wr * 16 * MI, n0 + wc * 16 * NI, lane, smem + wave * epi_stage_bytes(MI, NI, SKB),
                           rowstat + wr * 16 * MI, colvec + wc * 16 * NI, colvec + BN + wc * 16 * NI, split);
}

// ---------------------------------------------------------------------------------------------------------------
// hm_gemm_fp8: the same 256x256 tile, 8 waves (4x2, each 64x128), 2-stage LDS-DMA ring, on the block-scaled fp8 MFMA
// v_mfma_scale_f32_16x16x128_f8f6f4.  A K-step is 128 e4m3 bytes per row -- the byte geometry (128-byte rows, 16-byte
// chunk XOR swizzle, 64 KB per stage) is that of the 16-bit kernel at BK = 64, but a stage now carries twice the K
// and its 32 MFMAs per wave (2x the cycles each) four times the flops of a 16-bit sub-step.
// Operand lane map (probed with exact data, tools/probes/): lane (r = l & 15, g = l >> 4) holds, for row r, the 16 bytes
// k = 16g .. 16g+15 in VGPRs 0-3 and k = 64+16g .. 64+16g+15 in VGPRs 4-7, i.e. 16-byte chunks g and 4+g of the
// 128-byte row; the E8M0 scale byte supplied by lane (r, g) scales k = 32g .. 32g+31 of row r.  MFMA "A" = W rows
// (scale 1.0: W carries one f32 scale per output channel, applied to the accumulators), "B" = X rows with their
// MXFP8 block scales, which travel with the stage: [4 k-blocks][256 rows] bytes = one more 1-KiB LDS-DMA per K-step.
typedef __attribute__((ext_vector_type(8))) int v8i_t;
typedef __attribute__((ext_vector_type(4))) int v4i_t;

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_fp8_kernel(const KArgs g) {
  constexpr int WM = 4, WN = 2, MI = 4, NI = 8, NW = 8, BM = 256, BN = 256, BKB = 128;
  constexpr int TILE_BYTES = 256 * BKB, SCALE_BYTES = 4 * BM, STAGE_BYTES = 2 * TILE_BYTES + SCALE_BYTES;
  constexpr int XI = BM / NW / 8, WI = BN / NW / 8;              // 1-KiB DMA pieces (8 rows) per wave and operand: 4 + 4
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wave / WN, wc = wave % WN;
  const char* X = (const char*)g.X;
  const char* W = (const char*)g.W;

  const int srow = lane >> 3, chunk = (lane & 7) ^ (srow & 7);
  const char* xsrc[XI];
  const char* wsrc[WI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    int gm = m0 + wave * XI * 8 + i * 8 + srow;
    gm = gm < g.M ? gm : g.M - 1;
    xsrc[i] = X + (size_t)gm * g.ldx + chunk * 16;
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    int gn = n0 + wave * WI * 8 + i * 8 + srow;
    gn = gn < g.N ? gn : g.N - 1;
    wsrc[i] = W + (size_t)gn * g.ldw + chunk * 16;
  }
  // scales of a K-step: lane l copies the 16 row-scales [m0 + 16*(l&15), +16) of k-block l>>4 (rows past M: clamped)
  int srow0 = m0 + 16 * (lane & 15);
  srow0 = srow0 + 16 <= g.M ? srow0 : g.M - 16;
  const unsigned char* ssrc = g.xs + (size_t)(lane >> 4) * g.M + srow0;

  auto stage = [&](int buf, int kt) {
    char* lx = smem + buf * STAGE_BYTES + wave * XI * 1024;
    char* lw = smem + buf * STAGE_BYTES + TILE_BYTES + wave * WI * 1024;
#pragma unroll
    for (int i = 0; i < XI; ++i) glds16(xsrc[i] + (size_t)kt * BKB, lx + i * 1024);
#pragma unroll
    for (int i = 0; i < WI; ++i) glds16(wsrc[i] + (size_t)kt * BKB, lw + i * 1024);
    if (wave == 0) glds16(ssrc + (size_t)kt * 4 * g.M, smem + buf * STAGE_BYTES + 2 * TILE_BYTES);
  };

  f32x4_t acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fg = lane >> 4, fsw = lane & 7;
  const int c_lo = ((fg ^ fsw) * 16), c_hi = (((4 + fg) ^ fsw) * 16);
  auto kstep = [&](int buf) {
    const char* lx = smem + buf * STAGE_BYTES;
    const char* lw = lx + TILE_BYTES;
    const unsigned char* ls = (const unsigned char*)(lx + 2 * TILE_BYTES) + fg * BM + wr * 64 + frow;
    v8i_t xf[MI];
    int xsc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const char* r = lx + (wr * 64 + i * 16 + frow) * BKB;
      const v4i_t lo = *(const v4i_t*)(r + c_lo), hi = *(const v4i_t*)(r + c_hi);
      xf[i] = v8i_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      xsc[i] = ls[i * 16];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {                      // two halves of the wave's 128 columns: 4 W fragments live at a time
      v8i_t wf[NI / 2];
#pragma unroll
      for (int i = 0; i < NI / 2; ++i) {
        const char* r = lw + (wc * 128 + (h * 4 + i) * 16 + frow) * BKB;
        const v4i_t lo = *(const v4i_t*)(r + c_lo), hi = *(const v4i_t*)(r + c_hi);
        wf[i] = v8i_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < NI / 2; ++i)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[h * 4 + i][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[i], xf[mi], acc[h * 4 + i][mi], 0, 0, 0, 127, 0, xsc[mi]);
      __builtin_amdgcn_s_setprio(0);
    }
  };

  const int nk = g.K / BKB;
  stage(0, 0);
  constexpr int RING_BYTES = 2 * STAGE_BYTES;
  float2* rowstat = (float2*)(smem + RING_BYTES);                 // unused here, keeps the epilogue's LDS map
  float* colvec = (float*)(rowstat + BM);
  for (int c = tid; c < BN; c += 512) {
    const int n = min(n0 + c, g.N - 1);
    colvec[c] = g.bias ? g.bias[n] : 0.f;
    colvec[BN + c] = g.wscale[n];
  }
  int rd = 0;
  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) stage(rd ^ 1, kt + 1);
    kstep(rd);
    rd ^= 1;
  }
  __builtin_amdgcn_s_barrier();
  // per-output-channel weight scale: lane's 4 accumulator registers of tile ni are columns ni*16 + 4*(lane>>4) .. +3
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const f32x4_t ws = *(const f32x4_t*)(colvec + BN + wc * 128 + ni * 16 + 4 * fg);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] *= ws;
  }
  __builtin_amdgcn_s_barrier();                                   // (the ring is reused as epilogue staging)
  epilogue<TBf16, EPI, MI, NI, false>(g, acc, m0 + wr * 64, n0 + wc * 128, lane, smem + wave * epi_stage_bytes(MI, NI),
                               rowstat + wr * 64, colvec + wc * 128, colvec + BN + wc * 128, 0);
}

template <int EPI>
int launch_fp8(const KArgs& g, hipStream_t s) {
  constexpr int LDS = 2 * (2 * 256 * 128 + 4 * 256) + 256 * 8 + 256 * 8;
  static_assert(LDS >= 8 * epi_stage_bytes(4, 8), "epilogue staging fits in the ring");
  auto kern = gemm_fp8_kernel<EPI>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "hm_gemm_fp8: cannot raise the dynamic LDS limit")) return rc;
  const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), LDS, s, g);
  return hm_check_launch("hm_gemm_fp8");
}

// ---------------------------------------------------------------------------------------------------------------
// Deep-prefetch variant of the 256x256 tile (plain GEMM, 16-bit operands).  With a 2-stage ring the LDS-DMA of one
// K-step (64 KB) is all that is in flight, and the K loop runs at the latency of that burst (~1.6 us per step against
// 0.95 us of MFMA work; a 256x128 tile with 3 stages reaches the same FLOP rate with 1.5x the bytes per flop: the fill
// is bound by bytes in flight, not by a byte rate).  Two full stages are 128 KB of the 160 KB LDS; the remaining 32 KB
// are exactly one more X tile, so here X runs through a ring of THREE slots and is fetched two K-steps ahead (W: two
// slots, one step ahead): 96 KB in flight.  That uses every byte of LDS, so the per-column epilogue vectors wait in a
// register per thread and are written to LDS after the K loop.  All copies are issued from inline asm (saddr form:
// uniform base + 32-bit lane offset) and vmcnt is counted by hand: at the top of step t all but the newest 4 copies
// (X of step t+1, issued after W(t) in step t-1) must have landed.
// saddr forms: address = uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset
// (Round 4: M0 is overwritten, not saved and restored -- hipcc initialises M0 in front of each of its own uses and never reads it
//  back, and reading M0 right behind an LDS-DMA instruction is one more scalar dependency per piece.  Hot loops go further and
//  pass the LDS destination as a scalar byte offset, glds16_lean_s below: interleaved A/B of the persistent GEMM, -1.4 %.)
__device__ __forceinline__ void glds16_hidden_s(const char* base, unsigned off, void* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(base), "s"(dst) : "memory");
}

// The same copy with the LDS destination given as a byte offset already in a scalar register, and M0 simply overwritten: no
// generic -> LDS pointer cast (hipcc guards that with a 64-bit null test per piece), no save / restore of M0 (the compiler
// re-initialises M0 in front of each of its own uses and never reads it back) -- 3 scalar instructions per piece instead of 9.
__device__ __forceinline__ void glds16_lean_s(const char* base, unsigned off, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(base), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void glds4_hidden_s(const char* base, unsigned off, void* lds_wave_base) {   // 4 bytes per lane
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(off), "s"(base), "s"(dst) : "memory");
}

template <class T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_x3_kernel(const KArgs g) {
  constexpr int WM = 4, WN = 2, MI = 4, NI = 8, NW = 8, BM = 256, BN = 256, BK = 64, ROWB = 128;
  constexpr int TILE_BYTES = 256 * ROWB;                               // 32 KB
  constexpr int XRING = 0, WRING = 3 * TILE_BYTES;                      // X slots 0..2 | W slots 0..1  (= 160 KB)
  constexpr int XI = 4, WI = 4;                      // 1-KiB pieces (8 rows x 128 B) per wave and operand per K-step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using vec8 = typename T::vec8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wave / WN, wc = wave % WN;
  const char* X = (const char*)g.X;
  const char* W = (const char*)g.W;

  const int srow = lane >> 3, swz = (lane & 7) ^ (srow & 7);
  unsigned xoff[XI], woff[WI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    int gm = m0 + wave * XI * 8 + i * 8 + srow;
    gm = gm < g.M ? gm : g.M - 1;
    xoff[i] = (unsigned)gm * (unsigned)(g.ldx * 2) + swz * 16;
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    int gn = n0 + wave * WI * 8 + i * 8 + srow;
    gn = gn < g.N ? gn : g.N - 1;
    woff[i] = (unsigned)gn * (unsigned)(g.ldw * 2) + swz * 16;
  }
  auto dma_x = [&](int slot, int kt) {
    const char* base = X + (size_t)kt * ROWB;
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + XRING + slot * TILE_BYTES + wave * XI * 1024);
#pragma unroll
    for (int i = 0; i < XI; ++i) glds16_lean_s(base, xoff[i], l + i * 1024);
  };
  auto dma_w = [&](int slot, int kt) {
    const char* base = W + (size_t)kt * ROWB;
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + WRING + slot * TILE_BYTES + wave * WI * 1024);
#pragma unroll
    for (int i = 0; i < WI; ++i) glds16_lean_s(base, woff[i], l + i * 1024);
  };

  f32x4_t acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fsw = lane & 7, fch = lane >> 4;
  auto substep = [&](int xslot, int wslot, int ks) {
    const char* lx = smem + XRING + xslot * TILE_BYTES;
    const char* lw = smem + WRING + wslot * TILE_BYTES;
    const int coff = ((ks * 4 + fch) ^ fsw) * 16;
    vec8 wf[NI], xf[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) xf[i] = *(const vec8*)(lx + (wr * 16 * MI + i * 16 + frow) * ROWB + coff);
    __builtin_amdgcn_sched_barrier(0);                 // X fragments first (round 4): the first MFMAs need W0 and all of X
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[i] = *(const vec8*)(lw + (wc * 16 * NI + i * 16 + frow) * ROWB + coff);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = T::mfma(wf[ni], xf[mi], acc[ni][mi]);
    __builtin_amdgcn_s_setprio(0);
  };

  const int nk = g.K / BK;
  // prologue: X(0), W(0), X(1) in flight; this thread's column of the bias waits in a register for the epilogue
  dma_x(0, 0);
  dma_w(0, 0);
  if (nk > 1) dma_x(1, 1);
  float bias_reg = 0.f;
  if (g.bias) bias_reg = g.bias[min(n0 + (tid & (BN - 1)), g.N - 1)];   // every wave loads: one more outstanding operation each
                                                                        // (a plain load: hipcc waits for it at its first use, after the loop)
  // step 0 needs X(0) and W(0); X(1) (4 copies) and the bias load (the newest 5 operations) may stay in flight -- the bias
  // is first read after the K loop, behind the vmcnt(0) of the last step
  if (nk > 1 && g.bias) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int xs = 0;                                          // X slot of step kt = kt % 3
  for (int kt = 0; kt < nk; ++kt) {
    // step kt-1 issued W(kt) then X(kt+1): all but the newest 4 copies have landed = W(kt) and the older X(kt)
    if (kt >= 1) {
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                      // step kt complete in LDS; everyone is done reading step kt-1's slots
    const int xs2 = xs == 0 ? 2 : xs - 1;              // (kt + 2) % 3 = the slot step kt-1 read
    substep(xs, kt & 1, 0);
    if (kt + 1 < nk) dma_w((kt + 1) & 1, kt + 1);      // W first, then X: the counted wait above relies on this order
    if (kt + 2 < nk) dma_x(xs2, kt + 2);
    substep(xs, kt & 1, 1);
    xs = xs == 2 ? 0 : xs + 1;
  }

  __builtin_amdgcn_s_barrier();                        // the ring is free: per-column vectors, then epilogue staging
  constexpr int EPI_BYTES = NW * epi_stage_bytes(MI, NI);
  float2* rowstat = (float2*)(smem + EPI_BYTES);       // unused by these epilogues
  float* colvec = (float*)(rowstat + BM);
  if (tid < BN) colvec[tid] = bias_reg;
  __builtin_amdgcn_s_barrier();
  epilogue<T, EPI, MI, NI>(g, acc, m0 + wr * 16 * MI, n0 + wc * 16 * NI, lane, smem + wave * epi_stage_bytes(MI, NI),
                           rowstat + wr * 16 * MI, colvec + wc * 16 * NI, colvec + BN + wc * 16 * NI, 0);
}

template <class T, int EPI>
int launch_rs(const KArgs& g, hipStream_t s) {
  constexpr int LDS = 5 * 256 * 128;                                    // 163,840 B: all of it
  static_assert(LDS >= 8 * epi_stage_bytes(4, 8) + 256 * 8 + 2 * 256 * 4, "epilogue staging + vectors fit");
  auto kern = gemm_x3_kernel<T, EPI>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "gemm: cannot raise the dynamic LDS limit")) return rc;
  const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), LDS, s, g);
  return hm_check_launch("hm_gemm");
}

// ---------------------------------------------------------------------------------------------------------------
// gemm_x3_kernel with the fp32 residual fetched INSIDE the K loop (round 3; proj and fc2: 65 of the 130 launches of a step).
// C = acc + bias + resid moves 2 x 4 bytes per output element through HBM.  Fetched in the epilogue (round 2), the read is
// paid after the K loop, by every CU at the same moment, while the MFMA pipes idle; during the K loop HBM idles instead (the
// operands come from L2 / the Infinity Cache).  Here every wave requests its residual tile in the ACCUMULATOR layout -- piece
// (ni, mi) = 16 rows x 64 B, one global_load_dwordx4 per lane; two pieces of neighbouring ni form whole 128-byte lines --
// two pieces per K-step during steps 0..15, and adds them into the accumulators two steps later.  The epilogue is then the
// plain fp32 store epilogue (bias from LDS, no loads).
//   vmcnt.  Loads and LDS-DMA copies retire in issue order.  Step t issues W(t+1) [4], X(t+2) [4], R(t) [2].  The top of
//   step t+1 needs W(t+1): everything but the newest 6 (X(t+2), R(t)) -> vmcnt(6), R(t) may stay in flight.  The top of step
//   t+2 needs W(t+2), issued after R(t): that wait retires R(t) as a side effect, so R(t) has between one and two steps
//   (1.5-3 us) to arrive before it would stall anything, and is consumed right behind that wait.  The wait statement takes
//   the two registers as read-write operands: no use or copy of them can be scheduled above it
//   (tests/test_isa_audit.py checks the emitted ISA).
//   The 18 steps that issue or consume residual pieces are unrolled (the piece index selects accumulator registers); steps
//   18.. run in a loop.  Requires whole tiles, K >= 20 x 64, resid_mod == 0, 32-bit residual offsets (launcher: rin_ok).
template <class T>
__global__ __launch_bounds__(512, 2) void gemm_x3r_kernel(const KArgs g) {
  constexpr int WN = 2, MI = 4, NI = 8, NW = 8, BN = 256, BK = 64, ROWB = 128;
  constexpr int TILE_BYTES = 256 * ROWB;
  constexpr int XRING = 0, WRING = 3 * TILE_BYTES;
  constexpr int XI = 4, WI = 4;
  constexpr int RSTEPS = 16;                            // steps that issue residual pieces: 2 pieces each, 32 in all
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using vec8 = typename T::vec8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = g.N >> 8, tiles_m = g.M >> 8;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
  const int m0 = tm << 8, n0 = tn << 8;
  const int wr = wave / WN, wc = wave % WN;
  const char* X = (const char*)g.X;
  const char* W = (const char*)g.W;

  const int srow = lane >> 3, swz = (lane & 7) ^ (srow & 7);
  unsigned xoff[XI], woff[WI];
#pragma unroll
  for (int i = 0; i < XI; ++i) xoff[i] = (unsigned)(m0 + wave * XI * 8 + i * 8 + srow) * (unsigned)(g.ldx * 2) + swz * 16;
#pragma unroll
  for (int i = 0; i < WI; ++i) woff[i] = (unsigned)(n0 + wave * WI * 8 + i * 8 + srow) * (unsigned)(g.ldw * 2) + swz * 16;
  auto dma_x = [&](int slot, int kt) {
    const char* base = X + (size_t)kt * ROWB;
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + XRING + slot * TILE_BYTES + wave * XI * 1024);
#pragma unroll
    for (int i = 0; i < XI; ++i) glds16_lean_s(base, xoff[i], l + i * 1024);
  };
  auto dma_w = [&](int slot, int kt) {
    const char* base = W + (size_t)kt * ROWB;
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + WRING + slot * TILE_BYTES + wave * WI * 1024);
#pragma unroll
    for (int i = 0; i < WI; ++i) glds16_lean_s(base, woff[i], l + i * 1024);
  };
  // residual piece (ni, mi) of this wave: rows mb + 16 mi + (lane & 15), columns nb + 16 ni + 4 (lane >> 4) .. + 3
  const int mb = m0 + wr * 16 * MI, nb = n0 + wc * 16 * NI;
  unsigned roff[MI];                                    // byte offset of the lane's row / column quad for ni = 0
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) roff[mi] = (unsigned)(mb + mi * 16 + (lane & 15)) * (unsigned)(g.ldr * 4) + (unsigned)(nb + 4 * (lane >> 4)) * 4;
  const char* rbase = (const char*)g.resid;

  f32x4_t acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fsw = lane & 7, fch = lane >> 4;
  auto substep = [&](int xslot, int wslot, int ks) {
    const char* lx = smem + XRING + xslot * TILE_BYTES;
    const char* lw = smem + WRING + wslot * TILE_BYTES;
    const int coff = ((ks * 4 + fch) ^ fsw) * 16;
    vec8 wf[NI], xf[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) xf[i] = *(const vec8*)(lx + (wr * 16 * MI + i * 16 + frow) * ROWB + coff);
    __builtin_amdgcn_sched_barrier(0);                 // X fragments first (round 4): the first MFMAs need W0 and all of X
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[i] = *(const vec8*)(lw + (wc * 16 * NI + i * 16 + frow) * ROWB + coff);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = T::mfma(wf[ni], xf[mi], acc[ni][mi]);
    __builtin_amdgcn_s_setprio(0);
  };

  const int nk = g.K / BK;                              // launcher: nk >= RSTEPS + 4
  dma_x(0, 0);
  dma_w(0, 0);
  dma_x(1, 1);
  float bias_reg = 0.f;
  if (g.bias) bias_reg = g.bias[n0 + (tid & (BN - 1))];  // a plain load: hipcc waits for it at its first use, after the loop
  if (g.bias) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");     // X(0), W(0) landed; X(1) and the bias may stay in flight
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");

  f32x4_t rq[3][2];                                     // residual pieces in flight: issued in step t, consumed in step t + 2
  // One unrolled step.  T_ = step index (compile time): issues R(T_) when T_ < RSTEPS, consumes R(T_ - 2) when 2 <= T_ < RSTEPS + 2.
  auto ustep = [&](auto TC) {
    constexpr int t = decltype(TC)::value;
    constexpr bool CONSUME = t >= 2 && t < RSTEPS + 2;
    constexpr bool PREV_ISSUED = t >= 1 && t - 1 < RSTEPS;           // step t - 1 put 2 residual loads behind its copies
    if constexpr (t >= 1) {
      if constexpr (CONSUME) {
        f32x4_t (&r)[2] = rq[(t - 2) % 3];
        if constexpr (PREV_ISSUED) asm volatile("s_waitcnt vmcnt(6)" : "+v"(r[0]), "+v"(r[1]) :: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(r[0]), "+v"(r[1]) :: "memory");
        constexpr int p = t - 2, ni0 = 2 * (p / 4), mi = p % 4;
        acc[ni0][mi] += r[0];                           // (before the barrier: this wave would be waiting for the others anyway)
        acc[ni0 + 1][mi] += r[1];
      } else {
        if constexpr (PREV_ISSUED) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
    }
    __builtin_amdgcn_s_barrier();                      // step t complete in LDS; everyone is done reading step t-1's slots
    constexpr int xs = t % 3, xs2 = (t + 2) % 3;
    substep(xs, t & 1, 0);
    dma_w((t + 1) & 1, t + 1);                         // W first, then X, then the residual pieces: the counted waits rely on this order
    dma_x(xs2, t + 2);
    if constexpr (t < RSTEPS) {
      constexpr int ni0 = 2 * (t / 4), mi = t % 4;
      f32x4_t (&r)[2] = rq[t % 3];
      asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:64"
                   : "=&v"(r[0]), "=&v"(r[1]) : "v"(roff[mi]), "s"(rbase + ni0 * 64) : "memory");
    }
    substep(xs, t & 1, 1);
  };
#define HM_USTEP(n) ustep(std::integral_constant<int, n>{})
  HM_USTEP(0); HM_USTEP(1); HM_USTEP(2); HM_USTEP(3); HM_USTEP(4); HM_USTEP(5); HM_USTEP(6); HM_USTEP(7); HM_USTEP(8);
  HM_USTEP(9); HM_USTEP(10); HM_USTEP(11); HM_USTEP(12); HM_USTEP(13); HM_USTEP(14); HM_USTEP(15); HM_USTEP(16); HM_USTEP(17);
#undef HM_USTEP
  static_assert(RSTEPS + 2 == 18 && 18 % 3 == 0, "the loop below starts at step 18 with X slot 0");
  int xs = 0;
  for (int kt = RSTEPS + 2; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int xs2 = xs == 0 ? 2 : xs - 1;
    substep(xs, kt & 1, 0);
    if (kt + 1 < nk) dma_w((kt + 1) & 1, kt + 1);
    if (kt + 2 < nk) dma_x(xs2, kt + 2);
    substep(xs, kt & 1, 1);
    xs = xs == 2 ? 0 : xs + 1;
  }

  __builtin_amdgcn_s_barrier();                        // the ring is free: per-column vectors, then epilogue staging
  constexpr int EPI_BYTES = NW * epi_stage_bytes(MI, NI);
  float2* rowstat = (float2*)(smem + EPI_BYTES);       // unused by this epilogue
  float* colvec = (float*)(rowstat + 256);
  if (tid < BN) colvec[tid] = bias_reg;
  __builtin_amdgcn_s_barrier();
  epilogue<T, HM_EPI_F32, MI, NI>(g, acc, mb, nb, lane, smem + wave * epi_stage_bytes(MI, NI),
                                  rowstat + wr * 16 * MI, colvec + wc * 16 * NI, colvec + BN + wc * 16 * NI, 0);
}

// what gemm_x3r_kernel requires: whole 256 x 256 tiles, the 18 unrolled steps plus at least two more (so that every copy the
// unrolled steps issue exists), a plain residual (no row modulus), 32-bit residual offsets, 16-byte residual rows
bool rin_ok(const KArgs& g) {
  return g.M % 256 == 0 && g.N % 256 == 0 && g.K >= 20 * 64 && g.resid_mod == 0 && g.resid != nullptr && (g.ldr & 3) == 0 &&
         (size_t)g.M * g.ldr * 4 < (1ull << 32) && hm_option(HM_OPT_RESID_IN_EPILOGUE) == 0;
}

template <class T>
int launch_rin(const KArgs& g, hipStream_t s) {
  constexpr int LDS = 5 * 256 * 128;
  auto kern = gemm_x3r_kernel<T>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "gemm: cannot raise the dynamic LDS limit")) return rc;
  hipLaunchKernelGGL(kern, dim3((g.M >> 8) * (g.N >> 8)), dim3(512), LDS, s, g);
  return hm_check_launch("hm_gemm");
}

#ifdef HM_ABLATIONS   // experiments live in the tools-only build (python -m hamer_yolo_amd.build --ablations), not in the product
// ---------------------------------------------------------------------------------------------------------------
// EXPERIMENT (variant 28): 256x160 tile with BOTH operands two K-steps ahead.  The 256x256 K loop runs at the latency of
// its copies: W is requested one step before it is needed because a third 32 KB W slot does not fit beside three X slots
// (160 KB).  A 256x160 tile (waves 4x2, each 64x80 = 4x5 MFMA tiles) fits three slots of each operand (96 + 60 KB), so
// every copy has two full steps to land; it pays 27 % more operand bytes per flop.  Uses the generic epilogue.
template <class T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_d2_kernel(const KArgs g) {
  constexpr int WN = 2, MI = 4, NI = 5, NW = 8, BM = 256, BN = 160, ROWB = 128;
  constexpr int XT = BM * ROWB, WT = BN * ROWB;                         // 32 KB, 20 KB
  constexpr int XRING = 0, WRING = 3 * XT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using vec8 = typename T::vec8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wave / WN, wc = wave % WN;
  const char* X = (const char*)g.X;
  const char* W = (const char*)g.W;
  const int srow = lane >> 3, swz = (lane & 7) ^ (srow & 7);
  const int nwp = wave < 4 ? 3 : 2;                                     // W pieces of this wave: 20 pieces over 8 waves
  unsigned xoff[4], woff[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int gm = m0 + wave * 32 + i * 8 + srow;
    gm = gm < g.M ? gm : g.M - 1;
    xoff[i] = (unsigned)gm * (unsigned)(g.ldx * 2) + swz * 16;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int gn = n0 + (wave + 8 * i) * 8 + srow;                            // piece p = wave + 8 i
    gn = gn < g.N ? gn : g.N - 1;
    woff[i] = (unsigned)gn * (unsigned)(g.ldw * 2) + swz * 16;
  }
  float bias_reg = 0.f;
  if (g.bias) bias_reg = g.bias[min(n0 + (tid % BN), g.N - 1)];         // the oldest vector-memory operation: every wait below covers it
  auto dma = [&](int slot, int kt) {
    const char* xb = X + (size_t)kt * ROWB;
    const char* wb = W + (size_t)kt * ROWB;
    char* lx = smem + XRING + slot * XT + wave * 4 * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16_hidden_s(xb, xoff[i], lx + i * 1024);
    char* lw = smem + WRING + slot * WT;
    glds16_hidden_s(wb, woff[0], lw + wave * 1024);
    glds16_hidden_s(wb, woff[1], lw + (wave + 8) * 1024);
    if (wave < 4) glds16_hidden_s(wb, woff[2], lw + (wave + 16) * 1024);
  };
  f32x4_t acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fsw = lane & 7, fch = lane >> 4;
  auto substep = [&](int slot, int ks) {
    const char* lx = smem + XRING + slot * XT;
    const char* lw = smem + WRING + slot * WT;
    const int coff = ((ks * 4 + fch) ^ fsw) * 16;
    vec8 wf[NI], xf[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) xf[i] = *(const vec8*)(lx + (wr * 16 * MI + i * 16 + frow) * ROWB + coff);
    __builtin_amdgcn_sched_barrier(0);                 // X fragments first (round 4): the first MFMAs need W0 and all of X
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[i] = *(const vec8*)(lw + (wc * 16 * NI + i * 16 + frow) * ROWB + coff);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = T::mfma(wf[ni], xf[mi], acc[ni][mi]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto wait_step = [&](bool more) {                                     // all but the next step's copies (4 X + nwp W) have landed
    if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (nwp == 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  };
  const int nk = g.K / 64;
  dma(0, 0);
  if (nk > 1) dma(1, 1);
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    wait_step(kt + 1 < nk);
    __builtin_amdgcn_s_barrier();                      // step kt complete in LDS; everyone is done reading step kt-1's slot
    substep(slot, 0);
    if (kt + 2 < nk) dma(slot == 0 ? 2 : slot - 1, kt + 2);            // (kt + 2) % 3 = the slot step kt-1 read
    substep(slot, 1);
    slot = slot == 2 ? 0 : slot + 1;
  }
  __builtin_amdgcn_s_barrier();
  constexpr int EPI_BYTES = NW * epi_stage_bytes(MI, NI);
  float2* rowstat = (float2*)(smem + EPI_BYTES);       // unused by these epilogues
  float* colvec = (float*)(rowstat + BM);
  if (tid < BN) colvec[tid] = bias_reg;
  __builtin_amdgcn_s_barrier();
  epilogue<T, EPI, MI, NI>(g, acc, m0 + wr * 16 * MI, n0 + wc * 16 * NI, lane, smem + wave * epi_stage_bytes(MI, NI),
                           rowstat + wr * 16 * MI, colvec + wc * 16 * NI, colvec + BN + wc * 16 * NI, 0);
}

template <class T, int EPI>
int launch_d2(const KArgs& g, hipStream_t s) {
  constexpr int LDS = 3 * 256 * 128 + 3 * 160 * 128;                    // 159,744 B
  auto kern = gemm_d2_kernel<T, EPI>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "gemm: cannot raise the dynamic LDS limit")) return rc;
  const int tiles = ((g.M + 255) / 256) * ((g.N + 159) / 160);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), LDS, s, g);
  return hm_check_launch("hm_gemm");
}

#endif  // HM_ABLATIONS (variant 28)

// ---------------------------------------------------------------------------------------------------------------
// Persistent form of gemm_x3_kernel for the 16-bit store epilogues (qkv, fc1, to_kv: 3-5 tiles per CU at B = 64).
// Launched tile by tile, every tile pays its own start-up in series with everything else on its CU -- address set-up,
// the first HBM/L2 round trip of its operands (nothing to overlap it with: one workgroup per CU), the wait for its
// result stores to be acknowledged before the wave may end, the dispatch of the next workgroup: ~11 us per 32 us K loop
// at K = 1280.  Here one workgroup per CU walks its tiles (the same XCD-contiguous order the dispatcher produced) and the
// LDS-DMA pipeline never stops: step t of the concatenated (tile, k-step) sequence issues W(t+1) and X(t+2) whether or
// not they belong to the next tile, so a tile's first operands land during the previous tile's last K-steps and
// epilogue, and its stores retire under the next tile's K loop.
//   LDS: X ring of 3 slots + W ring of 2 (all 160 KB), slot = step % 3 / step & 1 across tiles.  The epilogue stages
//   through the X slot the tile's LAST step read (4 KB per wave: 32-row half-strips) and keeps the bias in that step's W
//   slot; the copies in flight meanwhile target the other slots, and those two are first written again after the next
//   step's barrier.
//   vmcnt: loads, stores and LDS-DMA count together, in issue order -- waiting for an operation waits for every older
//   one.  So the epilogue contains NO load (a wait for it would drain the stores before it: the bias travels through two
//   registers requested at the top of the tile) and exactly 16 stores per wave (interior tiles only: the launcher
//   guarantees M, N multiples of 256), issued after the copies W(t+1), X(t+2) of the last step; the first step of the next
//   tile therefore waits vmcnt(16 + 4): everything but those stores and X(t+2).  An extra vector-memory operation
//   anywhere (a spill, the next tile's bias request) only makes a counted wait stricter, never looser.
// DMAW (experiments library, variant 34): waves 0..3 issue ALL copies (their own rows and those of the wave that shares their
// SIMD, wave + 4), waves 4..7 none -- does a SIMD whose second wave never stalls in copy issue keep its MFMA pipe fuller?
// STAMP (experiments library, variant 35; a diagnostic build: in the product no stamp executes): wave 0 of every workgroup
// records s_memtime (shader cycles) and s_memrealtime (100 MHz) around the kernel and around every tile's K loop into
// g.ln_stats as [workgroup][6] uint64: {cycles, realtime ticks} of the whole kernel, {cycles, ticks, K-steps} of its K loops.
// LEAN (round 4, the default; variant 36 of the experiments library compares it with the old form = LEAN false): the copies
// through glds16_lean_s (LDS destinations as scalar byte offsets, M0 not saved / restored): -1.4 % per ViT block.
// XFIRST (round 4, the default; variant 37 of the experiments library = the old order): the four X fragments of a sub-step are
// read before the eight W fragments, so the first MFMAs (W0 x X0..3) issue behind five reads instead of nine and the remaining
// W reads retire under them: -1.6 % per ViT block (profiles/r04_gemm_xfirst_ab.log).  Same MFMA order: same bytes.
// R1EARLY (variant 38): the twelve fragment reads of the step's SECOND sub-step are issued in front of the step's copies instead of
// behind them, so their latency passes under the copies' issue and the second run of MFMAs starts at once.
// TWEAK (experiments): 1 = read order X0 W0 X1 X2 X3 W1.. (first MFMA behind two reads), 2 = no s_setprio around the MFMA runs.
template <class T, int EPI, bool DIRECT = true, bool DMAW = false, bool STAMP = false, bool LEAN = true, bool XFIRST = true, bool R1EARLY = false, int TWEAK = 0>
__global__ __launch_bounds__(512, 2) void gemm_px_kernel(const KArgs g) {
  static_assert(EPI == HM_EPI_STORE || EPI == HM_EPI_GELU || EPI == HM_EPI_SILU, "16-bit store epilogues only");
  static_assert(!(DMAW && LEAN), "the loader-wave experiment was written against the round-3 copy issue");
  constexpr int WN = 2, MI = 4, NI = 8, ROWB = 128;
  constexpr int TILE_BYTES = 256 * ROWB;                               // 32 KB
  constexpr int XRING = 0, WRING = 3 * TILE_BYTES;
  constexpr int XI = 4, WI = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using vec8 = typename T::vec8;
  using elem = typename T::elem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = g.N >> 8, tiles_m = g.M >> 8, tiles = tiles_m * tiles_n;
  // this workgroup's tiles: XCD `blockIdx & 7` owns a contiguous run of tile ids (as xcd_remap deals them), its G/8
  // workgroups take them round robin -- the order the dispatcher gave the one-tile workgroups
  const int G = gridDim.x, xcd = blockIdx.x & 7, li = blockIdx.x >> 3, per = G >> 3;
  const int tq = tiles >> 3, tr = tiles & 7;
  const int run_lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, run_len = tq + (xcd < tr ? 1 : 0);
  const int my = li < run_len ? (run_len - li + per - 1) / per : 0;
  if (my == 0) return;
  unsigned long long st_c0 = 0, st_r0 = 0, st_kc = 0, st_kr = 0, st_kn = 0;
  if constexpr (STAMP) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
  const int nk = g.K / 64, S = my * nk;
  const int wr = wave / WN, wc = wave % WN;
  const char* X = (const char*)g.X;
  const char* W = (const char*)g.W;

  const int srow = lane >> 3, swz = (lane & 7) ^ (srow & 7);
  unsigned xoff[XI], woff[WI];                        // lane part of the source address; the tile origin rides in the scalar base
#pragma unroll
  for (int i = 0; i < XI; ++i) xoff[i] = (unsigned)(wave * XI * 8 + i * 8 + srow) * (unsigned)(g.ldx * 2) + swz * 16;
#pragma unroll
  for (int i = 0; i < WI; ++i) woff[i] = (unsigned)(wave * WI * 8 + i * 8 + srow) * (unsigned)(g.ldw * 2) + swz * 16;
  auto origin = [&](int ti, int& m0, int& n0) {
    int tm, tn;
    tile_coords(run_lo + li + ti * per, tiles_m, tiles_n, g.group_m, tm, tn);
    m0 = tm << 8; n0 = tn << 8;
  };
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma_x = [&](int slot, const char* base) {
    if constexpr (DMAW) { if (wave >= 4) return; }
    char* l = smem + XRING + slot * TILE_BYTES + wave * XI * 1024;
    if constexpr (LEAN) {
      const unsigned d = __builtin_amdgcn_readfirstlane(lds_base + XRING + slot * TILE_BYTES + wave * XI * 1024);
#pragma unroll
      for (int i = 0; i < XI; ++i) glds16_lean_s(base, xoff[i], d + i * 1024);
      return;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) glds16_hidden_s(base, xoff[i], l + i * 1024);
    if constexpr (DMAW) {                                 // the rows of wave + 4: 4 * XI * 8 rows further on, same lane offsets
      const char* b2 = base + (size_t)(4 * XI * 8) * g.ldx * 2;
#pragma unroll
      for (int i = 0; i < XI; ++i) glds16_hidden_s(b2, xoff[i], l + 4 * XI * 1024 + i * 1024);
    }
  };
  auto dma_w = [&](int slot, const char* base) {
    if constexpr (DMAW) { if (wave >= 4) return; }
    char* l = smem + WRING + slot * TILE_BYTES + wave * WI * 1024;
    if constexpr (LEAN) {
      const unsigned d = __builtin_amdgcn_readfirstlane(lds_base + WRING + slot * TILE_BYTES + wave * WI * 1024);
#pragma unroll
      for (int i = 0; i < WI; ++i) glds16_lean_s(base, woff[i], d + i * 1024);
      return;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) glds16_hidden_s(base, woff[i], l + i * 1024);
    if constexpr (DMAW) {
      const char* b2 = base + (size_t)(4 * WI * 8) * g.ldw * 2;
#pragma unroll
      for (int i = 0; i < WI; ++i) glds16_hidden_s(b2, woff[i], l + 4 * WI * 1024 + i * 1024);
    }
  };
  const bool dbl = DMAW && wave < 4;                       // this wave's copy groups are 8 pieces, not 4: the counted waits double
  // cursors over the concatenated step sequence: X runs two steps ahead of the MFMAs, W one
  int xti = 0, xkt = 0, wti = 0, wkt = 0, m0, n0;
  origin(0, m0, n0);
  const char* xbase = X + (size_t)m0 * g.ldx * 2;
  const char* wbase = W + (size_t)n0 * g.ldw * 2;
  auto next_x = [&]() {
    if (++xkt == nk) {
      xkt = 0;
      if (++xti < my) { int a, b; origin(xti, a, b); xbase = X + (size_t)a * g.ldx * 2; }
    }
  };
  auto next_w = [&]() {
    if (++wkt == nk) {
      wkt = 0;
      if (++wti < my) { int a, b; origin(wti, a, b); wbase = W + (size_t)b * g.ldw * 2; }
    }
  };

  f32x4_t acc[NI][MI];
  const int frow = lane & 15, fsw = lane & 7, fch = lane >> 4;
  auto substep = [&](int xslot, int wslot, int ks) {
    const char* lx = smem + XRING + xslot * TILE_BYTES;
    const char* lw = smem + WRING + wslot * TILE_BYTES;
    const int coff = ((ks * 4 + fch) ^ fsw) * 16;
    vec8 wf[NI], xf[MI];
    if constexpr (TWEAK == 1) {
      xf[0] = *(const vec8*)(lx + (wr * 16 * MI + frow) * ROWB + coff);
      __builtin_amdgcn_sched_barrier(0);
      wf[0] = *(const vec8*)(lw + (wc * 16 * NI + frow) * ROWB + coff);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 1; i < MI; ++i) xf[i] = *(const vec8*)(lx + (wr * 16 * MI + i * 16 + frow) * ROWB + coff);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 1; i < NI; ++i) wf[i] = *(const vec8*)(lw + (wc * 16 * NI + i * 16 + frow) * ROWB + coff);
    } else {
    if constexpr (XFIRST) {
#pragma unroll
      for (int i = 0; i < MI; ++i) xf[i] = *(const vec8*)(lx + (wr * 16 * MI + i * 16 + frow) * ROWB + coff);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[i] = *(const vec8*)(lw + (wc * 16 * NI + i * 16 + frow) * ROWB + coff);
    if constexpr (!XFIRST) {
#pragma unroll
      for (int i = 0; i < MI; ++i) xf[i] = *(const vec8*)(lx + (wr * 16 * MI + i * 16 + frow) * ROWB + coff);
    }
    }
    if constexpr (TWEAK != 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = T::mfma(wf[ni], xf[mi], acc[ni][mi]);
    if constexpr (TWEAK != 2) __builtin_amdgcn_s_setprio(0);
  };
  auto frag_reads = [&](int xslot, int wslot, int ks, vec8 (&wf)[NI], vec8 (&xf)[MI]) {
    const char* lx = smem + XRING + xslot * TILE_BYTES;
    const char* lw = smem + WRING + wslot * TILE_BYTES;
    const int coff = ((ks * 4 + fch) ^ fsw) * 16;
#pragma unroll
    for (int i = 0; i < MI; ++i) xf[i] = *(const vec8*)(lx + (wr * 16 * MI + i * 16 + frow) * ROWB + coff);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[i] = *(const vec8*)(lw + (wc * 16 * NI + i * 16 + frow) * ROWB + coff);
  };
  auto frag_mfmas = [&](const vec8 (&wf)[NI], const vec8 (&xf)[MI]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = T::mfma(wf[ni], xf[mi], acc[ni][mi]);
    __builtin_amdgcn_s_setprio(0);
  };

  // prologue: X(0), W(0), X(1) (nk >= 2)
  dma_x(0, xbase); next_x();
  dma_w(0, wbase); next_w();
  dma_x(1, xbase + (size_t)xkt * ROWB); next_x();
  if (dbl) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");

  int gs = 0, xs = 0;                                  // global step, its X slot (gs % 3); W slot = gs & 1
  for (int ti = 0; ti < my; ++ti) {
    if (ti > 0) origin(ti, m0, n0);
    // this wave's 128 bias values, two per lane, requested now and moved to LDS after the loop.  Issued from asm so that
    // hipcc does not track them: a tracked load makes it drain vmcnt to 0 where the value is first touched (the epilogue's
    // top, with the next tile's copies in flight).  Always issued (no bias: any valid address, the values are replaced by
    // zeros behind the fence), so the counted waits below do not depend on g.bias.  The wait of step kt == 1 retires them
    // (they are older than the copies that wait is for); the fence statement in front of the epilogue names the two
    // registers, so nothing can touch them earlier (tests/test_isa_audit.py checks the emitted ISA).
    float bias_lo, bias_hi;
    {
      const float* bsrc = g.bias ? g.bias + n0 + wc * 128 : (const float*)g.W;
      asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %3 offset:256"
                   : "=&v"(bias_lo), "=&v"(bias_hi) : "v"(lane * 4), "s"(bsrc) : "memory");
    }
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int xs_last = 0, ws_last = 0;
    unsigned long long st_a = 0, st_b = 0;
    if constexpr (STAMP) { st_a = __builtin_amdgcn_s_memtime(); st_b = __builtin_amdgcn_s_memrealtime(); }
    for (int kt = 0; kt < nk; ++kt, ++gs) {
      if (gs > 0) {
        if (kt == 0) { if (dbl) asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); }   // 16 epilogue stores + X(gs+1) + the 2 bias loads may stay in flight
        else if (gs + 1 < S) { if (dbl) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }   // X(gs+1)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                    // step gs complete in LDS; everyone is done with step gs-1's slots (and epilogue)
      const int ws = gs & 1, xs2 = xs == 0 ? 2 : xs - 1;
      if constexpr (R1EARLY) {
        vec8 wf1[NI], xf1[MI];
        substep(xs, ws, 0);
        frag_reads(xs, ws, 1, wf1, xf1);
        if (gs + 1 < S) { dma_w(ws ^ 1, wbase + (size_t)wkt * ROWB); next_w(); }
        if (gs + 2 < S) { dma_x(xs2, xbase + (size_t)xkt * ROWB); next_x(); }
        frag_mfmas(wf1, xf1);
      } else {
      substep(xs, ws, 0);
      if (gs + 1 < S) { dma_w(ws ^ 1, wbase + (size_t)wkt * ROWB); next_w(); }   // W first, then X: the counted waits rely on this order
      if (gs + 2 < S) { dma_x(xs2, xbase + (size_t)xkt * ROWB); next_x(); }
      substep(xs, ws, 1);
      }
      xs_last = xs; ws_last = ws;
      xs = xs == 2 ? 0 : xs + 1;
    }
    if constexpr (STAMP) {
      st_kc += __builtin_amdgcn_s_memtime() - st_a; st_kr += __builtin_amdgcn_s_memrealtime() - st_b; st_kn += nk;
      if (ti + 1 == my && tid == 0) {                  // (the last tile's epilogue is not inside the kernel figure: stamped before it)
        unsigned long long* o = (unsigned long long*)g.ln_stats + (size_t)blockIdx.x * 6;
        o[0] = __builtin_amdgcn_s_memtime() - st_c0; o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
        o[2] = st_kc; o[3] = st_kr; o[4] = st_kn; o[5] = my;
      }
    }
    __builtin_amdgcn_s_barrier();                      // the last step's two slots are free: epilogue staging
    // The bias pair landed long ago (the wait of step kt == 1 retired it; >= 16 copies were issued behind it).  This statement
    // is its fence: at most 12 operations are in flight here (X(gs+1), W(gs+1), X(gs+2)), so it never stalls, and it names the
    // two registers as read-write operands -- no use, copy or spill of them can be scheduled above it.
    if constexpr (DMAW) asm volatile("s_waitcnt vmcnt(24)" : "+v"(bias_lo), "+v"(bias_hi) :: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" : "+v"(bias_lo), "+v"(bias_hi) :: "memory");

    // ---- epilogue: 64 x 128 per wave = 4 column groups of 32 x 2 half-strips of 32 rows, each through 4 KB of LDS (all
    // eight waves inside the last step's X slot; the wave's 128 bias values wait in 512 B of the last step's W slot).
    // No vector-memory instruction here but the stores: exactly 16 per wave.
    float* cb = (float*)(smem + WRING + ws_last * TILE_BYTES + wave * 512);
    cb[lane] = g.bias ? bias_lo : 0.f; cb[64 + lane] = g.bias ? bias_hi : 0.f;   // wave-private: written and read by this wave only
    char* wl = smem + XRING + xs_last * TILE_BYTES + wave * 4096;
    const int arow = lane & 15, apiece = lane >> 4, row0 = lane >> 2, j = lane & 3;
    const int mb = m0 + wr * 64, nb = n0 + wc * 128;
    if constexpr (DIRECT) {
      // Round 3: no LDS round trip.  A lane holds C[row li][4 columns] of each 16 x 16 accumulator tile -- 8 bytes of 16-bit
      // output, 32-byte runs per row if stored as is.  For a PAIR of neighbouring column tiles (ni, ni + 1) one
      // v_permlane16_swap per register (rows 1 / 3 of the first operand <-> rows 0 / 2 of the second; row = 16 lanes)
      // leaves every lane with 8 CONSECUTIVE columns: lane group g holds columns [8 (g >> 1), + 8) of tile ni + (g & 1), so one
      // 16-byte store per lane writes 64 contiguous bytes of each of 16 rows -- the same store shape as the staged epilogue,
      // the same 16 stores per wave (the counted waits do not change), the same arithmetic per value (bit-identical
      // results), and none of its 32 + 32 LDS instructions per wave.
      const int g4 = lane >> 4;
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      f32x4_t bv[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bv[ni] = *(const f32x4_t*)(cb + ni * 16 + 4 * g4);
      elem* crow = (elem*)g.C + (size_t)(mb + arow) * g.ldc + nb + (g4 & 1) * 16 + (g4 >> 1) * 8;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int np = 0; np < NI / 2; ++np) {
          unsigned pk[2][2];                                         // [tile of the pair][dword]: 4 x 16-bit values
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4_t a = acc[2 * np + h][mi], bb = bv[2 * np + h];
            float v0 = __fadd_rn(a[0], bb[0]), v1 = __fadd_rn(a[1], bb[1]), v2 = __fadd_rn(a[2], bb[2]), v3 = __fadd_rn(a[3], bb[3]);
            if constexpr (EPI == HM_EPI_GELU) {
              // (pairs as the staged epilogue forms them: values q and 4 + q of a lane's 8 columns -- there columns c, c + 4;
              //  gelu_fast2 evaluates its two values independently, so the pairing does not change a result)
              const f32x2_t g0 = gelu_fast2(f32x2_t{v0, v1}), g1 = gelu_fast2(f32x2_t{v2, v3});
              v0 = g0[0] * g.out_scale; v1 = g0[1] * g.out_scale; v2 = g1[0] * g.out_scale; v3 = g1[1] * g.out_scale;
            }
            if constexpr (EPI == HM_EPI_SILU) {                  // (1x1 convolutions of the detector: launch_conv)
              const f32x2_t g0 = silu2(f32x2_t{v0, v1}), g1 = silu2(f32x2_t{v2, v3});
              v0 = g0[0]; v1 = g0[1]; v2 = g1[0]; v3 = g1[1];
            }
            typename T::vec4 o;
            o[0] = (elem)v0; o[1] = (elem)v1; o[2] = (elem)v2; o[3] = (elem)v3;
            const u32x2 w = __builtin_bit_cast(u32x2, o);
            pk[h][0] = w[0]; pk[h][1] = w[1];
          }
          const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
          const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
          *(u32x4*)(crow + (size_t)(mi * 16) * g.ldc + np * 32) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
      continue;
    }
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
      const f32x4_t b0 = *(const f32x4_t*)(cb + cg * 32 + 8 * j), b1 = *(const f32x4_t*)(cb + cg * 32 + 8 * j + 4);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int nl = 0; nl < 2; ++nl)
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) {
            const int row = mm * 16 + arow;
            *(f32x4_t*)(wl + row * 128 + (((nl * 4 + apiece) ^ (row & 7)) << 4)) = acc[cg * 2 + nl][half * 2 + mm];
          }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 16 + row0, sw = row & 7;
          const f32x4_t v0 = *(const f32x4_t*)(wl + row * 128 + (((2 * j) ^ sw) << 4));
          const f32x4_t v1 = *(const f32x4_t*)(wl + row * 128 + (((2 * j + 1) ^ sw) << 4));
          vec8 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float a = __fadd_rn(v0[q], b0[q]), b = __fadd_rn(v1[q], b1[q]);           // as the one-tile kernels round
            if constexpr (EPI == HM_EPI_GELU) { const f32x2_t gq = gelu_fast2(f32x2_t{a, b}); a = gq[0] * g.out_scale; b = gq[1] * g.out_scale; }
            if constexpr (EPI == HM_EPI_SILU) { const f32x2_t sq = silu2(f32x2_t{a, b}); a = sq[0]; b = sq[1]; }
            o[q] = (elem)a; o[4 + q] = (elem)b;
          }
          *(vec8*)((elem*)g.C + (size_t)(mb + half * 32 + row) * g.ldc + nb + cg * 32 + 8 * j) = o;
        }
      }
    }
  }
}

#ifdef HM_ABLATIONS
// ---------------------------------------------------------------------------------------------------------------
// Round 4 EXPERIMENT (experiments library only, -DHM_ABLATIONS; tests opt-in): gemm_px_kernel with a SOFTWARE-PIPELINED K loop
// (variant 27, gemm_pp_kernel; 33 = the same with the copies of waves 4..7 issued four groups after those of waves 0..3).
// Bit-identical to variant 26 and NOT faster: interleaved A/B on one box (profiles/r04_gemm_pipelined_ab.log), store epilogue,
// B = 64: qkv 122.0 us (26) / 123.2 (27) / 132.4 (33), fc1 144.1 / 147.7 / 154.2, fc2 130.9 / 135.4 / 140.0, kv 172.6 / 178.6 /
// 184.4 -- the reasoning below (which the loop's ISA confirms: no MFMA waits for a fragment any more) predicted a third off the
// K loop; the measurement says the fragment reads were never what the loop waited for.  With rounds 1-3 (ring depth, tile
// shapes, wave counts, stagger, no-wait ablations) every structure lands on ~1.5 us per K-step on random operands: the guide's
// own 256 x 256 template rate, and the rate at which the chip holds its clock down under MFMA + LDS-DMA load (DVFS give-back).
// What the px / x3 loops do per K-step and wave, as hipcc emits them: [barrier] 12 ds_read_b128, lgkmcnt(3..0), 32 MFMAs,
// 8 copies, 12 ds_read_b128, wait, 32 MFMAs.  All eight waves leave the barrier together, so all of them read while both
// MFMA pipes of every SIMD idle, then all of them multiply while the LDS idles: 2 x (96 reads x 4 cycles + latency) + 2 x 1024
// MFMA cycles per SIMD = ~3050 cycles per step where the MFMAs alone need 2048 (v_mfma 16x16x32: 16 cycles, 64 per wave, two
// waves per SIMD) -- the 1.5 us per step that every tile structure of rounds 1-3 landed on.  Here the fragment reads never
// wait in front of the MFMAs that use them:
//   * a K-step is 16 GROUPS of 4 MFMAs (ks = g >> 3, ni = g & 7: W fragment (ks, ni) against the four X fragments of ks);
//   * fragment reads are issued from inline asm FIVE groups ahead of their use (W: an 8-entry register ring, slot = ni; X: two
//     sets of four, one per ks) and waited for by COUNT (LDS operations retire in order; the counts are derived below);
//   * the step's barrier moves from its top to group 12: by then every read of the step has been issued and is retired by one
//     lgkmcnt(0), so after the barrier the step's LDS slots are free -- the copies W(t+2), X(t+3) go out there, two pieces per
//     group over groups 12..15, and the first fragments of step t+1 (whose copies the same wait + barrier have completed) are
//     requested under the last 16 MFMAs of step t.  W therefore runs TWO steps ahead as well (px: one), in the same two slots.
//   * last step of a tile: no reads and no copies behind the barrier; after the epilogue (which stages through this wave's own
//     4 KB of the step's X slot and keeps its bias in its own 4 KB of the W slot -- exactly the regions this wave's next copies
//     overwrite) the copies, the next tile's bias request and its first nine reads are issued in the order the steady state
//     uses, so every group's wait count is the same in every step.
// Read stream of a step (L_g = reads issued in front of group g) and the count each group waits with (reads issued after the
// last one it needs, through L_g):
//   L0 W5 | L1 W6 | L2 W7 X1a | L3 W8 X1b | L4 W9 X1c | L5 W10 X1d | L6 W11 | L7 W12 | L8 W13 | L9 W14 | L10 W15 | L11 - |
//   [lgkmcnt(0) vmcnt barrier] L12 nW0 nW1 nX0a | L13 nW2 nX0b nX0c | L14 nW3 nX0d | L15 nW4
//   g0 2 (needs nX0d) | g1 9 | g2 9 | g3 8 | g4 8 | g5 9 | g6 9 | g7 9 | g8 3 (needs X1d) | g9 7 | g10 6 | g11 4 | g12 3 | g13 6 | g14 8 | g15 9
// vmcnt: copies retire in issue order: W then X, always.  At group 12 of step t everything but X(t+2) must have landed:
// vmcnt(4); in the first step of a tile the tile's two bias loads are younger than X(t+2): vmcnt(6); vmcnt(0) once no X(t+2)
// exists.  The epilogue's 16 stores are older than the copies issued behind it and retire with them.
constexpr int pp_wait_count(int g) {
  constexpr int c[16] = {2, 9, 9, 8, 8, 9, 9, 9, 3, 7, 6, 4, 3, 6, 8, 9};
  return c[g];
}

template <class T, int EPI, bool DIRECT = true, bool STAG = false>   // STAG: waves 4..7 issue a step's copies four groups later than waves 0..3 (groups 0..3 of the next step)
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const KArgs g) {
  static_assert(EPI == HM_EPI_STORE || EPI == HM_EPI_GELU, "16-bit store epilogues only");
  constexpr int WN = 2, MI = 4, NI = 8, ROWB = 128;
  constexpr int TILE_BYTES = 256 * ROWB;                               // 32 KB
  constexpr int XRING = 0, WRING = 3 * TILE_BYTES;
  constexpr int XI = 4, WI = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using vec8 = typename T::vec8;
  using elem = typename T::elem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = g.N >> 8, tiles_m = g.M >> 8, tiles = tiles_m * tiles_n;
  const int G = gridDim.x, xcd = blockIdx.x & 7, li = blockIdx.x >> 3, per = G >> 3;
  const int tq = tiles >> 3, tr = tiles & 7;
  const int run_lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, run_len = tq + (xcd < tr ? 1 : 0);
  const int my = li < run_len ? (run_len - li + per - 1) / per : 0;
  if (my == 0) return;
  const int nk = g.K / 64, S = my * nk;
  const int wr = wave / WN, wc = wave % WN;
  const char* X = (const char*)g.X;
  const char* W = (const char*)g.W;

  const int srow = lane >> 3, swz = (lane & 7) ^ (srow & 7);
  // lane part of a copy's source address (piece 0 of this wave); the piece index and the tile origin ride in the scalar base
  const unsigned xoff0 = (unsigned)(wave * XI * 8 + srow) * (unsigned)(g.ldx * 2) + swz * 16;
  const unsigned woff0 = (unsigned)(wave * WI * 8 + srow) * (unsigned)(g.ldw * 2) + swz * 16;
  const size_t xpiece = (size_t)8 * g.ldx * 2, wpiece = (size_t)8 * g.ldw * 2;
  auto origin = [&](int ti, int& m0, int& n0) {
    int tm, tn;
    tile_coords(run_lo + li + ti * per, tiles_m, tiles_n, g.group_m, tm, tn);
    m0 = tm << 8; n0 = tn << 8;
  };
  // cursors over the concatenated step sequence: W runs two steps ahead of the MFMAs, X three (one copy cursor each)
  int xti = 0, xkt = 0, wti = 0, wkt = 0, m0, n0;
  origin(0, m0, n0);
  const char* xbase = X + (size_t)m0 * g.ldx * 2;
  const char* wbase = W + (size_t)n0 * g.ldw * 2;
  auto next_x = [&]() {
    if (++xkt == nk) {
      xkt = 0;
      if (++xti < my) { int a, b; origin(xti, a, b); xbase = X + (size_t)a * g.ldx * 2; }
    }
  };
  auto next_w = [&]() {
    if (++wkt == nk) {
      wkt = 0;
      if (++wti < my) { int a, b; origin(wti, a, b); wbase = W + (size_t)b * g.ldw * 2; }
    }
  };
  auto piece_x = [&](int slot, int i) {
    glds16_hidden_s(xbase + (size_t)xkt * ROWB + i * xpiece, xoff0, smem + XRING + slot * TILE_BYTES + wave * XI * 1024 + i * 1024);
  };
  auto piece_w = [&](int slot, int i) {
    glds16_hidden_s(wbase + (size_t)wkt * ROWB + i * wpiece, woff0, smem + WRING + slot * TILE_BYTES + wave * WI * 1024 + i * 1024);
  };

  // fragment read addresses: X (ks, mi) = X slot + lx[ks] + mi * 2048, W (ks, ni) = W slot + lw[ks] + ni * 2048 (immediates)
  const int frow = lane & 15, fsw = lane & 7, fch = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned lx[2], lw[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    lx[ks] = lds0 + XRING + (unsigned)((wr * 16 * MI + frow) * ROWB + (((ks * 4 + fch) ^ fsw) << 4));
    lw[ks] = lds0 + WRING + (unsigned)((wc * 16 * NI + frow) * ROWB + (((ks * 4 + fch) ^ fsw) << 4));
  }

  f32x4_t acc[NI][MI];
  vec8 wf[NI], xf[2][MI];
#define PP_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  // the nine reads that open a step, in the order the steady state issues them behind its barrier (nW0 nW1 nX0a | nW2 nX0b nX0c
  // | nW3 nX0d | nW4): part 0..3 = what groups 12..15 of the previous step carry
  auto open_reads = [&](auto part, unsigned ax0, unsigned aw0) {
    constexpr int P = decltype(part)::value;
    if constexpr (P == 0) { PP_RD(wf[0], aw0, 0); PP_RD(wf[1], aw0, 2048); PP_RD(xf[0][0], ax0, 0); }
    if constexpr (P == 1) { PP_RD(wf[2], aw0, 4096); PP_RD(xf[0][1], ax0, 2048); PP_RD(xf[0][2], ax0, 4096); }
    if constexpr (P == 2) { PP_RD(wf[3], aw0, 6144); PP_RD(xf[0][3], ax0, 6144); }
    if constexpr (P == 3) { PP_RD(wf[4], aw0, 8192); }
  };

  // prologue: X(0), W(0), X(1) -> wait for the first two -> barrier -> W(1), X(2) -> bias request -> the first nine reads
  int gs = 0, xs = 0;                                  // global step, its X slot (gs % 3); W slot = gs & 1
#pragma unroll
  for (int i = 0; i < XI; ++i) piece_x(0, i);
  next_x();
#pragma unroll
  for (int i = 0; i < WI; ++i) piece_w(0, i);
  next_w();
#pragma unroll
  for (int i = 0; i < XI; ++i) piece_x(1, i);
  next_x();
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (1 < S) {
#pragma unroll
    for (int i = 0; i < WI; ++i) piece_w(1, i);
    next_w();
  }
  if (2 < S) {
#pragma unroll
    for (int i = 0; i < XI; ++i) piece_x(2, i);
    next_x();
  }

  for (int ti = 0; ti < my; ++ti) {
    if (ti > 0) origin(ti, m0, n0);
    // this wave's 128 bias values, two per lane (see gemm_px_kernel): always issued, from asm, fenced in front of the epilogue
    float bias_lo, bias_hi;
    {
      const float* bsrc = g.bias ? g.bias + n0 + wc * 128 : (const float*)g.W;
      asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %3 offset:256"
                   : "=&v"(bias_lo), "=&v"(bias_hi) : "v"(lane * 4), "s"(bsrc) : "memory");
    }
    {
      const unsigned ax0 = lx[0] + (unsigned)(xs * TILE_BYTES), aw0 = lw[0] + (unsigned)((gs & 1) * TILE_BYTES);
      open_reads(std::integral_constant<int, 0>{}, ax0, aw0); open_reads(std::integral_constant<int, 1>{}, ax0, aw0);
      open_reads(std::integral_constant<int, 2>{}, ax0, aw0); open_reads(std::integral_constant<int, 3>{}, ax0, aw0);
    }
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int xs_last = 0, ws_last = 0;
    // one K-step; `last` (the tile's last step) is a compile-time property: behind its barrier there are no reads and no copies,
    // and a runtime branch around asm statements that define fragment registers costs register copies in every group
    auto step = [&](auto last_c, const int kt) {
      constexpr bool last = decltype(last_c)::value;
      const int ws = gs & 1;
      const unsigned ax1 = lx[1] + (unsigned)(xs * TILE_BYTES);
      const unsigned aw0 = lw[0] + (unsigned)(ws * TILE_BYTES), aw1 = lw[1] + (unsigned)(ws * TILE_BYTES);
      const int xsn = xs == 2 ? 0 : xs + 1;
      const unsigned nax0 = lx[0] + (unsigned)(xsn * TILE_BYTES), naw0 = lw[0] + (unsigned)((ws ^ 1) * TILE_BYTES);
      auto group = [&](auto gc) {
        constexpr int GI = decltype(gc)::value, ks = GI >> 3, ni = GI & 7;
        // ---- reads (and, behind the barrier, copies) issued in front of this group
        if constexpr (GI <= 10) {                                     // W fragment five groups on (same step)
          constexpr int j = GI + 5, jks = j >> 3, jni = j & 7;
          if constexpr (jks == 0) PP_RD(wf[jni], aw0, jni * 2048); else PP_RD(wf[jni], aw1, jni * 2048);
        }
        if constexpr (GI >= 2 && GI <= 5) PP_RD(xf[1][GI - 2], ax1, (GI - 2) * 2048);     // X fragments of ks = 1
        if constexpr (GI == 12) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // every fragment read of this step has returned
          if (gs + 1 < S) {
            if (gs + 2 >= S) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (kt == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // X(gs+2) and the tile's bias pair may stay in flight
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                   // X(gs+2)
          }
          __builtin_amdgcn_s_barrier();      // step gs+1 complete in LDS; nobody reads step gs's slots any more
        }
        if constexpr (GI >= 12) {
          if constexpr (!last) {
            // copies into the slots this step just gave up: W(gs+2) -> its W slot, X(gs+3) -> its X slot; two pieces per group
            if (!STAG || wave < 4) {
              if constexpr (GI == 12 || GI == 13) {
                if (gs + 2 < S) { piece_w(ws, 2 * (GI - 12)); piece_w(ws, 2 * (GI - 12) + 1); if constexpr (GI == 13) next_w(); }
              } else {
                if (gs + 3 < S) { piece_x(xs, 2 * (GI - 14)); piece_x(xs, 2 * (GI - 14) + 1); if constexpr (GI == 15) next_x(); }
              }
            }
            open_reads(std::integral_constant<int, GI - 12>{}, nax0, naw0);     // the first fragments of step gs + 1
          }
        }
        if constexpr (STAG && GI <= 3) {
          // waves 4..7: the copies of the step before this one (its barrier has passed), unless that step was a tile's last
          // (kt == 0: everyone issued those behind the epilogue).  Same issue order per wave, so the counted waits do not change.
          if (wave >= 4 && kt > 0) {
            const int wprev = ws ^ 1, xprev = xs == 0 ? 2 : xs - 1;
            if constexpr (GI <= 1) {
              if (gs + 1 < S) { piece_w(wprev, 2 * GI); piece_w(wprev, 2 * GI + 1); if constexpr (GI == 1) next_w(); }
            } else {
              if (gs + 2 < S) { piece_x(xprev, 2 * (GI - 2)); piece_x(xprev, 2 * (GI - 2) + 1); if constexpr (GI == 3) next_x(); }
            }
          }
        }
        // ---- wait for this group's fragments by count, naming them so that no MFMA below can move above the wait
        {
          vec8 &w = wf[ni], &x0 = xf[ks][0], &x1 = xf[ks][1], &x2 = xf[ks][2], &x3 = xf[ks][3];
          if constexpr (ni == 0) {                                    // the group that opens a sub-step also releases its X set
            asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(w), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "n"(pp_wait_count(GI)));
          } else if constexpr (GI >= 12 && last) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w));
          } else {
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w) : "n"(pp_wait_count(GI)));
          }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = T::mfma(wf[ni], xf[ks][mi], acc[ni][mi]);
      };
      group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{}); group(std::integral_constant<int, 2>{});
      group(std::integral_constant<int, 3>{}); group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
      group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{}); group(std::integral_constant<int, 8>{});
      group(std::integral_constant<int, 9>{}); group(std::integral_constant<int, 10>{}); group(std::integral_constant<int, 11>{});
      group(std::integral_constant<int, 12>{}); group(std::integral_constant<int, 13>{}); group(std::integral_constant<int, 14>{});
      group(std::integral_constant<int, 15>{});
      xs_last = xs; ws_last = ws;
      xs = xsn;
      ++gs;
    };
    for (int kt = 0; kt + 1 < nk; ++kt) step(std::false_type{}, kt);
    step(std::true_type{}, nk - 1);
    // (no barrier here: the one at group 12 of the last step already separates every wave's last reads of these slots from the
    // staging below.)  Fence of the bias pair, as in gemm_px_kernel: at most 8 copies are in flight, so it never stalls.
    asm volatile("s_waitcnt vmcnt(12)" : "+v"(bias_lo), "+v"(bias_hi) :: "memory");

    // ---- epilogue (gemm_px_kernel's, with the bias in THIS wave's 4 KB of the last step's W slot)
    // (the epilogue's lane constants are re-derived from an opaque copy of the lane id: hoisted out of the tile loop they would
    //  live through the K loop, whose 192 accumulator + fragment registers leave no room -- hipcc spilled two of them)
    int elane = lane;
    asm volatile("" : "+v"(elane));
    float* cb = (float*)(smem + WRING + ws_last * TILE_BYTES + wave * 4096);
    cb[elane] = g.bias ? bias_lo : 0.f; cb[64 + elane] = g.bias ? bias_hi : 0.f;
    char* wl = smem + XRING + xs_last * TILE_BYTES + wave * 4096;
    const int arow = elane & 15, apiece = elane >> 4, row0 = elane >> 2, j = elane & 3;
    const int mb = m0 + wr * 64, nb = n0 + wc * 128;
    if constexpr (DIRECT) {
      const int g4 = elane >> 4;
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      f32x4_t bv[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bv[ni] = *(const f32x4_t*)(cb + ni * 16 + 4 * g4);
      elem* crow = (elem*)g.C + (size_t)(mb + arow) * g.ldc + nb + (g4 & 1) * 16 + (g4 >> 1) * 8;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int np = 0; np < NI / 2; ++np) {
          unsigned pk[2][2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4_t a = acc[2 * np + h][mi], bb = bv[2 * np + h];
            float v0 = __fadd_rn(a[0], bb[0]), v1 = __fadd_rn(a[1], bb[1]), v2 = __fadd_rn(a[2], bb[2]), v3 = __fadd_rn(a[3], bb[3]);
            if constexpr (EPI == HM_EPI_GELU) {
              const f32x2_t g0 = gelu_fast2(f32x2_t{v0, v1}), g1 = gelu_fast2(f32x2_t{v2, v3});
              v0 = g0[0] * g.out_scale; v1 = g0[1] * g.out_scale; v2 = g1[0] * g.out_scale; v3 = g1[1] * g.out_scale;
            }
            typename T::vec4 o;
            o[0] = (elem)v0; o[1] = (elem)v1; o[2] = (elem)v2; o[3] = (elem)v3;
            const u32x2 w = __builtin_bit_cast(u32x2, o);
            pk[h][0] = w[0]; pk[h][1] = w[1];
          }
          const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
          const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
          *(u32x4*)(crow + (size_t)(mi * 16) * g.ldc + np * 32) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
    } else {
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) {
        const f32x4_t b0 = *(const f32x4_t*)(cb + cg * 32 + 8 * j), b1 = *(const f32x4_t*)(cb + cg * 32 + 8 * j + 4);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int nl = 0; nl < 2; ++nl)
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
              const int row = mm * 16 + arow;
              *(f32x4_t*)(wl + row * 128 + (((nl * 4 + apiece) ^ (row & 7)) << 4)) = acc[cg * 2 + nl][half * 2 + mm];
            }
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int row = it * 16 + row0, sw = row & 7;
            const f32x4_t v0 = *(const f32x4_t*)(wl + row * 128 + (((2 * j) ^ sw) << 4));
            const f32x4_t v1 = *(const f32x4_t*)(wl + row * 128 + (((2 * j + 1) ^ sw) << 4));
            vec8 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float a = __fadd_rn(v0[q], b0[q]), b = __fadd_rn(v1[q], b1[q]);
              if constexpr (EPI == HM_EPI_GELU) { const f32x2_t gq = gelu_fast2(f32x2_t{a, b}); a = gq[0] * g.out_scale; b = gq[1] * g.out_scale; }
              o[q] = (elem)a; o[4 + q] = (elem)b;
            }
            *(vec8*)((elem*)g.C + (size_t)(mb + half * 32 + row) * g.ldc + nb + cg * 32 + 8 * j) = o;
          }
        }
      }
    }
    // the copies the last step held back: W(gs+1) .. wait: gs already counts the next step.  W(gs + 1), X(gs + 2) in the numbering
    // of the step that just ended + 1: the slots of the tile's last step (ws_last, xs_last), whose staging this wave has finished
    // (its LDS reads returned before the stores that carry their data were issued)
    if (gs + 1 < S) {
#pragma unroll
      for (int i = 0; i < WI; ++i) piece_w(ws_last, i);
      next_w();
    }
    if (gs + 2 < S) {
#pragma unroll
      for (int i = 0; i < XI; ++i) piece_x(xs_last, i);
      next_x();
    }
  }
#undef PP_RD
}

#endif  // HM_ABLATIONS (variants 27 / 33)

// Default persistent grid: one workgroup per CU on all but ONE CU of every XCD (248 of 256; see launch_px), unless all 256 save
// a whole round of tiles (1020 tiles = fc1 at 68 hands: 4 rounds instead of 5).
int px_default_grid(int tiles, int cus) {
  const int g248 = cus >= 64 ? cus - 8 : cus;
  return (tiles + cus - 1) / cus < (tiles + g248 - 1) / g248 ? cus : g248;
}

// The grid launch_px takes for `tiles` whole tiles on `cus` CUs (exported as hm_gemm_px_grid for the host tests).
// HM_OPT_PX_GRID (hm_set_option) wins while it is non-zero and is NOT copied into the start-up default: setting the option back
// to 0 restores the environment's grid (HM_PX_GRID, read once) or the default below.
// Default: one workgroup per CU on all but ONE CU of every XCD (248 of 256).  Round 3, interleaved whole-model A/B in one
// process (tools/bench_model_ab.py, profiles/r03_model_ab_px_grid.log): a lone forward takes 18.50 ms with 240 or 248
// workgroups against 19.84 with 256 -- the same number of tile rounds for qkv / fc1 (720 / 960 tiles), so it is not
// quantisation: a 256-workgroup persistent kernel needs every CU of the chip at once, and whichever CU is late (the previous
// kernel's last waves, the dispatcher) delays one workgroup's whole run of tiles.  With two batches in flight the other
// stream filled that hole already (17.55 ms either way).
int px_grid_for(int tiles, int cus) {
  if (g_px_grid == -2) {
    const char* e = getenv("HM_PX_GRID");                          // tuning runs, read ONCE: start-up default of HM_OPT_PX_GRID
    g_px_grid = e ? atoi(e) : -1;
  }
  const int opt_grid = hm_option(HM_OPT_PX_GRID);
  const int forced = opt_grid > 0 ? opt_grid : (g_px_grid > 0 ? g_px_grid : 0);
  int want = forced > 0 ? forced : px_default_grid(tiles, cus);
  if (want > cus) want = cus;
  // a multiple of the 8 XCDs, rounded UP when there are fewer tiles than workgroups (a workgroup without a tile returns at once;
  // rounded down, 180 tiles on 176 workgroups were two rounds: qkv at 16 hands 67 us against 43 for the 128 x 128 tile)
  return tiles < want ? ((tiles + 7) & ~7) : (want & ~7);
}
extern "C" int hm_gemm_px_grid(int tiles, int cus) { return px_grid_for(tiles, cus > 0 ? cus : 256); }

template <class T, int EPI, int PIPE = 0>      // PIPE: 1 = the software-pipelined K loop (gemm_pp_kernel), 2 = with the copy stagger between the wave halves
int launch_px(const KArgs& g, hipStream_t s) {
  constexpr int LDS = 5 * 256 * 128;
  // Epilogue form.  Both are bit-identical; measured in one process (profiles/r03_gemm_px_epilogue_ab.log): the lane-swap form
  // is 3 % faster with the GELU (fc1 157.6 vs 162.1 us: its VALU work no longer queues behind 64 LDS instructions per wave)
  // and 3 % slower for the plain store (qkv 119.8 vs 116.4, kv 177.2 vs 170.9) -- so each epilogue takes its faster form.
  // HM_OPT_PX_LDS_EPILOGUE: 0 = that choice, 1 = always through LDS, 2 = always lane swaps.
  const int form = hm_option(HM_OPT_PX_LDS_EPILOGUE);
  const bool staged = form == 1 || (form == 0 && ((PIPE == 1 || PIPE == 2) || (EPI != HM_EPI_GELU && EPI != HM_EPI_SILU)));
  if (PIPE == 4 && !g.ln_stats) return hm_set_error(HM_ERR_ARG, "hm_gemm: variant 35 (stamps) needs a device buffer of 6 x 8 bytes per workgroup in ln_stats");   // (pipelined kernel: the lane-swap GELU form does not fit the register file)
#ifdef HM_ABLATIONS
  void (*kern)(const KArgs) = nullptr;
  if constexpr (PIPE == 0) kern = staged ? gemm_px_kernel<T, EPI, false> : gemm_px_kernel<T, EPI, true>;     // (the only form a SiLU epilogue exists in)
  else kern = PIPE == 9 ? (staged ? gemm_px_kernel<T, EPI, false, false, false, true, true, false, 2> : gemm_px_kernel<T, EPI, true, false, false, true, true, false, 2>)
            : PIPE == 8 ? (staged ? gemm_px_kernel<T, EPI, false, false, false, true, true, false, 1> : gemm_px_kernel<T, EPI, true, false, false, true, true, false, 1>)
            : PIPE == 7 ? (staged ? gemm_px_kernel<T, EPI, false, false, false, true, true, true> : gemm_px_kernel<T, EPI, true, false, false, true, true, true>)
            : PIPE == 6 ? (staged ? gemm_px_kernel<T, EPI, false, false, false, true, false> : gemm_px_kernel<T, EPI, true, false, false, true, false>)
            : PIPE == 5 ? (staged ? gemm_px_kernel<T, EPI, false, false, false, false> : gemm_px_kernel<T, EPI, true, false, false, false>)
            : PIPE == 4 ? (staged ? gemm_px_kernel<T, EPI, false, false, true> : gemm_px_kernel<T, EPI, true, false, true>)
            : PIPE == 3 ? (staged ? gemm_px_kernel<T, EPI, false, true, false, false, false> : gemm_px_kernel<T, EPI, true, true, false, false, false>)   // (as measured: round 3's copy issue and read order)
            : PIPE == 2 ? (staged ? gemm_pp_kernel<T, EPI, false, true> : gemm_pp_kernel<T, EPI, true, true>)
            : PIPE ? (staged ? gemm_pp_kernel<T, EPI, false> : gemm_pp_kernel<T, EPI, true>)
                   : (staged ? gemm_px_kernel<T, EPI, false> : gemm_px_kernel<T, EPI, true>);
#else
  static_assert(PIPE == 0, "the pipelined K loop is an experiment (-DHM_ABLATIONS)");
  auto kern = staged ? gemm_px_kernel<T, EPI, false> : gemm_px_kernel<T, EPI, true>;
#endif
  static HmLdsOnce lds_once[2];      // (per instantiation of this launcher: one pair per (T, EPI, PIPE))
  if (const int rc = lds_once[staged ? 1 : 0].ensure((const void*)kern, LDS, "gemm: cannot raise the dynamic LDS limit")) return rc;
  const int tiles = (g.M >> 8) * (g.N >> 8);
  int cus = hm_device_cu_count();
  if (cus <= 0) cus = 256;
  const int grid = px_grid_for(tiles, cus);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, s, g);
  return hm_check_launch("hm_gemm");
}

// what gemm_px_kernel requires: whole 256 x 256 tiles, at least two K-steps, 16-byte output rows, 32-bit lane offsets
bool px_ok(const KArgs& g) {
  return g.M % 256 == 0 && g.N % 256 == 0 && g.K >= 128 && (g.ldc & 7) == 0 && (g.M >> 8) * (g.N >> 8) >= 8 &&
         256ull * g.ldx * 2 < (1ull << 32) && 256ull * g.ldw * 2 < (1ull << 32) && (((uintptr_t)g.C) & 15) == 0;
}

#ifdef HM_ABLATIONS
// ---------------------------------------------------------------------------------------------------------------
// Variant 29: the 256x256x64 tile on FOUR waves (one per SIMD), each owning a 128x128 quadrant = 8x8 MFMA tiles.
// Per K-step and CU that is 64 ds_read_b128 instead of the 96 of the 8-wave kernels (2/3 of the LDS traffic per flop) and
// a wave may use the whole 512-entry register file: 256 accumulator registers, and TWO sets of operand fragments, so the
// fragments of the next half-step are read while the MFMAs of the current one run -- inside one wave, without relying on
// a second wave of the SIMD being out of phase.  One barrier per K-step, in its middle:
//     first half : MFMAs on F0 = fragments (t, k 0..31)   | read F1 = (t, k 32..63)
//     lgkmcnt(0), vmcnt(8), barrier                        -> every wave is done with step t's slots; step t+1 has landed
//     second half: MFMAs on F1                             | copy W(t+2), X(t+3) into the freed slots | read F0 = (t+1, k 0..31)
// Ring as in gemm_x3_kernel (X: 3 slots, two steps ahead; W: 2 slots).  Copies, reads and MFMAs are laid out in 16 groups
// per half-step (1 copy, 1 read, 4 MFMAs) fenced by sched_barrier, so the issue order is the one written here.
template <class T, int EPI, int ABL = 0>     // ABL (HM_ABLATIONS builds only): 1 = no copies in the loop, 2 = copies and waits only
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const KArgs g) {
  constexpr int MI = 8, NI = 8, BM = 256, BN = 256, BK = 64, ROWB = 128;
  constexpr int TILE_BYTES = 256 * ROWB;
  constexpr int XRING = 0, WRING = 3 * TILE_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using vec8 = typename T::vec8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wr = wave >> 1, wc = wave & 1;
  const char* X = (const char*)g.X;
  const char* W = (const char*)g.W;

  // copies: wave w moves rows [64 w, 64 w + 64) of both tiles, 8 pieces of 8 rows x 128 B each
  const int srow = lane >> 3, swz = (lane & 7) ^ (srow & 7);
  unsigned xoff[8], woff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int gm = m0 + wave * 64 + i * 8 + srow;
    gm = gm < g.M ? gm : g.M - 1;
    xoff[i] = (unsigned)gm * (unsigned)(g.ldx * 2) + swz * 16;
    int gn = n0 + wave * 64 + i * 8 + srow;
    gn = gn < g.N ? gn : g.N - 1;
    woff[i] = (unsigned)gn * (unsigned)(g.ldw * 2) + swz * 16;
  }
  auto dma_x1 = [&](int slot, int kt, int i) { glds16_hidden_s(X + (size_t)kt * ROWB, xoff[i], smem + XRING + slot * TILE_BYTES + (wave * 8 + i) * 1024); };
  auto dma_w1 = [&](int slot, int kt, int i) { glds16_hidden_s(W + (size_t)kt * ROWB, woff[i], smem + WRING + slot * TILE_BYTES + (wave * 8 + i) * 1024); };

  f32x4_t acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fsw = lane & 7, fch = lane >> 4;
  const int fx = (wr * 128 + frow) * ROWB, fw = (wc * 128 + frow) * ROWB;
  const int c0 = (fch ^ fsw) * 16, c1 = ((4 + fch) ^ fsw) * 16;
  vec8 wf0[NI], xf0[MI], wf1[NI], xf1[MI];
  // fragment j of a half-step: j < 8 -> X rows 16 j.., else W rows 16 (j - 8)..
  auto rd = [&](const char* lx, const char* lw, int coff, int j, vec8 (&wf)[NI], vec8 (&xf)[MI]) {
    if constexpr (ABL == 2) return;
    if (j < 8) xf[j] = *(const vec8*)(lx + fx + j * 16 * ROWB + coff);
    else wf[j - 8] = *(const vec8*)(lw + fw + (j - 8) * 16 * ROWB + coff);
  };
  auto mma4 = [&](int j, const vec8 (&wf)[NI], const vec8 (&xf)[MI]) {      // group j of 16: W fragment j / 2, X fragments 4 (j & 1) .. + 3
    if constexpr (ABL == 2) return;
    const int ni = j >> 1, mb = (j & 1) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[ni][mb + q] = T::mfma(wf[ni], xf[mb + q], acc[ni][mb + q]);
  };

  const int nk = g.K / BK;                              // host: nk >= 3
  // prologue: X(0) W(0) X(1) W(1) X(2)
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_x1(0, 0, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_w1(0, 0, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_x1(1, 1, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_w1(1, 1, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_x1(2, 2, i);
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");    // X(0), W(0)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < 16; ++j) rd(smem + XRING, smem + WRING, c0, j, wf0, xf0);

  int xs = 0;                                          // X slot of step t (t % 3); W slot t & 1
  // one K-step; CW / CX: copies of W(t+2) / X(t+3) exist, MORE: a step t+1 exists (compile-time: the steady-state body has no branch)
  auto step = [&](auto CW, auto CX, auto MORE, int t) {
    const char* lx = smem + XRING + xs * TILE_BYTES;
    const char* lw = smem + WRING + (t & 1) * TILE_BYTES;
    // ---- first half
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      rd(lx, lw, c1, j, wf1, xf1);
      mma4(j, wf0, xf0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // F1 is in registers (this wave is done with step t's slots); all but X(t+2) of this wave's copies have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (decltype(CW)::value && ABL != 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- second half
    const int xn = xs == 2 ? 0 : xs + 1;               // slot of step t + 1
    const char* lx1 = smem + XRING + xn * TILE_BYTES;
    const char* lw1 = smem + WRING + ((t + 1) & 1) * TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < 8) { if constexpr (decltype(CW)::value && ABL != 1) dma_w1(t & 1, t + 2, j); }            // W first, then X: the counted wait relies on this order
      else { if constexpr (decltype(CX)::value && ABL != 1) dma_x1(xs, t + 3, j - 8); }
      if constexpr (decltype(MORE)::value) rd(lx1, lw1, c0, j, wf0, xf0);
      mma4(j, wf1, xf1);
      __builtin_amdgcn_sched_barrier(0);
    }
    xs = xn;
  };
  using Yes = std::true_type;
  using No = std::false_type;
  for (int t = 0; t < nk - 3; ++t) step(Yes{}, Yes{}, Yes{}, t);
  step(Yes{}, No{}, Yes{}, nk - 3);
  step(No{}, No{}, Yes{}, nk - 2);
  step(No{}, No{}, No{}, nk - 1);

  __builtin_amdgcn_s_barrier();                        // the ring is free: per-column vectors, then epilogue staging
  constexpr int EPI_BYTES = 4 * epi_stage_bytes(MI, NI);
  float2* rowstat = (float2*)(smem + EPI_BYTES);       // unused by these epilogues
  float* colvec = (float*)(rowstat + BM);
  colvec[tid] = g.bias ? g.bias[min(n0 + tid, g.N - 1)] : 0.f;
  __builtin_amdgcn_s_barrier();
  epilogue<T, EPI, MI, NI>(g, acc, m0 + wr * 128, n0 + wc * 128, lane, smem + wave * epi_stage_bytes(MI, NI),
                           rowstat + wr * 128, colvec + wc * 128, colvec + BN + wc * 128, 0);
}

template <class T, int EPI, int ABL = 0>
int launch_w4(const KArgs& g, hipStream_t s) {
  constexpr int LDS = 5 * 256 * 128;
  static_assert(LDS >= 4 * epi_stage_bytes(8, 8) + 256 * 8 + 2 * 256 * 4, "epilogue staging + vectors fit");
  auto kern = gemm_w4_kernel<T, EPI, ABL>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "gemm: cannot raise the dynamic LDS limit")) return rc;
  const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), LDS, s, g);
  return hm_check_launch("hm_gemm");
}

#endif  // HM_ABLATIONS (variant 29)

// ---------------------------------------------------------------------------------------------------------------
// Persistent form of gemm_fp8_kernel (the gemm_px_kernel idea on the fp8 MFMA) for the qkv (bf16 out) and fc1 (GELU ->
// MXFP8 out) GEMMs of BASELINE configs[4].  An fp8 K-step carries twice the flops of a 16-bit one, so a K = 1280 tile is
// only 10 steps (~15 us) and the per-tile start-up / drain of the one-tile kernel is ~40 % of its time at B = 256.
// Two-stage ring as in gemm_fp8_kernel; at a tile's LAST step the free stage receives the NEXT tile's first K-step, which
// lands under that step's MFMAs and the epilogue.  The epilogue stages through the stage the last step read (4 KB per
// wave + 1 KB of per-column bias / weight scales that travelled in four registers), issues no loads, and a fixed number
// of stores per wave: 16 (bf16 out) or 32 (MXFP8 out: 16 data + 16 block-scale stores), which the next tile's first wait
// counts around.  Whole tiles only (M, N multiples of 256; K >= 256).
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_fp8p_kernel(const KArgs g) {
  static_assert(EPI == HM_EPI_STORE || EPI == HM_EPI_GELU_MX8 || EPI == HM_EPI_RESID_F32, "qkv / fc1 / proj, fc2 epilogues");
  constexpr int WN = 2, MI = 4, NI = 8, BM = 256, BKB = 128;
  constexpr int TILE_BYTES = 256 * BKB, SCALE_BYTES = 4 * BM, STAGE_BYTES = 2 * TILE_BYTES + SCALE_BYTES;
  constexpr int NSTORE = EPI == HM_EPI_RESID_F32 ? 32 : 16;         // stores per wave and tile (RESID_F32: 32 x 16 B of fp32; MX8: 8 data + 8 scale)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = g.N >> 8, tiles_m = g.M >> 8, tiles = tiles_m * tiles_n;
  const int G = gridDim.x, xcd = blockIdx.x & 7, li = blockIdx.x >> 3, per = G >> 3;
  const int tq = tiles >> 3, tr = tiles & 7;
  const int run_lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, run_len = tq + (xcd < tr ? 1 : 0);
  const int my = li < run_len ? (run_len - li + per - 1) / per : 0;
  if (my == 0) return;
  const int nk = g.K / BKB, S = my * nk;
  const int wr = wave / WN, wc = wave % WN;
  const char* X = (const char*)g.X;
  const char* W = (const char*)g.W;
  const int srow = lane >> 3, chunk = (lane & 7) ^ (srow & 7);
  unsigned xoff[4], woff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xoff[i] = (unsigned)(wave * 32 + i * 8 + srow) * (unsigned)g.ldx + chunk * 16;
    woff[i] = (unsigned)(wave * 32 + i * 8 + srow) * (unsigned)g.ldw + chunk * 16;
  }
  const unsigned soff = (unsigned)(lane >> 4) * (unsigned)g.M + 16 * (lane & 15);      // block scales: 16 rows of one k-block per lane
  auto origin = [&](int ti, int& m0, int& n0) {
    int tm, tn;
    tile_coords(run_lo + li + ti * per, tiles_m, tiles_n, g.group_m, tm, tn);
    m0 = tm << 8; n0 = tn << 8;
  };
  int lti = 0, lkt = 0, m0, n0;                         // load cursor: one K-step ahead of the MFMAs
  origin(0, m0, n0);
  const char* xbase = X + (size_t)m0 * g.ldx;
  const char* wbase = W + (size_t)n0 * g.ldw;
  const char* sbase = (const char*)g.xs + m0;
  auto load_step = [&](int buf) {                       // the cursor's K-step -> stage `buf`, then advance the cursor
    char* st = smem + buf * STAGE_BYTES;
    const char* xb = xbase + (size_t)lkt * BKB;
    const char* wb = wbase + (size_t)lkt * BKB;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16_hidden_s(xb, xoff[i], st + (wave * 4 + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16_hidden_s(wb, woff[i], st + TILE_BYTES + (wave * 4 + i) * 1024);
    if (wave == 0) glds16_hidden_s(sbase + (size_t)lkt * 4 * g.M, soff, st + 2 * TILE_BYTES);
    if (++lkt == nk) {
      lkt = 0;
      if (++lti < my) {
        int a, b;
        origin(lti, a, b);
        xbase = X + (size_t)a * g.ldx; wbase = W + (size_t)b * g.ldw; sbase = (const char*)g.xs + a;
      }
    }
  };

  f32x4_t acc[NI][MI];
  const int frow = lane & 15, fg = lane >> 4, fsw = lane & 7;
  const int c_lo = ((fg ^ fsw) * 16), c_hi = (((4 + fg) ^ fsw) * 16);
  auto kstep = [&](int buf) {
    const char* lx = smem + buf * STAGE_BYTES;
    const char* lw = lx + TILE_BYTES;
    const unsigned char* ls = (const unsigned char*)(lx + 2 * TILE_BYTES) + fg * BM + wr * 64 + frow;
    v8i_t xf[MI];
    int xsc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const char* r = lx + (wr * 64 + i * 16 + frow) * BKB;
      const v4i_t lo = *(const v4i_t*)(r + c_lo), hi = *(const v4i_t*)(r + c_hi);
      xf[i] = v8i_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      xsc[i] = ls[i * 16];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      v8i_t wf[NI / 2];
#pragma unroll
      for (int i = 0; i < NI / 2; ++i) {
        const char* r = lw + (wc * 128 + (h * 4 + i) * 16 + frow) * BKB;
        const v4i_t lo = *(const v4i_t*)(r + c_lo), hi = *(const v4i_t*)(r + c_hi);
        wf[i] = v8i_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < NI / 2; ++i)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[h * 4 + i][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[i], xf[mi], acc[h * 4 + i][mi], 0, 0, 0, 127, 0, xsc[mi]);
      __builtin_amdgcn_s_setprio(0);
    }
  };

  float* cb = (float*)(smem + 2 * STAGE_BYTES) + wave * 256;          // [128 bias | 128 weight scales], wave-private
  load_step(0);
  int gs = 0, rd = 0;
  for (int ti = 0; ti < my; ++ti) {
    if (ti > 0) origin(ti, m0, n0);
    // bias and weight scale of this wave's 128 columns -> the wave's own 1 KB of LDS by dword LDS-DMA: no register, no
    // compiler-placed wait (a plain load here makes hipcc drain vmcnt to 0 wherever the value is first touched)
    {
      const char* bsrc = (const char*)(g.bias + n0 + wc * 128);
      const char* wsrc = (const char*)(g.wscale + n0 + wc * 128);
      glds4_hidden_s(bsrc, lane * 4, cb);
      glds4_hidden_s(bsrc + 256, lane * 4, cb + 64);
      glds4_hidden_s(wsrc, lane * 4, cb + 128);
      glds4_hidden_s(wsrc + 256, lane * 4, cb + 192);
    }
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt, ++gs) {
      // the loads of this step were issued one step ago -- before the previous tile's epilogue stores when kt == 0
      if (gs > 0 && kt == 0) {
        if constexpr (NSTORE == 16) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");     // stores + the 4 bias / scale copies above
        else asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (gs + 1 < S) load_step(rd ^ 1);
      kstep(rd);
      rd ^= 1;
    }
    __builtin_amdgcn_s_barrier();                      // the last step's stage (rd ^ 1 now) is free: epilogue staging
    char* stg = smem + (rd ^ 1) * STAGE_BYTES;
    char* wl = stg + wave * 4096;
    const int arow = lane & 15, apiece = lane >> 4, row0 = lane >> 2, j = lane & 3;
    const int mb = m0 + wr * 64, nb = n0 + wc * 128;
    if constexpr (EPI == HM_EPI_RESID_F32) {
      // fp32 out = acc * wscale + bias + residual.  The residual rows of (column group, half-strip) g + 1 are requested --
      // row-major, 2 x 16 B per lane and row, straight into registers the K loop no longer needs -- before group g is
      // processed.  The loads are issued from asm (a load hipcc tracks makes it drain vmcnt to 0, with the next tile's
      // copies and this tile's stores in flight) and waited for by count: newer than the loads of group g are the 4 stores
      // of group g - 1 and the 4 loads of group g + 1.  The wait statement takes the four registers as read-write operands,
      // so no use of them can be scheduled above it.  32 stores per wave (NSTORE).
      f32x4_t rr[2][4];
      auto rload = [&](int grp, f32x4_t (&dst)[4]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const float* p = g.resid + (size_t)(mb + (grp & 1) * 32 + it * 16 + row0) * g.ldr + nb + (grp >> 1) * 32 + 8 * j;
          asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                       : "=&v"(dst[2 * it]), "=&v"(dst[2 * it + 1]) : "v"(p) : "memory");
        }
      };
      rload(0, rr[0]);
#pragma unroll
      for (int grp = 0; grp < 8; ++grp) {
        const int cg = grp >> 1, half = grp & 1;
        if (grp + 1 < 8) rload(grp + 1, rr[(grp + 1) & 1]);
        const f32x4_t b0 = *(const f32x4_t*)(cb + cg * 32 + 8 * j), b1 = *(const f32x4_t*)(cb + cg * 32 + 8 * j + 4);
        const f32x4_t w0 = *(const f32x4_t*)(cb + 128 + cg * 32 + 4 * apiece), w1 = *(const f32x4_t*)(cb + 128 + cg * 32 + 16 + 4 * apiece);
#pragma unroll
        for (int nl = 0; nl < 2; ++nl)
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) {
            const int row = mm * 16 + arow;
            *(f32x4_t*)(wl + row * 128 + (((nl * 4 + apiece) ^ (row & 7)) << 4)) = acc[cg * 2 + nl][half * 2 + mm] * (nl ? w1 : w0);
          }
        f32x4_t (&r)[4] = rr[grp & 1];
        if (grp == 0 || grp == 7) asm volatile("s_waitcnt vmcnt(4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 16 + row0, sw = row & 7, m = mb + half * 32 + row, n = nb + cg * 32 + 8 * j;
          f32x4_t v0 = *(const f32x4_t*)(wl + row * 128 + (((2 * j) ^ sw) << 4));
          f32x4_t v1 = *(const f32x4_t*)(wl + row * 128 + (((2 * j + 1) ^ sw) << 4));
#pragma unroll
          for (int q = 0; q < 4; ++q) { v0[q] = __fadd_rn(v0[q], b0[q]); v1[q] = __fadd_rn(v1[q], b1[q]); }
          float* o = (float*)g.C + (size_t)m * g.ldc + n;
          *(f32x4_t*)o = v0 + r[2 * it];
          *(f32x4_t*)(o + 4) = v1 + r[2 * it + 1];
        }
      }
    } else if constexpr (EPI == HM_EPI_GELU_MX8) {
      // GELU -> MXFP8 in column groups of 64: a lane holds 16 consecutive columns of a row (half an MX block), the four
      // lanes of a row 64 bytes of e4m3 -- every data store is 16 B per lane and 64 B per row (the 32-column form stored
      // 8 B per lane, 32-B row segments, and twice as many instructions).  Staging: 32 rows x 256 B per wave = 8 KB, all
      // waves inside the consumed stage's X and W parts.  8 data + 8 scale stores per wave.
      char* wl8 = stg + wave * 8192;
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        const int cw = grp >> 1, half = grp & 1;                 // columns 64 cw .., rows 32 half ..
#pragma unroll
        for (int nl = 0; nl < 4; ++nl) {
          const f32x4_t wsv = *(const f32x4_t*)(cb + 128 + cw * 64 + nl * 16 + 4 * apiece);
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) {
            const int row = mm * 16 + arow;
            *(f32x4_t*)(wl8 + row * 256 + (((nl * 4 + apiece) ^ (row & 15)) << 4)) = acc[cw * 4 + nl][half * 2 + mm] * wsv;
          }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 16 + row0, sw = row & 15, m = mb + half * 32 + row, n = nb + cw * 64 + 16 * j;
          f32x4_t v[4];
          float amax = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            v[c] = *(const f32x4_t*)(wl8 + row * 256 + (((4 * j + c) ^ sw) << 4));
            const f32x4_t bi = *(const f32x4_t*)(cb + cw * 64 + 16 * j + 4 * c);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[c][q] = __fadd_rn(v[c][q], bi[q]);
          }
#pragma unroll
          for (int c = 0; c < 4; c += 2)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x2_t gq = gelu_fast2(f32x2_t{v[c][q], v[c + 1][q]});
              v[c][q] = gq[0]; v[c + 1][q] = gq[1];
              amax = fmaxf(amax, fmaxf(fabsf(gq[0]), fabsf(gq[1])));
            }
          const unsigned sb = mx8_scale_byte(pair_max(amax));    // lanes 2k, 2k+1 of a row hold one 32-column block
          const float inv = mx8_inv_scale(sb);
          int4 o8;
          o8.x = mx8_pack4(v[0][0] * inv, v[0][1] * inv, v[0][2] * inv, v[0][3] * inv);
          o8.y = mx8_pack4(v[1][0] * inv, v[1][1] * inv, v[1][2] * inv, v[1][3] * inv);
          o8.z = mx8_pack4(v[2][0] * inv, v[2][1] * inv, v[2][2] * inv, v[2][3] * inv);
          o8.w = mx8_pack4(v[3][0] * inv, v[3][1] * inv, v[3][2] * inv, v[3][3] * inv);
          *(int4*)((char*)g.C + (size_t)m * g.ldc + n) = o8;
          // one scale byte per (row, block): both lanes of the pair issue the store (same byte, same value): an unconditional
          // instruction, so the store count per wave is a constant
          g.out_scales[(size_t)(n >> 5) * g.M + m] = (unsigned char)sb;
        }
      }
    } else {
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
      const f32x4_t b0 = *(const f32x4_t*)(cb + cg * 32 + 8 * j), b1 = *(const f32x4_t*)(cb + cg * 32 + 8 * j + 4);
      const f32x4_t w0 = *(const f32x4_t*)(cb + 128 + cg * 32 + 4 * apiece), w1 = *(const f32x4_t*)(cb + 128 + cg * 32 + 16 + 4 * apiece);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int nl = 0; nl < 2; ++nl)
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) {
            const int row = mm * 16 + arow;
            *(f32x4_t*)(wl + row * 128 + (((nl * 4 + apiece) ^ (row & 7)) << 4)) = acc[cg * 2 + nl][half * 2 + mm] * (nl ? w1 : w0);
          }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 16 + row0, sw = row & 7, m = mb + half * 32 + row, n = nb + cg * 32 + 8 * j;
          f32x4_t v0 = *(const f32x4_t*)(wl + row * 128 + (((2 * j) ^ sw) << 4));
          f32x4_t v1 = *(const f32x4_t*)(wl + row * 128 + (((2 * j + 1) ^ sw) << 4));
#pragma unroll
          for (int q = 0; q < 4; ++q) { v0[q] = __fadd_rn(v0[q], b0[q]); v1[q] = __fadd_rn(v1[q], b1[q]); }
          if constexpr (EPI == HM_EPI_GELU_MX8) {
            float amax = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x2_t gq = gelu_fast2(f32x2_t{v0[q], v1[q]});
              v0[q] = gq[0]; v1[q] = gq[1];
              amax = fmaxf(amax, fmaxf(fabsf(v0[q]), fabsf(v1[q])));
            }
            const unsigned sb = mx8_scale_byte(quad_max(amax));        // the 4 lanes of a row hold one 32-column block
            const float inv = mx8_inv_scale(sb);
            int2 o8;
            o8.x = mx8_pack4(v0[0] * inv, v0[1] * inv, v0[2] * inv, v0[3] * inv);
            o8.y = mx8_pack4(v1[0] * inv, v1[1] * inv, v1[2] * inv, v1[3] * inv);
            *(int2*)((char*)g.C + (size_t)m * g.ldc + n) = o8;
            // one scale byte per (row, block): every lane of the quad issues the store, all to the same byte with the same
            // value -- an unconditional instruction, so the store count per wave is a constant
            g.out_scales[(size_t)(n >> 5) * g.M + m] = (unsigned char)sb;
          } else {
            bf16x8_t o;
#pragma unroll
            for (int q = 0; q < 4; ++q) { o[q] = (__bf16)v0[q]; o[4 + q] = (__bf16)v1[q]; }
            *(bf16x8_t*)((__bf16*)g.C + (size_t)m * g.ldc + n) = o;
          }
        }
      }
    }
    }
  }
}

template <int EPI>
int launch_fp8p(const KArgs& g, hipStream_t s) {
  constexpr int LDS = 2 * (2 * 256 * 128 + 4 * 256) + 8 * 1024;
  auto kern = gemm_fp8p_kernel<EPI>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "hm_gemm_fp8: cannot raise the dynamic LDS limit")) return rc;
  const int tiles = (g.M >> 8) * (g.N >> 8);
  int cus = hm_device_cu_count();
  if (cus <= 0) cus = 256;
  if (const int v = hm_option(HM_OPT_FP8P_GRID)) cus = v;   // tests: few workgroups, many tiles each (hm_set_option; a multiple of 8)
  const int grid = (tiles < cus ? tiles : cus) & ~7;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, s, g);
  return hm_check_launch("hm_gemm_fp8");
}

bool fp8p_ok(const KArgs& g) {
  return g.bias != nullptr && g.M % 256 == 0 && g.N % 256 == 0 && g.K >= 256 && (g.M >> 8) * (g.N >> 8) >= 8 && 256ull * g.ldx < (1ull << 32) &&
         256ull * g.ldw < (1ull << 32) && 4ull * g.M < (1ull << 32) && hm_option(HM_OPT_FP8_ONE_TILE) == 0;
}

// partial (sum, sum of squares) per 64 columns [P][M][2] -> (mean, rstd) per row [M][2]
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float2* __restrict__ part, float2* __restrict__ fin, int M, int P,
                                                          float invD, float eps) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float a = 0.f, b = 0.f;
#pragma unroll 8
  for (int p = 0; p < P; ++p) { const float2 v = part[(size_t)p * M + m]; a += v.x; b += v.y; }
  const float mean = a * invD;
  fin[m] = float2{mean, 1.0f / sqrtf(fmaxf(b * invD - mean * mean, 0.f) + eps)};
}

template <class T, int EPI, int WM, int WN, int MI, int NI, int STAGES, bool CONV, int BK = 64, int SCHED = 0, int WK = 1, bool KDUAL = false, bool KSER = false>
int launch_cfg(const KArgs& g, hipStream_t s, const char* what) {
  constexpr int BM = WM * MI * 16, BN = WN * NI * 16;
  constexpr int RING = WK * STAGES * (BM + BN) * BK * 2, EPIB = WM * WN * epi_stage_bytes(MI, NI, WM * WN > 8 ? 8 : 16);
  constexpr int LDS = (RING > EPIB ? RING : EPIB) + BM * 8 + BN * 8;     // + row statistics + column vectors
  static_assert(LDS <= 160 * 1024, "fits the CU's LDS");
  if ((WK > 1 || KDUAL) && (g.K / BK / (EPI == HM_EPI_F32 ? g.ksplit : (KSER ? g.kser : 1))) % 2 != 0) return hm_set_error(HM_ERR_ARG, "gemm: K tiles do not split over the K groups");
  if (KSER && (g.kser < 1 || (g.K / BK) % g.kser != 0)) return hm_set_error(HM_ERR_ARG, "gemm: K tiles do not split over the serial ranges");
  auto kern = gemm_tn_kernel<T, EPI, WM, WN, MI, NI, STAGES, CONV, BK, SCHED, WK, KDUAL, KSER>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "gemm: cannot raise the dynamic LDS limit")) return rc;
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * (EPI == HM_EPI_F32 ? g.ksplit : 1);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WM * WN * WK), LDS, s, g);
  return hm_check_launch(what);
}

// Tile configurations of the plain GEMM (hm_gemm).  HM_GEMM_VARIANT / hm_gemm_set_variant select one
// for tuning runs; the default is chosen per shape in pick_variant().
int g_group_m = 8;
int g_variant = -2;    // -2: read HM_GEMM_VARIANT on first use; -1: per-shape default

// what the 32-bit lane offsets of the asm-issued copies (gemm_x3_kernel and the experiments built like it) require
bool off32_ok(const KArgs& g) { return (size_t)g.M * g.ldx * 2 < (1ull << 32) && (size_t)g.N * g.ldw * 2 < (1ull << 32); }

template <class T, int EPI>
int launch_gemm(const KArgs& g, int variant, hipStream_t s) {
  switch (variant) {
    // ---- the product's tiles: what pick_variant() can choose
    case 0: return launch_cfg<T, EPI, 2, 2, 4, 4, 2, false>(g, s, "hm_gemm");          // 128x128, 4 waves, 2 stages (64 KB, 2 blocks/CU)
    case 10: return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 2>(g, s, "hm_gemm");  // 256x256, waves 4x2 (64x128 each), 2 stages, s_setprio, loads between the sub-steps
    case 26:                                                                           // persistent 256x256 (gemm_px_kernel), else as 24
      if constexpr (EPI == HM_EPI_STORE || EPI == HM_EPI_GELU) {
        if (px_ok(g)) return launch_px<T, EPI>(g, s);
      }
      [[fallthrough]];
    case 24:                                                                           // X two K-steps ahead in a 3-slot ring (gemm_x3_kernel)
      if constexpr (EPI == HM_EPI_RESID_F32) {
        if (off32_ok(g) && rin_ok(g)) return launch_rin<T>(g, s);   // residual rows fetched inside the K loop (gemm_x3r_kernel)
      }
      if constexpr (EPI == HM_EPI_STORE || EPI == HM_EPI_GELU || EPI == HM_EPI_RESID_F32) {
        if (off32_ok(g)) return launch_rs<T, EPI>(g, s);
      }
      return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 2>(g, s, "hm_gemm");
#ifdef HM_ABLATIONS
    // ---- experiments (correct results, measured and not adopted: DESIGN.md section 4); tools and opt-in tests only
    case 35:                                                                           // gemm_px_kernel with clock stamps (diagnostic)
      if constexpr (EPI == HM_EPI_STORE || EPI == HM_EPI_GELU) {
        if (px_ok(g)) return launch_px<T, EPI, 4>(g, s);
      }
      return hm_set_error(HM_ERR_ARG, "hm_gemm: variant 35 exists for the persistent kernel's shapes only");
    case 27: case 33: case 34: case 36: case 37: case 38: case 39: case 40:                                                         // persistent 256x256, software-pipelined K loop (gemm_pp_kernel; 33: + copy stagger); 34: gemm_px_kernel with all copies issued by waves 0..3; else as 24
      if constexpr (EPI == HM_EPI_STORE || EPI == HM_EPI_GELU) {
        if (px_ok(g)) return variant == 40 ? launch_px<T, EPI, 9>(g, s) : variant == 39 ? launch_px<T, EPI, 8>(g, s) : variant == 38 ? launch_px<T, EPI, 7>(g, s) : variant == 37 ? launch_px<T, EPI, 6>(g, s) : variant == 36 ? launch_px<T, EPI, 5>(g, s) : variant == 34 ? launch_px<T, EPI, 3>(g, s) : (variant == 33 ? launch_px<T, EPI, 2>(g, s) : launch_px<T, EPI, 1>(g, s));   // (36: gemm_px_kernel with round 3's copy issue -- pointer casts, M0 saved and restored)
      }
      if constexpr (EPI == HM_EPI_RESID_F32) {
        if (off32_ok(g) && rin_ok(g)) return launch_rin<T>(g, s);
      }
      if constexpr (EPI == HM_EPI_STORE || EPI == HM_EPI_GELU || EPI == HM_EPI_RESID_F32) {
        if (off32_ok(g)) return launch_rs<T, EPI>(g, s);
      }
      return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 2>(g, s, "hm_gemm");
    case 1: return launch_cfg<T, EPI, 4, 2, 4, 4, 3, false>(g, s, "hm_gemm");   // 256x128, 8 waves, 3 stages (144 KB)
    case 2: return launch_cfg<T, EPI, 2, 4, 8, 4, 2, false>(g, s, "hm_gemm");   // 256x256, 8 waves, 2 stages (128 KB)
    case 3: return launch_cfg<T, EPI, 4, 2, 4, 4, 2, false>(g, s, "hm_gemm");   // 256x128, 8 waves, 2 stages (96 KB)
    case 4: return launch_cfg<T, EPI, 2, 2, 4, 4, 3, false>(g, s, "hm_gemm");   // 128x128, 4 waves, 3 stages (96 KB, 1 block/CU)
    case 5: return launch_cfg<T, EPI, 2, 2, 8, 4, 3, false>(g, s, "hm_gemm");   // 256x128, 4 waves (128x64 each), 3 stages (144 KB)
    case 6: return launch_cfg<T, EPI, 2, 4, 8, 4, 4, false, 32>(g, s, "hm_gemm");   // 256x256x32, 8 waves, 4 stages (128 KB)
    case 7: return launch_cfg<T, EPI, 2, 4, 8, 4, 3, false, 32>(g, s, "hm_gemm");   // 256x256x32, 8 waves, 3 stages (96 KB)
    case 8: return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false>(g, s, "hm_gemm");       // 256x256, waves 4x2 (64x128 each), 2 stages
    case 9: return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 1>(g, s, "hm_gemm");   // + s_setprio around the MFMA cluster
    case 11: return launch_cfg<T, EPI, 4, 2, 4, 8, 4, false, 32, 1>(g, s, "hm_gemm");  // 256x256x32, 4 stages, setprio
    case 12: return launch_cfg<T, EPI, 4, 4, 4, 4, 2, false, 64, 2>(g, s, "hm_gemm");  // 256x256 on SIXTEEN waves of 64x64 (4 per SIMD, <= 128 VGPRs), 2 stages
    case 21: return launch_cfg<T, EPI, 4, 2, 4, 4, 2, false, 32, 1>(g, s, "hm_gemm");  // 256x128x32, 8 waves, 2 stages (48 KB): 2 blocks/CU
    case 22: return launch_cfg<T, EPI, 4, 2, 4, 4, 3, false, 32, 1>(g, s, "hm_gemm");  // 256x128x32, 8 waves, 3 stages (72 KB): 2 blocks/CU
    case 23: return launch_cfg<T, EPI, 2, 2, 4, 4, 2, false, 32, 1>(g, s, "hm_gemm");  // 128x128x32, 4 waves, 2 stages (32 KB): 4 blocks/CU
    case 25: return launch_cfg<T, EPI, 4, 2, 4, 10, 2, false, 64, 2>(g, s, "hm_gemm"); // 256x320, waves 4x2 (64x160 each), 2 stages (144 KB)
    case 28:                                                                           // 256x160, both operands two steps ahead
      if constexpr (EPI == HM_EPI_STORE || EPI == HM_EPI_GELU || EPI == HM_EPI_RESID_F32) {
        if (off32_ok(g) && g.K >= 192) return launch_d2<T, EPI>(g, s);
      }
      return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 2>(g, s, "hm_gemm");
    case 29:                                                                           // 256x256 on four waves of 128x128 (gemm_w4_kernel)
      if constexpr (EPI == HM_EPI_STORE || EPI == HM_EPI_GELU || EPI == HM_EPI_RESID_F32) {
        if (off32_ok(g) && g.K >= 192) return launch_w4<T, EPI>(g, s);
      }
      return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 2>(g, s, "hm_gemm");
    // ---- timing ablations with WRONG results (tools/bench_gemm_ab.py only).  The w4 forms carry the guards of case 29:
    // gemm_w4_kernel's prologue issues K-steps 0..2 unconditionally and its lane offsets are 32-bit.
    case 30: case 31:
      if constexpr (EPI == HM_EPI_STORE) {
        if (!off32_ok(g) || g.K < 192) return hm_set_error(HM_ERR_ARG, "hm_gemm: ablations 30 / 31 need K >= 192 and operands below 4 GB");
        return variant == 30 ? launch_w4<T, EPI, 1>(g, s) : launch_w4<T, EPI, 2>(g, s);     // w4 without copies / copies only
      } else {
        return hm_set_error(HM_ERR_ARG, "hm_gemm: ablations 30 / 31 exist for the store epilogue only");
      }
    case 16: return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 94>(g, s, "hm_gemm"); // everything, but no wave waits for its copies
    case 17: return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 95>(g, s, "hm_gemm"); // ... and no barrier
    case 14: return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 92>(g, s, "hm_gemm"); // LDS-DMA + waits + barriers only
    case 15: return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 93>(g, s, "hm_gemm"); // ds_read + MFMA + barriers, no loads
    case 20: return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 96>(g, s, "hm_gemm"); // no epilogue
    case 18: return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 97>(g, s, "hm_gemm"); // every tile loads operand panel 0 (pure L2 hits)
    case 32: return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 98>(g, s, "hm_gemm"); // K loop on 32x32x16 MFMAs (half the MFMA issues), no epilogue
#endif
    default: return hm_set_error(HM_ERR_ARG, "hm_gemm: unknown tile variant");
  }
}

// The deferred-LayerNorm epilogues exist for the two one-tile production tiles (0, 10).
template <class T, int EPI>
int launch_gemm_ln(const KArgs& g, int variant, hipStream_t s) {
  if (variant == 0) return launch_cfg<T, EPI, 2, 2, 4, 4, 2, false>(g, s, "hm_gemm");
  return launch_cfg<T, EPI, 4, 2, 4, 8, 2, false, 64, 2>(g, s, "hm_gemm");
}

// Default tile choice (measured on MI355X, interleaved A/B on the ViT-H shapes, random data): the 256x256 tile halves
// the LDS-DMA bytes per flop of the 128x128 one (the K loop is bound by the global->LDS fill, not by the MFMA pipe) and
// wins by 15 % at M = 12288 -- as long as its tiles fill the CUs; small and mid-sized problems keep the 128x128 tile
// (less padding and round waste, 2 workgroups per CU).
bool variant_ok(int v) {
  if (v == -1 || v == 0 || v == 10 || v == 24 || v == 26) return true;
#ifdef HM_ABLATIONS
  if ((v >= 1 && v <= 12) || (v >= 14 && v <= 18) || v == 20 || (v >= 21 && v <= 23) || v == 25 || (v >= 27 && v <= 40)) return true;
#endif
  return false;
}

int pick_variant(const KArgs& g, int epilogue) {
  if (g_variant == -2) {
    const char* e = getenv("HM_GEMM_VARIANT");     // tuning runs only; an unknown or ablation id is ignored, never obeyed
    const int v = e ? atoi(e) : -1;
    g_variant = variant_ok(v) ? v : -1;
  }
  if (g_variant >= 0) return g_variant;
  if (g.M < 1024 || g.N < 512) return 0;
  const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
  if (hm_option(HM_OPT_GEMM_TILE_RULE) == 1) {
    // round 2's rule: the 256x256 tile only when its tiles fill whole rounds of the 256 CUs to 85 %
    const int rounds = (tiles + 255) / 256;
    return tiles * 100 < rounds * 256 * 85 ? 0 : 26;
  }
  // Round 3: whichever tile finishes sooner under a small model fitted to per-shape measurements at 16..96 hands
  // (tools/gpu/r03_s.sh, profiles/r03_gemm_tile_rule_sweep.log).  The 256 x 256 kernels take whole rounds of equal tiles --
  // the persistent gemm_px_kernel (16-bit store / GELU epilogues) on its grid of 248 or 256, the one-tile kernels (gemm_x3r /
  // gemm_x3: fp32 residual and the rest) one per CU -- at ~1000 TFLOP/s of a full round; the 128 x 128 tile has 512 slots that
  // refill as workgroups finish (fractional rounds, at least one) at ~700 (store epilogues) / ~520 (fp32 residual) of the same
  // unit (isolated launches put the residual shapes at 560; with two forwards in flight the other stream fills a big tile's idle
  // last round, and 72-80 hands ran 1-6 % faster with the big tile there).  Round 2's 85 % rule sent 40, 48, 56, 72 and 76 hands to the small tile for fc1 / qkv (fc1 at 72 hands: 1080 tiles =
  // 84 % of five rounds) and cost 9-15 % of the whole forward there (tools/probes/forward_vs_batch.py); 16-32 hands keep the
  // small tile for proj / fc2 (60-120 big tiles) under both rules.
  int cus = hm_device_cu_count();
  if (cus <= 0) cus = 256;
  const bool persistent = epilogue == HM_EPI_STORE || epilogue == HM_EPI_GELU;
  const int big_slots = persistent ? px_default_grid(tiles, cus) : cus;
  const int small_tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
  const double big_rounds = (double)((tiles + big_slots - 1) / big_slots);
  double small_rounds = (double)small_tiles / (2.0 * cus);
  if (small_rounds < 1.0) small_rounds = 1.0;
  const double t_big = big_rounds / 1000.0, t_small = small_rounds * 0.5 / (epilogue == HM_EPI_RESID_F32 ? 520.0 : 700.0);
  return t_big <= t_small ? 26 : 0;
  // 26: persistent 256x256 (gemm_px_kernel) for the 16-bit store epilogues on whole tiles; otherwise the one-tile kernel with X
  // two K-steps ahead (gemm_x3_kernel / gemm_x3r_kernel, variant 24), else the variant-10 tile.  Measured at B = 64, fp16
  // (interleaved A/B, us per launch): fc1 172 (24) / 164 (25: 256x320, 3.0 rounds and 10 % fewer operand bytes per flop) / 162
  // (26); kv 181 / 180 / 174; qkv 119 / 143 / 119.
}

template <class T>
int launch_gemm_epi(const KArgs& g, int epilogue, hipStream_t s) {
  const int v = pick_variant(g, epilogue);
  switch (epilogue) {
    case HM_EPI_STORE: return launch_gemm<T, HM_EPI_STORE>(g, v, s);
    case HM_EPI_GELU: return launch_gemm<T, HM_EPI_GELU>(g, v, s);
    case HM_EPI_RESID_F32: return launch_gemm<T, HM_EPI_RESID_F32>(g, v, s);
    case HM_EPI_F32: return launch_gemm<T, HM_EPI_F32>(g, v, s);
    case HM_EPI_SILU: return launch_gemm<T, HM_EPI_SILU>(g, v, s);
    case HM_EPI_RESID_LN: return launch_gemm_ln<T, HM_EPI_RESID_LN>(g, v, s);
    case HM_EPI_LN_STORE: return launch_gemm_ln<T, HM_EPI_LN_STORE>(g, v, s);
    case HM_EPI_LN_GELU: return launch_gemm_ln<T, HM_EPI_LN_GELU>(g, v, s);
    default: return hm_set_error(HM_ERR_ARG, "hm_gemm: unknown epilogue");
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Direct 3x3 convolution for the first layer of the detector (round 3): 3(8) -> 32 at 384 x 640 -- 6.5 % of the conv stack's time for
// no flops to speak of.  As an implicit GEMM it is 30,720 workgroups of two K-steps each: all prologue and epilogue, at 3x its
// HBM time.  (Written for Cin 8 / 32 / 64; the two layers behind the first were measured with it too and stay on the implicit
// GEMM, see try_conv_direct.)  Here:
//   * the WHOLE weight matrix ([Cout][9 Cin], <= 72 KB) sits in LDS for the life of a persistent workgroup (rows padded by 16 B:
//     the 16 rows of an A fragment fall on 16 different 16-byte slots of the 256-byte bank row);
//   * activations never touch LDS: the MFMA B operand of a K-step is, per lane, 16 contiguous bytes of ONE input pixel (8
//     channels of tap (ky, kx)), so every lane loads its fragment straight from global memory -- 16 lanes x consecutive
//     pixels = whole lines, the nine taps of a pixel hit L1 / L2 after the first;
//   * a wave computes 64 consecutive output pixels of one row (4 tiles of 16) x all Cout channels, so a weight fragment read
//     from LDS feeds 4 MFMAs, and its output is contiguous in NHWC; lane-swap epilogue (v_permlane16_swap) -> 16-byte stores.
// K order, MFMA and epilogue arithmetic are the implicit GEMM's: results are bit-identical to it (test_conv_direct_stem_...).
template <class T, int CIN, int COUT, int STRIDE, int ACT>
__global__ __launch_bounds__(256, 2) void conv3x3_direct_kernel(const KArgs g) {
  using elem = typename T::elem;
  using vec8 = typename T::vec8;
  constexpr int NKS = (9 * CIN + 31) / 32;              // K-steps of 32: 3 (Cin 8: four taps per step), 9, 18
  constexpr int WROW = NKS * 64 + 16;                   // bytes per weight row in LDS (padded)
  constexpr int NT = COUT / 16, MP = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g4 = lane >> 4;
  // weights -> LDS (K index = tap * CIN + ci as stored; columns >= 9 CIN of the padded rows are zero in the tensor itself)
  for (int c = tid; c < COUT * NKS * 4; c += 256) {
    const int row = c / (NKS * 4), ch = c - row * (NKS * 4);
    *(vec8*)(smem + row * WROW + ch * 16) = *(const vec8*)((const elem*)g.W + (size_t)row * g.ldw + ch * 8);
  }
  f32x4_t bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bv[nt] = *(const f32x4_t*)(g.bias + nt * 16 + 4 * g4);
  __syncthreads();
  vec8 zero8;
#pragma unroll
  for (int q = 0; q < 8; ++q) zero8[q] = (elem)0.0f;
  const int xblocks = (g.Wout + 63) / 64;
  const int groups = (g.M / g.Wout) * xblocks;          // (image, output row) pairs x 64-pixel blocks
  const elem* X = (const elem*)g.X;
  // XCD-aware order: workgroups b, b + 8, .. share an XCD and its L2 -- give them consecutive rows, so that the three output rows
  // that read an input row do so through one L2 (dealt round robin every XCD fetched its own copy: 3.1x the input, PMC)
  for (int grp = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave; grp < groups; grp += gridDim.x * 4) {
    const int rowi = grp / xblocks, x0 = (grp - rowi * xblocks) * 64;
    const int n = rowi / g.Hout, oy = rowi - n * g.Hout;
    const elem* img = X + (size_t)n * g.H * g.Wd * g.ldx;
    f32x4_t acc[MP][NT];
#pragma unroll
    for (int p = 0; p < MP; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[p][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // this lane's B fragment of K-step ks for pixel tile p: 8 channels of one tap of pixel x0 + 16 p + li
    auto load_b = [&](int ks, vec8 (&xf)[MP]) {
      int tap, coff;
      if (CIN == 8) { tap = 4 * ks + g4; coff = 0; }
      else if (CIN == 32) { tap = ks; coff = 8 * g4; }
      else { tap = ks >> 1; coff = (ks & 1) * 32 + 8 * g4; }
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int iy = oy * STRIDE - 1 + ky;
      const bool rowok = tap < 9 && iy >= 0 && iy < g.H;
#pragma unroll
      for (int p = 0; p < MP; ++p) {
        const int ox = x0 + 16 * p + li, ix = ox * STRIDE - 1 + kx;
        const bool ok = rowok && ix >= 0 && ix < g.Wd && ox < g.Wout;
        xf[p] = ok ? *(const vec8*)(img + ((size_t)iy * g.Wd + ix) * g.ldx + coff) : zero8;
      }
    };
    vec8 xa[MP], xb[MP];
    load_b(0, xa);
    auto kstep = [&](int ks, const vec8 (&xf)[MP]) {
      vec8 wf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wf[nt] = *(const vec8*)(smem + (nt * 16 + li) * WROW + ks * 64 + g4 * 16);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < MP; ++p) acc[p][nt] = T::mfma(wf[nt], xf[p], acc[p][nt]);
    };
#pragma unroll 1
    for (int ks = 0; ks < NKS; ks += 2) {                // next K-step's fragments in flight while this one's MFMAs run (a real
                                                         // loop: unrolled, hipcc hoists every load and takes 256 registers)
      if (ks + 1 < NKS) load_b(ks + 1, xb);
      kstep(ks, xa);
      if (ks + 1 < NKS) {
        if (ks + 2 < NKS) load_b(ks + 2, xa);
        kstep(ks + 1, xb);
      }
    }
    // epilogue: + bias, activation, 16-bit; pairs of channel tiles -> 8 consecutive channels per lane by one lane swap per register
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    elem* yrow = (elem*)g.C + (size_t)rowi * g.Wout * g.ldc;
#pragma unroll
    for (int p = 0; p < MP; ++p) {
      const int ox = x0 + 16 * p + li;
#pragma unroll
      for (int np = 0; np < NT / 2; ++np) {
        unsigned pk[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4_t a = acc[p][2 * np + h], bb = bv[2 * np + h];
          typename T::vec4 o;
          f32x4_t v4 = add4(a, bb);
          if (ACT == 1 || ACT == 3) v4 = silu4(v4);
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (elem)(ACT == 2 ? fmaxf(v4[q], 0.f) : v4[q]);
          const u32x2 w = __builtin_bit_cast(u32x2, o);
          pk[h][0] = w[0]; pk[h][1] = w[1];
        }
        const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        if (ox < g.Wout && (ACT != 3 || g.kser == 0x7fff))          // (ACT 3: bound diagnosis of the experiments library, nothing is stored)
          *(u32x4*)(yrow + (size_t)ox * g.ldc + (2 * np + (g4 & 1)) * 16 + (g4 >> 1) * 8) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
    }
  }
}

template <class T, int CIN, int COUT, int STRIDE, int ACT>
int launch_conv_direct(const KArgs& g, hipStream_t s) {
  constexpr int NKS = (9 * CIN + 31) / 32, LDS = COUT * (NKS * 64 + 16);
  auto kern = conv3x3_direct_kernel<T, CIN, COUT, STRIDE, ACT>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "hm_conv2d_nhwc: cannot raise the dynamic LDS limit")) return rc;
  int cus = hm_device_cu_count();
  if (cus <= 0) cus = 256;
  const int groups = (g.M / g.Wout) * ((g.Wout + 63) / 64);
  const int per_cu = LDS > 64 * 1024 ? 2 : 4;                  // resident workgroups per CU (LDS-limited for the 72 KB matrix)
  int grid = (groups + 3) / 4;
  if (grid > cus * per_cu) grid = cus * per_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, s, g);
  return hm_check_launch("hm_conv2d_nhwc (direct 3x3)");
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3, stride 1, 64 -> 64 channels on the large maps (round 3): the second stem layer at 384 x 640 and the 3x3s of the first two
// E-ELAN stages -- eight layers, 13 % of the conv stack's time at a quarter of the rate of the deep layers.  As an implicit
// GEMM (K = 576: nine K-steps) every output tile re-fetches each input pixel nine times through LDS-DMA and streams 72 KB of
// weights per 128 pixels, with a prologue and an epilogue per nine steps.  Here:
//   * the weights never touch LDS: a wave owns 32 output channels and keeps their whole [32][576] slice as MFMA A fragments in
//     registers (36 fragments = 144 VGPRs) for the life of a persistent workgroup;
//   * the input is staged ONCE per output tile: the 10 x 18 pixel halo of an 8 x 16 pixel tile (180 rows of 128 bytes, the
//     GEMM's 16-byte-chunk XOR swizzle) comes in by LDS-DMA, double-buffered -- the next tile's halo is fetched while this
//     tile's nine taps run; the B fragment of (tap, pixel row) is the same LDS rows shifted by (ky, kx).  The swizzle key of a
//     fragment is (lane + m) & 7 with m a compile-time constant of (tap, pixel row), and the second K sub-step's key is m + 4:
//     eight lane addresses, computed once, and an immediate offset serve all 72 reads of a tile;
//   * the copy addresses are tile-invariant too (per-lane byte offsets from the tile's first halo pixel, one register per copy;
//     a tile on the map's border takes a slower per-lane path that substitutes the zero line);
//   * one barrier per TILE (none per K-step): within a tile nothing is in flight that the MFMAs wait for;
//   * the wait before that barrier is counted: the four stores of the previous tile's epilogue are the newest operations and may
//     stay in flight (whole tiles; a tile that overhangs the map masks lanes, so it takes vmcnt(0)).
// Four waves = 2 pixel groups (4 map rows each) x 2 channel halves; wave tile 64 pixels x 32 channels.  K order (tap-major, 32
// deep MFMA steps) and the epilogue arithmetic are the implicit GEMM's: results are bit-identical to it.
__device__ __forceinline__ void glds16_hidden_v(const void* src, void* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory");
}

// asm_store_note: the epilogues of the kernels below issue their 16-byte stores from inline asm (uniform base + lane offset).  A
// store of more than 64 bits followed by a write of its data registers needs a wait state that hipcc's hazard recogniser inserts
// for ITS stores but cannot for one it does not see: without the s_nop, a short epilogue (conv1x1_wreg_kernel, K = 128) let the
// moves that assemble the next store's data overtake the read of the previous one's (lanes 12-15 of a row, first dwords wrong).
template <class T, int ACT>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(const KArgs g) {
  using elem = typename T::elem;
  using vec8 = typename T::vec8;
  constexpr int TH = 8, TW = 16, HWID = TW + 2, HPIX = (TH + 2) * HWID;
  constexpr int RING = 2;                               // fragment sets in flight (3 = two steps ahead: 16 more registers)
  constexpr int PIECES = (HPIX + 7) / 8, HBUF = PIECES * 1024, NJ = (PIECES + 3) / 4;   // 180 px -> 23 pieces of 8 rows x 128 B; 6 copies per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];                 // 2 x HBUF | bias [64] f32
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g4 = lane >> 4, pg = wave >> 1, chh = wave & 1;
  const elem* X = (const elem*)g.X;
  // this wave's weight slice -> registers: fragment (tap, ks, ni) = rows 32 chh + 16 ni + li, K = 64 tap + 32 ks + 8 g4 .. +7
  vec8 wf[9][2][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        wf[tap][ks][ni] = *(const vec8*)((const elem*)g.W + (size_t)(32 * chh + 16 * ni + li) * g.ldw + 64 * tap + 32 * ks + 8 * g4);
  if (tid < 64) ((float*)(smem + 2 * HBUF))[tid] = g.bias[tid];

  const int tiles_x = (g.Wout + TW - 1) / TW, tiles_y = (g.Hout + TH - 1) / TH, tpi = tiles_x * tiles_y;
  const int ntiles = (g.M / (g.Hout * g.Wout)) * tpi;
  // LDS-DMA copy j of this wave = piece 4 j + wave = halo pixels 8 (4 j + wave) + (lane >> 3), physical chunk lane & 7 = logical
  // chunk (lane & 7) ^ (pixel & 7).  voff[j]: byte offset of that source chunk from the tile's first halo pixel (tile-invariant)
  unsigned voff[NJ];
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int h = 8 * (4 * j + wave) + srow, hy = h / HWID, hx = h - hy * HWID;
    voff[j] = h < HPIX ? (unsigned)(((hy * g.Wd + hx) * g.ldx + schunk * 8) * 2) : 0u;        // (rows past the halo: any valid address)
  }
  auto issue_halo = [&](int tile, int buf) {
    const int n = tile / tpi, rem = tile - n * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int y0 = ty * TH - 1, x0 = tx * TW - 1;
    char* dst = smem + buf * HBUF;
    if (y0 >= 0 && x0 >= 0 && y0 + TH + 2 <= g.H && x0 + HWID <= g.Wd) {      // the whole halo lies inside the map (uniform)
      const char* base = (const char*)(X + (((size_t)n * g.H + y0) * g.Wd + x0) * g.ldx);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (4 * j + wave < PIECES) glds16_hidden_s(base, voff[j], dst + (4 * j + wave) * 1024);
    } else {
      const elem* img = X + (size_t)n * g.H * g.Wd * g.ldx;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (4 * j + wave < PIECES) {
          const int h = 8 * (4 * j + wave) + srow, hy = h / HWID, hx = h - hy * HWID;
          const int iy = y0 + hy, ix = x0 + hx;
          const bool ok = h < HPIX && iy >= 0 && iy < g.H && ix >= 0 && ix < g.Wd;
          const elem* src = ok ? img + ((size_t)iy * g.Wd + ix) * g.ldx + schunk * 8 : (const elem*)g.zeros;
          glds16_hidden_v(src, dst + (4 * j + wave) * 1024);
        }
    }
  };
  // fragment read addresses: pixel row (4 pg + mi + ky) * 18 + kx + li of the halo, chunk (4 ks + g4) ^ (row & 7); row & 7 =
  // (li + m) & 7 with m = (2 (mi + ky) + kx) & 7 known at compile time (4 pg * 18 = 0 mod 8), and (li + m + 4) & 7 flips the
  // key's bit 2 exactly as ks = 1 flips the chunk's: faddr[(m + 4 ks) & 7] + the compile-time row offset addresses every fragment
  unsigned faddr[8];
#pragma unroll
  for (int m = 0; m < 8; ++m)
    faddr[m] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem +
               (unsigned)(((4 * pg) * HWID + li) * 128 + ((g4 ^ ((li + m) & 7)) << 4));

  const unsigned yoff = (unsigned)((li * g.ldc + (g4 & 1) * 16 + (g4 >> 1) * 8) * 2);   // output: pixel li of a row, this lane's 8 channels
  int tile = xcd_remap(blockIdx.x, gridDim.x), buf = 0;     // neighbouring tiles (shared halo columns / rows) through one XCD's L2
  bool whole = false;
  if (tile < ntiles) issue_halo(tile, 0);
  // the weight loads complete HERE (hipcc would otherwise wait for them one by one inside the tile loop, with counts that also
  // cover -- and so serialise -- the halo copies and the stores)
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) asm volatile("" : "+v"(wf[tap][ks][ni]));
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    // this tile's halo (issued a tile ago) has landed; the previous epilogue's stores (the newest four) may stay in flight
    if (whole) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (also the first tile: `whole` starts false)
    __builtin_amdgcn_s_barrier();                      // everyone's pieces landed; everyone is done reading the other buffer
    if (tile + (int)gridDim.x < ntiles) issue_halo(tile + gridDim.x, buf ^ 1);
    const int n = tile / tpi, rem = tile - n * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    f32x4_t acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // 18 steps (tap, ks) of 4 fragment reads + 8 MFMAs.  The reads are issued from inline asm TWO steps ahead into a ring of three
    // fragment sets and waited for by count (LDS operations retire in order): left to hipcc every read sinks to just before its
    // first use and the MFMAs wait out the LDS latency; fully hoisted they spill.  The wait statement names the registers it
    // releases as read-write operands, so no MFMA that reads them can be scheduled above it.
    vec8 xr[RING][4];
    auto rd = [&](auto sc) {
      constexpr int s = decltype(sc)::value, tap = s >> 1, ks = s & 1, r = s % RING, ky = tap / 3, kx = tap % 3;
      vec8 (&x)[4] = xr[r];                             // (asm operands inside a generic lambda must be named through a local)
      unsigned (&fa)[8] = faddr;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[0]) : "v"(fa[(2 * (0 + ky) + kx + 4 * ks) & 7]), "n"(((0 + ky) * HWID + kx) * 128));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[1]) : "v"(fa[(2 * (1 + ky) + kx + 4 * ks) & 7]), "n"(((1 + ky) * HWID + kx) * 128));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[2]) : "v"(fa[(2 * (2 + ky) + kx + 4 * ks) & 7]), "n"(((2 + ky) * HWID + kx) * 128));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[3]) : "v"(fa[(2 * (3 + ky) + kx + 4 * ks) & 7]), "n"(((3 + ky) * HWID + kx) * 128));
    };
    auto step = [&](auto sc) {
      constexpr int s = decltype(sc)::value, r = s % RING, AHEAD = RING - 1;
      if constexpr (s + AHEAD < 18) rd(std::integral_constant<int, s + AHEAD>{});
      vec8 (&x)[4] = xr[r];
      constexpr int newer = (18 - 1 - s < AHEAD ? 18 - 1 - s : AHEAD) * 4;       // reads of later steps that may stay in flight
      if constexpr (newer == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
      else if constexpr (newer == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = T::mfma(wf[s >> 1][s & 1][ni], xr[r][mi], acc[mi][ni]);
    };
    rd(std::integral_constant<int, 0>{});
    if constexpr (RING == 3) rd(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
    step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
    step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{});
    step(std::integral_constant<int, 15>{}); step(std::integral_constant<int, 16>{}); step(std::integral_constant<int, 17>{});
    // epilogue: + bias, activation, 16-bit; the two channel tiles -> 8 consecutive channels per lane by one lane swap per register
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int ox = tx * TW + li;
    whole = (ty + 1) * TH <= g.Hout && (tx + 1) * TW <= g.Wout;
    char* ytile = (char*)g.C + ((((size_t)n * g.Hout + ty * TH + 4 * pg) * g.Wout + tx * TW) * g.ldc + 32 * chh) * 2;
    f32x4_t bv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) bv[h] = *(const f32x4_t*)(smem + 2 * HBUF + (32 * chh + 16 * h + 4 * g4) * 4);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int oy = ty * TH + 4 * pg + mi;
      unsigned pk[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4_t a = acc[mi][h], bb = bv[h];
        typename T::vec4 o;
        f32x4_t v4 = add4(a, bb);
        if (ACT == 1) v4 = silu4(v4);
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (elem)(ACT == 2 ? fmaxf(v4[q], 0.f) : v4[q]);
        const u32x2 w = __builtin_bit_cast(u32x2, o);
        pk[h][0] = w[0]; pk[h][1] = w[1];
      }
      const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
      const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
      if (whole || (ox < g.Wout && oy < g.Hout)) {      // uniform row base + this lane's tile-invariant 32-bit offset (from asm:
        const u32x4 o = u32x4{s0[0], s1[0], s0[1], s1[1]};   // hipcc would add them into a 64-bit register pair per store)
        const char* yrow = ytile + (size_t)mi * g.Wout * g.ldc * 2;
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(yoff), "v"(o), "s"(yrow) : "memory");   // (s_nop: see asm_store_note)
      }
    }
    {                                                  // the fragment addresses follow the halo buffer
      const unsigned d = buf ? (unsigned)-HBUF : (unsigned)HBUF;
#pragma unroll
      for (int m = 0; m < 8; ++m) faddr[m] += d;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3, STRIDE 2, 32 -> 64 channels (round 3): the layer behind the stem (384 x 640 -> 192 x 320), 252 MB in and 126 MB out for
// 36 GFLOP -- as an implicit GEMM 122 us for 16 frames, 2.6x its HBM time.  The scheme of conv3x3_c64_kernel with two changes:
//   * the 17 x 33 pixel halo of an 8 x 16 output tile is staged as TWO planes, even and odd input columns, so that the 16 pixels of
//     a fragment (input columns 2 x + kx) are consecutive 64-byte rows of one plane (kx = 0: even plane, 1: odd plane, 2: even
//     plane, one entry on);
//   * 64-byte rows (32 channels = one 32-deep MFMA step per tap) put four rows in a 256-byte bank row, and the swizzle that keeps
//     ds_read_b128's four non-contiguous 16-lane groups conflict-free is: 16-byte chunk c of row e sits at c ^ 2 ((e >> 2) & 1)
//     (found by exhaustive search over chunk permutations keyed on e; the GEMM's c ^ (e & 7) does not apply to 4-chunk rows).
//     (e >> 2) & 1 of a fragment's row is ((lane + m) >> 2) & 1 with m a compile-time constant of (tap, pixel row): again eight
//     precomputed lane addresses and an immediate offset serve every read.
// Weights (this wave's 32 channels x 288) stay in registers (18 fragments), nine steps of 4 reads + 8 MFMAs per tile.  K order and
// epilogue arithmetic are the implicit GEMM's (its K-step of 64 is two taps, each one 32-deep MFMA): bit-identical to it.
template <class T, int ACT>
__global__ __launch_bounds__(256, 2) void conv3x3_s2c32_kernel(const KArgs g) {
  using elem = typename T::elem;
  using vec8 = typename T::vec8;
  constexpr int TH = 8, TW = 16, HH = 2 * TH + 1, HWD = 2 * TW + 1;        // halo 17 x 33 input pixels
  constexpr int PITCH = 17, PLANE = 296;                                   // entries per halo row of a plane; entries per plane (17 x 17 = 289 -> 296)
  constexpr int ENTRIES = 2 * PLANE, PIECES = ENTRIES / 16, HBUF = PIECES * 1024, NJ = (PIECES + 3) / 4;   // 592 rows of 64 B = 37 pieces
  static_assert(ENTRIES % 16 == 0 && PLANE % 8 == 0, "whole 1-KB pieces; plane offsets keep the swizzle phase");
  extern __shared__ __attribute__((aligned(16))) char smem[];                 // 2 x HBUF | bias [64] f32
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g4 = lane >> 4, pg = wave >> 1, chh = wave & 1;
  const elem* X = (const elem*)g.X;
  vec8 wf[9][2];                                       // fragment (tap, ni): rows 32 chh + 16 ni + li, K = 32 tap + 8 g4 .. +7
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
      wf[tap][ni] = *(const vec8*)((const elem*)g.W + (size_t)(32 * chh + 16 * ni + li) * g.ldw + 32 * tap + 8 * g4);
  if (tid < 64) ((float*)(smem + 2 * HBUF))[tid] = g.bias[tid];

  const int tiles_x = (g.Wout + TW - 1) / TW, tiles_y = (g.Hout + TH - 1) / TH, tpi = tiles_x * tiles_y;
  const int ntiles = (g.M / (g.Hout * g.Wout)) * tpi;
  // copy j of this wave = piece 4 j + wave = rows 16 (4 j + wave) + (lane >> 2), physical chunk lane & 3
  const int srow = lane >> 2;
  auto entry_geom = [&](int e, int& hy, int& hx, bool& real) {
    const int pl = e >= PLANE ? 1 : 0, r = e - pl * PLANE;
    hy = r / PITCH;
    const int idx = r - hy * PITCH;
    hx = 2 * idx + pl;
    real = r < HH * PITCH && hx < HWD;
  };
  unsigned voff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int e = 16 * (4 * j + wave) + srow, c = (lane & 3) ^ (((e >> 2) & 1) << 1);
    int hy, hx; bool real;
    entry_geom(e, hy, hx, real);
    voff[j] = (real && e < ENTRIES) ? (unsigned)(((hy * g.Wd + hx) * g.ldx + c * 8) * 2) : 0u;
  }
  auto issue_halo = [&](int tile, int buf) {
    const int n = tile / tpi, rem = tile - n * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int y0 = 2 * ty * TH - 1, x0 = 2 * tx * TW - 1;
    char* dst = smem + buf * HBUF;
    if (y0 >= 0 && x0 >= 0 && y0 + HH <= g.H && x0 + HWD <= g.Wd) {          // the whole halo lies inside the map (uniform)
      const char* base = (const char*)(X + (((size_t)n * g.H + y0) * g.Wd + x0) * g.ldx);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (4 * j + wave < PIECES) glds16_hidden_s(base, voff[j], dst + (4 * j + wave) * 1024);
    } else {
      const elem* img = X + (size_t)n * g.H * g.Wd * g.ldx;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (4 * j + wave < PIECES) {
          const int e = 16 * (4 * j + wave) + srow, c = (lane & 3) ^ (((e >> 2) & 1) << 1);
          int hy, hx; bool real;
          entry_geom(e, hy, hx, real);
          const int iy = y0 + hy, ix = x0 + hx;
          const bool ok = real && iy >= 0 && iy < g.H && ix >= 0 && ix < g.Wd;
          const elem* src = ok ? img + ((size_t)iy * g.Wd + ix) * g.ldx + c * 8 : (const elem*)g.zeros;
          glds16_hidden_v(src, dst + (4 * j + wave) * 1024);
        }
    }
  };
  // fragment of (tap, pixel row mi): plane kx & 1, rows (2 (4 pg + mi) + ky) * 17 + (kx == 2) + li, chunk g4 ^ 2 ((row >> 2) & 1);
  // (row >> 2) & 1 = ((li + m) >> 2) & 1 with m = ((2 mi + ky) * 17 + (kx == 2)) & 7 (8 pg * 17 and the plane offset are 0 mod 8)
  unsigned faddr[8];
#pragma unroll
  for (int m = 0; m < 8; ++m)
    faddr[m] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem +
               (unsigned)((8 * pg * PITCH + li) * 64 + ((g4 ^ ((((li + m) >> 2) & 1) << 1)) << 4));
  const unsigned yoff = (unsigned)((li * g.ldc + (g4 & 1) * 16 + (g4 >> 1) * 8) * 2);

  int tile = xcd_remap(blockIdx.x, gridDim.x), buf = 0;
  bool whole = false;
  if (tile < ntiles) issue_halo(tile, 0);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) asm volatile("" : "+v"(wf[tap][ni]));      // the weight loads complete here (see conv3x3_c64_kernel)
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    if (whole) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + (int)gridDim.x < ntiles) issue_halo(tile + gridDim.x, buf ^ 1);
    const int n = tile / tpi, rem = tile - n * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    f32x4_t acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    vec8 xr[3][4];
    auto rd = [&](auto sc) {
      constexpr int tap = decltype(sc)::value, r = tap % 3, ky = tap / 3, kx = tap % 3;
      vec8 (&x)[4] = xr[r];
      unsigned (&fa)[8] = faddr;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[0]) : "v"(fa[((0 + ky) * PITCH + (kx == 2)) & 7]), "n"(((kx & 1) * PLANE + (0 + ky) * PITCH + (kx == 2)) * 64));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[1]) : "v"(fa[((2 + ky) * PITCH + (kx == 2)) & 7]), "n"(((kx & 1) * PLANE + (2 + ky) * PITCH + (kx == 2)) * 64));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[2]) : "v"(fa[((4 + ky) * PITCH + (kx == 2)) & 7]), "n"(((kx & 1) * PLANE + (4 + ky) * PITCH + (kx == 2)) * 64));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[3]) : "v"(fa[((6 + ky) * PITCH + (kx == 2)) & 7]), "n"(((kx & 1) * PLANE + (6 + ky) * PITCH + (kx == 2)) * 64));
    };
    auto step = [&](auto sc) {
      constexpr int tap = decltype(sc)::value, r = tap % 3;
      if constexpr (tap + 2 < 9) rd(std::integral_constant<int, tap + 2>{});
      vec8 (&x)[4] = xr[r];
      constexpr int newer = (9 - 1 - tap < 2 ? 9 - 1 - tap : 2) * 4;
      if constexpr (newer == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
      else if constexpr (newer == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = T::mfma(wf[tap][ni], xr[r][mi], acc[mi][ni]);
    };
    rd(std::integral_constant<int, 0>{});
    rd(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int ox = tx * TW + li;
    whole = ACT != 3 && (ty + 1) * TH <= g.Hout && (tx + 1) * TW <= g.Wout;
    char* ytile = (char*)g.C + ((((size_t)n * g.Hout + ty * TH + 4 * pg) * g.Wout + tx * TW) * g.ldc + 32 * chh) * 2;
    f32x4_t bv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) bv[h] = *(const f32x4_t*)(smem + 2 * HBUF + (32 * chh + 16 * h + 4 * g4) * 4);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int oy = ty * TH + 4 * pg + mi;
      unsigned pk[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4_t a = acc[mi][h], bb = bv[h];
        typename T::vec4 o;
        f32x4_t v4 = add4(a, bb);
        if (ACT == 1 || ACT == 3) v4 = silu4(v4);
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (elem)(ACT == 2 ? fmaxf(v4[q], 0.f) : v4[q]);
        const u32x2 w = __builtin_bit_cast(u32x2, o);
        pk[h][0] = w[0]; pk[h][1] = w[1];
      }
      const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
      const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
      if ((whole || (ox < g.Wout && oy < g.Hout)) && (ACT != 3 || g.kser == 0x7fff)) {
        const u32x4 o = u32x4{s0[0], s1[0], s0[1], s1[1]};
        const char* yrow = ytile + (size_t)mi * g.Wout * g.ldc * 2;
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(yoff), "v"(o), "s"(yrow) : "memory");   // (s_nop: see asm_store_note)
      }
    }
    {
      const unsigned d = buf ? (unsigned)-HBUF : (unsigned)HBUF;
#pragma unroll
      for (int m = 0; m < 8; ++m) faddr[m] += d;
    }
  }
}

template <class T, int ACT>
int launch_conv_s2c32(const KArgs& g, hipStream_t s) {
  constexpr int LDS = 2 * 37 * 1024 + 256;
  auto kern = conv3x3_s2c32_kernel<T, ACT>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "hm_conv2d_nhwc: cannot raise the dynamic LDS limit")) return rc;
  int cus = hm_device_cu_count();
  if (cus <= 0) cus = 256;
  const int ntiles = (g.M / (g.Hout * g.Wout)) * ((g.Wout + 15) / 16) * ((g.Hout + 7) / 8);
  const int grid = ntiles < 2 * cus ? ntiles : 2 * cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, s, g);
  return hm_check_launch("hm_conv2d_nhwc (3x3 stride 2, 32 -> 64)");
}

// ---------------------------------------------------------------------------------------------------------------
// The two stem layers as ONE launch (round 4): 3(8) -> 32 at full size (3x3, stride 1) and 32 -> 64 (3x3, stride 2) behind it.
// Apart they are the two most expensive launches of a 48-frame pass (265 + 253 us of 5.4 ms) for 5 % of its flops: the first
// writes 755 MB, the second reads them again, and nobody else does.  With the stores removed the second layer alone runs in
// 165 us and the first in 219 (latency of its per-lane global loads; its SiLU is not what it waits for:
// profiles/r04_yolo_stem_bound.log).  Here the intermediate never leaves the CU: conv3x3_s2c32_kernel's two-plane halo of an
// 8 x 16 output tile (17 x 33 pixels x 32 channels, 37 KB) is PRODUCED in LDS instead of copied from HBM --
//   * the 19 x 35 input pixels under it (16 bytes each: 8 channels, 3 real) arrive by LDS-DMA, double-buffered, a tile ahead;
//   * a wave takes every fourth group of 16 halo entries: per group three 32-deep MFMA steps (K = tap * 8 + ci, four taps per
//     step, as conv3x3_direct_kernel orders it) x two channel tiles, the B fragment of a step being one ds_read_b128 per lane
//     (the 8 channels of input pixel (hy + ky, hx + kx)); + bias, SiLU, 16-bit, and the lane's four channels go to the entry's
//     row in the halo layout; entries outside the first layer's map (the second layer's zero padding) are written as zeros;
//   * barrier; then the second layer exactly as conv3x3_s2c32_kernel runs it (weights in registers, nine steps of 4 reads +
//     8 MFMAs, counted waits, lane-swap epilogue).
// Both layers keep their K order, MFMA shape and epilogue arithmetic, and the intermediate is rounded to 16 bits as the stored
// tensor was: the output is bit-identical to the two launches (test_stem_pair_bit_identical_to_two_launches).  ~9 % of the first
// layer is computed twice (halo overlap of neighbouring tiles).
struct StemPairArgs {
  const void* X; const void* W0; const float* b0; const void* W1; const float* b1; void* C; const void* zeros;
  int ldw0, ldw1, ldc, NB, H, Wd, Hout, Wout;
};

template <class T>
__global__ __launch_bounds__(256, 2) void conv_stem_pair_kernel(const StemPairArgs g) {
  using elem = typename T::elem;
  using vec8 = typename T::vec8;
  constexpr int TH = 8, TW = 16, HH = 2 * TH + 1, HWD = 2 * TW + 1;        // halo of first-layer outputs: 17 x 33
  constexpr int PITCH = 17, PLANE = 296, ENTRIES = 2 * PLANE, GROUPS = ENTRIES / 16, HBUF = ENTRIES * 64;   // as conv3x3_s2c32_kernel
  constexpr int IH = HH + 2, IW = HWD + 2, IPIX = IH * IW;                 // input pixels under the halo: 19 x 35 = 665
  constexpr int IPIECES = (IPIX + 63) / 64, IBUF = IPIECES * 1024, NJ = (IPIECES + 3) / 4;   // 11 pieces of 64 pixels; <= 3 copies per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];              // halo HBUF | input 2 x IBUF | bias1 [64] | bias0 [32] f32 | entry table [592] x 8 B
  char* const inb = smem + HBUF;
  float* const bias1 = (float*)(smem + HBUF + 2 * IBUF);
  float* const bias0 = bias1 + 64;
  char* const etab = (char*)(bias0 + 32);               // per halo entry: where its pixel sits in the input block, (hy, hx), flags
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g4 = lane >> 4, pg = wave >> 1, chh = wave & 1;
  const elem* X = (const elem*)g.X;
  // second layer's weight slice (this wave's 32 channels x 288) and the first layer's whole matrix (32 x 96) as A fragments
  vec8 wf[9][2], wf0[3][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
      wf[tap][ni] = *(const vec8*)((const elem*)g.W1 + (size_t)(32 * chh + 16 * ni + li) * g.ldw1 + 32 * tap + 8 * g4);
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      wf0[ks][nt] = *(const vec8*)((const elem*)g.W0 + (size_t)(16 * nt + li) * g.ldw0 + 32 * ks + 8 * g4);
  if (tid < 64) bias1[tid] = g.b1[tid];
  else if (tid < 96) bias0[tid - 64] = g.b0[tid - 64];
  for (int e = tid; e < ENTRIES; e += 256) {            // tile-invariant geometry of the halo entries (conv3x3_s2c32_kernel's two planes)
    const int pl = e >= PLANE ? 1 : 0, r = e - pl * PLANE, hy = r / PITCH, idx = r - hy * PITCH, hx = 2 * idx + pl;
    const bool real = r < HH * PITCH && hx < HWD;
    uint2 tb;
    tb.x = real ? (unsigned)((hy * IW + hx) * 16) : 0u;
    tb.y = real ? (unsigned)(hy | (hx << 8) | 0x10000 | (((e >> 2) & 1) << 24)) : (unsigned)(((e >> 2) & 1) << 24);
    *(uint2*)(etab + e * 8) = tb;
  }
  const unsigned wlane = (unsigned)(((g4 >> 1) << 4) + 8 * (g4 & 1));   // this lane's place in an entry's 64-byte row: chunk g4 >> 1 (+ 2 nt), half g4 & 1

  const int tiles_x = (g.Wout + TW - 1) / TW, tiles_y = (g.Hout + TH - 1) / TH, tpi = tiles_x * tiles_y;
  const int ntiles = g.NB * tpi;
  // input copy j of this wave = piece 4 j + wave = pixels 64 (4 j + wave) + lane of the 19 x 35 block, row-major, 16 B each
  unsigned voff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int q = 64 * (4 * j + wave) + lane, iy = (q * 1873) >> 16, ix = q - iy * IW;       // q / 35 for q < 720
    voff[j] = q < IPIX ? (unsigned)((iy * g.Wd + ix) * 16) : 0u;
  }
  auto issue_input = [&](int tile, int buf) {
    const int n = tile / tpi, rem = tile - n * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int y0 = 2 * ty * TH - 2, x0 = 2 * tx * TW - 2;                  // first input pixel of the block
    char* dst = inb + buf * IBUF;
    if (y0 >= 0 && x0 >= 0 && y0 + IH <= g.H && x0 + IW <= g.Wd) {        // the whole block lies inside the image (uniform)
      const char* base = (const char*)(X + (((size_t)n * g.H + y0) * g.Wd + x0) * 8);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (4 * j + wave < IPIECES) glds16_hidden_s(base, voff[j], dst + (4 * j + wave) * 1024);
    } else {
      const elem* img = X + (size_t)n * g.H * g.Wd * 8;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (4 * j + wave < IPIECES) {
          const int q = 64 * (4 * j + wave) + lane, iy = (q * 1873) >> 16, ix = q - iy * IW;
          const int yy = y0 + iy, xx = x0 + ix;
          const bool ok = q < IPIX && yy >= 0 && yy < g.H && xx >= 0 && xx < g.Wd;
          const elem* src = ok ? img + ((size_t)yy * g.Wd + xx) * 8 : (const elem*)g.zeros;
          glds16_hidden_v(src, dst + (4 * j + wave) * 1024);
        }
    }
  };
  // first layer: byte offset of tap (4 ks + g4) inside the input block; taps >= 9 (zero weight columns) re-read tap 8
  unsigned tapoff[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int tap = 4 * ks + g4 < 9 ? 4 * ks + g4 : 8, ky = tap / 3, kx = tap - 3 * ky;
    tapoff[ks] = (unsigned)((ky * IW + kx) * 16);
  }
  // second layer: fragment addresses as in conv3x3_s2c32_kernel (one halo buffer here)
  unsigned faddr[8];
#pragma unroll
  for (int m = 0; m < 8; ++m)
    faddr[m] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem +
               (unsigned)((8 * pg * PITCH + li) * 64 + ((g4 ^ ((((li + m) >> 2) & 1) << 1)) << 4));
  const unsigned yoff = (unsigned)((li * g.ldc + (g4 & 1) * 16 + (g4 >> 1) * 8) * 2);

  int tile = xcd_remap(blockIdx.x, gridDim.x), buf = 0;
  bool whole = false;
  if (tile < ntiles) issue_input(tile, 0);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) asm volatile("" : "+v"(wf[tap][ni]));      // the weight loads complete here (see conv3x3_c64_kernel)
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) asm volatile("" : "+v"(wf0[ks][nt]));
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    // this tile's input block (issued a tile ago) has landed; the previous epilogue's four stores may stay in flight
    if (whole) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // everyone's pieces landed; everyone is done with the halo and the other input block
    if (tile + (int)gridDim.x < ntiles) issue_input(tile + gridDim.x, buf ^ 1);
    const int n = tile / tpi, rem = tile - n * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    // ---- first layer into the halo
    {
      const char* ib = inb + buf * IBUF;
      const int hy0 = 2 * ty * TH - 1, hx0 = 2 * tx * TW - 1;              // first-layer coordinates of halo entry (0, 0)
      f32x4_t bb0[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bb0[nt] = *(const f32x4_t*)(bias0 + 16 * nt + 4 * g4);
      // NG groups at a time, in phases: all their fragment reads, then the MFMAs (2 NG independent chains), then the SiLUs, then
      // the stores -- written group by group hipcc keeps every group's reads behind the previous group's stores (both are LDS)
      // and the wave sits out a read latency, an MFMA chain and a SiLU chain per group.  An entry's geometry comes from the
      // table (one ds_read_b64 instead of ~20 vector instructions: the kernel is bound by vector issue, profiles/r04_pmc_stem_pair.json);
      // EDGE: the tile touches the map's border -- only there can an entry lie outside the map and must be written as zero.
      auto do_groups = [&](auto ngc, auto edgec, int grp0) {
        constexpr int NG = decltype(ngc)::value;
        constexpr bool EDGE = decltype(edgec)::value;
        unsigned wr[NG]; bool inmap[NG];
        vec8 xf[NG][3];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
          const int e = 16 * (grp0 + 4 * u) + li;
          const uint2 tb = *(const uint2*)(etab + e * 8);               // x: byte offset of the entry's pixel in the input block | y: hy | hx << 8 | real << 16
          wr[u] = ((unsigned)(e * 64) + wlane) ^ ((tb.y >> 19) & 32u);     // halo row + this lane's place; the swizzle bit rides in tb.y bit 24
          if constexpr (EDGE) {
            const int hy = tb.y & 255, hx = (tb.y >> 8) & 255;
            inmap[u] = (tb.y & 0x10000u) != 0 && (unsigned)(hy0 + hy) < (unsigned)g.H && (unsigned)(hx0 + hx) < (unsigned)g.Wd;
          }
          const char* px = ib + tb.x;
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) xf[u][ks] = *(const vec8*)(px + tapoff[ks]);
        }
        f32x4_t a0[NG][2];
#pragma unroll
        for (int u = 0; u < NG; ++u)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) a0[u][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
          for (int u = 0; u < NG; ++u)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) a0[u][nt] = T::mfma(wf0[ks][nt], xf[u][ks], a0[u][nt]);
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        u32x2 w[NG][2];
#pragma unroll
        for (int u = 0; u < NG; ++u)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            typename T::vec4 o;
            const f32x4_t v4 = silu4(add4(a0[u][nt], bb0[nt]));      // (silu2 pins the fp32 product: the 16-bit rounding is a second one, as everywhere)
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (elem)v4[q];
            w[u][nt] = __builtin_bit_cast(u32x2, o);
            if constexpr (EDGE) {                                    // outside the first layer's map: the second layer's zero padding
              w[u][nt][0] = inmap[u] ? w[u][nt][0] : 0u;
              w[u][nt][1] = inmap[u] ? w[u][nt][1] : 0u;
            }
          }
#pragma unroll
        for (int u = 0; u < NG; ++u) {
          *(u32x2*)(smem + wr[u]) = w[u][0];
          *(u32x2*)(smem + (wr[u] ^ 32u)) = w[u][1];
        }
      };
      static_assert(GROUPS == 37, "nine groups per wave and one more for wave 0");
      // (entries of the halo's padding rows -- never read by the second layer -- keep whatever finite value their lanes compute)
      const bool edge = ty == 0 || tx == 0 || hy0 + HH > g.H || hx0 + HWD > g.Wd;
      if (edge) {
#pragma unroll 1
        for (int j = 0; j < 9; j += 3) do_groups(std::integral_constant<int, 3>{}, std::true_type{}, 4 * j + wave);
        if (wave == 0) do_groups(std::integral_constant<int, 1>{}, std::true_type{}, 36);
      } else {
#pragma unroll 1
        for (int j = 0; j < 9; j += 3) do_groups(std::integral_constant<int, 3>{}, std::false_type{}, 4 * j + wave);
        if (wave == 0) do_groups(std::integral_constant<int, 1>{}, std::false_type{}, 36);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // the halo is complete
    // ---- second layer out of it
    f32x4_t acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    vec8 xr[3][4];
    auto rd = [&](auto sc) {
      constexpr int tap = decltype(sc)::value, r = tap % 3, ky = tap / 3, kx = tap % 3;
      vec8 (&x)[4] = xr[r];
      unsigned (&fa)[8] = faddr;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[0]) : "v"(fa[((0 + ky) * PITCH + (kx == 2)) & 7]), "n"(((kx & 1) * PLANE + (0 + ky) * PITCH + (kx == 2)) * 64));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[1]) : "v"(fa[((2 + ky) * PITCH + (kx == 2)) & 7]), "n"(((kx & 1) * PLANE + (2 + ky) * PITCH + (kx == 2)) * 64));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[2]) : "v"(fa[((4 + ky) * PITCH + (kx == 2)) & 7]), "n"(((kx & 1) * PLANE + (4 + ky) * PITCH + (kx == 2)) * 64));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[3]) : "v"(fa[((6 + ky) * PITCH + (kx == 2)) & 7]), "n"(((kx & 1) * PLANE + (6 + ky) * PITCH + (kx == 2)) * 64));
    };
    auto step = [&](auto sc) {
      constexpr int tap = decltype(sc)::value, r = tap % 3;
      if constexpr (tap + 2 < 9) rd(std::integral_constant<int, tap + 2>{});
      vec8 (&x)[4] = xr[r];
      constexpr int newer = (9 - 1 - tap < 2 ? 9 - 1 - tap : 2) * 4;
      if constexpr (newer == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
      else if constexpr (newer == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = T::mfma(wf[tap][ni], xr[r][mi], acc[mi][ni]);
    };
    rd(std::integral_constant<int, 0>{});
    rd(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int ox = tx * TW + li;
    whole = (ty + 1) * TH <= g.Hout && (tx + 1) * TW <= g.Wout;
    char* ytile = (char*)g.C + ((((size_t)n * g.Hout + ty * TH + 4 * pg) * g.Wout + tx * TW) * g.ldc + 32 * chh) * 2;
    f32x4_t bv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) bv[h] = *(const f32x4_t*)(bias1 + 32 * chh + 16 * h + 4 * g4);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int oy = ty * TH + 4 * pg + mi;
      unsigned pk[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4_t a = acc[mi][h], bb = bv[h];
        typename T::vec4 o;
        const f32x4_t v4 = silu4(add4(a, bb));
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (elem)v4[q];
        const u32x2 w = __builtin_bit_cast(u32x2, o);
        pk[h][0] = w[0]; pk[h][1] = w[1];
      }
      const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
      const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
      if (whole || (ox < g.Wout && oy < g.Hout)) {
        const u32x4 o = u32x4{s0[0], s1[0], s0[1], s1[1]};
        const char* yrow = ytile + (size_t)mi * g.Wout * g.ldc * 2;
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(yoff), "v"(o), "s"(yrow) : "memory");   // (s_nop: see asm_store_note)
      }
    }
  }
}

template <class T>
int launch_conv_stem_pair(const StemPairArgs& g, hipStream_t s) {
  constexpr int LDS = 592 * 64 + 2 * 11 * 1024 + 96 * 4 + 592 * 8;
  auto kern = conv_stem_pair_kernel<T>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "hm_conv2d_stem_pair: cannot raise the dynamic LDS limit")) return rc;
  int cus = hm_device_cu_count();
  if (cus <= 0) cus = 256;
  const int ntiles = g.NB * ((g.Wout + 15) / 16) * ((g.Hout + 7) / 8);
  const int grid = ntiles < 2 * cus ? ntiles : 2 * cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, s, g);
  return hm_check_launch("hm_conv2d_stem_pair");
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 convolutions of the large maps with K <= 256 (round 3): 245760 x 256 x 256 and its relatives move 120-250 MB for 8-32 GFLOP
// and ran at 2.5x their HBM time as implicit GEMMs -- four K-steps between a prologue and an epilogue whose SiLU alone costs
// more VALU cycles than the K loop costs MFMA cycles.  Same recipe as the 3x3 kernels: a persistent workgroup keeps the WEIGHTS
// in registers (wave w owns Cout / 4 channels: NI x K/32 fragments), streams 64-pixel tiles of X through a double-buffered LDS
// image by LDS-DMA with tile-invariant lane offsets, reads fragments through K/32 precomputed lane addresses + immediates (rows
// are K x 2 = 256 / 512 bytes, a multiple of the 256-byte bank row, so the 16-byte chunk index is XORed with row & 15), one
// barrier per tile, counted waits, lane-swap epilogue.  K order = the implicit GEMM's: bit-identical.
template <class T, int K32, int NI, int ACT>
__global__ __launch_bounds__(256, 2) void conv1x1_wreg_kernel(const KArgs g) {
  using elem = typename T::elem;
  using vec8 = typename T::vec8;
  constexpr int K = K32 * 32, ROWB = K * 2, BM = 64, TILEB = BM * ROWB, PIECES = TILEB / 1024, NJ = PIECES / 4, RPP = 1024 / ROWB;   // rows per piece: 2 / 4
  constexpr int CPR = ROWB / 16;                       // 16-byte chunks per row: 32 / 16
  constexpr int NST = 4 * NI / 2;                      // 16-byte stores per wave and tile: 4 pixel tiles x NI / 2 channel pairs
  static_assert(K32 == 4 || K32 == 8, "K = 128 or 256");
  extern __shared__ __attribute__((aligned(16))) char smem[];                 // 2 x TILEB | bias [64 NI] f32
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g4 = lane >> 4;
  const int nb = wave * NI * 16;                       // this wave's first output channel
  vec8 wf[K32][NI];
#pragma unroll
  for (int ks = 0; ks < K32; ++ks)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      wf[ks][ni] = *(const vec8*)((const elem*)g.W + (size_t)(nb + 16 * ni + li) * g.ldw + 32 * ks + 8 * g4);
  for (int c = tid; c < 64 * NI; c += 256) ((float*)(smem + 2 * TILEB))[c] = g.bias[c];
  // copy j of this wave = piece 4 j + wave = rows RPP (4 j + wave) + lane / CPR, physical chunk lane % CPR = logical chunk ^ (row & 15)
  unsigned voff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = RPP * (4 * j + wave) + lane / CPR, c = (lane % CPR) ^ (row & 15);
    voff[j] = (unsigned)((row * g.ldx + c * 8) * 2);
  }
  const int ntiles = g.M / BM;                         // (host: M % 64 == 0)
  auto issue = [&](int tile, int buf) {
    const char* base = (const char*)((const elem*)g.X + (size_t)tile * BM * g.ldx);
    char* dst = smem + buf * TILEB;
#pragma unroll
    for (int j = 0; j < NJ; ++j) glds16_hidden_s(base, voff[j], dst + (4 * j + wave) * 1024);
  };
  // fragment (ks, mi): row 16 mi + li, chunk (4 ks + g4) ^ li  ->  one lane address per ks, 16 mi rows as an immediate
  unsigned faddr[K32];
#pragma unroll
  for (int ks = 0; ks < K32; ++ks)
    faddr[ks] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(li * ROWB + (((4 * ks + g4) ^ li) << 4));
  const unsigned yoff = (unsigned)((li * g.ldc + (g4 & 1) * 16 + (g4 >> 1) * 8) * 2);

  int tile = xcd_remap(blockIdx.x, gridDim.x), buf = 0;
  if (tile < ntiles) issue(tile, 0);
#pragma unroll
  for (int ks = 0; ks < K32; ++ks)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+v"(wf[ks][ni]));      // the weight loads complete here (see conv3x3_c64_kernel)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ... and the first tile's rows, so that the loop's wait is the same every time
  constexpr int MIT = K32 * NI > 16 ? 2 : 4;           // pixel tiles of 16 per pass over K (two passes of 32 pixels when the weights
                                                       // fill half the register file: 64 accumulator registers would spill)
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    // this tile's rows (issued a tile ago) have landed; the previous epilogue's NST stores (the newest operations) may stay in flight
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x, buf ^ 1);
    char* ytile = (char*)g.C + ((size_t)tile * BM * g.ldc + nb) * 2;
#pragma unroll
    for (int half = 0; half < 4 / MIT; ++half) {
      f32x4_t acc[MIT][NI];
#pragma unroll
      for (int mi = 0; mi < MIT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      vec8 xr[2][MIT];
      auto rd = [&](auto sc, auto hc) {
        constexpr int ks = decltype(sc)::value, r = ks & 1, h0 = decltype(hc)::value * MIT;
        vec8 (&x)[MIT] = xr[r];
        const unsigned a = faddr[ks];
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[0]) : "v"(a), "n"((h0 + 0) * 16 * ROWB));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[1]) : "v"(a), "n"((h0 + 1) * 16 * ROWB));
        if constexpr (MIT == 4) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[2]) : "v"(a), "n"((h0 + 2) * 16 * ROWB));
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[3]) : "v"(a), "n"((h0 + 3) * 16 * ROWB));
        }
      };
      auto step = [&](auto sc, auto hc) {
        constexpr int ks = decltype(sc)::value, r = ks & 1;
        if constexpr (ks + 1 < K32) rd(std::integral_constant<int, ks + 1>{}, hc);
        vec8 (&x)[MIT] = xr[r];
        if constexpr (MIT == 4) {
          if constexpr (ks + 1 < K32) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
        } else {
          if constexpr (ks + 1 < K32) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(x[0]), "+v"(x[1]));
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MIT; ++mi) acc[mi][ni] = T::mfma(wf[ks][ni], xr[r][mi], acc[mi][ni]);
      };
      auto run = [&](auto hc) {
        rd(std::integral_constant<int, 0>{}, hc);
        step(std::integral_constant<int, 0>{}, hc); step(std::integral_constant<int, 1>{}, hc);
        step(std::integral_constant<int, 2>{}, hc); step(std::integral_constant<int, 3>{}, hc);
        if constexpr (K32 == 8) {
          step(std::integral_constant<int, 4>{}, hc); step(std::integral_constant<int, 5>{}, hc);
          step(std::integral_constant<int, 6>{}, hc); step(std::integral_constant<int, 7>{}, hc);
        }
      };
      if (half == 0) run(std::integral_constant<int, 0>{});
      else run(std::integral_constant<int, 1>{});
#pragma unroll
      for (int np = 0; np < NI / 2; ++np) {
        f32x4_t bv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) bv[h] = *(const f32x4_t*)(smem + 2 * TILEB + (nb + 16 * (2 * np + h) + 4 * g4) * 4);
#pragma unroll
        for (int mi = 0; mi < MIT; ++mi) {
          unsigned pk[2][2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4_t a = acc[mi][2 * np + h], bb = bv[h];
            typename T::vec4 o;
            f32x4_t v4 = add4(a, bb);
            if (ACT == 1) v4 = silu4(v4);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (elem)(ACT == 2 ? fmaxf(v4[q], 0.f) : v4[q]);
            const u32x2 w = __builtin_bit_cast(u32x2, o);
            pk[h][0] = w[0]; pk[h][1] = w[1];
          }
          const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
          const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
          const u32x4 o = u32x4{s0[0], s1[0], s0[1], s1[1]};
          const char* yrow = ytile + ((size_t)(half * MIT + mi) * 16 * g.ldc + np * 32) * 2;
          asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(yoff), "v"(o), "s"(yrow) : "memory");   // (s_nop: see asm_store_note)
        }
      }
    }
    {
      const unsigned d = buf ? (unsigned)-TILEB : (unsigned)TILEB;
#pragma unroll
      for (int ks = 0; ks < K32; ++ks) faddr[ks] += d;
    }
  }
}

template <class T, int K32, int NI, int ACT>
int launch_conv1x1_wreg(const KArgs& g, hipStream_t s) {
  constexpr int LDS = 2 * 64 * K32 * 64 + 64 * NI * 4;
  auto kern = conv1x1_wreg_kernel<T, K32, NI, ACT>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "hm_conv2d_nhwc: cannot raise the dynamic LDS limit")) return rc;
  int cus = hm_device_cu_count();
  if (cus <= 0) cus = 256;
  const int ntiles = g.M / 64;
  const int grid = ntiles < 2 * cus ? ntiles : 2 * cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, s, g);
  return hm_check_launch("hm_conv2d_nhwc (1x1, weights in registers)");
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3, STRIDE 2, 64 -> 128 channels (the second down-sampling layer, 192 x 320 -> 96 x 160): conv3x3_s2c32_kernel's two planes of
// input columns with conv3x3_c64_kernel's 128-byte rows (chunk ^ (row & 7), two 32-deep steps per tap), on EIGHT waves = 2 pixel
// groups x 4 channel groups of 32 (a wave's weight slice is again 36 fragments), one workgroup per CU (two halo buffers of 74 KB).
// Bit-identical to the implicit GEMM.
template <class T, int ACT>
__global__ __launch_bounds__(512, 1) void conv3x3_s2c64_kernel(const KArgs g) {
  using elem = typename T::elem;
  using vec8 = typename T::vec8;
  constexpr int TH = 8, TW = 16, HH = 2 * TH + 1, HWD = 2 * TW + 1;        // halo 17 x 33 input pixels
  constexpr int PITCH = 17, PLANE = 296, NWV = 8;
  constexpr int ENTRIES = 2 * PLANE, PIECES = ENTRIES / 8, HBUF = PIECES * 1024, NJ = (PIECES + NWV - 1) / NWV;   // 592 rows of 128 B = 74 pieces
  extern __shared__ __attribute__((aligned(16))) char smem[];                 // 2 x HBUF | bias [128] f32
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g4 = lane >> 4, pg = wave >> 2, chh = wave & 3;
  const elem* X = (const elem*)g.X;
  vec8 wf[9][2][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        wf[tap][ks][ni] = *(const vec8*)((const elem*)g.W + (size_t)(32 * chh + 16 * ni + li) * g.ldw + 64 * tap + 32 * ks + 8 * g4);
  if (tid < 128) ((float*)(smem + 2 * HBUF))[tid] = g.bias[tid];

  const int tiles_x = (g.Wout + TW - 1) / TW, tiles_y = (g.Hout + TH - 1) / TH, tpi = tiles_x * tiles_y;
  const int ntiles = (g.M / (g.Hout * g.Wout)) * tpi;
  const int srow = lane >> 3;                          // copy j of this wave = piece 8 j + wave = rows 8 (8 j + wave) + srow, physical chunk lane & 7
  auto entry_geom = [&](int e, int& hy, int& hx, bool& real) {
    const int pl = e >= PLANE ? 1 : 0, r = e - pl * PLANE;
    hy = r / PITCH;
    const int idx = r - hy * PITCH;
    hx = 2 * idx + pl;
    real = r < HH * PITCH && hx < HWD;
  };
  unsigned voff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int e = 8 * (NWV * j + wave) + srow, c = (lane & 7) ^ (e & 7);
    int hy, hx; bool real;
    entry_geom(e, hy, hx, real);
    voff[j] = (real && e < ENTRIES) ? (unsigned)(((hy * g.Wd + hx) * g.ldx + c * 8) * 2) : 0u;
  }
  auto issue_halo = [&](int tile, int buf) {
    const int n = tile / tpi, rem = tile - n * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int y0 = 2 * ty * TH - 1, x0 = 2 * tx * TW - 1;
    char* dst = smem + buf * HBUF;
    if (y0 >= 0 && x0 >= 0 && y0 + HH <= g.H && x0 + HWD <= g.Wd) {
      const char* base = (const char*)(X + (((size_t)n * g.H + y0) * g.Wd + x0) * g.ldx);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (NWV * j + wave < PIECES) glds16_hidden_s(base, voff[j], dst + (NWV * j + wave) * 1024);
    } else {
      const elem* img = X + (size_t)n * g.H * g.Wd * g.ldx;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (NWV * j + wave < PIECES) {
          const int e = 8 * (NWV * j + wave) + srow, c = (lane & 7) ^ (e & 7);
          int hy, hx; bool real;
          entry_geom(e, hy, hx, real);
          const int iy = y0 + hy, ix = x0 + hx;
          const bool ok = real && iy >= 0 && iy < g.H && ix >= 0 && ix < g.Wd;
          const elem* src = ok ? img + ((size_t)iy * g.Wd + ix) * g.ldx + c * 8 : (const elem*)g.zeros;
          glds16_hidden_v(src, dst + (NWV * j + wave) * 1024);
        }
    }
  };
  // fragment of (tap, ks, pixel row mi): plane kx & 1, row (2 (4 pg + mi) + ky) * 17 + (kx == 2) + li, chunk (4 ks + g4) ^ (row & 7);
  // row & 7 = (li + m) & 7 with m = ((2 mi + ky) * 17 + (kx == 2)) & 7, and ks = 1 is key m + 4 (as in conv3x3_c64_kernel)
  unsigned faddr[8];
#pragma unroll
  for (int m = 0; m < 8; ++m)
    faddr[m] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem +
               (unsigned)((8 * pg * PITCH + li) * 128 + ((g4 ^ ((li + m) & 7)) << 4));
  const unsigned yoff = (unsigned)((li * g.ldc + (g4 & 1) * 16 + (g4 >> 1) * 8) * 2);

  int tile = xcd_remap(blockIdx.x, gridDim.x), buf = 0;
  bool whole = false;
  if (tile < ntiles) issue_halo(tile, 0);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) asm volatile("" : "+v"(wf[tap][ks][ni]));
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    if (whole) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + (int)gridDim.x < ntiles) issue_halo(tile + gridDim.x, buf ^ 1);
    const int n = tile / tpi, rem = tile - n * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int ox = tx * TW + li;
    whole = (ty + 1) * TH <= g.Hout && (tx + 1) * TW <= g.Wout;
    char* ytile = (char*)g.C + ((((size_t)n * g.Hout + ty * TH + 4 * pg) * g.Wout + tx * TW) * g.ldc + 32 * chh) * 2;
    // two passes over K, two pixel rows each (four rows at once need 32 + 32 more registers than the 144 of the weights leave)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4_t acc[2][2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      vec8 xr[2][2];
      auto rd = [&](auto sc, auto hc) {
        constexpr int s = decltype(sc)::value, tap = s >> 1, ks = s & 1, r = s & 1, ky = tap / 3, kx = tap % 3, r0 = 4 * decltype(hc)::value;
        vec8 (&x)[2] = xr[r];
        unsigned (&fa)[8] = faddr;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[0]) : "v"(fa[((r0 + 0 + ky) * PITCH + (kx == 2) + 4 * ks) & 7]), "n"(((kx & 1) * PLANE + (r0 + 0 + ky) * PITCH + (kx == 2)) * 128));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[1]) : "v"(fa[((r0 + 2 + ky) * PITCH + (kx == 2) + 4 * ks) & 7]), "n"(((kx & 1) * PLANE + (r0 + 2 + ky) * PITCH + (kx == 2)) * 128));
      };
      auto step = [&](auto sc, auto hc) {
        constexpr int s = decltype(sc)::value, r = s & 1;
        if constexpr (s + 1 < 18) rd(std::integral_constant<int, s + 1>{}, hc);
        vec8 (&x)[2] = xr[r];
        if constexpr (s + 1 < 18) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(x[0]), "+v"(x[1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = T::mfma(wf[s >> 1][s & 1][ni], xr[r][mi], acc[mi][ni]);
      };
      auto run = [&](auto hc) {
        rd(std::integral_constant<int, 0>{}, hc);
        step(std::integral_constant<int, 0>{}, hc); step(std::integral_constant<int, 1>{}, hc); step(std::integral_constant<int, 2>{}, hc);
        step(std::integral_constant<int, 3>{}, hc); step(std::integral_constant<int, 4>{}, hc); step(std::integral_constant<int, 5>{}, hc);
        step(std::integral_constant<int, 6>{}, hc); step(std::integral_constant<int, 7>{}, hc); step(std::integral_constant<int, 8>{}, hc);
        step(std::integral_constant<int, 9>{}, hc); step(std::integral_constant<int, 10>{}, hc); step(std::integral_constant<int, 11>{}, hc);
        step(std::integral_constant<int, 12>{}, hc); step(std::integral_constant<int, 13>{}, hc); step(std::integral_constant<int, 14>{}, hc);
        step(std::integral_constant<int, 15>{}, hc); step(std::integral_constant<int, 16>{}, hc); step(std::integral_constant<int, 17>{}, hc);
      };
      if (half == 0) run(std::integral_constant<int, 0>{});
      else run(std::integral_constant<int, 1>{});
      f32x4_t bv[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) bv[h] = *(const f32x4_t*)(smem + 2 * HBUF + (32 * chh + 16 * h + 4 * g4) * 4);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int oy = ty * TH + 4 * pg + 2 * half + mi;
        unsigned pk[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4_t a = acc[mi][h], bb = bv[h];
          typename T::vec4 o;
          f32x4_t v4 = add4(a, bb);
          if (ACT == 1) v4 = silu4(v4);
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (elem)(ACT == 2 ? fmaxf(v4[q], 0.f) : v4[q]);
          const u32x2 w = __builtin_bit_cast(u32x2, o);
          pk[h][0] = w[0]; pk[h][1] = w[1];
        }
        const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        if (whole || (ox < g.Wout && oy < g.Hout)) {
          const u32x4 o = u32x4{s0[0], s1[0], s0[1], s1[1]};
          const char* yrow = ytile + (size_t)(2 * half + mi) * g.Wout * g.ldc * 2;
          asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(yoff), "v"(o), "s"(yrow) : "memory");   // (s_nop: see asm_store_note)
        }
      }
    }
    {
      const unsigned d = buf ? (unsigned)-HBUF : (unsigned)HBUF;
#pragma unroll
      for (int m = 0; m < 8; ++m) faddr[m] += d;
    }
  }
}

template <class T, int ACT>
int launch_conv_s2c64(const KArgs& g, hipStream_t s) {
  constexpr int LDS = 2 * 74 * 1024 + 512;
  auto kern = conv3x3_s2c64_kernel<T, ACT>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "hm_conv2d_nhwc: cannot raise the dynamic LDS limit")) return rc;
  int cus = hm_device_cu_count();
  if (cus <= 0) cus = 256;
  const int ntiles = (g.M / (g.Hout * g.Wout)) * ((g.Wout + 15) / 16) * ((g.Hout + 7) / 8);
  const int grid = ntiles < cus ? ntiles : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, s, g);
  return hm_check_launch("hm_conv2d_nhwc (3x3 stride 2, 64 -> 128)");
}

int conv_c64_tiles(const KArgs& g) { return (g.M / (g.Hout * g.Wout)) * ((g.Wout + 15) / 16) * ((g.Hout + 7) / 8); }

template <class T, int ACT>
int launch_conv_c64(const KArgs& g, hipStream_t s) {
  constexpr int LDS = 2 * 23 * 1024 + 256;
  auto kern = conv3x3_c64_kernel<T, ACT>;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "hm_conv2d_nhwc: cannot raise the dynamic LDS limit")) return rc;
  int cus = hm_device_cu_count();
  if (cus <= 0) cus = 256;
  const int ntiles = conv_c64_tiles(g);
  const int grid = ntiles < 2 * cus ? ntiles : 2 * cus;                       // two workgroups per CU (registers: 2 waves per SIMD)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, s, g);
  return hm_check_launch("hm_conv2d_nhwc (3x3, 64 -> 64)");
}

// the shapes the direct kernels are built for; everything else (and HM_OPT_CONV_DIRECT = 1) takes the implicit GEMM
template <class T>
int try_conv_direct(const KArgs& g, int epilogue, hipStream_t s, bool& taken) {
  taken = false;
  if (hm_option(HM_OPT_CONV_DIRECT) == 1 || epilogue != HM_EPI_SILU || g.bias == nullptr || (g.ldc & 7) != 0 || (((uintptr_t)g.C) & 15) != 0)
    return HM_OK;
  if (g.ksz == 1 && g.stride == 1 && hm_option(HM_OPT_CONV_DIRECT) != 2) {
    // 1x1 with K = 128 / 256 and Cout = 128 / 256 (M % 64 == 0, from 200 tiles of 64 pixels): 16 frames: 245760 x 256 x 256 76 -> 60 us,
    // x 128 x 256 48 -> 37, x 128 x 128 40 -> 30; 61440 rows 25 -> 23; 15360 rows 11.3 -> 8.8; 3840 rows (60 tiles) 9.5 -> 11.4: not
    // taken.  Bit-identical to the implicit GEMM, so the choice may depend on the batch.
    const int cin1 = 1 << g.cin_log2;
    const bool big = hm_option(HM_OPT_CONV_DIRECT) == 3 || g.M / 64 >= 200;
    if (g.M % 64 == 0 && g.K == cin1 && g.ldw >= g.K && big && (size_t)64 * g.ldx * 2 < (1ull << 31)) {
      taken = true;
      if (cin1 == 256 && g.N == 256) return launch_conv1x1_wreg<T, 8, 4, 1>(g, s);
      if (cin1 == 256 && g.N == 128) return launch_conv1x1_wreg<T, 8, 2, 1>(g, s);
      if (cin1 == 128 && g.N == 128) return launch_conv1x1_wreg<T, 4, 2, 1>(g, s);
      if (cin1 == 128 && g.N == 256) return launch_conv1x1_wreg<T, 4, 4, 1>(g, s);
      taken = false;
    }
    return HM_OK;
  }
  if (g.ksz != 3 || g.ldw < ((9 << g.cin_log2) + 31) / 32 * 32) return HM_OK;
  const int cin = 1 << g.cin_log2;
  taken = true;
#ifdef HM_ABLATIONS
  // bound diagnosis of the two stem layers (WRONG results): HM_OPT_CONV_DIRECT = 4: no activation, 5: SiLU computed, nothing stored
  if (hm_option(HM_OPT_CONV_DIRECT) == 4 || hm_option(HM_OPT_CONV_DIRECT) == 5) {
    const bool nostore = hm_option(HM_OPT_CONV_DIRECT) == 5;
    if (cin == 8 && g.N == 32 && g.stride == 1) return nostore ? launch_conv_direct<T, 8, 32, 1, 3>(g, s) : launch_conv_direct<T, 8, 32, 1, 0>(g, s);
    if (cin == 32 && g.N == 64 && g.stride == 2 && g.pad == 1) return nostore ? launch_conv_s2c32<T, 3>(g, s) : launch_conv_s2c32<T, 0>(g, s);
  }
#endif
  // 3x3 stride 1, 64 -> 64, when every one of the 512 persistent workgroups gets at least two 8 x 16 tiles (its weight slice is
  // loaded into registers once per workgroup): 16 frames, 192 x 320: 138 -> 79 us; 96 x 160 (3.75 tiles each): 40.4 -> 30.5 us;
  // 48 x 80 (480 tiles, not taken): 16.8 -> 18.2 us.  The two kernels agree to the bit, so the choice may depend on the batch.
  // (32-bit byte offsets inside one tile's halo: 10 rows of the input map)
  if (cin == 64 && g.N == 64 && g.stride == 1 && g.pad == 1 && hm_option(HM_OPT_CONV_DIRECT) != 2 && (size_t)12 * g.Wd * g.ldx * 2 < (1ull << 31) &&
      (hm_option(HM_OPT_CONV_DIRECT) == 3 || conv_c64_tiles(g) >= 1024))
    return launch_conv_c64<T, 1>(g, s);
  // Measured per layer, 16 frames of 1080p (tools/prof_yolo.py): 3(8) -> 32: 192 -> 93 us (one frame 18 -> 11.5); 32 -> 64 stride 2:
  // 128 -> 134 and 64 -> 64: 143 -> 161 -- with 9 / 18 K-steps of four dependent global loads each and two waves per SIMD the
  // direct form is bound by load latency there (it would need a ring of fragments many K-steps deep): only the first layer takes it.
  if (cin == 64 && g.N == 128 && g.stride == 2 && g.pad == 1 && hm_option(HM_OPT_CONV_DIRECT) != 2 && (size_t)20 * g.Wd * g.ldx * 2 < (1ull << 31) &&
      (hm_option(HM_OPT_CONV_DIRECT) == 3 || conv_c64_tiles(g) >= 1024))
    return launch_conv_s2c64<T, 1>(g, s);
  if (cin == 32 && g.N == 64 && g.stride == 2 && g.pad == 1 && hm_option(HM_OPT_CONV_DIRECT) != 2 && (size_t)20 * g.Wd * g.ldx * 2 < (1ull << 31) &&
      (hm_option(HM_OPT_CONV_DIRECT) == 3 || conv_c64_tiles(g) >= 1024))
    return launch_conv_s2c32<T, 1>(g, s);
  if (cin == 8 && g.N == 32 && g.stride == 1) return launch_conv_direct<T, 8, 32, 1, 1>(g, s);
  taken = false;
  return HM_OK;
}

// Convolution tiles.  Round 2 had the 128-row, four-wave, two-workgroups-per-CU tile only (128 x 128 / 64 / 32); round 3 adds the
// eight-wave 256-row tiles of the ViT GEMM (256 x 256 / 128 / 64: half the operand bytes per flop) for the layers whose output
// has enough of them to fill the chip, and split-K (below) for the layers that have too few tiles of any shape.
enum { CT_128x128 = 0, CT_128x64 = 1, CT_128x32 = 2, CT_256x128 = 3, CT_256x256 = 4, CT_256x64 = 5,
       CT_128x32_D = 6, CT_128x64_D = 7, CT_128x128_D = 8,                      // _D: deep ring (4 / 4 / 3 stages) for lone workgroups
       CT_128x32_K2 = 9, CT_128x64_K2 = 10, CT_128x128_K2 = 11,                 // _K2: two K groups of four waves (3 / 3 / 2 stages each)
       CT_128x32_P2 = 12, CT_128x64_P2 = 13, CT_128x128_P2 = 14, CT_COUNT = 15,    // _P2: one group, two accumulator sets -- the K2 order
       // (not forceable, chosen by launch_conv only) _S: the split-K ranges summed serially by one workgroup; _SP2: with the K2 order inside a range
       CT_128x32_S = 15, CT_128x64_S = 16, CT_128x128_S = 17, CT_128x32_SP2 = 18, CT_128x64_SP2 = 19, CT_128x128_SP2 = 20 };
constexpr int ct_bm(int t) { return (t >= CT_256x128 && t <= CT_256x64) ? 256 : 128; }
constexpr int ct_bn(int t) {
  return (t == CT_128x128 || t == CT_256x128 || t == CT_128x128_D || t == CT_128x128_K2 || t == CT_128x128_P2) ? 128
         : (t == CT_256x256 ? 256 : ((t == CT_128x32 || t == CT_128x32_D || t == CT_128x32_K2 || t == CT_128x32_P2) ? 32 : 64));
}
constexpr bool ct_k2(int t) { return t >= CT_128x32_K2 && t <= CT_128x128_P2; }      // the tiles with the two-group summation order

int conv_tiles(const KArgs& g, int t) { return ((g.M + ct_bm(t) - 1) / ct_bm(t)) * ((g.N + ct_bn(t) - 1) / ct_bn(t)); }

// Tile choice, from per-layer sweeps of the YOLOv7 shapes with every tile forced in turn (tools/prof_yolo.py with CONV_TILE;
// profiles/r03_yolo_tile_sweep_*.txt; HM_OPT_CONV_TILE forces one):
//  * 256 x 256 (the ViT tile) wins by ~10 % where Cout >= 256 and its tiles fill the chip; 256 x 128 / 256 x 64 never won
//    (they stay selectable for sweeps);
//  * otherwise the 128-row tiles, two to four workgroups per CU, narrowed while most CUs would stay without a tile;
//  * when even the narrow tile leaves at most one workgroup per CU, nothing hides the global -> LDS round trip of a two-stage
//    ring (0.8 us per 64-deep K-step measured on the 12x20 maps): those layers take the deep ring (`ks` = K ranges of split-K).
// K groups for the 12 x 20 / 24 x 40 maps.  WHETHER a layer sums its even and odd K tiles separately is, like split-K, a rule on
// ONE image's output (the summation order changes, and a frame must get the same bytes alone and in a batch): at most 1024
// output pixels per image (4096 for Cout <= 128) and at least 8 K tiles per K range, an even number of them.  HOW it does so depends on the launch:
// few workgroups (one frame; 16 frames of a narrow layer) take two groups of four waves with a ring each
// (gemm_tn_kernel<..., WK = 2>: half the serial K-steps, twice the bytes in flight), many workgroups one group with two
// accumulator sets (<..., KDUAL>: two workgroups per CU overlap each other) -- bit-identical by construction.  One frame: the
// small-map layers -10..-30 % (0.94 ms for the whole pass instead of 1.08); 16 frames: -14..-33 % for the narrow layers, while
// K groups everywhere cost the wide ones +10..+40 % (profiles/r03_yolo_kgroups_*.txt).  HM_OPT_CONV_KGROUPS = 1: never.
int conv_kgroups(const KArgs& g, int ks) {
  if (hm_option(HM_OPT_CONV_KGROUPS) == 1) return 1;
  const int nk = g.K / 64 / (ks > 1 ? ks : 1);
  const int m_img = g.Hout * g.Wout;
  // (48 x 80 maps too where the layer is narrow: one frame has 30 row tiles there; wide layers of that size keep the 256 x 256
  //  tile for batched passes, which cannot hold two accumulator sets)
  return ((m_img <= 1024 || (m_img <= 4096 && g.N <= 128)) && nk >= 8 && nk % 2 == 0) ? 2 : 1;
}

int pick_conv_tile(const KArgs& g, int ks_hint = 1, int wk = 1) {
  const int forced = hm_option(HM_OPT_CONV_TILE);
  if (forced > 0 && forced <= CT_COUNT && (ct_k2(forced - 1) ? wk == 2 : wk == 1)) return forced - 1;   // (a forced tile never changes the K order)
  if (wk == 1 && g.N >= 256 && conv_tiles(g, CT_256x256) >= 200) return CT_256x256;
  int t = g.N > 64 ? CT_128x128 : (g.N > 32 ? CT_128x64 : CT_128x32);
  while (t < CT_128x32 && conv_tiles(g, t) < 192) ++t;
  if (wk == 2) {
    const int wgs = conv_tiles(g, t) * ks_hint;
    const bool groups = wgs <= 128 || (ks_hint == 1 && wgs <= 256);       // few workgroups: two K groups each; else two accumulator sets
    if (groups) return t == CT_128x128 ? CT_128x128_K2 : (t == CT_128x64 ? CT_128x64_K2 : CT_128x32_K2);
    return t == CT_128x128 ? CT_128x128_P2 : (t == CT_128x64 ? CT_128x64_P2 : CT_128x32_P2);
  }
  if (conv_tiles(g, t) * ks_hint <= 288 && g.K >= 4 * 64) t = t == CT_128x128 ? CT_128x128_D : (t == CT_128x64 ? CT_128x64_D : CT_128x32_D);
  return t;
}

// `deep` (round 4): the 128 x 64 / 128 x 32 tiles of the two-accumulator-set and serial-range forms with a THREE-stage ring (72 /
// 60 KB: still two workgroups per CU).  On the 12 x 20 maps of a 48-frame pass those layers are a few hundred workgroups of 18-72
// K-steps each, and with one tile in flight a K-step lasts as long as its copy's round trip (~1 us); two tiles in flight halve
// that.  Same K order: same bytes.  Chosen by launch_conv when the launch is at most one round of two workgroups per CU.
template <class T, int EPI>
int launch_conv_tile(const KArgs& g, int t, hipStream_t s, bool deep = false) {
  if (deep) {
    if constexpr (EPI == HM_EPI_STORE || EPI == HM_EPI_SILU || EPI == HM_EPI_RELU || EPI == HM_EPI_F32) {
      switch (t) {
        case CT_128x32_P2: return launch_cfg<T, EPI, 2, 2, 4, 1, 3, true, 64, 0, 1, true>(g, s, "hm_conv2d_nhwc");
        case CT_128x64_P2: return launch_cfg<T, EPI, 2, 2, 4, 2, 3, true, 64, 0, 1, true>(g, s, "hm_conv2d_nhwc");
        default: break;
      }
    }
    if constexpr (EPI == HM_EPI_STORE || EPI == HM_EPI_SILU || EPI == HM_EPI_RELU) {
      switch (t) {
        case CT_128x32_S: return launch_cfg<T, EPI, 2, 2, 4, 1, 3, true, 64, 0, 1, false, true>(g, s, "hm_conv2d_nhwc");
        case CT_128x64_S: return launch_cfg<T, EPI, 2, 2, 4, 2, 3, true, 64, 0, 1, false, true>(g, s, "hm_conv2d_nhwc");
        case CT_128x32_SP2: return launch_cfg<T, EPI, 2, 2, 4, 1, 3, true, 64, 0, 1, true, true>(g, s, "hm_conv2d_nhwc");
        case CT_128x64_SP2: return launch_cfg<T, EPI, 2, 2, 4, 2, 3, true, 64, 0, 1, true, true>(g, s, "hm_conv2d_nhwc");
        default: break;
      }
    }
  }
  switch (t) {
    case CT_128x128: return launch_cfg<T, EPI, 2, 2, 4, 4, 2, true>(g, s, "hm_conv2d_nhwc");
    case CT_128x64: return launch_cfg<T, EPI, 2, 2, 4, 2, 2, true>(g, s, "hm_conv2d_nhwc");
    case CT_128x32: return launch_cfg<T, EPI, 2, 2, 4, 1, 2, true>(g, s, "hm_conv2d_nhwc");
    case CT_256x128: return launch_cfg<T, EPI, 4, 2, 4, 4, 2, true, 64, 2>(g, s, "hm_conv2d_nhwc");
    case CT_256x256: return launch_cfg<T, EPI, 4, 2, 4, 8, 2, true, 64, 2>(g, s, "hm_conv2d_nhwc");
    case CT_256x64: return launch_cfg<T, EPI, 4, 2, 4, 2, 2, true, 64, 2>(g, s, "hm_conv2d_nhwc");
    case CT_128x32_D: return launch_cfg<T, EPI, 2, 2, 4, 1, 4, true>(g, s, "hm_conv2d_nhwc");      // 4 x 20 KB
    case CT_128x64_D: return launch_cfg<T, EPI, 2, 2, 4, 2, 4, true>(g, s, "hm_conv2d_nhwc");      // 4 x 24 KB
    case CT_128x128_D: return launch_cfg<T, EPI, 2, 2, 4, 4, 3, true>(g, s, "hm_conv2d_nhwc");     // 3 x 32 KB
    case CT_128x32_K2: return launch_cfg<T, EPI, 2, 2, 4, 1, 3, true, 64, 0, 2>(g, s, "hm_conv2d_nhwc");     // 2 x 3 x 20 KB
    case CT_128x64_K2: return launch_cfg<T, EPI, 2, 2, 4, 2, 3, true, 64, 0, 2>(g, s, "hm_conv2d_nhwc");     // 2 x 3 x 24 KB
    case CT_128x128_K2: return launch_cfg<T, EPI, 2, 2, 4, 4, 2, true, 64, 0, 2>(g, s, "hm_conv2d_nhwc");    // 2 x 2 x 32 KB
    case CT_128x32_P2: return launch_cfg<T, EPI, 2, 2, 4, 1, 2, true, 64, 0, 1, true>(g, s, "hm_conv2d_nhwc");
    case CT_128x64_P2: return launch_cfg<T, EPI, 2, 2, 4, 2, 2, true, 64, 0, 1, true>(g, s, "hm_conv2d_nhwc");
    case CT_128x128_P2: return launch_cfg<T, EPI, 2, 2, 4, 4, 2, true, 64, 0, 1, true>(g, s, "hm_conv2d_nhwc");
    default: break;
  }
  if constexpr (EPI == HM_EPI_STORE || EPI == HM_EPI_SILU || EPI == HM_EPI_RELU) {
    switch (t) {
      case CT_128x32_S: return launch_cfg<T, EPI, 2, 2, 4, 1, 2, true, 64, 0, 1, false, true>(g, s, "hm_conv2d_nhwc");
      case CT_128x64_S: return launch_cfg<T, EPI, 2, 2, 4, 2, 2, true, 64, 0, 1, false, true>(g, s, "hm_conv2d_nhwc");
      case CT_128x128_S: return launch_cfg<T, EPI, 2, 2, 4, 4, 2, true, 64, 0, 1, false, true>(g, s, "hm_conv2d_nhwc");
      case CT_128x32_SP2: return launch_cfg<T, EPI, 2, 2, 4, 1, 2, true, 64, 0, 1, true, true>(g, s, "hm_conv2d_nhwc");
      case CT_128x64_SP2: return launch_cfg<T, EPI, 2, 2, 4, 2, 2, true, 64, 0, 1, true, true>(g, s, "hm_conv2d_nhwc");
      case CT_128x128_SP2: return launch_cfg<T, EPI, 2, 2, 4, 4, 2, true, 64, 0, 1, true, true>(g, s, "hm_conv2d_nhwc");
      default: break;
    }
  }
  switch (t) {
    default: return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: unknown tile");
  }
}

// Split-K for convolutions with few output tiles and a long K (the 12x20 / 24x40 maps of the YOLOv7 neck: 30-120 row tiles,
// K = 2304..4608): the K tiles are cut into `ks` ranges, each range's workgroups write an fp32 partial slab [ks][M][N] into
// the caller's workspace (HM_EPI_F32 path of the kernel), and one small kernel adds the slabs in order, adds the bias, applies
// the activation and writes the 16-bit NHWC result.  Deterministic (fixed summation order), no atomics.
template <class T, int ACT>     // ACT: 0 none, 1 SiLU, 2 ReLU
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                  typename T::elem* __restrict__ y, int M, int N, int ldy, int ks) {
  const int n8 = N >> 3;                                   // host: N % 8 == 0
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * n8) return;
  const int m = (int)(i / n8), n = (int)(i - (size_t)m * n8) * 8;
  f32x4_t a0 = *(const f32x4_t*)(bias + n), a1 = *(const f32x4_t*)(bias + n + 4);
  f32x4_t s0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, s1 = s0;
  for (int s = 0; s < ks; ++s) {
    const float* p = part + ((size_t)s * M + m) * N + n;
    s0 += *(const f32x4_t*)p; s1 += *(const f32x4_t*)(p + 4);
  }
  typename T::vec8 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float u = __fadd_rn(s0[q], a0[q]), v = __fadd_rn(s1[q], a1[q]);          // acc + bias, then the activation: as the fused epilogue rounds
    if (ACT == 1) { const f32x2_t sq = silu2(f32x2_t{u, v}); u = sq[0]; v = sq[1]; }
    if (ACT == 2) { u = fmaxf(u, 0.f); v = fmaxf(v, 0.f); }
    o[q] = (typename T::elem)u; o[4 + q] = (typename T::elem)v;
  }
  *(typename T::vec8*)(y + (size_t)m * ldy + n) = o;
}

// K ranges for a convolution.  The choice depends on the PER-IMAGE problem only (output pixels of one image, Cout, K) -- never on
// the number of images in the batch -- so a frame gives bit-identical results whether it runs alone or in a batched pass
// (split-K changes the fp32 summation order; a different order per batch size would move fp16 roundings and, through them,
// boxes by a pixel).  The rule is a compromise between one frame (few tiles: every serial K-step costs ~0.8 us of exposed
// latency, partial slabs are tiny) and sixteen (slab traffic 2 x ranges x M x N x 4 bytes counts):
//   12x20-sized maps (<= 256 pixels):  K >= 2048 -> 4 ranges, K >= 1024 -> 2
//   24x40-sized maps (<= 1024 pixels): K >= 2048 and Cout <= 256 -> 2
int conv_split_rule(const KArgs& g) {
  const int forced = hm_option(HM_OPT_CONV_SPLITK);
  const int m_img = g.Hout * g.Wout, nk = g.K / 64;
  if (forced == 1 || g.N % 8 != 0 || (g.ldc & 7) != 0 || nk < 8) return 1;
  int want = 1;
  if (forced > 1) want = forced;
  else if (m_img <= 256) want = nk >= 32 ? 4 : (nk >= 16 ? 2 : 1);
  else if (m_img <= 1024) want = (nk >= 32 && g.N <= 256) ? 2 : 1;
  if (want > nk / 4) want = nk / 4;                        // at least 4 K tiles per range: the ring's start-up is two tiles
  while (want > 1 && nk % want != 0) --want;
  return want < 1 ? 1 : want;
}

// deep ring for the narrow two-set / serial tiles: at most one round of two workgroups per CU, and enough K-steps per range
bool conv_deep_ring(const KArgs& g, int t, int wgs, int nk_range) {
  if (hm_option(HM_OPT_CONV_TILE) != 0) return false;                 // (a forced tile is taken as is: sweeps)
  const bool narrow = t == CT_128x32_P2 || t == CT_128x64_P2 || t == CT_128x32_S || t == CT_128x64_S || t == CT_128x32_SP2 || t == CT_128x64_SP2;
  int cus = hm_device_cu_count();
  if (cus <= 0) cus = 256;
  return narrow && wgs <= 2 * cus && nk_range >= 6;
}

template <class T>
int launch_conv(const KArgs& g0, int epilogue, void* ws, size_t ws_bytes, hipStream_t s) {
  KArgs g = g0;
  {
    bool taken = false;
    const int rc = try_conv_direct<T>(g, epilogue, s, taken);
    if (taken || rc != HM_OK) return rc;
  }
  const bool act_ok = epilogue == HM_EPI_STORE || epilogue == HM_EPI_SILU || epilogue == HM_EPI_RELU;
  int ks = (ws && g.bias && act_ok && ((((uintptr_t)ws) | ((uintptr_t)g.C)) & 15) == 0) ? conv_split_rule(g) : 1;   // (the reduce kernel stores 16 bytes per lane)
  if (ks > 1 && (size_t)ks * g.M * g.N * 4 > ws_bytes)
    return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: splitk_ws too small (hm_conv_splitk_bytes gives the size; the split must not depend on what fits)");
  const int kgr = conv_kgroups(g, ks);
  // Many tiles already (a pass of many frames): the split-K ranges one after the other in ONE workgroup -- the same bytes as the
  // split (gemm_tn_kernel<..., KSER>), without slabs and without the reduce launch.  HM_OPT_CONV_SPLITK = 2..: tuning runs force
  // the parallel split.
  if (ks > 1 && hm_option(HM_OPT_CONV_SPLITK) == 0) {
    int ts = g.N > 64 ? CT_128x128 : (g.N > 32 ? CT_128x64 : CT_128x32);
    while (ts < CT_128x32 && conv_tiles(g, ts) < 192) ++ts;           // narrowed as pick_conv_tile narrows it
    if (conv_tiles(g, ts) >= 256) {
      g.kser = ks;
      const int t_ser = kgr == 2 ? (ts == CT_128x128 ? CT_128x128_SP2 : (ts == CT_128x64 ? CT_128x64_SP2 : CT_128x32_SP2))
                                 : (ts == CT_128x128 ? CT_128x128_S : (ts == CT_128x64 ? CT_128x64_S : CT_128x32_S));
      const bool deep = conv_deep_ring(g, t_ser, conv_tiles(g, t_ser), g.K / 64 / ks);
      switch (epilogue) {
        case HM_EPI_STORE: return launch_conv_tile<T, HM_EPI_STORE>(g, t_ser, s, deep);
        case HM_EPI_SILU: return launch_conv_tile<T, HM_EPI_SILU>(g, t_ser, s, deep);
        default: return launch_conv_tile<T, HM_EPI_RELU>(g, t_ser, s, deep);
      }
    }
  }
  // 1x1, stride 1, SiLU, Cout % 256 == 0 over at least a round and a half of whole 256 x 256 tiles (a batched pass: the 512 -> 512 /
  // 1024 -> 1024 / 512 -> 256 transitions of the 48 x 80 and 24 x 40 maps): a plain GEMM -- the PERSISTENT kernel of the ViT
  // (gemm_px_kernel, SiLU in its lane-swap epilogue).  As one-tile workgroups these layers spend as long outside their 8-16
  // K-steps (launch, ring fill, epilogue; one workgroup per CU) as inside.  Same K order and epilogue arithmetic: same bytes
  // (test_conv_1x1_persistent_kernel_equals_the_tile_kernel).  HM_OPT_CONV_TILE = 16 forces it where it applies, any other forced tile
  // excludes it.
  if (epilogue == HM_EPI_SILU && ks <= 1 && kgr == 1 && g.ksz == 1 && g.stride == 1 && g.K == (1 << g.cin_log2) && px_ok(g)) {   // (kgr: the persistent kernel has the one-group summation order only)
    const int forced = hm_option(HM_OPT_CONV_TILE), tiles = (g.M >> 8) * (g.N >> 8);
    int cus = hm_device_cu_count();
    if (cus <= 0) cus = 256;
    if (forced == CT_COUNT + 1 || (forced == 0 && 2 * tiles >= 3 * cus && g.K >= 256)) return launch_px<T, HM_EPI_SILU>(g, s);
  }
  const int t = pick_conv_tile(g, ks, kgr);
  const bool deep = conv_deep_ring(g, t, conv_tiles(g, t) * (ks > 1 ? ks : 1), g.K / 64 / (ks > 1 ? ks : 1));
  if (ks > 1) {
    void* y = g.C; const int ldy = g.ldc; const float* bias = g.bias;
    g.C = ws; g.ldc = g.N; g.bias = nullptr; g.ksplit = ks;
    if (const int rc = launch_conv_tile<T, HM_EPI_F32>(g, t, s, deep)) return rc;
    const size_t n = (size_t)g.M * (g.N >> 3);
    const dim3 grid((unsigned)((n + 255) / 256));
    using elem = typename T::elem;
    if (epilogue == HM_EPI_SILU) hipLaunchKernelGGL((conv_splitk_reduce_kernel<T, 1>), grid, dim3(256), 0, s, (const float*)ws, bias, (elem*)y, g.M, g.N, ldy, ks);
    else if (epilogue == HM_EPI_RELU) hipLaunchKernelGGL((conv_splitk_reduce_kernel<T, 2>), grid, dim3(256), 0, s, (const float*)ws, bias, (elem*)y, g.M, g.N, ldy, ks);
    else hipLaunchKernelGGL((conv_splitk_reduce_kernel<T, 0>), grid, dim3(256), 0, s, (const float*)ws, bias, (elem*)y, g.M, g.N, ldy, ks);
    return hm_check_launch("hm_conv2d_nhwc (split-K reduce)");
  }
  switch (epilogue) {
    case HM_EPI_STORE: return launch_conv_tile<T, HM_EPI_STORE>(g, t, s, deep);
    case HM_EPI_SILU: return launch_conv_tile<T, HM_EPI_SILU>(g, t, s, deep);
    case HM_EPI_F32: return launch_conv_tile<T, HM_EPI_F32>(g, t, s, deep);
    case HM_EPI_RELU: return launch_conv_tile<T, HM_EPI_RELU>(g, t, s, deep);
    case HM_EPI_ADD_RELU: return launch_conv_tile<T, HM_EPI_ADD_RELU>(g, t, s);
    default: return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: unsupported epilogue");
  }
}

}  // namespace

extern "C" int hm_ln_finalize(const float* partials, float* row_stats, int M, int D, float eps, void* stream_) {
  if (!partials || !row_stats || M <= 0 || D <= 0 || D % 64 != 0) return hm_set_error(HM_ERR_ARG, "hm_ln_finalize: bad arguments");
  hipLaunchKernelGGL(ln_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream_, (const float2*)partials,
                     (float2*)row_stats, M, D / 64, 1.0f / (float)D, eps);
  return hm_check_launch("hm_ln_finalize");
}

extern "C" int hm_gemm_set_group_m(int gm) {
  if (gm < 1 || gm > 64) return hm_set_error(HM_ERR_ARG, "hm_gemm_set_group_m: 1..64");
  g_group_m = gm;
  return HM_OK;
}

extern "C" int hm_gemm_set_variant(int v) {
  if (!variant_ok(v)) return hm_set_error(HM_ERR_ARG, "hm_gemm_set_variant: -1 (default) or a shipped tile variant: 0, 10, 24, 26 (experiments and ablations: libhamer_hip_abl.so)");
  g_variant = v;
  return HM_OK;
}

extern "C" int hm_gemm(const hm_gemm_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a) return hm_set_error(HM_ERR_ARG, "hm_gemm: null args");
  const hm_gemm_args& g = *a;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return hm_set_error(HM_ERR_ARG, "hm_gemm: empty problem");
  if (g.K % BK_DEFAULT != 0) return hm_set_error(HM_ERR_ARG, "hm_gemm: K must be a multiple of 64");
  if (g.N % 4 != 0 || g.ldc % 4 != 0) return hm_set_error(HM_ERR_ARG, "hm_gemm: N and ldc must be multiples of 4");
  if (g.ldx % 8 != 0 || g.ldw % 8 != 0) return hm_set_error(HM_ERR_ARG, "hm_gemm: ldx/ldw must be multiples of 8 elements");
  if (g.ldx < g.K || g.ldw < g.K || g.ldc < g.N) return hm_set_error(HM_ERR_ARG, "hm_gemm: leading dimension too small");
  if (!g.X || !g.W || !g.C) return hm_set_error(HM_ERR_ARG, "hm_gemm: null operand");
  const bool ln_out = g.epilogue == HM_EPI_RESID_LN, ln_in = g.epilogue == HM_EPI_LN_STORE || g.epilogue == HM_EPI_LN_GELU;
  if ((g.epilogue == HM_EPI_RESID_F32 || ln_out) && (!g.resid || g.ldr < g.N || g.ldr % 4 != 0))
    return hm_set_error(HM_ERR_ARG, "hm_gemm: residual epilogue needs resid and ldr >= N, ldr % 4 == 0");
  if (ln_out && (!g.ln_gamma || !g.ln_xg || !g.ln_stats || g.N % 64 != 0))
    return hm_set_error(HM_ERR_ARG, "hm_gemm: HM_EPI_RESID_LN needs ln_gamma, ln_xg, ln_stats and N % 64 == 0");
  if (ln_in && (!g.ln_stats || !g.ln_colsum || g.N % 8 != 0 || g.ldc % 8 != 0))
    return hm_set_error(HM_ERR_ARG, "hm_gemm: HM_EPI_LN_* need ln_stats, ln_colsum, N % 8 == 0, ldc % 8 == 0");
  if ((ln_out || ln_in) && (((uintptr_t)g.ln_gamma | (uintptr_t)g.ln_xg | (uintptr_t)g.ln_stats | (uintptr_t)g.ln_colsum) & 15))
    return hm_set_error(HM_ERR_ARG, "hm_gemm: deferred-LN pointers must be 16-byte aligned");
  if (((uintptr_t)g.X | (uintptr_t)g.W | (uintptr_t)g.C | (uintptr_t)g.bias | (uintptr_t)g.resid) & 15)
    return hm_set_error(HM_ERR_ARG, "hm_gemm: pointers must be 16-byte aligned");
  KArgs k{};
  k.X = g.X; k.W = g.W; k.C = g.C; k.bias = g.bias; k.resid = g.resid;
  k.M = g.M; k.N = g.N; k.K = g.K; k.ldx = g.ldx; k.ldw = g.ldw; k.ldc = g.ldc; k.ldr = g.ldr; k.resid_mod = g.resid_mod;
  k.group_m = g_group_m;
  k.ksplit = 1;
  k.out_scale = g.out_scale != 0.0f ? g.out_scale : 1.0f;
  if (k.out_scale != 1.0f && g.epilogue != HM_EPI_GELU) return hm_set_error(HM_ERR_ARG, "hm_gemm: out_scale applies to HM_EPI_GELU only");
  if (g.k_split > 1) {
    if (g.epilogue != HM_EPI_F32 || g.bias || g.K % (BK_DEFAULT * g.k_split) != 0)
      return hm_set_error(HM_ERR_ARG, "hm_gemm: k_split needs HM_EPI_F32, no bias and K % (64 * k_split) == 0");
    k.ksplit = g.k_split;
  }
  if (ln_out) { k.ln_gamma = g.ln_gamma; k.ln_xg = g.ln_xg; k.ln_stats = g.ln_stats; k.ln_P = g.N / 64; }
#ifdef HM_ABLATIONS
  if (g_variant == 35) k.ln_stats = g.ln_stats;      // (diagnostic build: the stamp buffer of gemm_px_kernel<..., STAMP>)
#endif
  if (ln_in) {
    k.ln_stats = g.ln_stats; k.ln_colsum = g.ln_colsum;
  }
  HmProfScope prof(HM_K_GEMM, g.epilogue, g.M, g.N, g.K, stream);
  if (g.dtype == HM_DTYPE_BF16) return launch_gemm_epi<TBf16>(k, g.epilogue, stream);
  if (g.dtype == HM_DTYPE_F16) return launch_gemm_epi<TF16>(k, g.epilogue, stream);
  return hm_set_error(HM_ERR_ARG, "hm_gemm: dtype must be HM_DTYPE_BF16 or HM_DTYPE_F16");
}

extern "C" int hm_gemm_fp8(const hm_gemm_fp8_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a) return hm_set_error(HM_ERR_ARG, "hm_gemm_fp8: null args");
  const hm_gemm_fp8_args& g = *a;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.M % 16 != 0 || g.N % 64 != 0 || g.K % 128 != 0)
    return hm_set_error(HM_ERR_ARG, "hm_gemm_fp8: need M % 16 == 0, N % 64 == 0, K % 128 == 0");
  if (!g.X8 || !g.x_scales || !g.W8 || !g.w_scale || !g.C) return hm_set_error(HM_ERR_ARG, "hm_gemm_fp8: null operand");
  if (g.ldx % 16 != 0 || g.ldw % 16 != 0 || g.ldx < g.K || g.ldw < g.K || g.ldc < g.N || g.ldc % 8 != 0)
    return hm_set_error(HM_ERR_ARG, "hm_gemm_fp8: ldx/ldw must be multiples of 16 bytes covering K, ldc % 8 == 0 covering N");
  if (((uintptr_t)g.X8 | (uintptr_t)g.x_scales | (uintptr_t)g.W8 | (uintptr_t)g.w_scale | (uintptr_t)g.C | (uintptr_t)g.bias |
       (uintptr_t)g.resid | (uintptr_t)g.out_scales) & 15)
    return hm_set_error(HM_ERR_ARG, "hm_gemm_fp8: pointers must be 16-byte aligned");
  if (g.epilogue == HM_EPI_RESID_F32 && (!g.resid || g.ldr < g.N || g.ldr % 4 != 0))
    return hm_set_error(HM_ERR_ARG, "hm_gemm_fp8: residual epilogue needs resid and ldr >= N, ldr % 4 == 0");
  if (g.epilogue == HM_EPI_GELU_MX8 && !g.out_scales) return hm_set_error(HM_ERR_ARG, "hm_gemm_fp8: HM_EPI_GELU_MX8 needs out_scales");
  KArgs k{};
  k.X = g.X8; k.W = g.W8; k.C = g.C; k.bias = g.bias; k.resid = g.resid;
  k.M = g.M; k.N = g.N; k.K = g.K; k.ldx = g.ldx; k.ldw = g.ldw; k.ldc = g.ldc; k.ldr = g.ldr; k.resid_mod = 0;
  k.group_m = g_group_m; k.ksplit = 1;
  k.xs = (const unsigned char*)g.x_scales; k.wscale = g.w_scale; k.out_scales = (unsigned char*)g.out_scales;
  HmProfScope prof(HM_K_GEMM, 16 + g.epilogue, g.M, g.N, g.K, stream);
  switch (g.epilogue) {
    case HM_EPI_STORE:
      if (g.out_dtype != HM_DTYPE_BF16) return hm_set_error(HM_ERR_ARG, "hm_gemm_fp8: HM_EPI_STORE writes bf16 (out_dtype HM_DTYPE_BF16)");
      if (fp8p_ok(k) && (g.ldc & 7) == 0) return launch_fp8p<HM_EPI_STORE>(k, stream);
      return launch_fp8<HM_EPI_STORE>(k, stream);
    case HM_EPI_RESID_F32:
      // persistent form: bit-identical, measured NOT faster (all workgroups reach their 512 KB-per-tile epilogues in
      // lockstep; the one-tile kernel's workgroups drift apart and spread that traffic): opt-in, hm_set_option(HM_OPT_FP8P_RESID, 1)
      if (hm_option(HM_OPT_FP8P_RESID) != 0 && fp8p_ok(k) && k.resid_mod == 0 && (k.ldr & 3) == 0 && (k.ldc & 3) == 0)
        return launch_fp8p<HM_EPI_RESID_F32>(k, stream);
      return launch_fp8<HM_EPI_RESID_F32>(k, stream);
    case HM_EPI_GELU_MX8:
      if (fp8p_ok(k) && (g.ldc & 15) == 0) return launch_fp8p<HM_EPI_GELU_MX8>(k, stream);
      return launch_fp8<HM_EPI_GELU_MX8>(k, stream);
    default: return hm_set_error(HM_ERR_ARG, "hm_gemm_fp8: epilogue must be STORE, RESID_F32 or GELU_MX8");
  }
}

extern "C" size_t hm_conv_splitk_bytes(const hm_conv_args* a) {
  if (!a || a->out_f32 || a->resid || !a->bias || a->ksize <= 0 || a->stride <= 0 || a->Kpad % 64 != 0) return 0;
  const int pad = a->ksize / 2;
  KArgs k{};
  k.Hout = (a->H + 2 * pad - a->ksize) / a->stride + 1; k.Wout = (a->W_in + 2 * pad - a->ksize) / a->stride + 1;
  k.M = a->N * k.Hout * k.Wout; k.N = a->Cout; k.K = a->Kpad; k.ldc = a->ldy;
  const int ks = conv_split_rule(k);
  return ks > 1 ? (size_t)ks * k.M * k.N * 4 : 0;
}

extern "C" int hm_conv2d_nhwc(const hm_conv_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: null args");
  const hm_conv_args& c = *a;
  if (!c.X || !c.W || !c.Y || !c.zeros) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: null operand");
  if (c.N <= 0 || c.H <= 0 || c.W_in <= 0 || c.Cin <= 0 || c.Cout <= 0) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: empty problem");
  if ((c.ksize != 1 && c.ksize != 3 && c.ksize != 5 && c.ksize != 7) || (c.stride != 1 && c.stride != 2))
    return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: kernel size 1, 3, 5 or 7, stride 1 or 2");
  if (c.act < 0 || c.act > 2 || (c.resid && (c.act != 2 || c.ldr % 8 != 0 || c.ldr < c.Cout || c.Cout % 8 != 0 || ((uintptr_t)c.resid & 15))))
    return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: act is 0 / 1 (SiLU) / 2 (ReLU); resid needs act == 2, Cout % 8 == 0, ldr % 8 == 0, 16-byte alignment");
  int lg = 0;
  while ((1 << lg) < c.Cin) ++lg;
  if ((1 << lg) != c.Cin || c.Cin < 8) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: Cin must be a power of two >= 8");
  const int taps = c.ksize * c.ksize, ktrue = taps * c.Cin;
  if (c.Kpad % BK_DEFAULT != 0 || c.Kpad < ktrue) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: Kpad must be a multiple of 64 covering k*k*Cin");
  if (c.Cout % 4 != 0 || c.ldy % 4 != 0 || c.ldy < c.Cout || c.ldx % 8 != 0 || c.ldx < c.Cin)
    return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: Cout % 4, ldy % 4, ldx % 8 must be 0 and cover the channels");
  if ((((uintptr_t)c.X | (uintptr_t)c.W | (uintptr_t)c.zeros | (uintptr_t)c.bias) & 15) || ((uintptr_t)c.Y & 7))
    return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: X/W/zeros/bias 16-byte aligned, Y 8-byte aligned");
  const int pad = c.ksize / 2;
  const int Hout = (c.H + 2 * pad - c.ksize) / c.stride + 1, Wout = (c.W_in + 2 * pad - c.ksize) / c.stride + 1;
  KArgs k{};
  k.X = c.X; k.W = c.W; k.C = c.Y; k.bias = c.bias; k.resid = nullptr;
  k.M = c.N * Hout * Wout; k.N = c.Cout; k.K = c.Kpad; k.ldx = c.ldx; k.ldw = c.Kpad; k.ldc = c.ldy; k.ldr = 0; k.resid_mod = 0;
  k.zeros = c.zeros; k.H = c.H; k.Wd = c.W_in; k.Hout = Hout; k.Wout = Wout; k.ksz = c.ksize; k.stride = c.stride; k.pad = pad;
  k.cin_log2 = lg; k.taps = taps; k.group_m = g_group_m;
  k.general_loader = hm_option(HM_OPT_CONV_GENERAL_LOADER) == 1;
  if (c.out_f32 && c.act) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: f32 output has no activation");
  k.resid16 = c.resid; k.ldr16 = c.ldr;
  const int epi = c.out_f32 ? HM_EPI_F32 : (c.act == 1 ? HM_EPI_SILU : (c.act == 2 ? (c.resid ? HM_EPI_ADD_RELU : HM_EPI_RELU) : HM_EPI_STORE));
  k.ksplit = 1;
  HmProfScope prof(HM_K_CONV, c.ksize * 10 + c.stride, k.M, k.N, ktrue, stream);
  if (c.dtype == HM_DTYPE_BF16) return launch_conv<TBf16>(k, epi, c.splitk_ws, c.splitk_ws_bytes, stream);
  if (c.dtype == HM_DTYPE_F16) return launch_conv<TF16>(k, epi, c.splitk_ws, c.splitk_ws_bytes, stream);
  return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: bad dtype");
}

extern "C" int hm_conv2d_stem_pair(const hm_conv_args* first, const hm_conv_args* second, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!first || !second) return hm_set_error(HM_ERR_ARG, "hm_conv2d_stem_pair: null args");
  const hm_conv_args &a = *first, &b = *second;
  const bool fused = hm_option(HM_OPT_CONV_STEM_PAIR) == 0 && hm_option(HM_OPT_CONV_DIRECT) != 1 &&
      a.X && a.W && a.Y && a.bias && a.zeros && b.W && b.Y && b.bias && a.dtype == b.dtype && (a.dtype == HM_DTYPE_F16 || a.dtype == HM_DTYPE_BF16) &&
      a.ksize == 3 && a.stride == 1 && a.Cin == 8 && a.ldx == 8 && a.Cout == 32 && a.ldy == 32 && a.act == 1 && !a.out_f32 && !a.resid && a.Kpad >= 96 && a.Kpad % 8 == 0 &&
      b.X == a.Y && b.ksize == 3 && b.stride == 2 && b.Cin == 32 && b.ldx == 32 && b.Cout == 64 && b.act == 1 && !b.out_f32 && !b.resid && b.Kpad >= 288 && b.Kpad % 8 == 0 &&
      b.N == a.N && b.H == a.H && b.W_in == a.W_in && a.N > 0 && a.H > 0 && a.W_in > 0 && b.ldy % 8 == 0 && b.ldy >= 64 &&
      ((((uintptr_t)a.X | (uintptr_t)a.W | (uintptr_t)b.W | (uintptr_t)a.bias | (uintptr_t)b.bias | (uintptr_t)b.Y | (uintptr_t)a.zeros) & 15) == 0) &&
      (size_t)20 * a.W_in * 16 < (1ull << 31) && (size_t)16 * ((a.W_in - 1) / 2 + 1) * b.ldy * 2 < (1ull << 31);
  if (!fused) {
    const int rc = hm_conv2d_nhwc(first, stream);
    return rc != HM_OK ? rc : hm_conv2d_nhwc(second, stream);
  }
  StemPairArgs g{};
  g.X = a.X; g.W0 = a.W; g.b0 = a.bias; g.W1 = b.W; g.b1 = b.bias; g.C = b.Y; g.zeros = a.zeros;
  g.ldw0 = a.Kpad; g.ldw1 = b.Kpad; g.ldc = b.ldy; g.NB = a.N; g.H = a.H; g.Wd = a.W_in;
  g.Hout = (a.H - 1) / 2 + 1; g.Wout = (a.W_in - 1) / 2 + 1;
  // one profile record for the launch: the second layer's rows and columns, K = 288 + 4 * 72 / 2: 2 M N K is the flop count of both layers
  HmProfScope prof(HM_K_CONV, 32, g.NB * g.Hout * g.Wout, 64, 288 + 144, stream);
  return a.dtype == HM_DTYPE_BF16 ? launch_conv_stem_pair<TBf16>(g, stream) : launch_conv_stem_pair<TF16>(g, stream);
}
