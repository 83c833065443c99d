// C[M,N] = epilogue(X[M,K] . W[N,K]^T): the ViT-H patch-embed / QKV / proj / MLP GEMMs and
// the decoder's to_kv projection (reference: nn.Linear calls at vit.py:82-87,:110-126,
// pose_transformer.py:114; >97 % of the FLOPs of HAMER.forward_step).
//
// gfx950 design: 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 output tile as 4x4
// MFMA 16x16x32 accumulators.  Both operands are K-contiguous, staged global->LDS with
// 16-byte LDS-DMA (global_load_lds_dwordx4) into two 32 KB buffers; the LDS image is
// lane-linear, so the bank-conflict XOR swizzle (16-B chunk ^= row&7 inside each 128-B row)
// is applied to the per-lane SOURCE address and again on the ds_read_b128 fragment reads.
// The MFMA "A" operand is the W tile and "B" the X tile, so a lane's 4 accumulator
// registers are 4 consecutive output columns of one row: bias/residual/outputs move as
// 8- or 16-byte vectors.
#include "common.h"
#include "hamer_hip_internal.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // X tile + W tile

template <class T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const hm_gemm_args g) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  using vec8 = typename T::vec8;
  using elem = typename T::elem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int wgid = xcd_remap(blockIdx.x, nwg);
  const int m0 = (wgid / tiles_n) * BM, n0 = (wgid % tiles_n) * BN;
  const int wr = wave >> 1, wc = wave & 1;

  const elem* __restrict__ X = (const elem*)g.X;
  const elem* __restrict__ W = (const elem*)g.W;

  // per-lane source rows for the 4 staging instructions of this wave (rows clamped at the edge:
  // edge rows are loaded from valid memory and never stored)
  const int srow = wave * 32 + (lane >> 3);               // + i*8
  const int chunk_phys = lane & 7;
  const elem* xsrc[4];
  const elem* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = srow + i * 8;
    const int c = chunk_phys ^ (r & 7);
    int gm = m0 + r; gm = gm < g.M ? gm : g.M - 1;
    int gn = n0 + r; gn = gn < g.N ? gn : g.N - 1;
    xsrc[i] = X + (size_t)gm * g.ldx + c * 8;
    wsrc[i] = W + (size_t)gn * g.ldw + c * 8;
  }

  auto stage = [&](int buf, int kt) {
    char* lx = smem + buf * STAGE_BYTES + wave * 32 * 128;
    char* lw = lx + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(xsrc[i] + kt * BK, lx + i * 8 * 128);
      glds16(wsrc[i] + kt * BK, lw + i * 8 * 128);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = tile_row + (lane&15); 16-B chunk = ks*4 + (lane>>4), swizzled by row&7
  const int frow = lane & 15;
  const int fsw = lane & 7;
  const int fch = lane >> 4;

  auto compute = [&](int buf) {
    const char* lx = smem + buf * STAGE_BYTES;
    const char* lw = lx + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fch) ^ fsw) * 16;
      vec8 wf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wf[i] = *(const vec8*)(lw + (wc * 64 + i * 16 + frow) * 128 + coff);
        xf[i] = *(const vec8*)(lx + (wr * 64 + i * 16 + frow) * 128 + coff);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = T::mfma(wf[ni], xf[mi], acc[ni][mi]);
    }
  };

  const int nk = g.K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = ..+(lane&15), n = ..+4*(lane>>4)
  const float* __restrict__ bias = g.bias;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = n0 + wc * 64 + ni * 16 + (lane >> 4) * 4;
    if (n >= g.N) continue;
    f32x4_t bv = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *(const f32x4_t*)(bias + n);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = m0 + wr * 64 + mi * 16 + (lane & 15);
      if (m >= g.M) continue;
      f32x4_t v = acc[ni][mi] + bv;
      if (EPI == HM_EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      } else if (EPI == HM_EPI_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu(v[r]);
      }
      if (EPI == HM_EPI_RESID_F32) {
        const int rm = g.resid_mod > 0 ? m % g.resid_mod : m;
        v += *(const f32x4_t*)(g.resid + (size_t)rm * g.ldr + n);
        *(f32x4_t*)((float*)g.C + (size_t)m * g.ldc + n) = v;
      } else if (EPI == HM_EPI_F32) {
        *(f32x4_t*)((float*)g.C + (size_t)m * g.ldc + n) = v;
      } else {
        typename T::vec4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (elem)v[r];
        *(typename T::vec4*)((elem*)g.C + (size_t)m * g.ldc + n) = o;
      }
    }
  }
}

template <class T>
int launch_t(const hm_gemm_args& g, hipStream_t s) {
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  dim3 grid(tiles), block(256);
  switch (g.epilogue) {
    case HM_EPI_STORE: hipLaunchKernelGGL((gemm_tn_kernel<T, HM_EPI_STORE>), grid, block, 0, s, g); break;
    case HM_EPI_GELU: hipLaunchKernelGGL((gemm_tn_kernel<T, HM_EPI_GELU>), grid, block, 0, s, g); break;
    case HM_EPI_SILU: hipLaunchKernelGGL((gemm_tn_kernel<T, HM_EPI_SILU>), grid, block, 0, s, g); break;
    case HM_EPI_RESID_F32: hipLaunchKernelGGL((gemm_tn_kernel<T, HM_EPI_RESID_F32>), grid, block, 0, s, g); break;
    case HM_EPI_F32: hipLaunchKernelGGL((gemm_tn_kernel<T, HM_EPI_F32>), grid, block, 0, s, g); break;
    default: return hm_set_error(HM_ERR_ARG, "hm_gemm: unknown epilogue");
  }
  return hm_check_launch("hm_gemm");
}

}  // namespace

extern "C" int hm_gemm(const hm_gemm_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a) return hm_set_error(HM_ERR_ARG, "hm_gemm: null args");
  const hm_gemm_args& g = *a;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return hm_set_error(HM_ERR_ARG, "hm_gemm: empty problem");
  if (g.K % BK != 0) return hm_set_error(HM_ERR_ARG, "hm_gemm: K must be a multiple of 64");
  if (g.N % 4 != 0 || g.ldc % 4 != 0) return hm_set_error(HM_ERR_ARG, "hm_gemm: N and ldc must be multiples of 4");
  if (g.ldx % 8 != 0 || g.ldw % 8 != 0) return hm_set_error(HM_ERR_ARG, "hm_gemm: ldx/ldw must be multiples of 8 elements");
  if (g.ldx < g.K || g.ldw < g.K || g.ldc < g.N) return hm_set_error(HM_ERR_ARG, "hm_gemm: leading dimension too small");
  if (!g.X || !g.W || !g.C) return hm_set_error(HM_ERR_ARG, "hm_gemm: null operand");
  if (g.epilogue == HM_EPI_RESID_F32 && (!g.resid || g.ldr < g.N || g.ldr % 4 != 0))
    return hm_set_error(HM_ERR_ARG, "hm_gemm: residual epilogue needs resid and ldr >= N, ldr % 4 == 0");
  if (((uintptr_t)g.X | (uintptr_t)g.W | (uintptr_t)g.C | (uintptr_t)g.bias | (uintptr_t)g.resid) & 15)
    return hm_set_error(HM_ERR_ARG, "hm_gemm: pointers must be 16-byte aligned");
  HmProfScope prof(HM_K_GEMM, g.epilogue, g.M, g.N, g.K, stream);
  if (g.dtype == HM_DTYPE_BF16) return launch_t<TBf16>(g, stream);
  if (g.dtype == HM_DTYPE_F16) return launch_t<TF16>(g, stream);
  return hm_set_error(HM_ERR_ARG, "hm_gemm: dtype must be HM_DTYPE_BF16 or HM_DTYPE_F16");
}
