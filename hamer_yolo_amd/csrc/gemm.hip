// C[M,N] = epilogue(X[M,K] . W[N,K]^T) on MFMA, in two flavours sharing one kernel body:
//  * hm_gemm          : plain row-major X -- the ViT-H patch-embed / QKV / proj / MLP GEMMs and the
//                       decoder's to_kv projection (nn.Linear calls at vit.py:82-87,:110-126,
//                       pose_transformer.py:114; >97 % of the FLOPs of HAMER.forward_step).
//  * hm_conv2d_nhwc   : implicit GEMM for the YOLOv7 convolutions (Conv.fuseforward common.py:114,
//                       RepConv common.py:502-504, Detect yolo.py:151): X rows are output pixels, the K
//                       axis walks (ky, kx, ci) of an NHWC tensor, out-of-image taps read a zero line.
//
// gfx950 design: 128 x (32*NI) x 64 block tile, 4 waves (2x2), each wave 64 x (16*NI) outputs as
// NI x 4 MFMA 16x16x32 accumulators.  Both operands are K-contiguous and staged global->LDS with
// 16-byte LDS-DMA (global_load_lds_dwordx4) into two buffers; the LDS image is lane-linear, so the
// bank-conflict XOR swizzle (16-B chunk ^= row&7 inside each 128-B row) is applied to the per-lane
// SOURCE address and again on the ds_read_b128 fragment reads.  The MFMA "A" operand is the W tile
// and "B" the X tile, so a lane's 4 accumulator registers are 4 consecutive output columns of one
// row: bias / residual / outputs move as 8- or 16-byte vectors.
#include "common.h"
#include "hamer_hip_internal.h"

namespace {

constexpr int BM = 128, BK = 64;
constexpr int XTILE_BYTES = BM * BK * 2;          // 16 KB

struct KArgs {                                    // kernel-side view of either entry point
  const void* X; const void* W; void* C; const float* bias; const float* resid;
  int M, N, K, ldx, ldw, ldc, ldr, resid_mod;
  // convolution geometry (CONV only)
  const void* zeros;
  int H, Wd, Hout, Wout, ksz, stride, pad, cin_log2, taps;
};

template <class T, int EPI, int NI, bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const KArgs g) {
  constexpr int BN = 32 * NI;
  constexpr int WTILE_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = XTILE_BYTES + WTILE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  using vec8 = typename T::vec8;
  using elem = typename T::elem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.N + BN - 1) / BN;
  const int wgid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wgid / tiles_n) * BM, n0 = (wgid % tiles_n) * BN;
  const int wr = wave >> 1, wc = wave & 1;

  const elem* __restrict__ X = (const elem*)g.X;
  const elem* __restrict__ W = (const elem*)g.W;

  // staging geometry: wave w loads X rows [32w, 32w+32) and W rows [8*NI*w, 8*NI*(w+1)); one
  // instruction = 8 rows x 128 B; lane -> (row = lane>>3, physical 16-B chunk = lane&7)
  const int chunk = (lane & 7) ^ ((lane >> 3) & 7);       // logical chunk of this lane (same for all its rows)
  const elem* xsrc[4];
  int pix_y[4], pix_x[4];
  const elem* wsrc[NI];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int gm = m0 + wave * 32 + i * 8 + (lane >> 3);
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
  for (int i = 0; i < NI; ++i) {
    int gn = n0 + wave * 8 * NI + i * 8 + (lane >> 3);
    gn = gn < g.N ? gn : g.N - 1;
    wsrc[i] = W + (size_t)gn * g.ldw + chunk * 8;
  }

  auto stage = [&](int buf, int kt) {
    char* lx = smem + buf * STAGE_BYTES + wave * 32 * 128;
    char* lw = smem + buf * STAGE_BYTES + XTILE_BYTES + wave * 8 * NI * 128;
    if (CONV) {
      const int k = kt * BK + chunk * 8;
      const int tap = k >> g.cin_log2, ci = k & ((1 << g.cin_log2) - 1);
      const int ky = tap / g.ksz, kx = tap - ky * g.ksz;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = pix_y[i] + ky, ix = pix_x[i] + kx;
        const bool ok = tap < g.taps && iy >= 0 && iy < g.H && ix >= 0 && ix < g.Wd;
        const elem* src = ok ? xsrc[i] + ((size_t)iy * g.Wd + ix) * g.ldx + ci : (const elem*)g.zeros;
        glds16(src, lx + i * 8 * 128);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16(xsrc[i] + kt * BK, lx + i * 8 * 128);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) glds16(wsrc[i] + kt * BK, lw + i * 8 * 128);
  };

  f32x4_t acc[NI][4];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment reads: row = tile_row + (lane&15); 16-B chunk = ks*4 + (lane>>4), swizzled by row&7
  const int frow = lane & 15, fsw = lane & 7, fch = lane >> 4;
  auto compute = [&](int buf) {
    const char* lx = smem + buf * STAGE_BYTES;
    const char* lw = lx + XTILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fch) ^ fsw) * 16;
      vec8 wf[NI], xf[4];
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[i] = *(const vec8*)(lw + (wc * 16 * NI + i * 16 + frow) * 128 + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[i] = *(const vec8*)(lx + (wr * 64 + i * 16 + frow) * 128 + coff);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
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
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wc * 16 * NI + ni * 16 + (lane >> 4) * 4;
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

template <class T, int NI, bool CONV>
int launch_epi(const KArgs& g, int epilogue, hipStream_t s, const char* what) {
  constexpr int BN = 32 * NI;
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  dim3 grid(tiles), block(256);
  switch (epilogue) {
    case HM_EPI_STORE: hipLaunchKernelGGL((gemm_tn_kernel<T, HM_EPI_STORE, NI, CONV>), grid, block, 0, s, g); break;
    case HM_EPI_SILU: hipLaunchKernelGGL((gemm_tn_kernel<T, HM_EPI_SILU, NI, CONV>), grid, block, 0, s, g); break;
    case HM_EPI_F32: hipLaunchKernelGGL((gemm_tn_kernel<T, HM_EPI_F32, NI, CONV>), grid, block, 0, s, g); break;
    case HM_EPI_GELU:
      if (CONV) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: unsupported epilogue");
      hipLaunchKernelGGL((gemm_tn_kernel<T, HM_EPI_GELU, NI, false>), grid, block, 0, s, g); break;
    case HM_EPI_RESID_F32:
      if (CONV) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: unsupported epilogue");
      hipLaunchKernelGGL((gemm_tn_kernel<T, HM_EPI_RESID_F32, NI, false>), grid, block, 0, s, g); break;
    default: return hm_set_error(HM_ERR_ARG, "unknown epilogue");
  }
  return hm_check_launch(what);
}

template <class T, bool CONV>
int launch_ni(const KArgs& g, int epilogue, hipStream_t s, const char* what) {
  if constexpr (!CONV) {
    return launch_epi<T, 4, false>(g, epilogue, s, what);
  } else {
    if (g.N > 64) return launch_epi<T, 4, true>(g, epilogue, s, what);
    if (g.N > 32) return launch_epi<T, 2, true>(g, epilogue, s, what);
    return launch_epi<T, 1, true>(g, epilogue, s, what);
  }
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
  KArgs k{};
  k.X = g.X; k.W = g.W; k.C = g.C; k.bias = g.bias; k.resid = g.resid;
  k.M = g.M; k.N = g.N; k.K = g.K; k.ldx = g.ldx; k.ldw = g.ldw; k.ldc = g.ldc; k.ldr = g.ldr; k.resid_mod = g.resid_mod;
  HmProfScope prof(HM_K_GEMM, g.epilogue, g.M, g.N, g.K, stream);
  if (g.dtype == HM_DTYPE_BF16) return launch_ni<TBf16, false>(k, g.epilogue, stream, "hm_gemm");
  if (g.dtype == HM_DTYPE_F16) return launch_ni<TF16, false>(k, g.epilogue, stream, "hm_gemm");
  return hm_set_error(HM_ERR_ARG, "hm_gemm: dtype must be HM_DTYPE_BF16 or HM_DTYPE_F16");
}

extern "C" int hm_conv2d_nhwc(const hm_conv_args* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: null args");
  const hm_conv_args& c = *a;
  if (!c.X || !c.W || !c.Y || !c.zeros) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: null operand");
  if (c.N <= 0 || c.H <= 0 || c.W_in <= 0 || c.Cin <= 0 || c.Cout <= 0) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: empty problem");
  if ((c.ksize != 1 && c.ksize != 3) || (c.stride != 1 && c.stride != 2))
    return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: kernel size 1 or 3, stride 1 or 2");
  int lg = 0;
  while ((1 << lg) < c.Cin) ++lg;
  if ((1 << lg) != c.Cin || c.Cin < 8) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: Cin must be a power of two >= 8");
  const int taps = c.ksize * c.ksize, ktrue = taps * c.Cin;
  if (c.Kpad % BK != 0 || c.Kpad < ktrue) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: Kpad must be a multiple of 64 covering k*k*Cin");
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
  k.cin_log2 = lg; k.taps = taps;
  if (c.out_f32 && c.act) return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: f32 output has no activation");
  const int epi = c.out_f32 ? HM_EPI_F32 : (c.act ? HM_EPI_SILU : HM_EPI_STORE);
  HmProfScope prof(HM_K_CONV, c.ksize * 10 + c.stride, k.M, k.N, ktrue, stream);
  if (c.dtype == HM_DTYPE_BF16) return launch_ni<TBf16, true>(k, epi, stream, "hm_conv2d_nhwc");
  if (c.dtype == HM_DTYPE_F16) return launch_ni<TF16, true>(k, epi, stream, "hm_conv2d_nhwc");
  return hm_set_error(HM_ERR_ARG, "hm_conv2d_nhwc: bad dtype");
}
