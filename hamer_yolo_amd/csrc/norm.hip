// nn.LayerNorm over the last dimension (vit.py:136,:144,:252 with eps 1e-6; the decoder's
// PreNorm, pose_transformer.py:33-37 / t_cond_mlp.py:51-52, eps 1e-5).
// HBM-bound: one wavefront per row, the row lives in registers (16-byte loads), two-pass
// mean / variance in fp32, output written as bf16/fp16 (the next GEMM's operand) or f32.
#include "common.h"
#include "hamer_hip_internal.h"

namespace {

constexpr int MAXJ_LIMIT = 8;  // D <= 64 lanes * 4 floats * 8 = 2048

template <class OutT>
__device__ __forceinline__ void store4(OutT* p, f32x4_t v);
template <>
__device__ __forceinline__ void store4<float>(float* p, f32x4_t v) { *(f32x4_t*)p = v; }
template <>
__device__ __forceinline__ void store4<__bf16>(__bf16* p, f32x4_t v) {
  bf16x4_t o; o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
  *(bf16x4_t*)p = o;
}
template <>
__device__ __forceinline__ void store4<_Float16>(_Float16* p, f32x4_t v) {
  f16x4_t o; o[0] = (_Float16)v[0]; o[1] = (_Float16)v[1]; o[2] = (_Float16)v[2]; o[3] = (_Float16)v[3];
  *(f16x4_t*)p = o;
}

// MAXJ = ceil(D / 256) 16-byte pieces per lane.  The affine parameters are fetched after the reductions: fetching
// them with the row (15 loads in flight per lane) measured 1.5x SLOWER on MI355X -- the kernel lives on occupancy.
// ACC: the residual add of a split-K producer rides along: x[row] += bias + sum_s part[s][row] (s ascending, so the
// sum has one order whatever the launch), written back before the statistics are taken.
template <class OutT, int MAXJ, bool ACC = false>
__global__ __launch_bounds__(256, 8) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, OutT* __restrict__ out, int M,
                                                        int D, float eps, const float* __restrict__ part = nullptr,
                                                        int n_part = 0, const float* __restrict__ bias = nullptr) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * D;
  f32x4_t v[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int i = lane * 4 + j * 256;
    v[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (i < D) v[j] = *(const f32x4_t*)(xr + i);
  }
  if constexpr (ACC) {
    f32x4_t a[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int i = lane * 4 + j * 256;
      a[j] = (bias && i < D) ? *(const f32x4_t*)(bias + i) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 4
    for (int s = 0; s < n_part; ++s) {             // slab by slab, four slabs (4 x MAXJ loads) in flight together
      const float* pr = part + ((size_t)s * M + row) * D;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int i = lane * 4 + j * 256;
        if (i < D) a[j] += *(const f32x4_t*)(pr + i);
      }
    }
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int i = lane * 4 + j * 256;
      if (i < D) { v[j] += a[j]; *(f32x4_t*)(const_cast<float*>(xr) + i) = v[j]; }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int i = lane * 4 + j * 256;
    if (i < D) {
      const f32x4_t d = v[j] - mean;
      q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  OutT* orow = out + (size_t)row * D;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int i = lane * 4 + j * 256;
    if (i < D) store4<OutT>(orow + i, (v[j] - mean) * rstd * *(const f32x4_t*)(gamma + i) + *(const f32x4_t*)(beta + i));
  }
}

// LayerNorm with MXFP8 output for hm_gemm_fp8: the 8 lanes that hold 32 consecutive columns agree on the block's E8M0
// scale (DPP max), every lane packs its 4 values into one dword of e4m3 bytes.  Scales go to [D/32][M].
template <int MAXJ>
__global__ __launch_bounds__(256, 8) void layernorm_mx8_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int* __restrict__ out8,
                                                            unsigned char* __restrict__ scales, int M, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * D;
  f32x4_t v[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int i = lane * 4 + j * 256;
    v[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (i < D) v[j] = *(const f32x4_t*)(xr + i);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int i = lane * 4 + j * 256;
    if (i < D) {
      const f32x4_t d = v[j] - mean;
      q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int i = lane * 4 + j * 256;
    const bool ok = i < D;                             // whole 8-lane groups are in or out (D % 32 == 0)
    f32x4_t y = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (ok) y = (v[j] - mean) * rstd * *(const f32x4_t*)(gamma + i) + *(const f32x4_t*)(beta + i);
    const float amax = row8_max(fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))));
    const unsigned sb = mx8_scale_byte(amax);
    const float inv = mx8_inv_scale(sb);
    if (ok) {
      out8[((size_t)row * D + i) >> 2] = mx8_pack4(y[0] * inv, y[1] * inv, y[2] * inv, y[3] * inv);
      if ((lane & 7) == 0) scales[(size_t)(i >> 5) * M + row] = (unsigned char)sb;
    }
  }
}

// The production LayerNorm (no split-K accumulation): the same arithmetic, two changes for bandwidth.
//  * Rows per wave.  12288 rows on one-row waves are 1.5 x the 8192 waves the chip holds: the second half-round leaves
//    half of the machine idle.  Here the grid is sized to be resident at once and every wave walks `rows_per_wave` rows
//    (stride = waves in the grid, so neighbouring waves stream neighbouring rows), the next row's loads issued before the
//    current row's reductions.
//  * 16-byte stores of the 16-bit output.  A lane holds columns 4l..4l+3 of every 256-column piece (16-byte loads); for the
//    output it trades halves with its neighbour (DPP swap inside lane pairs) so that the even lane stores columns
//    4l..4l+7 of piece j and the odd lane those of piece j+1: every store instruction is 16 bytes per lane.
__device__ __forceinline__ unsigned pair_swap(unsigned v) {
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
}
template <class OutT> __device__ __forceinline__ uint2 pack4h(f32x4_t v);
template <> __device__ __forceinline__ uint2 pack4h<__bf16>(f32x4_t v) {
  bf16x4_t o; o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
  return __builtin_bit_cast(uint2, o);
}
template <> __device__ __forceinline__ uint2 pack4h<_Float16>(f32x4_t v) {
  f16x4_t o; o[0] = (_Float16)v[0]; o[1] = (_Float16)v[1]; o[2] = (_Float16)v[2]; o[3] = (_Float16)v[3];
  return __builtin_bit_cast(uint2, o);
}

template <class OutT, int MAXJ>
__global__ __launch_bounds__(256, 4) void layernorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, OutT* __restrict__ out, int M,
                                                                int D, float eps, int rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const bool full = D == MAXJ * 256;                 // every piece complete: the paired 16-byte stores apply
  f32x4_t v[MAXJ], nx[MAXJ];
  auto fetch = [&](int row, f32x4_t (&dst)[MAXJ]) {
    const float* xr = x + (size_t)row * D;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int i = lane * 4 + j * 256;
      dst[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (i < D) dst[j] = *(const f32x4_t*)(xr + i);
    }
  };
  int row = wave;
  if (row < M) fetch(row, v);
  for (int k = 0; k < rows_per_wave && row < M; ++k, row += nwaves) {
    const int nrow = row + nwaves;
    if (k + 1 < rows_per_wave && nrow < M) fetch(nrow, nx);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int i = lane * 4 + j * 256;
      if (i < D) {
        const f32x4_t d = v[j] - mean;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    OutT* orow = out + (size_t)row * D;
    if constexpr (sizeof(OutT) == 2) {
      if (full) {
        const bool odd = lane & 1;
#pragma unroll
        for (int j = 0; j + 1 < MAXJ; j += 2) {
          const int i0 = lane * 4 + j * 256, i1 = i0 + 256;
          const uint2 p0 = pack4h<OutT>((v[j] - mean) * rstd * *(const f32x4_t*)(gamma + i0) + *(const f32x4_t*)(beta + i0));
          const uint2 p1 = pack4h<OutT>((v[j + 1] - mean) * rstd * *(const f32x4_t*)(gamma + i1) + *(const f32x4_t*)(beta + i1));
          const uint2 send = odd ? p0 : p1;
          const uint2 recv = uint2{pair_swap(send.x), pair_swap(send.y)};
          const uint4 o = odd ? uint4{recv.x, recv.y, p1.x, p1.y} : uint4{p0.x, p0.y, recv.x, recv.y};
          *(uint4*)(orow + (odd ? i1 - 4 : i0)) = o;
        }
        if constexpr (MAXJ & 1) {
          const int i = lane * 4 + (MAXJ - 1) * 256;
          *(uint2*)(orow + i) = pack4h<OutT>((v[MAXJ - 1] - mean) * rstd * *(const f32x4_t*)(gamma + i) + *(const f32x4_t*)(beta + i));
        }
        goto next_row;
      }
    }
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int i = lane * 4 + j * 256;
      if (i < D) store4<OutT>(orow + i, (v[j] - mean) * rstd * *(const f32x4_t*)(gamma + i) + *(const f32x4_t*)(beta + i));
    }
  next_row:
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) v[j] = nx[j];
  }
}

template <class OutT>
void launch_ln(const float* x, const float* g, const float* b, OutT* out, int M, int D, float eps, hipStream_t s) {
  const int mj = (D + 255) / 256;
  // all waves resident at once: <= 16 waves per CU (4 workgroups of 256), every wave ceil(M / waves) rows
  const int cus = hm_device_cu_count() > 0 ? hm_device_cu_count() : 256;
  const int max_waves = cus * 16;
  const int rpw = (M + max_waves - 1) / max_waves;
  const int waves = (M + rpw - 1) / rpw;
  dim3 grid((waves + 3) / 4), block(256);
  if (mj <= 1) hipLaunchKernelGGL((layernorm_rows_kernel<OutT, 1>), grid, block, 0, s, x, g, b, out, M, D, eps, rpw);
  else if (mj <= 2) hipLaunchKernelGGL((layernorm_rows_kernel<OutT, 2>), grid, block, 0, s, x, g, b, out, M, D, eps, rpw);
  else if (mj <= 4) hipLaunchKernelGGL((layernorm_rows_kernel<OutT, 4>), grid, block, 0, s, x, g, b, out, M, D, eps, rpw);
  else if (mj <= 5) hipLaunchKernelGGL((layernorm_rows_kernel<OutT, 5>), grid, block, 0, s, x, g, b, out, M, D, eps, rpw);
  else hipLaunchKernelGGL((layernorm_rows_kernel<OutT, 8>), grid, block, 0, s, x, g, b, out, M, D, eps, rpw);
}

__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float* __restrict__ vec, float* __restrict__ out,
                                                             int B, int D) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (size_t)B * D) out[i] = vec[i % D];
}

}  // namespace

extern "C" int hm_layernorm(const float* x, const float* gamma, const float* beta, void* out, int out_dtype, int M,
                            int D, float eps, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!x || !gamma || !beta || !out) return hm_set_error(HM_ERR_ARG, "hm_layernorm: null pointer");
  if (M <= 0 || D <= 0 || D % 4 != 0 || D > 256 * MAXJ_LIMIT)
    return hm_set_error(HM_ERR_ARG, "hm_layernorm: need 0 < D <= 2048, D % 4 == 0, M > 0");
  HmProfScope prof(HM_K_LAYERNORM, out_dtype, M, D, 0, s);
  if (out_dtype == HM_DTYPE_BF16) launch_ln<__bf16>(x, gamma, beta, (__bf16*)out, M, D, eps, s);
  else if (out_dtype == HM_DTYPE_F16) launch_ln<_Float16>(x, gamma, beta, (_Float16*)out, M, D, eps, s);
  else if (out_dtype == HM_OUT_F32) launch_ln<float>(x, gamma, beta, (float*)out, M, D, eps, s);
  else return hm_set_error(HM_ERR_ARG, "hm_layernorm: bad out_dtype");
  return hm_check_launch("hm_layernorm");
}

template <class OutT>
void launch_ln_acc(float* x, const float* part, int n_part, const float* bias, const float* g, const float* b, OutT* out, int M,
                   int D, float eps, hipStream_t s) {
  dim3 grid((M + 3) / 4), block(256);
  const int mj = (D + 255) / 256;
  if (mj <= 2) hipLaunchKernelGGL((layernorm_kernel<OutT, 2, true>), grid, block, 0, s, x, g, b, out, M, D, eps, part, n_part, bias);
  else if (mj <= 5) hipLaunchKernelGGL((layernorm_kernel<OutT, 5, true>), grid, block, 0, s, x, g, b, out, M, D, eps, part, n_part, bias);
  else hipLaunchKernelGGL((layernorm_kernel<OutT, 8, true>), grid, block, 0, s, x, g, b, out, M, D, eps, part, n_part, bias);
}

extern "C" int hm_layernorm_accum(float* x, const float* partials, int n_partials, const float* bias, const float* gamma,
                                  const float* beta, void* out, int out_dtype, int M, int D, float eps, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!x || !partials || !gamma || !beta || !out || n_partials <= 0) return hm_set_error(HM_ERR_ARG, "hm_layernorm_accum: null pointer");
  if (M <= 0 || D <= 0 || D % 4 != 0 || D > 256 * MAXJ_LIMIT)
    return hm_set_error(HM_ERR_ARG, "hm_layernorm_accum: need 0 < D <= 2048, D % 4 == 0, M > 0");
  HmProfScope prof(HM_K_LAYERNORM, out_dtype, M, D, n_partials, s);
  if (out_dtype == HM_DTYPE_BF16) launch_ln_acc<__bf16>(x, partials, n_partials, bias, gamma, beta, (__bf16*)out, M, D, eps, s);
  else if (out_dtype == HM_DTYPE_F16) launch_ln_acc<_Float16>(x, partials, n_partials, bias, gamma, beta, (_Float16*)out, M, D, eps, s);
  else if (out_dtype == HM_OUT_F32) launch_ln_acc<float>(x, partials, n_partials, bias, gamma, beta, (float*)out, M, D, eps, s);
  else return hm_set_error(HM_ERR_ARG, "hm_layernorm_accum: bad out_dtype");
  return hm_check_launch("hm_layernorm_accum");
}

extern "C" int hm_layernorm_mx8(const float* x, const float* gamma, const float* beta, void* out8, void* out_scales, int M,
                                int D, float eps, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!x || !gamma || !beta || !out8 || !out_scales) return hm_set_error(HM_ERR_ARG, "hm_layernorm_mx8: null pointer");
  if (M <= 0 || D <= 0 || D % 32 != 0 || D > 256 * MAXJ_LIMIT)
    return hm_set_error(HM_ERR_ARG, "hm_layernorm_mx8: need 0 < D <= 2048, D % 32 == 0, M > 0");
  HmProfScope prof(HM_K_LAYERNORM, 3, M, D, 0, s);
  dim3 grid((M + 3) / 4), block(256);
  const int mj = (D + 255) / 256;
  if (mj <= 2) hipLaunchKernelGGL((layernorm_mx8_kernel<2>), grid, block, 0, s, x, gamma, beta, (int*)out8, (unsigned char*)out_scales, M, D, eps);
  else if (mj <= 5) hipLaunchKernelGGL((layernorm_mx8_kernel<5>), grid, block, 0, s, x, gamma, beta, (int*)out8, (unsigned char*)out_scales, M, D, eps);
  else hipLaunchKernelGGL((layernorm_mx8_kernel<8>), grid, block, 0, s, x, gamma, beta, (int*)out8, (unsigned char*)out_scales, M, D, eps);
  return hm_check_launch("hm_layernorm_mx8");
}

extern "C" int hm_broadcast_rows(const float* vec, float* out, int B, int D, void* stream_) {
  if (!vec || !out || B <= 0 || D <= 0) return hm_set_error(HM_ERR_ARG, "hm_broadcast_rows: bad arguments");
  const size_t n = (size_t)B * D;
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, vec,
                     out, B, D);
  return hm_check_launch("hm_broadcast_rows");
}

// ---------------------------------------------------------------------------------------------------------------
// Range probe (load-time calibration of the fp16 prescale, hm_hamer_weights.range_stats): max |x| of a column range of a 16-bit
// matrix.  Non-negative floats order like their bit patterns, so one integer atomic max per workgroup folds the result; an
// infinity or NaN in the data (bit pattern above every finite one) reads back as non-finite, which is what the caller tests.
template <class E>
__global__ __launch_bounds__(256) void absmax16_kernel(const E* __restrict__ x, int ld, int M, int col0, int ncols, float* slot) {
  const size_t total = (size_t)M * ncols;
  float m = 0.f;
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / ncols, c = i - r * ncols;
    const float v = fabsf((float)x[r * ld + col0 + c]);
    bad |= !(v <= 3.0e38f);
    m = fmaxf(m, v);
  }
  if (bad) m = __builtin_inff();
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned*)slot, __float_as_uint(m));
}

extern "C" int hm_absmax16(const void* x, int ld, int M, int col0, int ncols, int dtype, float* slot, void* stream_) {
  if (!x || !slot || M <= 0 || ncols <= 0 || ld < col0 + ncols || col0 < 0) return hm_set_error(HM_ERR_ARG, "hm_absmax16: bad arguments");
  const size_t total = (size_t)M * ncols;
  const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  if (dtype == HM_DTYPE_BF16) hipLaunchKernelGGL((absmax16_kernel<__bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const __bf16*)x, ld, M, col0, ncols, slot);
  else if (dtype == HM_DTYPE_F16) hipLaunchKernelGGL((absmax16_kernel<_Float16>), dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const _Float16*)x, ld, M, col0, ncols, slot);
  else return hm_set_error(HM_ERR_ARG, "hm_absmax16: dtype must be HM_DTYPE_BF16 or HM_DTYPE_F16");
  return hm_check_launch("hm_absmax16");
}
