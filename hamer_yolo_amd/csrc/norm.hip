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

template <class OutT>
void launch_ln(const float* x, const float* g, const float* b, OutT* out, int M, int D, float eps, hipStream_t s) {
  dim3 grid((M + 3) / 4), block(256);
  const int mj = (D + 255) / 256;
  if (mj <= 1) hipLaunchKernelGGL((layernorm_kernel<OutT, 1>), grid, block, 0, s, x, g, b, out, M, D, eps);
  else if (mj <= 2) hipLaunchKernelGGL((layernorm_kernel<OutT, 2>), grid, block, 0, s, x, g, b, out, M, D, eps);
  else if (mj <= 4) hipLaunchKernelGGL((layernorm_kernel<OutT, 4>), grid, block, 0, s, x, g, b, out, M, D, eps);
  else if (mj <= 5) hipLaunchKernelGGL((layernorm_kernel<OutT, 5>), grid, block, 0, s, x, g, b, out, M, D, eps);
  else hipLaunchKernelGGL((layernorm_kernel<OutT, 8>), grid, block, 0, s, x, g, b, out, M, D, eps);
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
