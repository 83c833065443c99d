// The MANO transformer-decoder head in fp32 (heads/mano_head.py:61-95,
// components/pose_transformer.py:40-124,:191-201): one query token per hand, so apart from
// to_kv (a big MFMA GEMM, gemm.hip) every layer is M = batch rows of small matrices.
//   hm_linear_f32     : out = act(x . W^T + b) (+ resid) on the f32-input MFMA
//                       (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains, no precision loss).
//   hm_cross_attention: softmax(q k^T * scale) v for one query over 192 context tokens.
#include "common.h"
#include "hamer_hip_internal.h"

namespace {

// block = 4 waves, one 16(m) x 16(n) output tile; the waves split K in 16-wide blocks and the
// partial tiles are summed through LDS.  MFMA A operand = W rows (n), B operand = x rows (m):
// lane l ends with out[m = l&15][n0 + 4*(l>>4) .. +3].  Within a k-block every lane loads 4
// consecutive k (one 16-byte load per operand); MFMA i of the block consumes element i, i.e.
// k-slot g of MFMA i is k = kb + 4g + i on both operands.
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ W, int ldw,
                                                         const float* __restrict__ bias, const float* resid, int ldr,
                                                         float* out, int ldo, int M, int N, int K, int act) {
  __shared__ f32x4_t part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int g = lane >> 4, li = lane & 15;
  int mr = m0 + li; mr = mr < M ? mr : M - 1;
  int nr = n0 + li; nr = nr < N ? nr : N - 1;
  const float* xp = x + (size_t)mr * ldx + 4 * g;
  const float* wp = W + (size_t)nr * ldw + 4 * g;
  f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // the loop is latency-bound (few waves, dependent MFMA chain): fetch 4 k-blocks ahead so 8 loads are in flight
  constexpr int PF = 4;
  int kb = wave * 16;
  for (; kb + 64 * (PF - 1) < K; kb += 64 * PF) {
    f32x4_t xv[PF], wv[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) { xv[p] = *(const f32x4_t*)(xp + kb + 64 * p); wv[p] = *(const f32x4_t*)(wp + kb + 64 * p); }
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[p][i], xv[p][i], acc, 0, 0, 0);
  }
  for (; kb < K; kb += 64) {
    const f32x4_t xv = *(const f32x4_t*)(xp + kb);
    const f32x4_t wv = *(const f32x4_t*)(wp + kb);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i], xv[i], acc, 0, 0, 0);
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
  f32x4_t v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  const int m = m0 + li, n = n0 + 4 * g;
  if (m >= M || n >= N) return;
  if (bias) v += *(const f32x4_t*)(bias + n);
  if (act == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
  }
  if (resid) v += *(const f32x4_t*)(resid + (size_t)m * ldr + n);
  *(f32x4_t*)(out + (size_t)m * ldo + n) = v;
}

constexpr int CA_MAXT = 256;

// One wavefront (= one workgroup, so all 256 CUs get work at B = 64) per (hand, head); dim_head = 64.  Scores: lane
// = token (3 tokens per lane), the token's 128-byte K row against q held in LDS.  Output: lane = channel, tokens
// walked 8 at a time into 4 independent accumulators (the 192-deep dependent FMA chain was most of the old 32 us).
template <class E>
__global__ __launch_bounds__(64) void cross_attention_kernel(const float* __restrict__ q, const E* __restrict__ kv,
                                                             int ldkv, int k_off, int v_off, float* __restrict__ out,
                                                             int B, int tokens, int heads, float scale) {
  __shared__ float ps[CA_MAXT];
  __shared__ float qs[64];
  typedef __attribute__((ext_vector_type(8))) E vec8;
  const int lane = threadIdx.x;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int inner = heads * 64;
  qs[lane] = q[(size_t)b * inner + h * 64 + lane];
  __syncthreads();
  const E* kbase = kv + (size_t)b * tokens * ldkv + k_off + h * 64;
  const E* vbase = kv + (size_t)b * tokens * ldkv + v_off + h * 64;
  float sc[CA_MAXT / 64];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < CA_MAXT / 64; ++i) {
    const int t = lane + 64 * i;
    sc[i] = -3.0e38f;
    if (t < tokens) {
      const E* kr = kbase + (size_t)t * ldkv;
      vec8 kk[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) kk[c] = *(const vec8*)(kr + c * 8);
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          d0 = fmaf(qs[c * 8 + e], (float)kk[c][e], d0);
          d1 = fmaf(qs[c * 8 + e + 1], (float)kk[c][e + 1], d1);
        }
      sc[i] = (d0 + d1) * scale;
      mx = fmaxf(mx, sc[i]);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < CA_MAXT / 64; ++i) {
    const int t = lane + 64 * i;
    if (t < tokens) {
      const float p = __expf(sc[i] - mx);
      ps[t] = p;
      sum += p;
    }
  }
  sum = wave_sum(sum);
  __syncthreads();
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  int t = 0;
  for (; t + 8 <= tokens; t += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)vbase[(size_t)(t + j) * ldkv + lane];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j & 3] = fmaf(ps[t + j], v[j], o[j & 3]);
  }
  for (; t < tokens; ++t) o[0] = fmaf(ps[t], (float)vbase[(size_t)t * ldkv + lane], o[0]);
  out[(size_t)b * inner + h * 64 + lane] = ((o[0] + o[1]) + (o[2] + o[3])) / sum;
}

__global__ void split_head_kernel(const float* __restrict__ head, int ldh, float* __restrict__ pose6d,
                                  float* __restrict__ betas, float* __restrict__ cam, int B) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * 109) return;
  const int b = i / 109, c = i % 109;
  const float v = head[(size_t)b * ldh + c];
  if (c < 96) pose6d[b * 96 + c] = v;
  else if (c < 106) betas[b * 10 + (c - 96)] = v;
  else cam[b * 3 + (c - 106)] = v;
}

}  // namespace

extern "C" int hm_linear_f32(const float* x, int ldx, const float* W, int ldw, const float* bias, const float* resid,
                             int ldr, float* out, int ldo, int M, int N, int K, int act, void* stream_) {
  if (!x || !W || !out || M <= 0 || N <= 0 || K <= 0) return hm_set_error(HM_ERR_ARG, "hm_linear_f32: bad arguments");
  if (K % 16 != 0 || N % 4 != 0 || ldx % 4 != 0 || ldw % 4 != 0 || ldo % 4 != 0 || (resid && ldr % 4 != 0))
    return hm_set_error(HM_ERR_ARG, "hm_linear_f32: K % 16, N % 4 and ld % 4 must be 0");
  if (ldx < K || ldw < K || ldo < N) return hm_set_error(HM_ERR_ARG, "hm_linear_f32: leading dimension too small");
  if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)resid) & 15)
    return hm_set_error(HM_ERR_ARG, "hm_linear_f32: pointers must be 16-byte aligned");
  dim3 grid((N + 15) / 16, (M + 15) / 16), block(256);
  HmProfScope prof(HM_K_LINEAR_F32, act, M, N, K, (hipStream_t)stream_);
  hipLaunchKernelGGL(linear_f32_kernel, grid, block, 0, (hipStream_t)stream_, x, ldx, W, ldw, bias, resid, ldr, out, ldo,
                     M, N, K, act);
  return hm_check_launch("hm_linear_f32");
}

extern "C" int hm_cross_attention(const float* q, const void* kv, int ldkv, int k_off, int v_off, float* out, int B,
                                  int tokens, int heads, int dim_head, float scale, int dtype, void* stream_) {
  if (!q || !kv || !out || B <= 0 || heads <= 0) return hm_set_error(HM_ERR_ARG, "hm_cross_attention: bad arguments");
  if (dim_head != 64 || tokens <= 0 || tokens > CA_MAXT)
    return hm_set_error(HM_ERR_ARG, "hm_cross_attention: dim_head must be 64 and tokens <= 256");
  if (ldkv % 8 != 0 || k_off % 8 != 0 || v_off % 8 != 0 || ((uintptr_t)kv & 15))
    return hm_set_error(HM_ERR_ARG, "hm_cross_attention: kv rows must be 16-byte aligned");
  dim3 grid(B * heads), block(64);
  hipStream_t s = (hipStream_t)stream_;
  HmProfScope prof(HM_K_CROSS_ATTN, 0, B, tokens, heads, s);
  if (dtype == HM_DTYPE_BF16)
    hipLaunchKernelGGL(cross_attention_kernel<__bf16>, grid, block, 0, s, q, (const __bf16*)kv, ldkv, k_off, v_off, out, B, tokens, heads, scale);
  else if (dtype == HM_DTYPE_F16)
    hipLaunchKernelGGL(cross_attention_kernel<_Float16>, grid, block, 0, s, q, (const _Float16*)kv, ldkv, k_off, v_off, out, B, tokens, heads, scale);
  else
    return hm_set_error(HM_ERR_ARG, "hm_cross_attention: bad dtype");
  return hm_check_launch("hm_cross_attention");
}

int hm_split_head(const float* head, int ldh, float* pose6d, float* betas, float* cam, int B, hipStream_t s) {
  hipLaunchKernelGGL(split_head_kernel, dim3((B * 109 + 255) / 256), dim3(256), 0, s, head, ldh, pose6d, betas, cam, B);
  return hm_check_launch("hm_split_head");
}
