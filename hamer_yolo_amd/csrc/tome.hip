// Token merging ("ToMe") between attention and MLP of every ViT block -- the reference's optional backbone variant
// HAMER_INFER(token_merge=True) (hamer/hamer/models/hamer.py:468-483: apply_patch + r = (8, -1)), implemented from
// hamer/hamer/models/backbones/selective_vit_adapter.py: ToMeAttention (:159-198, proportional attention and the
// head-averaged keys as matching metric), bipartite_soft_matching (:17-96), merge_wavg (:98-113), ToMeBlock (:200-235).
//
// The token count changes from block to block (192 -> 176 -> 161 ... for ViT-H with the reference's schedule) but is the
// same for every crop, so the activations stay token-major and compact, [B * T_i][C], and the GEMM / LayerNorm kernels are
// the dense path's.  What is new here is small, latency- / HBM-bound work per crop:
//   tome_attention_kernel  any T <= 192, + log(size) on the key axis; K and V of one (crop, head) in LDS, fp32 softmax
//   tome_metric_kernel     metric[b][t][d] = mean over heads of k
//   tome_match_kernel      one workgroup per crop: normalise, A (even) x B (odd) similarities, best partner per A token,
//                          stable descending rank of the proposals -> unmerged / merged A tokens and their B partners
//   tome_merge_kernel      size-weighted merge into a new compact residual stream (fp32) and the new sizes
#include <math.h>
#include <stdlib.h>
#include "common.h"
#include "hamer_hip_internal.h"

namespace {

constexpr int TM_MAXT = 192;          // tokens per crop (16 x 12 patches) before any merge
constexpr int TM_HD = 80;             // head dim of ViT-H/16 (and of the test geometries)
constexpr int TM_KSTR = 82;           // LDS row stride in elements: 41 dwords, odd -> conflict-free column walks

// One workgroup (4 waves) per (crop, head).  Wave w takes queries w, w+4, ...: lane = key (3 keys per lane at T = 192) for
// the scores, lane = channel for the output.
template <class E>
__global__ __launch_bounds__(256) void tome_attention_kernel(const E* __restrict__ qkv, const float* __restrict__ size,
                                                             E* __restrict__ out, int T, int heads, float scale) {
  __shared__ E Ks[TM_MAXT * TM_KSTR];
  __shared__ E Vs[TM_MAXT * TM_KSTR];
  __shared__ float lsz[TM_MAXT];
  __shared__ float ps[4][TM_MAXT];
  __shared__ float qs[4][TM_HD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int C = heads * TM_HD;
  const size_t ld = (size_t)3 * C;
  const E* base = qkv + (size_t)b * T * ld + (size_t)h * TM_HD;
  for (int i = tid; i < T * TM_HD; i += 256) {
    const int t = i / TM_HD, d = i - t * TM_HD;
    Ks[t * TM_KSTR + d] = base[(size_t)t * ld + C + d];
    Vs[t * TM_KSTR + d] = base[(size_t)t * ld + 2 * C + d];
  }
  for (int t = tid; t < T; t += 256) lsz[t] = size ? logf(size[(size_t)b * T + t]) : 0.0f;      // attn + size.log() (:185-187)
  __syncthreads();
  for (int q = wave; q < T; q += 4) {
    for (int d = lane; d < TM_HD; d += 64) qs[wave][d] = (float)base[(size_t)q * ld + d];
    __builtin_amdgcn_wave_barrier();
    float sc[TM_MAXT / 64], mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < TM_MAXT / 64; ++i) {
      const int t = lane + 64 * i;
      sc[i] = -3.0e38f;
      if (t < T) {
        float d0 = 0.f, d1 = 0.f;
#pragma unroll 8
        for (int d = 0; d < TM_HD; d += 2) {
          d0 = fmaf(qs[wave][d], (float)Ks[t * TM_KSTR + d], d0);
          d1 = fmaf(qs[wave][d + 1], (float)Ks[t * TM_KSTR + d + 1], d1);
        }
        sc[i] = (d0 + d1) * scale + lsz[t];
        mx = fmaxf(mx, sc[i]);
      }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TM_MAXT / 64; ++i) {
      const int t = lane + 64 * i;
      if (t < T) { const float p = expf(sc[i] - mx); ps[wave][t] = p; sum += p; }
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < TM_HD; d += 64) {
      float o0 = 0.f, o1 = 0.f;
      int t = 0;
      for (; t + 2 <= T; t += 2) {
        o0 = fmaf(ps[wave][t], (float)Vs[t * TM_KSTR + d], o0);
        o1 = fmaf(ps[wave][t + 1], (float)Vs[(t + 1) * TM_KSTR + d], o1);
      }
      if (t < T) o0 = fmaf(ps[wave][t], (float)Vs[t * TM_KSTR + d], o0);
      out[((size_t)b * T + q) * C + h * TM_HD + d] = (E)((o0 + o1) / sum);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <class E>
__global__ __launch_bounds__(256) void tome_metric_kernel(const E* __restrict__ qkv, float* __restrict__ metric, int rows, int heads) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows * TM_HD) return;
  const size_t r = i / TM_HD;
  const int d = (int)(i - r * TM_HD), C = heads * TM_HD;
  const E* k = qkv + r * 3 * C + C + d;
  float s = 0.f;
  for (int h = 0; h < heads; ++h) s += (float)k[h * TM_HD];
  metric[i] = s / (float)heads;                                     // k.mean(1) (:198)
}

// idx layout per crop: [3][TM_MAXT / 2] ints = unm_idx | src_idx | dst_idx
__global__ __launch_bounds__(128) void tome_match_kernel(const float* __restrict__ metric, int* __restrict__ idx, int T, int r,
                                                         int ld, int lo_off) {
  __shared__ float m[TM_MAXT * (TM_HD + 1)];
  __shared__ float nmax[TM_MAXT / 2];
  __shared__ int nidx[TM_MAXT / 2];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int Na = (T + 1) / 2, Nb = T / 2;
  const float* mb = metric + (size_t)b * T * ld;                    // row stride ld; lo_off > 0: value = hi + lo (columns d and lo_off + d)
  for (int t = tid; t < T; t += 128) {                              // metric / metric.norm(dim=-1) (:47)
    float ss = 0.f;
    for (int d = 0; d < TM_HD; ++d) {
      const float v = lo_off ? mb[(size_t)t * ld + d] + mb[(size_t)t * ld + lo_off + d] : mb[(size_t)t * ld + d];
      m[t * (TM_HD + 1) + d] = v;
      ss = fmaf(v, v, ss);
    }
    const float n = sqrtf(ss);
    for (int d = 0; d < TM_HD; ++d) m[t * (TM_HD + 1) + d] /= n;
  }
  __syncthreads();
  for (int a = tid; a < Na; a += 128) {                             // scores = a @ b^T; node_max, node_idx = scores.max(-1)
    const float* ar = m + (2 * a) * (TM_HD + 1);
    float best = -INFINITY; int bi = 0;
    for (int j = 0; j < Nb; ++j) {
      const float* br = m + (2 * j + 1) * (TM_HD + 1);
      float s = 0.f;
      for (int d = 0; d < TM_HD; ++d) s = fmaf(ar[d], br[d], s);
      if (s > best) { best = s; bi = j; }                           // first maximum
    }
    nmax[a] = best; nidx[a] = bi;
  }
  __syncthreads();
  int* ib = idx + (size_t)b * 3 * (TM_MAXT / 2);
  for (int a = tid; a < Na; a += 128) {                             // edge_idx = node_max.argsort(descending), stable
    const float v = nmax[a];
    int rank = 0;
    for (int o = 0; o < Na; ++o) rank += (nmax[o] > v || (nmax[o] == v && o < a)) ? 1 : 0;
    if (rank < r) { ib[TM_MAXT / 2 + rank] = a; ib[2 * (TM_MAXT / 2) + rank] = nidx[a]; }     // src_idx, dst_idx
    else ib[rank - r] = a;                                                                     // unm_idx
  }
}

// out token j of crop b: j < Na - r: the unmerged A token unm_idx[j]; else B token j - (Na - r) plus the A tokens merged into
// it, in proposal order.  x * size summed, sizes summed, divided (merge_wavg :106-112).  size == nullptr: all ones.
__global__ __launch_bounds__(256) void tome_merge_kernel(const float* __restrict__ x, const float* __restrict__ size,
                                                         const int* __restrict__ idx, float* __restrict__ xo,
                                                         float* __restrict__ so, int T, int r, int D) {
  const int To = T - r, Na = (T + 1) / 2;
  const int b = blockIdx.x / To, j = blockIdx.x % To;
  const int* ib = idx + (size_t)b * 3 * (TM_MAXT / 2);
  const float* xb = x + (size_t)b * T * D;
  const float* sb = size ? size + (size_t)b * T : nullptr;
  float* orow = xo + ((size_t)b * To + j) * D;
  if (j < Na - r) {
    const int t = 2 * ib[j];
    const float s = sb ? sb[t] : 1.0f;
    for (int d = threadIdx.x; d < D; d += 256) orow[d] = (xb[(size_t)t * D + d] * s) / s;
    if (threadIdx.x == 0) so[(size_t)b * To + j] = s;
    return;
  }
  const int jb = j - (Na - r), tb = 2 * jb + 1;
  const float s0 = sb ? sb[tb] : 1.0f;
  float stot = s0;
  for (int k = 0; k < r; ++k)
    if (ib[2 * (TM_MAXT / 2) + k] == jb) stot += sb ? sb[2 * ib[TM_MAXT / 2 + k]] : 1.0f;
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = xb[(size_t)tb * D + d] * s0;
    for (int k = 0; k < r; ++k)
      if (ib[2 * (TM_MAXT / 2) + k] == jb) {
        const int ta = 2 * ib[TM_MAXT / 2 + k];
        acc += xb[(size_t)ta * D + d] * (sb ? sb[ta] : 1.0f);
      }
    orow[d] = acc / stot;
  }
  if (threadIdx.x == 0) so[(size_t)b * To + j] = stot;
}

}  // namespace

extern "C" size_t hm_tome_index_bytes(int B) { return B > 0 ? (size_t)B * 3 * (TM_MAXT / 2) * sizeof(int) : 0; }

extern "C" int hm_tome_attention(const void* qkv, const float* size, void* out, int B, int tokens, int heads, int head_dim,
                                 float scale, int dtype, void* stream_) {
  if (!qkv || !out || B <= 0 || heads <= 0) return hm_set_error(HM_ERR_ARG, "hm_tome_attention: bad arguments");
  if (head_dim != TM_HD || tokens <= 0 || tokens > TM_MAXT) return hm_set_error(HM_ERR_ARG, "hm_tome_attention: head_dim 80, 0 < tokens <= 192");
  hipStream_t s = (hipStream_t)stream_;
  HmProfScope prof(HM_K_ATTENTION, 1, B, tokens, heads, s);
  // the MFMA kernel of the dense path (attention.hip) with a runtime token count; HM_TOME_SCALAR_ATTENTION=1 keeps the
  // fp32 lane-per-key kernel below (the first implementation: 7x slower, kept as a second opinion for the tests)
  if ((((uintptr_t)qkv | (uintptr_t)out) & 15) == 0 && hm_option(HM_OPT_TOME_SCALAR_ATTENTION) == 0)
    return hm_attention_tome_launch(qkv, size, out, B, tokens, heads, scale, dtype, s);
  if (dtype == HM_DTYPE_BF16)
    hipLaunchKernelGGL(tome_attention_kernel<__bf16>, dim3(B * heads), dim3(256), 0, s, (const __bf16*)qkv, size, (__bf16*)out, tokens, heads, scale);
  else if (dtype == HM_DTYPE_F16)
    hipLaunchKernelGGL(tome_attention_kernel<_Float16>, dim3(B * heads), dim3(256), 0, s, (const _Float16*)qkv, size, (_Float16*)out, tokens, heads, scale);
  else return hm_set_error(HM_ERR_ARG, "hm_tome_attention: bad dtype");
  return hm_check_launch("hm_tome_attention");
}

extern "C" int hm_tome_merge(const void* qkv, const float* x, const float* size, float* x_out, float* size_out, float* metric_ws,
                             int* index_ws, int B, int tokens, int r, int heads, int head_dim, int D, int dtype, void* stream_) {
  if (!qkv || !x || !x_out || !size_out || !metric_ws || !index_ws || B <= 0) return hm_set_error(HM_ERR_ARG, "hm_tome_merge: null pointer");
  if (head_dim != TM_HD || tokens <= 1 || tokens > TM_MAXT || heads <= 0 || D <= 0 || r <= 0 || r > tokens / 2)
    return hm_set_error(HM_ERR_ARG, "hm_tome_merge: head_dim 80, 1 < tokens <= 192, 0 < r <= tokens / 2");
  hipStream_t s = (hipStream_t)stream_;
  HmProfScope prof(HM_K_OTHER, 6, B, tokens, r, s);
  const size_t n = (size_t)B * tokens * TM_HD;
  if (dtype == HM_DTYPE_BF16)
    hipLaunchKernelGGL(tome_metric_kernel<__bf16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const __bf16*)qkv, metric_ws, B * tokens, heads);
  else if (dtype == HM_DTYPE_F16)
    hipLaunchKernelGGL(tome_metric_kernel<_Float16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const _Float16*)qkv, metric_ws, B * tokens, heads);
  else return hm_set_error(HM_ERR_ARG, "hm_tome_merge: bad dtype");
  hipLaunchKernelGGL(tome_match_kernel, dim3(B), dim3(128), 0, s, metric_ws, index_ws, tokens, r, TM_HD, 0);
  hipLaunchKernelGGL(tome_merge_kernel, dim3(B * (tokens - r)), dim3(256), 0, s, x, size, index_ws, x_out, size_out, tokens, r, D);
  return hm_check_launch("hm_tome_merge");
}

extern "C" int hm_tome_merge_metric(const float* metric, int ld_metric, int lo_off, const float* x, const float* size, float* x_out,
                                    float* size_out, int* index_ws, int B, int tokens, int r, int D, void* stream_) {
  if (!metric || !x || !x_out || !size_out || !index_ws || B <= 0) return hm_set_error(HM_ERR_ARG, "hm_tome_merge_metric: null pointer");
  if (tokens <= 1 || tokens > TM_MAXT || D <= 0 || r <= 0 || r > tokens / 2 || ld_metric < TM_HD || lo_off < 0 || (lo_off && lo_off + TM_HD > ld_metric))
    return hm_set_error(HM_ERR_ARG, "hm_tome_merge_metric: 1 < tokens <= 192, 0 < r <= tokens / 2, metric rows of >= 80 (+ lo part) floats");
  hipStream_t s = (hipStream_t)stream_;
  HmProfScope prof(HM_K_OTHER, 6, B, tokens, r, s);
  hipLaunchKernelGGL(tome_match_kernel, dim3(B), dim3(128), 0, s, metric, index_ws, tokens, r, ld_metric, lo_off);
  hipLaunchKernelGGL(tome_merge_kernel, dim3(B * (tokens - r)), dim3(256), 0, s, x, size, index_ws, x_out, size_out, tokens, r, D);
  return hm_check_launch("hm_tome_merge_metric");
}
