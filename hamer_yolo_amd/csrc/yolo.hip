// Detector-side kernels around the implicit-GEMM convolutions (gemm.hip):
//   letterbox (utils/datasets.py:999-1029 + LoadImage.process_img :137-141 + /255 detector.py:121-125)
//   MaxPool / nearest Upsample on NHWC (common.py:34-40,:275; yolov7.yaml:78,:92)
//   Detect decode (yolo.py:148-184), non_max_suppression + scale_coords (general.py:611-703,:323-344)
// All HBM/latency-bound byte and index work: coalesced 16-byte NHWC accesses, integer arithmetic
// for the u8 resize, one workgroup with an LDS bitonic sort for NMS.
#include <math.h>
#include <string.h>
#include <type_traits>
#include "common.h"
#include "hamer_hip_internal.h"

namespace {

// ------------------------------------------------------------------------------- pooling / upsample
template <class E>
__global__ __launch_bounds__(256) void maxpool_kernel(const E* __restrict__ x, int ldx, E* __restrict__ y, int ldy, int N,
                                                      int H, int W, int C, int k, int stride, int pad, int Ho, int Wo) {
  typedef __attribute__((ext_vector_type(8))) E vec8;
  const int c8 = C / 8;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)N * Ho * Wo * c8) return;
  const int c = (int)(gid % c8) * 8;
  const size_t p = gid / c8;
  const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), n = (int)(p / ((size_t)Wo * Ho));
  float m[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
  for (int ky = 0; ky < k; ++ky) {
    const int iy = oy * stride - pad + ky;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int ix = ox * stride - pad + kx;
      if (ix < 0 || ix >= W) continue;
      const vec8 v = *(const vec8*)(x + (((size_t)n * H + iy) * W + ix) * ldx + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)v[e]);
    }
  }
  vec8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (E)m[e];
  *(vec8*)(y + p * ldy + c) = o;
}

template <class E>
__global__ __launch_bounds__(256) void upsample2x_kernel(const E* __restrict__ x, int ldx, E* __restrict__ y, int ldy,
                                                         int N, int H, int W, int C) {
  typedef __attribute__((ext_vector_type(8))) E vec8;
  const int c8 = C / 8;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)N * 2 * H * 2 * W * c8) return;
  const int c = (int)(gid % c8) * 8;
  const size_t p = gid / c8;
  const int ox = (int)(p % (2 * W)), oy = (int)((p / (2 * W)) % (2 * H)), n = (int)(p / ((size_t)4 * W * H));
  *(vec8*)(y + p * ldy + c) = *(const vec8*)(x + (((size_t)n * H + (oy >> 1)) * W + (ox >> 1)) * ldx + c);
}

// ------------------------------------------------------------------------------- letterbox
template <class E>
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ frame, hm_letterbox_plan pl,
                                                        const int32_t* __restrict__ tab, E* __restrict__ x8,
                                                        uint8_t* __restrict__ u8, size_t frame_stride) {
  typedef __attribute__((ext_vector_type(8))) E vec8;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= pl.out_h * pl.out_w) return;
  frame += (size_t)blockIdx.y * frame_stride;           // frame blockIdx.y of a batched pass; its output follows the previous one's
  x8 += (size_t)blockIdx.y * pl.out_h * pl.out_w * 8;
  const int oy = pix / pl.out_w, ox = pix - oy * pl.out_w;
  const int dy = oy - pl.top, dx = ox - pl.left;
  int v[3] = {114, 114, 114};                               // BGR, letterbox colour (datasets.py:999)
  if (dy >= 0 && dy < pl.new_h && dx >= 0 && dx < pl.new_w) {
    const int32_t *tx = tab, *ty = tab + 3 * pl.new_w;
    const int x0 = tx[dx], ax0 = tx[pl.new_w + dx], ax1 = tx[2 * pl.new_w + dx];
    const int y0 = ty[dy], ay0 = ty[pl.new_h + dy], ay1 = ty[2 * pl.new_h + dy];
    const int x1 = min(x0 + 1, pl.src_w - 1), y1 = min(y0 + 1, pl.src_h - 1);
    const uint8_t* r0 = frame + (size_t)y0 * pl.src_w * 3;
    const uint8_t* r1 = frame + (size_t)y1 * pl.src_w * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0 = r0[x0 * 3 + c] * ax0 + r0[x1 * 3 + c] * ax1;
      const int h1 = r1[x0 * 3 + c] * ax0 + r1[x1 * 3 + c] * ax1;
      const int o = (((ay0 * (h0 >> 4)) >> 16) + ((ay1 * (h1 >> 4)) >> 16) + 2) >> 2;
      v[c] = min(max(o, 0), 255);
    }
  }
  vec8 o8;
  o8[0] = (E)((float)v[2] / 255.0f); o8[1] = (E)((float)v[1] / 255.0f); o8[2] = (E)((float)v[0] / 255.0f);
#pragma unroll
  for (int e = 3; e < 8; ++e) o8[e] = (E)0.0f;
  *(vec8*)(x8 + (size_t)pix * 8) = o8;
  if (u8) {
    const size_t plane = (size_t)pl.out_h * pl.out_w;
    u8[pix] = (uint8_t)v[2]; u8[plane + pix] = (uint8_t)v[1]; u8[2 * plane + pix] = (uint8_t)v[0];
  }
}

// ------------------------------------------------------------------------------- decode
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ raw, int ldraw, float* __restrict__ pred,
                                                     int row0, int ny, int nx, int no, float stride, float a0w, float a0h,
                                                     float a1w, float a1h, float a2w, float a2h, size_t raw_img, size_t pred_img) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= 3 * ny * nx) return;
  raw += (size_t)blockIdx.y * raw_img;                  // image blockIdx.y of a batched pass (element strides between images)
  pred += (size_t)blockIdx.y * pred_img;
  const int a = gid / (ny * nx), p = gid - a * ny * nx;
  const int y = p / nx, x = p - y * nx;
  const float aw = a == 0 ? a0w : (a == 1 ? a1w : a2w), ah = a == 0 ? a0h : (a == 1 ? a1h : a2h);
  const float* r = raw + (size_t)p * ldraw + a * no;
  float* o = pred + ((size_t)row0 + gid) * no;
  for (int c = 0; c < no; ++c) {
    const float s = 1.0f / (1.0f + expf(-r[c]));
    float v = s;
    if (c == 0) v = (s * 2.0f - 0.5f + (float)x) * stride;
    else if (c == 1) v = (s * 2.0f - 0.5f + (float)y) * stride;
    else if (c == 2) { const float t = s * 2.0f; v = __fmul_rn(__fmul_rn(t, t), aw); }
    else if (c == 3) { const float t = s * 2.0f; v = __fmul_rn(__fmul_rn(t, t), ah); }
    o[c] = v;
  }
}

// ------------------------------------------------------------------------------- NMS
// general.py:611-703 takes any (1, n, 5+nc): 15120 rows for a 384x640 letterbox, 18900 for 480x640, 25200 for 640x640.
// The confidence filter writes the surviving box of row i to cand[i] (no compaction of the boxes) and appends one sort key
// per survivor -- score in the high word, ~row in the low word -- to a compact key list.  One workgroup then sorts the
// keys (in LDS when <= 16384 survive, in the workspace otherwise: only a degenerate prediction does that), keeps the best
// 30000 as the reference does (max_nms), and runs the greedy suppression.  Equal scores: lower row first, as a stable
// descending argsort gives.
struct Cand { float x1, y1, x2, y2, conf, cls; int pad0, pad1; };
constexpr int NMS_LDS_KEYS = 16384;  // survivors sorted in LDS
constexpr int NMS_MAX_NMS = 30000;   // general.py:625
constexpr int NMS_MAX_ROWS = 1 << 20;
constexpr int NMS_BOXCACHE = 1024;
constexpr int NMS_SUPP_WORDS = 1024; // 32768 bits >= NMS_MAX_NMS
constexpr int NMS_LDS = NMS_LDS_KEYS * 8 + NMS_BOXCACHE * 16 + NMS_SUPP_WORDS * 4 + 16 + 4096;

__host__ __device__ inline int nms_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

__device__ __forceinline__ unsigned f2sortable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void nms_filter_kernel(const float* __restrict__ pred, int n, int nc, float conf_thres,
                                                         unsigned class_mask, Cand* __restrict__ cand,
                                                         unsigned long long* __restrict__ keys, int* __restrict__ counter) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = pred + (size_t)i * (5 + nc);
  const float obj = p[4];
  if (!(obj > conf_thres)) return;                                  // xc = prediction[..., 4] > conf_thres
  float best = -1.0f; int bj = 0;
  for (int c = 0; c < nc; ++c) {
    const float s = nc == 1 ? obj : __fmul_rn(p[5 + c], obj);        // x[:, 5:] *= x[:, 4:5]
    if (s > best) { best = s; bj = c; }                              // first maximum, as torch.max
  }
  if (!(best > conf_thres)) return;
  if (!((class_mask >> bj) & 1u)) return;
  const float hw = p[2] / 2, hh = p[3] / 2;                          // xywh2xyxy (general.py:268-275)
  cand[i] = Cand{p[0] - hw, p[1] - hh, p[0] + hw, p[1] + hh, best, (float)bj, 0, 0};
  keys[atomicAdd(counter, 1)] = ((unsigned long long)f2sortable(best) << 32) | (0xFFFFFFFFu - (unsigned)i);
}

__global__ __launch_bounds__(1024) void nms_kernel(const Cand* __restrict__ cand, unsigned long long* gkeys, const int* __restrict__ counter,
                                                   float iou_thres, int agnostic, int max_det, hm_letterbox_plan pl,
                                                   int do_scale, float* __restrict__ dets, int* __restrict__ count) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* bcache = (float4*)(smem + NMS_LDS_KEYS * 8);                              // NMS_BOXCACHE
  unsigned* supp = (unsigned*)(smem + NMS_LDS_KEYS * 8 + NMS_BOXCACHE * 16);        // NMS_SUPP_WORDS
  int* kept = (int*)(smem + NMS_LDS_KEYS * 8 + NMS_BOXCACHE * 16 + NMS_SUPP_WORDS * 4 + 16);   // <= 1024 entries
  const int tid = threadIdx.x;
  int n = *counter;
  const int P = nms_pow2(n);
  const bool in_lds = n <= NMS_LDS_KEYS;
  unsigned long long* keys = in_lds ? (unsigned long long*)smem : gkeys;            // the workspace holds nms_pow2(rows) keys
  if (in_lds) { for (int j = tid; j < P; j += 1024) keys[j] = j < n ? gkeys[j] : 0ull; }
  else { for (int j = n + tid; j < P; j += 1024) keys[j] = 0ull; }
  for (int j = tid; j < NMS_SUPP_WORDS; j += 1024) supp[j] = 0u;
  __syncthreads();
  // bitonic sort, descending (a real key has its top bit set: scores are > conf_thres >= 0, so the zero padding sorts last)
  for (int k = 2; k <= P; k <<= 1)
    for (int s = k >> 1; s > 0; s >>= 1) {
      for (int j = tid; j < P; j += 1024) {
        const int l = j ^ s;
        if (l > j) {
          const unsigned long long a = keys[j], b = keys[l];
          const bool desc = (j & k) == 0;
          if (desc ? (a < b) : (a > b)) { keys[j] = b; keys[l] = a; }
        }
      }
      __syncthreads();
    }
  n = n < NMS_MAX_NMS ? n : NMS_MAX_NMS;                                        // x[x[:, 4].argsort(descending=True)[:max_nms]]
  const float off = agnostic ? 0.0f : 4096.0f;                                 // c = cls * max_wh (general.py:684)
  auto row_of = [&](int j) { return (int)(0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull)); };
  auto load_box = [&](int j) {
    const Cand c = cand[row_of(j)];
    const float o = c.cls * off;
    return make_float4(c.x1 + o, c.y1 + o, c.x2 + o, c.y2 + o);
  };
  for (int j = tid; j < n && j < NMS_BOXCACHE; j += 1024) bcache[j] = load_box(j);
  __syncthreads();
  int nk = 0;
  for (int i = 0; i < n && nk < max_det; ++i) {
    if ((supp[i >> 5] >> (i & 31)) & 1u) continue;                            // uniform: read after a barrier
    if (tid == 0) kept[nk] = i;
    ++nk;
    const float4 bi = i < NMS_BOXCACHE ? bcache[i] : load_box(i);
    const float iarea = __fmul_rn(bi.z - bi.x, bi.w - bi.y);
    for (int j = i + 1 + tid; j < n; j += 1024) {
      if ((supp[j >> 5] >> (j & 31)) & 1u) continue;
      const float4 bj = j < NMS_BOXCACHE ? bcache[j] : load_box(j);
      const float w = fmaxf(0.0f, fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x));
      const float h = fmaxf(0.0f, fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y));
      const float inter = __fmul_rn(w, h);
      const float jarea = __fmul_rn(bj.z - bj.x, bj.w - bj.y);
      const float ovr = inter / (__fadd_rn(iarea, jarea) - inter);
      if (ovr > iou_thres) atomicOr(&supp[j >> 5], 1u << (j & 31));
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0) *count = nk;
  for (int r = tid; r < nk; r += 1024) {
    const Cand c = cand[row_of(kept[r])];
    float b[4] = {c.x1, c.y1, c.x2, c.y2};
    if (do_scale) {                                                            // scale_coords + clip + round
      const float lim[4] = {(float)pl.src_w, (float)pl.src_h, (float)pl.src_w, (float)pl.src_h};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = (b[e] - ((e & 1) ? pl.pad_y : pl.pad_x)) / pl.gain;
        t = fminf(fmaxf(t, 0.0f), lim[e]);
        b[e] = rintf(t);
      }
    }
    float* o = dets + (size_t)r * 6;
    o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3]; o[4] = c.conf; o[5] = c.cls;
  }
}

template <class F>
int with_dtype(int dtype, F&& f) {
  if (dtype == HM_DTYPE_BF16) return f((__bf16*)nullptr);
  if (dtype == HM_DTYPE_F16) return f((_Float16*)nullptr);
  return hm_set_error(HM_ERR_ARG, "bad dtype");
}

}  // namespace

extern "C" int hm_maxpool_nhwc(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, int k, int stride,
                               int pad, int dtype, void* stream_) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || pad < 0)
    return hm_set_error(HM_ERR_ARG, "hm_maxpool_nhwc: bad arguments");
  if (C % 8 || ldx % 8 || ldy % 8 || (((uintptr_t)x | (uintptr_t)y) & 15)) return hm_set_error(HM_ERR_ARG, "hm_maxpool_nhwc: C, ld % 8 and 16-byte alignment");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const size_t total = (size_t)N * Ho * Wo * (C / 8);
  hipStream_t s = (hipStream_t)stream_;
  HmProfScope prof(HM_K_OTHER, 1, N * Ho * Wo, C, k, s);
  const int rc = with_dtype(dtype, [&](auto* tag) {
    using E = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL(maxpool_kernel<E>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const E*)x, ldx, (E*)y, ldy, N, H, W, C, k, stride, pad, Ho, Wo);
    return HM_OK;
  });
  return rc != HM_OK ? rc : hm_check_launch("hm_maxpool_nhwc");
}

template <class E>
__global__ __launch_bounds__(256) void nchw3_to_nhwc8_kernel(const float* __restrict__ x, E* __restrict__ y, int B, int HW) {
  typedef __attribute__((ext_vector_type(8))) E vec8;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)B * HW) return;
  const size_t b = gid / HW, p = gid % HW;
  const float* xb = x + b * 3 * (size_t)HW + p;
  vec8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (E)0.0f;
  o[0] = (E)xb[0]; o[1] = (E)xb[HW]; o[2] = (E)xb[2 * (size_t)HW];
  *(vec8*)(y + gid * 8) = o;
}

// one workgroup per image: channel sums over HW (coalesced over channels), dot with w, + bias, * k
template <class E>
__global__ __launch_bounds__(256) void gap_linear_kernel(const E* __restrict__ feat, int HW, int C, const float* __restrict__ w,
                                                         float bias, const float* __restrict__ kv, float* __restrict__ depth) {
  __shared__ float part[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const E* f = feat + (size_t)b * HW * C;
  float acc = 0.f;
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += (float)f[(size_t)p * C + c];
    acc = fmaf(s / (float)HW, w[c], acc);
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) part[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) depth[b] = ((part[0] + part[1]) + (part[2] + part[3]) + bias) * kv[b];
}

extern "C" int hm_nchw3_to_nhwc8(const float* x, void* y, int B, int H, int W, int dtype, void* stream_) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || ((uintptr_t)y & 15)) return hm_set_error(HM_ERR_ARG, "hm_nchw3_to_nhwc8: bad arguments");
  hipStream_t s = (hipStream_t)stream_;
  const size_t total = (size_t)B * H * W;
  const int rc = with_dtype(dtype, [&](auto* tag) {
    using E = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL(nchw3_to_nhwc8_kernel<E>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, (E*)y, B, H * W);
    return HM_OK;
  });
  return rc != HM_OK ? rc : hm_check_launch("hm_nchw3_to_nhwc8");
}

extern "C" int hm_gap_linear(const void* feat, int HW, int C, const float* w, float bias, const float* k_value, float* depth, int B,
                             int dtype, void* stream_) {
  if (!feat || !w || !k_value || !depth || HW <= 0 || C <= 0 || B <= 0) return hm_set_error(HM_ERR_ARG, "hm_gap_linear: bad arguments");
  hipStream_t s = (hipStream_t)stream_;
  const int rc = with_dtype(dtype, [&](auto* tag) {
    using E = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL(gap_linear_kernel<E>, dim3(B), dim3(256), 0, s, (const E*)feat, HW, C, w, bias, k_value, depth);
    return HM_OK;
  });
  return rc != HM_OK ? rc : hm_check_launch("hm_gap_linear");
}

extern "C" int hm_upsample2x_nhwc(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, int dtype, void* stream_) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return hm_set_error(HM_ERR_ARG, "hm_upsample2x_nhwc: bad arguments");
  if (C % 8 || ldx % 8 || ldy % 8 || (((uintptr_t)x | (uintptr_t)y) & 15)) return hm_set_error(HM_ERR_ARG, "hm_upsample2x_nhwc: C, ld % 8 and 16-byte alignment");
  const size_t total = (size_t)N * 4 * H * W * (C / 8);
  hipStream_t s = (hipStream_t)stream_;
  HmProfScope prof(HM_K_OTHER, 2, N * 4 * H * W, C, 0, s);
  const int rc = with_dtype(dtype, [&](auto* tag) {
    using E = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL(upsample2x_kernel<E>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const E*)x, ldx, (E*)y, ldy, N, H, W, C);
    return HM_OK;
  });
  return rc != HM_OK ? rc : hm_check_launch("hm_upsample2x_nhwc");
}

extern "C" int hm_letterbox_plan_make(int H, int W, int new_shape, int stride, hm_letterbox_plan* p) {
  if (!p || H <= 0 || W <= 0 || new_shape <= 0 || stride <= 0) return hm_set_error(HM_ERR_ARG, "hm_letterbox_plan_make: bad arguments");
  const double r = fmin((double)new_shape / H, (double)new_shape / W);         // datasets.py:1006
  p->src_h = H; p->src_w = W;
  p->new_w = (int)nearbyint(W * r); p->new_h = (int)nearbyint(H * r);          // int(round(.)), half to even
  double dw = (double)((new_shape - p->new_w) % stride) / 2, dh = (double)((new_shape - p->new_h) % stride) / 2;
  const int top = (int)nearbyint(dh - 0.1), bottom = (int)nearbyint(dh + 0.1);
  const int left = (int)nearbyint(dw - 0.1), right = (int)nearbyint(dw + 0.1);
  p->top = top; p->left = left;
  p->out_h = p->new_h + top + bottom; p->out_w = p->new_w + left + right;
  const double gain = fmin((double)p->out_h / H, (double)p->out_w / W);         // scale_coords, general.py:326-327
  p->gain = (float)gain;
  p->pad_x = (float)((p->out_w - W * gain) / 2); p->pad_y = (float)((p->out_h - H * gain) / 2);
  return HM_OK;
}

extern "C" int hm_letterbox_tables(const hm_letterbox_plan* p, int32_t* tab) {
  if (!p || !tab) return hm_set_error(HM_ERR_ARG, "hm_letterbox_tables: null pointer");
  auto fill = [](int dn, int sn, int32_t* t) {
    const double scale = 1.0 / ((double)dn / sn);
    for (int d = 0; d < dn; ++d) {
      float f = (float)((d + 0.5) * scale - 0.5);
      int s = (int)floorf(f);
      f -= (float)s;
      if (s < 0) { s = 0; f = 0.f; }
      if (s >= sn - 1) { s = sn - 1; f = 0.f; }
      t[d] = s;
      t[dn + d] = (int32_t)lrintf((1.f - f) * 2048.f);
      t[2 * dn + d] = (int32_t)lrintf(f * 2048.f);
    }
  };
  fill(p->new_w, p->src_w, tab);
  fill(p->new_h, p->src_h, tab + 3 * p->new_w);
  return HM_OK;
}

extern "C" int hm_letterbox(const uint8_t* frame, const hm_letterbox_plan* plan, const int32_t* tab_dev, void* x8, int dtype,
                            uint8_t* u8_chw, void* stream_) {
  if (!frame || !plan || !tab_dev || !x8) return hm_set_error(HM_ERR_ARG, "hm_letterbox: null pointer");
  if ((uintptr_t)x8 & 15) return hm_set_error(HM_ERR_ARG, "hm_letterbox: x8 must be 16-byte aligned");
  const int total = plan->out_h * plan->out_w;
  hipStream_t s = (hipStream_t)stream_;
  HmProfScope prof(HM_K_OTHER, 3, plan->out_h, plan->out_w, 0, s);
  const int rc = with_dtype(dtype, [&](auto* tag) {
    using E = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL(letterbox_kernel<E>, dim3((total + 255) / 256), dim3(256), 0, s, frame, *plan, tab_dev, (E*)x8, u8_chw, (size_t)0);
    return HM_OK;
  });
  return rc != HM_OK ? rc : hm_check_launch("hm_letterbox");
}

extern "C" int hm_letterbox_batch(const uint8_t* frames, size_t frame_stride_bytes, int nb, const hm_letterbox_plan* plan,
                                  const int32_t* tab_dev, void* x8, int dtype, void* stream_) {
  if (!frames || !plan || !tab_dev || !x8 || nb <= 0 || nb > 65535) return hm_set_error(HM_ERR_ARG, "hm_letterbox_batch: bad arguments");
  if ((uintptr_t)x8 & 15) return hm_set_error(HM_ERR_ARG, "hm_letterbox_batch: x8 must be 16-byte aligned");
  if (frame_stride_bytes < (size_t)plan->src_h * plan->src_w * 3) return hm_set_error(HM_ERR_ARG, "hm_letterbox_batch: frame stride smaller than a frame");
  const int total = plan->out_h * plan->out_w;
  hipStream_t s = (hipStream_t)stream_;
  HmProfScope prof(HM_K_OTHER, 3, plan->out_h, plan->out_w, nb, s);
  const int rc = with_dtype(dtype, [&](auto* tag) {
    using E = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL(letterbox_kernel<E>, dim3((total + 255) / 256, nb), dim3(256), 0, s, frames, *plan, tab_dev, (E*)x8, (uint8_t*)nullptr,
                       frame_stride_bytes);
    return HM_OK;
  });
  return rc != HM_OK ? rc : hm_check_launch("hm_letterbox_batch");
}

extern "C" int hm_yolo_decode(const float* raw, int ldraw, float* pred, int row0, int ny, int nx, int nc, float stride,
                              const float* a, void* stream_) {
  if (!raw || !pred || !a || ny <= 0 || nx <= 0 || nc <= 0 || row0 < 0) return hm_set_error(HM_ERR_ARG, "hm_yolo_decode: bad arguments");
  if (ldraw < 3 * (5 + nc)) return hm_set_error(HM_ERR_ARG, "hm_yolo_decode: ldraw too small");
  hipStream_t s = (hipStream_t)stream_;
  HmProfScope prof(HM_K_OTHER, 4, ny, nx, nc, s);
  hipLaunchKernelGGL(decode_kernel, dim3((3 * ny * nx + 255) / 256), dim3(256), 0, s, raw, ldraw, pred, row0, ny, nx, 5 + nc,
                     stride, a[0], a[1], a[2], a[3], a[4], a[5], (size_t)0, (size_t)0);
  return hm_check_launch("hm_yolo_decode");
}

extern "C" int hm_yolo_decode_batch(const float* raw, int ldraw, float* pred, int row0, int ny, int nx, int nc, float stride,
                                    const float* a, int nb, size_t pred_rows_per_image, void* stream_) {
  if (!raw || !pred || !a || ny <= 0 || nx <= 0 || nc <= 0 || row0 < 0 || nb <= 0 || nb > 65535)
    return hm_set_error(HM_ERR_ARG, "hm_yolo_decode_batch: bad arguments");
  if (ldraw < 3 * (5 + nc)) return hm_set_error(HM_ERR_ARG, "hm_yolo_decode_batch: ldraw too small");
  hipStream_t s = (hipStream_t)stream_;
  HmProfScope prof(HM_K_OTHER, 4, ny, nx, nc, s);
  hipLaunchKernelGGL(decode_kernel, dim3((3 * ny * nx + 255) / 256, nb), dim3(256), 0, s, raw, ldraw, pred, row0, ny, nx, 5 + nc,
                     stride, a[0], a[1], a[2], a[3], a[4], a[5], (size_t)ny * nx * ldraw, pred_rows_per_image * (5 + nc));
  return hm_check_launch("hm_yolo_decode_batch");
}

extern "C" size_t hm_nms_workspace_bytes(int n) {
  if (n <= 0 || n > NMS_MAX_ROWS) return 0;
  return 256 + (size_t)nms_pow2(n) * 8 + (size_t)n * sizeof(Cand);      // counter | sort keys | one box per prediction row
}

extern "C" int hm_yolo_nms(const float* pred, int n, int nc, float conf_thres, float iou_thres, unsigned class_mask,
                           int agnostic, int max_det, const hm_letterbox_plan* plan, float* dets, int* count, void* workspace,
                           size_t workspace_bytes, void* stream_) {
  if (!pred || !dets || !count || !workspace) return hm_set_error(HM_ERR_ARG, "hm_yolo_nms: null pointer");
  if (n <= 0 || n > NMS_MAX_ROWS || nc <= 0 || nc > 32 || max_det <= 0 || max_det > 1024)
    return hm_set_error(HM_ERR_ARG, "hm_yolo_nms: need 0 < n <= 1048576, 0 < nc <= 32, 0 < max_det <= 1024");
  if (!(conf_thres >= 0.0f)) return hm_set_error(HM_ERR_ARG, "hm_yolo_nms: conf_thres must be >= 0");
  if (workspace_bytes < hm_nms_workspace_bytes(n) || ((uintptr_t)workspace & 15))
    return hm_set_error(HM_ERR_ARG, "hm_yolo_nms: workspace too small or misaligned");
  hipStream_t s = (hipStream_t)stream_;
  static HmLdsOnce lds_once;
  if (const int rc = lds_once.ensure((const void*)nms_kernel, NMS_LDS, "hm_yolo_nms: cannot raise dynamic LDS limit")) return rc;
  int* counter = (int*)workspace;
  unsigned long long* keys = (unsigned long long*)((char*)workspace + 256);
  Cand* cand = (Cand*)((char*)workspace + 256 + (size_t)nms_pow2(n) * 8);
  if (hipMemsetAsync(counter, 0, 256, s) != hipSuccess) return hm_set_error(HM_ERR_HIP, "hm_yolo_nms: memset failed");
  HmProfScope prof(HM_K_OTHER, 5, n, nc, max_det, s);
  hipLaunchKernelGGL(nms_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pred, n, nc, conf_thres, class_mask, cand, keys, counter);
  hm_letterbox_plan pl;
  memset(&pl, 0, sizeof(pl));
  if (plan) pl = *plan;
  hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(1024), NMS_LDS, s, cand, keys, counter, iou_thres, agnostic, max_det, pl, plan ? 1 : 0, dets, count);
  return hm_check_launch("hm_yolo_nms");
}

extern "C" int hm_yolo_run(const hm_yolo_op* ops, int n_ops, void* stream) {
  if (!ops || n_ops <= 0) return hm_set_error(HM_ERR_ARG, "hm_yolo_run: empty op list");
  for (int i = 0; i < n_ops; ++i) {
    const hm_yolo_op& o = ops[i];
    const hm_conv_args& c = o.conv;
    int rc;
    if (o.kind == HM_OP_CONV) rc = hm_conv2d_nhwc(&c, stream);
    else if (o.kind == HM_OP_CONV_PAIR) {
      if (i + 1 >= n_ops || ops[i + 1].kind != HM_OP_CONV) return hm_set_error(HM_ERR_ARG, "hm_yolo_run: HM_OP_CONV_PAIR needs a convolution behind it");
      rc = hm_conv2d_stem_pair(&c, &ops[i + 1].conv, stream);
      ++i;
    }
    else if (o.kind == HM_OP_MAXPOOL) rc = hm_maxpool_nhwc(c.X, c.ldx, c.Y, c.ldy, c.N, c.H, c.W_in, c.Cin, c.ksize, c.stride, o.pool_pad, c.dtype, stream);
    else if (o.kind == HM_OP_UPSAMPLE2X) rc = hm_upsample2x_nhwc(c.X, c.ldx, c.Y, c.ldy, c.N, c.H, c.W_in, c.Cin, c.dtype, stream);
    else return hm_set_error(HM_ERR_ARG, "hm_yolo_run: unknown op kind");
    if (rc != HM_OK) return rc;
  }
  return HM_OK;
}
