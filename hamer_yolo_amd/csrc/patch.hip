// Image-side kernels of the HaMeR path (HBM-bound byte/float shuffles):
//  * hm_crop_batch   - prepare_batch_bbox (infer.py:154-259): affine bilinear crop of every
//                      hand box out of ONE resident frame (the reference copies the frame per
//                      hand, infer.py:208), BGR->RGB, optional flip, mean/std normalisation.
//  * hm_patch_im2col - the 16x16/stride-16/pad-2 patch gather of PatchEmbed (vit.py:168-176)
//                      on the 192-wide window (hamer.py:119), written as the 16-bit GEMM operand.
#include <math.h>
#include "common.h"
#include "hamer_hip_internal.h"

namespace {

// ---- crop: cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT 0) restated in its fixed-point form
// (coordinates in 1/1024 px, rounded to 1/32 px; weights (32-fx)(32-fy)/1024) -- integer
// arithmetic, so the result is bit-identical to the oracle's numpy restatement.
constexpr int AB_BITS = 10, INTER_BITS = 5, INTER_TAB = 32;

__global__ __launch_bounds__(256) void crop_kernel(const uint8_t* __restrict__ frame, int H, int W,
                                                   const hm_crop_box* __restrict__ boxes, float* __restrict__ out,
                                                   int P, float m0, float m1, float m2, float s0, float s1, float s2) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= P * P) return;
  const int y = pix / P, x = pix % P;
  const hm_crop_box bx = boxes[b];
  const int xs = bx.flip ? (P - 1 - x) : x;   // cv2.flip(patch, 1) after the crop (infer.py:230)
  const int adelta = (int)rint(bx.m0 * (double)xs * 1024.0);
  const int bdelta = (int)rint(bx.m4 * (double)y * 1024.0);
  const int X = (bx.x0 + adelta) >> (AB_BITS - INTER_BITS);
  const int Y = (bx.y0 + bdelta) >> (AB_BITS - INTER_BITS);
  const int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
  const int fx = X & (INTER_TAB - 1), fy = Y & (INTER_TAB - 1);
  const int w00 = (INTER_TAB - fx) * (INTER_TAB - fy), w01 = fx * (INTER_TAB - fy);
  const int w10 = (INTER_TAB - fx) * fy, w11 = fx * fy;
  const bool x0ok = sx >= 0 && sx < W, x1ok = sx + 1 >= 0 && sx + 1 < W;
  const bool y0ok = sy >= 0 && sy < H, y1ok = sy + 1 >= 0 && sy + 1 < H;
  int acc[3] = {512, 512, 512};
  if (y0ok && x0ok) { const uint8_t* p = frame + ((size_t)sy * W + sx) * 3; acc[0] += w00 * p[0]; acc[1] += w00 * p[1]; acc[2] += w00 * p[2]; }
  if (y0ok && x1ok) { const uint8_t* p = frame + ((size_t)sy * W + sx + 1) * 3; acc[0] += w01 * p[0]; acc[1] += w01 * p[1]; acc[2] += w01 * p[2]; }
  if (y1ok && x0ok) { const uint8_t* p = frame + ((size_t)(sy + 1) * W + sx) * 3; acc[0] += w10 * p[0]; acc[1] += w10 * p[1]; acc[2] += w10 * p[2]; }
  if (y1ok && x1ok) { const uint8_t* p = frame + ((size_t)(sy + 1) * W + sx + 1) * 3; acc[0] += w11 * p[0]; acc[1] += w11 * p[1]; acc[2] += w11 * p[2]; }
  const float bl = (float)(acc[0] >> 10), gr = (float)(acc[1] >> 10), rd = (float)(acc[2] >> 10);
  float* o = out + (size_t)b * 3 * P * P + pix;
  o[0] = (rd - m0) / s0;                       // channel 0 = R (BGR -> RGB, infer.py:228)
  o[(size_t)P * P] = (gr - m1) / s1;
  o[(size_t)2 * P * P] = (bl - m2) / s2;
}

template <class E>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, E* __restrict__ patches, int B,
                                                     int img_h, int img_w_full, int x0, int win_w, int patch, int pad,
                                                     int gh, int gw) {
  // one thread = 8 consecutive kx of one (token, c, ky): a 16-byte store
  const int kper = 3 * patch * patch;
  const int chunks = kper / 8;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)B * gh * gw * chunks;
  if (gid >= total) return;
  const int ch = (int)(gid % chunks);
  const size_t tok = gid / chunks;
  const int j = (int)(tok % gw), i = (int)((tok / gw) % gh), b = (int)(tok / ((size_t)gw * gh));
  const int k = ch * 8;
  const int c = k / (patch * patch), ky = (k / patch) % patch, kx = k % patch;
  const int yy = i * patch - pad + ky;
  const int xw = j * patch - pad + kx;             // column inside the window
  const float* src = img + (((size_t)b * 3 + c) * img_h + yy) * img_w_full + x0 + xw;
  typedef __attribute__((ext_vector_type(8))) E vec8;
  vec8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool ok = yy >= 0 && yy < img_h && (xw + e) >= 0 && (xw + e) < win_w;
    o[e] = (E)(ok ? src[e] : 0.0f);
  }
  *(vec8*)(patches + tok * kper + k) = o;
}

}  // namespace

// Host helper: the affine map of gen_trans_from_patch_cv / generate_image_patch_cv2
// (datasets/utils.py:82-129,:318-376) for rot = 0, scale = 1, with the float32 roundings of
// the three control points, inverted the way cv2.warpAffine does.
extern "C" int hm_crop_box_from_bbox(double cx, double cy, double size, int flip, int P, hm_crop_box* out) {
  if (!out || P <= 0 || !(size > 0)) return hm_set_error(HM_ERR_ARG, "hm_crop_box_from_bbox: bad arguments");
  const float half = (float)(size * 0.5);                       // rotate_2d result, float32
  const float p0x = (float)cx, p0y = (float)cy;
  const float p2x = (float)(cx + (double)half), p1y = (float)(cy + (double)half);
  const double hp = 0.5 * P;
  const double m0 = ((double)p2x - (double)p0x) / hp;           // d src_x / d dst_x
  const double m4 = ((double)p1y - (double)p0y) / hp;
  const double m2 = (double)p0x - hp * m0, m5 = (double)p0y - hp * m4;
  const int round_delta = (1 << AB_BITS) / INTER_TAB / 2;       // 16
  out->m0 = m0; out->m4 = m4;
  out->x0 = (int)rint(m2 * 1024.0) + round_delta;
  out->y0 = (int)rint(m5 * 1024.0) + round_delta;
  out->flip = flip ? 1 : 0; out->reserved = 0;
  return HM_OK;
}

extern "C" int hm_crop_batch(const uint8_t* frame, int H, int W, const hm_crop_box* boxes, float* out, int B, int P,
                             const float* mean3_host, const float* std3_host, void* stream_) {
  if (!frame || !boxes || !out || !mean3_host || !std3_host) return hm_set_error(HM_ERR_ARG, "hm_crop_batch: null pointer");
  if (H <= 0 || W <= 0 || B <= 0 || P <= 0 || B > 65535) return hm_set_error(HM_ERR_ARG, "hm_crop_batch: bad sizes");
  dim3 grid((P * P + 255) / 256, B), block(256);
  HmProfScope prof(HM_K_CROP, 0, B, P, P, (hipStream_t)stream_);
  hipLaunchKernelGGL(crop_kernel, grid, block, 0, (hipStream_t)stream_, frame, H, W, boxes, out, P, mean3_host[0],
                     mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
  return hm_check_launch("hm_crop_batch");
}

extern "C" int hm_patch_im2col(const float* img, void* patches, int B, int img_h, int img_w_full, int x0, int win_w,
                               int patch, int pad, int dtype, void* stream_) {
  if (!img || !patches || B <= 0) return hm_set_error(HM_ERR_ARG, "hm_patch_im2col: bad arguments");
  if (patch % 8 != 0 || pad < 0 || x0 < 0 || x0 + win_w > img_w_full)
    return hm_set_error(HM_ERR_ARG, "hm_patch_im2col: patch % 8 == 0 and window inside the image required");
  const int gh = (img_h + 2 * pad - patch) / patch + 1, gw = (win_w + 2 * pad - patch) / patch + 1;
  const size_t total = (size_t)B * gh * gw * (3 * patch * patch / 8);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream_;
  HmProfScope prof(HM_K_IM2COL, 0, B, gh * gw, 3 * patch * patch, s);
  if (dtype == HM_DTYPE_BF16)
    hipLaunchKernelGGL(im2col_kernel<__bf16>, grid, block, 0, s, img, (__bf16*)patches, B, img_h, img_w_full, x0, win_w, patch, pad, gh, gw);
  else if (dtype == HM_DTYPE_F16)
    hipLaunchKernelGGL(im2col_kernel<_Float16>, grid, block, 0, s, img, (_Float16*)patches, B, img_h, img_w_full, x0, win_w, patch, pad, gh, gw);
  else
    return hm_set_error(HM_ERR_ARG, "hm_patch_im2col: bad dtype");
  return hm_check_launch("hm_patch_im2col");
}
