// Attention.forward core of the ViT backbone (vit.py:115-123): softmax(scale * q k^T) v for
// 192 tokens, 16 heads of head_dim 80.  One workgroup (4 waves) per (crop, head); K and V of the
// head are staged once in LDS (63 KB, plain 16-byte copies), each wave owns 48 queries (3 MFMA
// tiles).  V is consumed TRANSPOSED (the A operand of O^T = V^T . P^T) straight from its row-major
// image with ds_read_b64_tr_b16, the gfx950 transposing LDS read.
//
// Layout trick (no cross-lane traffic for P): scores are computed TRANSPOSED,
// S^T = K . Q^T with mfma_16x16x32 (A = K rows, B = Q^T), so lane l holds, for query l&15,
// keys 4*(l>>4)+r of every 16-key tile.  Two such tiles are exactly the 8 k-slots of lane l
// for the next MFMA's B operand (P^T) if the 32 keys of a k-step are permuted as
//   slot (g, j) -> key 4g + j (j < 4),  16 + 4g + (j - 4) (j >= 4),
// and the A operand (V^T) is read with the same permutation: two transposing 8-byte reads, each
// returning 4 consecutive keys of one head-dim column.
// O^T = V^T . P^T leaves 4 consecutive head-dim values of one query per lane, so the row sum
// stays lane-local and the output goes out as 8-byte vectors.  Softmax statistics in fp32
// (wavefront shuffles across the 4 lane groups), head_dim 80 is padded to 96 with zeros.
#include "common.h"
#include "hamer_hip_internal.h"

namespace {

constexpr int T = 192;        // tokens (16 x 12 patches)
constexpr int HD = 80;        // head dim
constexpr int KSTR = 88;      // K row stride in LDS (elements): 176 B rows, conflict-free b128 reads
constexpr int VSTR = 80;      // V row stride in LDS (elements): 160 B = 40 dwords, 8 rows x 32 B tile the 64 banks
constexpr int KS_BYTES = T * KSTR * 2;
constexpr int VT_BYTES = T * VSTR * 2;
constexpr int ATT_LDS = KS_BYTES + VT_BYTES;   // 64,512 B -> 2 workgroups per CU

// ds_read_b64_tr_b16: within each 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
// 4 x 16 block; lane i receives column i of the 4 rows.  EXEC must be all ones.
template <class V4> __device__ __forceinline__ V4 lds_read_tr4(const void* p) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(V4, v);
}

template <class TT>
__global__ __launch_bounds__(256, 2) void vit_attention_kernel(const typename TT::elem* __restrict__ qkv,
                                                               typename TT::elem* __restrict__ out, int heads,
                                                               float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using elem = typename TT::elem;
  using vec8 = typename TT::vec8;
  using vec4 = typename TT::vec4;
  elem* Ks = (elem*)smem;
  elem* Vs = (elem*)(smem + KS_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int C = heads * HD;            // embed dim
  const size_t ld = (size_t)3 * C;
  const elem* base = qkv + (size_t)b * T * ld + (size_t)h * HD;

  // ---- stage K and V (both row-major) for this head
  for (int c = tid; c < T * (HD / 8); c += 256) {
    const int t = c / (HD / 8), ch = c % (HD / 8);
    *(vec8*)(Ks + t * KSTR + ch * 8) = *(const vec8*)(base + (size_t)t * ld + C + ch * 8);
    *(vec8*)(Vs + t * VSTR + ch * 8) = *(const vec8*)(base + (size_t)t * ld + 2 * C + ch * 8);
  }
  __syncthreads();

  const int g = lane >> 4, li = lane & 15;
  vec8 zero8;
#pragma unroll
  for (int i = 0; i < 8; ++i) zero8[i] = (elem)0.0f;

  for (int qt = 0; qt < 3; ++qt) {
    const int q = wave * 48 + qt * 16 + li;   // this lane's query row
    const elem* qrow = base + (size_t)q * ld;
    vec8 qf[3];
    qf[0] = *(const vec8*)(qrow + 8 * g);
    qf[1] = *(const vec8*)(qrow + 32 + 8 * g);
    qf[2] = g < 2 ? *(const vec8*)(qrow + 64 + 8 * g) : zero8;

    // S^T tiles: st[kt][r] = score(key kt*16 + 4g + r, query li)
    f32x4_t st[12];
#pragma unroll
    for (int kt = 0; kt < 12; ++kt) {
      const elem* krow = Ks + (kt * 16 + li) * KSTR + 8 * g;
      f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
      acc = TT::mfma(*(const vec8*)(krow), qf[0], acc);
      acc = TT::mfma(*(const vec8*)(krow + 32), qf[1], acc);
      const vec8 k2 = g < 2 ? *(const vec8*)(krow + 64) : zero8;
      acc = TT::mfma(k2, qf[2], acc);
      st[kt] = acc;
    }
    // softmax over the 192 keys of query li (spread over 4 lane groups x 48 registers)
    float m = st[0][0];
#pragma unroll
    for (int kt = 0; kt < 12; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, st[kt][r]);
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 12; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f((st[kt][r] - m) * scale_log2e);
        st[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // P^T fragments (B operand): k-step kk holds key tiles 2kk (slots j<4) and 2kk+1 (slots j>=4)
    vec8 pf[6];
#pragma unroll
    for (int kk = 0; kk < 6; ++kk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf[kk][r] = (elem)st[2 * kk][r];
        pf[kk][4 + r] = (elem)st[2 * kk + 1][r];
      }

    // O^T = V^T . P^T : lane holds O[query li][d = dt*16 + 4g + r]
    elem* orow = out + ((size_t)b * T + q) * C + h * HD;
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) {
      // this lane's address inside the 4-key x 16-column block of its lane group: key 4g + (li>>2), column 4*(li&3)
      const elem* vblk = Vs + (4 * g + (li >> 2)) * VSTR + dt * 16 + 4 * (li & 3);
      f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 6; ++kk) {
        const vec4 lo = lds_read_tr4<vec4>(vblk + (32 * kk) * VSTR);         // keys 32kk + 4g .. +3, column dt*16 + li
        const vec4 hi = lds_read_tr4<vec4>(vblk + (32 * kk + 16) * VSTR);    // keys 32kk + 16 + 4g .. +3
        vec8 vf;
#pragma unroll
        for (int i = 0; i < 4; ++i) { vf[i] = lo[i]; vf[4 + i] = hi[i]; }
        acc = TT::mfma(vf, pf[kk], acc);
      }
      vec4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (elem)(acc[r] * inv);
      *(vec4*)(orow + dt * 16 + 4 * g) = o;
    }
  }
}

template <class TT>
int launch_att(const void* qkv, void* out, int B, int heads, float scale, hipStream_t s) {
  static bool attr_set = false;
  auto kern = vit_attention_kernel<TT>;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS) != hipSuccess)
      return hm_set_error(HM_ERR_HIP, "hm_vit_attention: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(B * heads), dim3(256), ATT_LDS, s, (const typename TT::elem*)qkv,
                     (typename TT::elem*)out, heads, scale * 1.44269504088896340736f);
  return hm_check_launch("hm_vit_attention");
}

}  // namespace

extern "C" int hm_vit_attention(const void* qkv, void* out, int B, int tokens, int heads, int head_dim, float scale,
                                int dtype, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!qkv || !out || B <= 0 || heads <= 0) return hm_set_error(HM_ERR_ARG, "hm_vit_attention: bad arguments");
  if (tokens != T || head_dim != HD)
    return hm_set_error(HM_ERR_ARG, "hm_vit_attention: built for 192 tokens and head_dim 80 (ViT-H/16 on 256x192)");
  if (((uintptr_t)qkv | (uintptr_t)out) & 15) return hm_set_error(HM_ERR_ARG, "hm_vit_attention: 16-byte alignment");
  HmProfScope prof(HM_K_ATTENTION, 0, B, heads, head_dim, s);
  if (dtype == HM_DTYPE_BF16) return launch_att<TBf16>(qkv, out, B, heads, scale, s);
  if (dtype == HM_DTYPE_F16) return launch_att<TF16>(qkv, out, B, heads, scale, s);
  return hm_set_error(HM_ERR_ARG, "hm_vit_attention: bad dtype");
}
