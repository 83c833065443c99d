// Shared device helpers for the gfx950 (CDNA4, MI355X) kernels of libhamer_hip.
// Wavefront = 64 lanes everywhere; MFMA fragment maps follow the 16x16x32 bf16/f16 shape:
//   A operand: lane l holds A[row l&15][k = 8*(l>>4) + j], j = 0..7
//   B operand: lane l holds B[k = 8*(l>>4) + j][col l&15]
//   C/D      : lane l holds D[row 4*(l>>4) + r][col l&15], r = 0..3
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// Element-type policies: both run on the same MFMA pipe at the same rate.
struct TBf16 {
  using elem = __bf16;
  using vec8 = bf16x8_t;
  using vec4 = bf16x4_t;
  static __device__ __forceinline__ f32x4_t mfma(vec8 a, vec8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
struct TF16 {
  using elem = _Float16;
  using vec8 = f16x8_t;
  using vec4 = f16x4_t;
  static __device__ __forceinline__ f32x4_t mfma(vec8 a, vec8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// 16-byte asynchronous global -> LDS copy (global_load_lds_dwordx4).  The LDS destination
// is the wave-uniform `lds_base` plus lane*16; the global source is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU (torch.nn.GELU default; vit.py:77, pose_transformer.py:45)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // v_rcp_f32, 1 ulp
// The same SiLU on two / four values: the products and the sum go out as v_pk_mul_f32 / v_pk_add_f32 (two results per issue
// slot), the two transcendentals stay per value -- ~24 issue cycles per value instead of ~32, and the convolution epilogues are
// bound by exactly these (round 4: the stem pair kernel runs 102 SiLUs per lane and tile next to 128 MFMAs).  Every operation is
// the one silu() performs (__expf(-x) IS v_exp_f32 of x * -log2(e)), so the results are bit-identical to it; the product is
// pinned in fp32 before anything converts it (left alone hipcc may fuse a multiply and a 16-bit conversion into
// v_fma_mixlo_f16: one rounding instead of two, other bits in one of ~10^4 values).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t silu2(f32x2_t x) {
  const f32x2_t t = x * f32x2_t{-1.44269504088896340736f, -1.44269504088896340736f};
  f32x2_t e;
  e[0] = __builtin_amdgcn_exp2f(t[0]); e[1] = __builtin_amdgcn_exp2f(t[1]);
  const f32x2_t d = e + f32x2_t{1.0f, 1.0f};
  f32x2_t r;
  r[0] = __builtin_amdgcn_rcpf(d[0]); r[1] = __builtin_amdgcn_rcpf(d[1]);
  f32x2_t p = x * r;
  asm volatile("" : "+v"(p));
  return p;
}
__device__ __forceinline__ f32x4_t silu4(f32x4_t x) {
  const f32x2_t lo = silu2(f32x2_t{x[0], x[1]}), hi = silu2(f32x2_t{x[2], x[3]});
  return f32x4_t{lo[0], lo[1], hi[0], hi[1]};
}
// acc + bias for four values as two packed adds (the same IEEE additions as four __fadd_rn)
__device__ __forceinline__ f32x4_t add4(f32x4_t a, f32x4_t b) {
  const f32x2_t lo = f32x2_t{a[0], a[1]} + f32x2_t{b[0], b[1]}, hi = f32x2_t{a[2], a[3]} + f32x2_t{b[2], b[3]};
  return f32x4_t{lo[0], lo[1], hi[0], hi[1]};
}

// XCD-aware, bijective block remap: blocks b and b+8 share an XCD (and its L2) under the
// round-robin dispatch, so give each XCD a contiguous run of tile ids.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// MXFP8 block quantisation (OCP e4m3 elements, one E8M0 scale per 32 elements along K).  The scale byte of a block whose
// largest magnitude is amax is ceil(log2(amax / 448)) + 127, read off the float's exponent field (E8M0 and fp32 share the
// bias); elements are multiplied by the exact inverse power of two and rounded to e4m3 (RNE), so |element| <= 448.
__device__ __forceinline__ unsigned mx8_scale_byte(float amax) {
  const unsigned bits = __float_as_uint(amax * (1.0f / 448.0f));
  return (bits >> 23) + ((bits & 0x7FFFFFu) != 0u);
}
__device__ __forceinline__ float mx8_inv_scale(unsigned sb) { return __uint_as_float((254u - sb) << 23); }
__device__ __forceinline__ int mx8_pack4(float a, float b, float c, float d) {
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
}
// max over each aligned group of 4 lanes / 8 lanes on the DPP path
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  return v;
}
__device__ __forceinline__ float pair_max(float v) {      // max over lanes 2k, 2k+1
  return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
}
__device__ __forceinline__ float row8_max(float v) {
  v = quad_max(v);
  return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
}

#define HM_OK 0
#define HM_ERR_ARG (-1)
#define HM_ERR_HIP (-2)
