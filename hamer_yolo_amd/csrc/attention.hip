// Attention.forward core of the ViT backbone (vit.py:115-123): softmax(scale * q k^T) v for
// 192 tokens, 16 heads of head_dim 80.  HBM-bound (94 MB of QKV in, 31 MB out per layer at B = 64,
// 12 GFLOP), so the kernel is organised around the copy, not the math: PERSISTENT workgroups (one per
// CU) walk the (crop, head) items; K and V of item i+1 stream into the second half of a 126 KB LDS
// double buffer by LDS-DMA (and its Q rows into registers) while the 12 wavefronts compute item i,
// one 16-query MFMA tile each.  (One workgroup per item with the copy in front ran every CU's copy
// at the same moment and left HBM idle during the math: 38-43 us per layer against 19 us for the
// copy alone; this form: 31.5 us, of which 28 us are the copies and stores by themselves.)  V is consumed TRANSPOSED (the A operand of O^T = V^T . P^T) straight from its
// row-major image with ds_read_b64_tr_b16, the gfx950 transposing LDS read.
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
constexpr int BUF_BYTES = KS_BYTES + VT_BYTES;  // 64,512 B per (crop, head)
constexpr int NWAVE = T / 16;                   // 12 wavefronts, one 16-query tile each
constexpr int OTILE_BYTES = 16 * HD * 2;        // a wave's output tile, staged so rows leave as 160-byte runs
constexpr int ATT_LDS = 2 * BUF_BYTES + NWAVE * OTILE_BYTES;   // double buffer 129,024 B + 30,720 B: one workgroup per CU
constexpr int ATT_LDS_TOME = ATT_LDS + 2 * T * 4;              // + log2(token size) of the two items in flight

// ds_read_b64_tr_b16: within each 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
// 4 x 16 block; lane i receives column i of the 4 rows.  EXEC must be all ones.
template <class V4> __device__ __forceinline__ V4 lds_read_tr4(const void* p) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(V4, v);
}

// LDS-DMA hidden from hipcc (inline asm; M0 = destination base, written in the same statement).  Issued
// through the builtin, hipcc knows an LDS write is pending and puts vmcnt(0) in front of the next LDS read that might
// alias it -- here the V reads of the item being computed, i.e. it would wait for the NEXT item's copy in the middle
// of the math.  The copy's completion is counted by hand instead: wait_vm0() + barrier at the top of every step.
__device__ __forceinline__ void glds16_hidden(const void* gsrc, void* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(dst) : "memory");   // (M0 overwritten, not saved: round 4, see gemm.hip)
}

// s_waitcnt vmcnt(0) through the builtin (simm16: expcnt and lgkmcnt fields at their maxima): hipcc's waitcnt pass
// sees it.  After an inline-asm wait it still believes the Q loads of the previous step are pending and puts its own
// vmcnt(0) in front of their first use -- behind the next item's copies, which serialises copy and math.
__device__ __forceinline__ void wait_vm0() {
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" ::: "memory");
}

// MX8: the output is the MXFP8 operand of an fp8 proj GEMM instead of 16-bit rows.  A head's 80 columns do not align with
// 32-element scale blocks, so each head is widened to HDP = 96 columns (3 blocks, the last 16 columns zero): out8 is
// [B*T][heads*96] e4m3 bytes, out_scales [heads*3][B*T] E8M0 -- the proj weight is laid out with the same K order.
constexpr int HDP = 96;
typedef __attribute__((ext_vector_type(4))) int v4i_att;

// TOME (token merging, hm_tome_attention): the same kernel for Tn <= 192 tokens per crop and "proportional attention",
// softmax(scale q.k + log(size_k)) (selective_vit_adapter.py:185-187).  The waves keep their fixed 16-query tiles (those
// past Tn only copy), the twelve key tiles stay (keys >= Tn are masked by a select, so whatever their LDS rows hold never
// reaches an exp; their V rows are zeroed once -- the copies never touch them), log2(size) of an item's tokens travels
// one register per thread with the item's copies and is put into LDS in front of the step's barrier.
template <class TT, bool MX8 = false, bool TOME = false>
__global__ __launch_bounds__(64 * NWAVE, 3) void vit_attention_kernel(const typename TT::elem* __restrict__ qkv,
                                                                    typename TT::elem* __restrict__ out, int heads,
                                                                    int items, float scale_log2e,
                                                                    unsigned char* __restrict__ out_scales = nullptr,
                                                                    const float* __restrict__ size = nullptr, int Tn = T) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using elem = typename TT::elem;
  using vec8 = typename TT::vec8;
  using vec4 = typename TT::vec4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = heads * HD;            // embed dim
  const size_t ld = (size_t)3 * C;
  const int g = lane >> 4, li = lane & 15;
  const int Tt = TOME ? Tn : T;                    // tokens per crop in this launch
  // TOME: [2][T] floats at smem + ATT_LDS hold log2(size) of the items in the two buffers (addressed from `smem` at every
  // use: kept in a pointer variable, hipcc loses the LDS address space and then emits an illegal flat-address test)
  vec8 zero8;
#pragma unroll
  for (int i = 0; i < 8; ++i) zero8[i] = (elem)0.0f;
  if constexpr (TOME) {                            // V rows of keys that do not exist: zero, once, in both buffers
    for (int buf = 0; buf < 2; ++buf) {
      char* Vb = smem + buf * BUF_BYTES + KS_BYTES;
      for (int c = Tt * (VSTR / 8) + tid; c < T * (VSTR / 8); c += 64 * NWAVE) *(vec8*)(Vb + c * 16) = zero8;
    }
  }

  // Copy of one item: K and V (both row-major) by LDS-DMA -- the LDS image is lane-linear (16 B per lane, 1 KiB per
  // instruction), so chunk c of the image maps to (row c / CPR, 16-byte column c % CPR); K rows carry one pad chunk
  // (CPR = 11: it re-reads column 0, never used), V rows none (CPR = 10) -- and this wave's 16 query rows to registers.
  auto issue = [&](int item, int buf, vec8 (&qf)[3], float& szl) {
    const elem* base = qkv + (size_t)(item / heads) * Tt * ld + (size_t)(item % heads) * HD;
    char* Kb = smem + buf * BUF_BYTES;
    char* Vb = Kb + KS_BYTES;
    constexpr int KCH = T * (KSTR / 8), VCH = T * (VSTR / 8);       // 2112 and 1920 chunks at 192 tokens
    const int kch = Tt * (KSTR / 8), vch = Tt * (VSTR / 8);          // TOME: the chunks of the Tn rows that exist
    // (loop bounds stay compile-time constants, the runtime count is a guard inside: with runtime bounds this hipcc emits an
    // illegal flat-address test for the LDS destination.)  A partial last group copies its rows >= Tn from row Tn - 1 --
    // in bounds, finite, and multiplied by P = 0.
    for (int c0 = wave * 64; c0 < KCH; c0 += 64 * NWAVE) {
      const int c = c0 + lane, ch = c % (KSTR / 8);
      int t = c / (KSTR / 8);
      if (TOME) t = t < Tt ? t : Tt - 1;
      if (!TOME || c0 < kch) glds16_hidden(base + (size_t)t * ld + C + (ch < HD / 8 ? ch : 0) * 8, Kb + c0 * 16);
    }
    for (int c0 = wave * 64; c0 < VCH; c0 += 64 * NWAVE) {
      const int c = c0 + lane, ch = c % (VSTR / 8);
      int t = c / (VSTR / 8);
      if (TOME) t = t < Tt ? t : Tt - 1;
      if (!TOME || c0 < vch) glds16_hidden(base + (size_t)t * ld + 2 * C + ch * 8, Vb + c0 * 16);
    }
    const int qr = wave * 16 + li;
    const elem* qrow = base + (size_t)(TOME ? (qr < Tt ? qr : Tt - 1) : qr) * ld;
    if constexpr (TOME) szl = (size != nullptr && tid < Tt) ? size[(size_t)(item / heads) * Tt + tid] : 1.0f;
    qf[0] = *(const vec8*)(qrow + 8 * g);
    qf[1] = *(const vec8*)(qrow + 32 + 8 * g);
    qf[2] = g < 2 ? *(const vec8*)(qrow + 64 + 8 * g) : zero8;
  };

  auto compute = [&](int item, int buf, const vec8 (&qf)[3]) {
    const elem* Ks = (const elem*)(smem + buf * BUF_BYTES);
    const elem* Vs = (const elem*)(smem + buf * BUF_BYTES + KS_BYTES);
    const int b = item / heads, h = item % heads;
    const int q = wave * 16 + li;             // this lane's query row
    if constexpr (TOME) { if (wave * 16 >= Tt) return; }      // (wave-uniform: this wave has no query of the item)
    // S^T tiles: st[kt][r] = score(key kt*16 + 4g + r, query li).  K fragments are fetched two key tiles ahead of
    // the MFMAs that use them (left to itself hipcc waits lgkmcnt(0) in front of every MFMA: ~200 cycles each)
    f32x4_t st[12];
    vec8 kf[2][2][3];
    auto load_k = [&](int kb, int pair) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const elem* krow = Ks + ((pair * 2 + j) * 16 + li) * KSTR + 8 * g;
        kf[kb][j][0] = *(const vec8*)(krow);
        kf[kb][j][1] = *(const vec8*)(krow + 32);
        kf[kb][j][2] = g < 2 ? *(const vec8*)(krow + 64) : zero8;
      }
    };
    load_k(0, 0);
#pragma unroll
    for (int pair = 0; pair < 6; ++pair) {
      if (pair + 1 < 6) load_k((pair + 1) & 1, pair + 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
        acc = TT::mfma(kf[pair & 1][j][0], qf[0], acc);
        acc = TT::mfma(kf[pair & 1][j][1], qf[1], acc);
        acc = TT::mfma(kf[pair & 1][j][2], qf[2], acc);
        st[pair * 2 + j] = acc;
      }
    }
    if constexpr (TOME) {                      // z = scale q.k + log(size) in the log2 domain; keys past Tn: -inf
#pragma unroll
      for (int kt = 0; kt < 12; ++kt) {
        const f32x4_t lz = *(const f32x4_t*)(smem + ATT_LDS + (buf * T + kt * 16 + 4 * g) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) st[kt][r] = (kt * 16 + 4 * g + r < Tt) ? fmaf(st[kt][r], scale_log2e, lz[r]) : -INFINITY;
      }
    }
    // softmax over the 192 keys of query li (spread over 4 lane groups x 48 registers)
    float m = st[0][0];
#pragma unroll
    for (int kt = 0; kt < 12; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, st[kt][r]);
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mneg = TOME ? -m : -m * scale_log2e;
    float sum4[4] = {0.f, 0.f, 0.f, 0.f};          // four partial sums: no 48-deep dependent add chain
#pragma unroll
    for (int kt = 0; kt < 12; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (TOME: p * 2^14 -- with p <= 1 the lo part of a typical p ~ 1/192 would be an fp16 subnormal; the factor leaves
        //  through `inv`)
        const float p = __builtin_amdgcn_exp2f(TOME ? st[kt][r] + (mneg + 14.0f) : fmaf(st[kt][r], scale_log2e, mneg));
        st[kt][r] = p;
        sum4[r] += p;
      }
    float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // P^T fragments (B operand): k-step kk holds key tiles 2kk (slots j<4) and 2kk+1 (slots j>=4).
    // TOME: P enters the PV product as hi + lo 16-bit parts (two MFMAs), i.e. with ~2x the mantissa: token matching is a
    // discrete decision on the keys of the NEXT block, and a P rounded to 11 (8) bits moves those keys enough to flip
    // near-ties against the oracle; the FLOPs are free here (the kernel is bound by its copies).
    vec8 pf[6], pl[TOME ? 6 : 1];
#pragma unroll
    for (int kk = 0; kk < 6; ++kk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf[kk][r] = (elem)st[2 * kk][r];
        pf[kk][4 + r] = (elem)st[2 * kk + 1][r];
        if constexpr (TOME) {
          pl[kk][r] = (elem)(st[2 * kk][r] - (float)pf[kk][r]);
          pl[kk][4 + r] = (elem)(st[2 * kk + 1][r] - (float)pf[kk][4 + r]);
        }
      }

    // O^T = V^T . P^T : lane holds O[query li][d = dt*16 + 4g + r].  Five independent accumulators (one per 16
    // head-dim columns), V fragments of the next 32 keys in flight while this k-step's MFMAs run.
    // this lane's address inside the 4-key x 16-column block of its lane group: key 4g + (li>>2), column 4*(li&3)
    const elem* vblk = Vs + (4 * g + (li >> 2)) * VSTR + 4 * (li & 3);
    vec4 vlo[2][5], vhi[2][5];
    auto load_v = [&](int vb, int kk) {
#pragma unroll
      for (int dt = 0; dt < 5; ++dt) {
        vlo[vb][dt] = lds_read_tr4<vec4>(vblk + (32 * kk) * VSTR + dt * 16);        // keys 32kk + 4g .. +3, column dt*16 + li
        vhi[vb][dt] = lds_read_tr4<vec4>(vblk + (32 * kk + 16) * VSTR + dt * 16);   // keys 32kk + 16 + 4g .. +3
      }
    };
    f32x4_t oacc[5];
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) oacc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (!TOME) load_v(0, 0);
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) {
      if constexpr (TOME) load_v(kk & 1, kk);        // (TOME holds hi and lo P fragments: no registers for a V set in flight)
      else if (kk + 1 < 6) load_v((kk + 1) & 1, kk + 1);
#pragma unroll
      for (int dt = 0; dt < 5; ++dt) {
        vec8 vf;
#pragma unroll
        for (int i = 0; i < 4; ++i) { vf[i] = vlo[kk & 1][dt][i]; vf[4 + i] = vhi[kk & 1][dt][i]; }
        oacc[dt] = TT::mfma(vf, pf[kk], oacc[dt]);
        if constexpr (TOME) oacc[dt] = TT::mfma(vf, pl[kk], oacc[dt]);
      }
    }
    // through this wave's LDS tile: the accumulator layout gives 8 bytes per lane and 32-byte runs per row; written
    // from LDS a lane carries 16 bytes and a row leaves as one 160-byte run (3 store instructions instead of 5)
    if constexpr (MX8) {
      const int Mrows = (items / heads) * T;
      // block amax over this lane's values, then over the 4 lane groups that share query li
      float am[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < 5; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) am[dt >> 1] = fmaxf(am[dt >> 1], fabsf(oacc[dt][r] * inv));
      unsigned sb[3];
      float isc[3];
#pragma unroll
      for (int bk = 0; bk < 3; ++bk) {
        float a = am[bk];
        a = fmaxf(a, __shfl_xor(a, 16, 64));
        a = fmaxf(a, __shfl_xor(a, 32, 64));
        sb[bk] = mx8_scale_byte(a);
        isc[bk] = mx8_inv_scale(sb[bk]);
      }
      char* ot8 = smem + 2 * BUF_BYTES + wave * OTILE_BYTES;            // [16][96] bytes (1.5 KB of the 2.5 KB tile)
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) {
        int p = 0;
        if (dt < 5) {
          const float sc = inv * isc[dt >> 1];
          p = mx8_pack4(oacc[dt][0] * sc, oacc[dt][1] * sc, oacc[dt][2] * sc, oacc[dt][3] * sc);
        }
        *(int*)(ot8 + li * HDP + dt * 16 + 4 * g) = p;
      }
      char* obase8 = (char*)out + ((size_t)b * T + wave * 16) * (heads * HDP) + h * HDP;
#pragma unroll
      for (int c0 = 0; c0 < 16 * (HDP / 16); c0 += 64) {
        const int c = c0 + lane;
        if (c < 16 * (HDP / 16)) {
          const int row = c / (HDP / 16), ch = c % (HDP / 16);
          *(v4i_att*)(obase8 + (size_t)row * (heads * HDP) + ch * 16) = *(const v4i_att*)(ot8 + row * HDP + ch * 16);
        }
      }
      if (g == 0) {
#pragma unroll
        for (int bk = 0; bk < 3; ++bk) out_scales[(size_t)(h * 3 + bk) * Mrows + (size_t)b * T + q] = (unsigned char)sb[bk];
      }
      return;
    }
    elem* ot = (elem*)(smem + 2 * BUF_BYTES + wave * OTILE_BYTES);
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) {
      vec4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (elem)(oacc[dt][r] * inv);
      *(vec4*)(ot + li * HD + dt * 16 + 4 * g) = o;
    }
    elem* obase = out + ((size_t)b * Tt + wave * 16) * C + h * HD;
#pragma unroll
    for (int c0 = 0; c0 < 16 * (HD / 8); c0 += 64) {
      const int c = c0 + lane;
      if (c < 16 * (HD / 8)) {
        const int row = c / (HD / 8), ch = c % (HD / 8);
        if (!TOME || wave * 16 + row < Tt) *(vec8*)(obase + (size_t)row * C + ch * 8) = *(const vec8*)(ot + row * HD + ch * 8);
      }
    }
  };

  // Item loop, unrolled by two so the double buffer and the two Q register sets have fixed roles.  At the top of a
  // step everything this wave issued a step ago is waited for (K/V copies and Q of the current item; the previous
  // item's output stores), the barrier then (a) publishes all waves' copies and (b) says every wave is done reading
  // the other buffer, which the next item's copies overwrite while this item is computed.
  vec8 qa[3], qb[3];
  float sza = 1.0f, szb = 1.0f;
  auto publish_size = [&](int buf, float szl) {    // TOME: this thread's token of the item that just landed in `buf`
    if constexpr (TOME) { if (tid < Tt) *(float*)(smem + ATT_LDS + (buf * T + tid) * 4) = __builtin_amdgcn_logf(szl); }   // v_log_f32 = log2
  };
  // XCD-aware item order: workgroups b, b+8, .. share an XCD and its L2; give them consecutive items, i.e. all heads of
  // the same crops at the same time -- a head's rows are 160-byte slices of 7680-byte token rows, so neighbouring heads
  // share 128-byte lines both in QKV (fetched once per XCD instead of once per head) and in the output (merged in L2)
  int item = xcd_remap(blockIdx.x, gridDim.x);
  if (item < items) issue(item, 0, qa, sza);
  while (item < items) {
    wait_vm0();
    publish_size(0, sza);
    __builtin_amdgcn_s_barrier();
    int nxt = item + gridDim.x;
    if (nxt < items) issue(nxt, 1, qb, szb);
    compute(item, 0, qa);
    item = nxt;
    if (item >= items) break;
    wait_vm0();
    publish_size(1, szb);
    __builtin_amdgcn_s_barrier();
    nxt = item + gridDim.x;
    if (nxt < items) issue(nxt, 0, qa, sza);
    compute(item, 1, qb);
    item = nxt;
  }
}

template <class TT, bool MX8 = false, bool TOME = false>
int launch_att(const void* qkv, void* out, int B, int heads, float scale, hipStream_t s, void* out_scales = nullptr,
               const float* size = nullptr, int Tn = T) {
  static HmLdsOnce lds_once;
  auto kern = vit_attention_kernel<TT, MX8, TOME>;
  constexpr int LDS = TOME ? ATT_LDS_TOME : ATT_LDS;
  if (const int rc = lds_once.ensure((const void*)kern, LDS, "hm_vit_attention: cannot raise dynamic LDS limit")) return rc;
  const int n_cu = hm_device_cu_count();
  if (n_cu <= 0) return hm_set_error(HM_ERR_HIP, "hm_vit_attention: cannot query the device");
  const int items = B * heads;
  // persistent: one workgroup per CU, every workgroup the same number of items when items % CUs == 0
  const int per = (items + n_cu - 1) / n_cu, grid = (items + per - 1) / per;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NWAVE), LDS, s, (const typename TT::elem*)qkv,
                     (typename TT::elem*)out, heads, items, scale * 1.44269504088896340736f, (unsigned char*)out_scales, size, Tn);
  return hm_check_launch("hm_vit_attention");
}

}  // namespace

// hm_tome_attention's MFMA path (tome.hip): tokens <= 192 per crop, head_dim 80, log(size) on the key axis
int hm_attention_tome_launch(const void* qkv, const float* size, void* out, int B, int tokens, int heads, float scale, int dtype,
                             hipStream_t s) {
  if (tokens < 1 || tokens > T) return hm_set_error(HM_ERR_ARG, "hm_tome_attention: 0 < tokens <= 192");
  if (dtype == HM_DTYPE_BF16) return launch_att<TBf16, false, true>(qkv, out, B, heads, scale, s, nullptr, size, tokens);
  if (dtype == HM_DTYPE_F16) return launch_att<TF16, false, true>(qkv, out, B, heads, scale, s, nullptr, size, tokens);
  return hm_set_error(HM_ERR_ARG, "hm_tome_attention: bad dtype");
}

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

extern "C" int hm_vit_attention_mx8(const void* qkv, void* out8, void* out_scales, int B, int tokens, int heads, int head_dim,
                                    float scale, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!qkv || !out8 || !out_scales || B <= 0 || heads <= 0) return hm_set_error(HM_ERR_ARG, "hm_vit_attention_mx8: bad arguments");
  if (tokens != T || head_dim != HD)
    return hm_set_error(HM_ERR_ARG, "hm_vit_attention_mx8: built for 192 tokens and head_dim 80 (ViT-H/16 on 256x192)");
  if (((uintptr_t)qkv | (uintptr_t)out8 | (uintptr_t)out_scales) & 15) return hm_set_error(HM_ERR_ARG, "hm_vit_attention_mx8: 16-byte alignment");
  HmProfScope prof(HM_K_ATTENTION, 1, B, heads, head_dim, s);
  return launch_att<TBf16, true>(qkv, out8, B, heads, scale, s, out_scales);
}
