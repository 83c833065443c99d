// hm_hamer_forward: HAMER.forward_step (hamer.py:99-156) as one stream enqueue -- patch gather,
// 1 + 4*depth + 1 MFMA GEMMs, LayerNorms, fused attention, the fp32 decoder head and the fused
// MANO tail.  Host code only sequences kernels; no allocation, no sync (graph-capturable).
#include "common.h"
#include <stdlib.h>
#include "hamer_hip_internal.h"

int hm_split_head(const float* head, int ldh, float* pose6d, float* betas, float* cam, int B, hipStream_t s);

namespace {

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
  size_t patches, x, h, qkv, att, mlp, kv, xd, hd, t1, t2, head, stats, rowstats, hs, mlps, atts, partials, total;
  size_t x2, tsize, tmetric, tindex;          // token merging: second residual stream, sizes [2][M], metric [M][80], indices
  int ksplit_proj, ksplit_fc2;
};

// K ranges for x += X.W^T at small M: as many as fill the 256 CUs with 128x128 tiles, whole 64-deep K tiles per range.
// proj / fc2 at 4 hands: 60 tiles x 20 / 80 K-steps on 60 CUs -> 240 workgroups x 5 / 20 K-steps.
int pick_split_k(int M, int N, int K) {
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128), nk = K / 64;
  if (M > 2048 || N < 512 || K % 64 != 0 || tiles * 2 > 256) return 1;      // few hands: hm_gemm picks its 128x128 tile there
  int want = 256 / tiles;
  if (want > 8) want = 8;                               // the consumer reads every slab: diminishing returns
  if (want > nk) want = nk;
  while (want > 1 && nk % want != 0) --want;
  return want;
}

Layout make_layout(const hm_hamer_weights& w, int B) {
  const int gh = (w.img_h + 2 * w.pad - w.patch) / w.patch + 1, gw = (w.win_w + 2 * w.pad - w.patch) / w.patch + 1;
  const size_t M = (size_t)B * gh * gw, D = w.embed_dim;
  const size_t inner = (size_t)w.dec_heads * w.dec_dim_head;
  const size_t dmax = (size_t)(w.dec_dim > w.dec_mlp ? w.dec_dim : w.dec_mlp);
  Layout L;
  size_t o = 0;
  L.patches = o; o += align256(M * 3 * w.patch * w.patch * 2);
  L.x = o; o += align256(M * D * 4);
  L.h = o; o += align256(M * D * 2);
  L.qkv = o; o += align256(M * 3 * D * 2);
  L.att = o; o += align256(M * D * 2);
  L.mlp = o; o += align256(M * (size_t)w.mlp_dim * 2);
  L.kv = o; o += align256(M * (size_t)w.dec_depth * 2 * inner * 2);
  L.xd = o; o += align256((size_t)B * w.dec_dim * 4);
  L.hd = o; o += align256((size_t)B * w.dec_dim * 4);
  L.t1 = o; o += align256((size_t)B * (inner > dmax ? inner : dmax) * 4);
  L.t2 = o; o += align256((size_t)B * (inner > dmax ? inner : dmax) * 4);
  L.head = o; o += align256((size_t)B * 112 * 4);
  L.stats = o; o += align256(M * ((D + 63) / 64) * 8);      // deferred-LN partials [D/64][M][2]
  L.rowstats = o; o += align256(M * 8);                      // (mean, rstd) per row
  L.hs = o; o += align256(((D + 31) / 32) * M);               // fp8 path: E8M0 scales of the MXFP8 LayerNorm output [D/32][M]
  L.mlps = o; o += align256((((size_t)w.mlp_dim + 31) / 32) * M);   // ... and of the GELU output [mlp/32][M]
  L.atts = o; o += align256((size_t)w.heads * 3 * M);           // ... and of the attention output [heads*3][M] (fp8 proj)
  // split-K of the two N = D GEMMs of a block when their 128x128 tiles would leave most CUs idle (a few hands)
  L.ksplit_proj = pick_split_k((int)M, (int)D, (int)D);
  L.ksplit_fc2 = pick_split_k((int)M, (int)D, w.mlp_dim);
  const int smax = L.ksplit_proj > L.ksplit_fc2 ? L.ksplit_proj : L.ksplit_fc2;
  L.partials = o; o += smax > 1 ? align256((size_t)smax * M * D * 4) : 0;
  L.x2 = L.tsize = L.tmetric = L.tindex = 0;
  if (w.tome_r) {
    L.x2 = o; o += align256(M * D * 4);
    L.tsize = o; o += align256(2 * M * 4);
    L.tmetric = o; o += align256(M * (D / w.heads) * 4);
    L.tindex = o; o += align256(hm_tome_index_bytes(B));
    L.ksplit_proj = L.ksplit_fc2 = 1;
    // the token count shrinks block by block: late blocks have a few hundred rows, and their N = D GEMMs are split over K
    // like the few-hands path (pick_split_k per block: <= 8 slabs of <= 2048 rows)
    const size_t prow = M < 2048 ? M : 2048;
    L.partials = o; o += align256(8 * prow * D * 4);
  }
  L.total = o;
  return L;
}

int check_weights(const hm_hamer_weights* w) {
  if (!w) return hm_set_error(HM_ERR_ARG, "hm_hamer_forward: null weights");
  if (!w->blocks || !w->layers || !w->patch_w || !w->patch_b || !w->pos || !w->last_g || !w->last_b || !w->token0 ||
      !w->kv_w || !w->head_w || !w->head_b)
    return hm_set_error(HM_ERR_ARG, "hm_hamer_forward: incomplete weights");
  if (w->embed_dim % w->heads != 0) return hm_set_error(HM_ERR_ARG, "hm_hamer_forward: embed_dim % heads != 0");
  return HM_OK;
}

}  // namespace

extern "C" size_t hm_hamer_workspace_bytes(const hm_hamer_weights* w, int B) {
  if (!w || B <= 0) return 0;
  return make_layout(*w, B).total;
}

#define HM_TRY(expr) do { int _rc = (expr); if (_rc != HM_OK) return _rc; } while (0)

extern "C" int hm_hamer_forward(const hm_hamer_weights* w, const float* img, int B, const hm_hamer_outputs* out,
                                void* workspace, size_t workspace_bytes, void* stream) {
  HM_TRY(check_weights(w));
  if (!img || !out || !workspace || B <= 0) return hm_set_error(HM_ERR_ARG, "hm_hamer_forward: bad arguments");
  if (!out->pose6d || !out->betas || !out->cam || !out->rotmats || !out->verts || !out->joints || !out->cam_t || !out->kp2d)
    return hm_set_error(HM_ERR_ARG, "hm_hamer_forward: incomplete outputs");
  const Layout L = make_layout(*w, B);
  if (workspace_bytes < L.total) return hm_set_error(HM_ERR_ARG, "hm_hamer_forward: workspace too small");
  if ((uintptr_t)workspace & 255) return hm_set_error(HM_ERR_ARG, "hm_hamer_forward: workspace must be 256-byte aligned");
  char* ws = (char*)workspace;
  const int gh = (w->img_h + 2 * w->pad - w->patch) / w->patch + 1, gw = (w->win_w + 2 * w->pad - w->patch) / w->patch + 1;
  const int tokens = gh * gw, M = B * tokens, D = w->embed_dim, dt = w->dtype;
  const int kpe = 3 * w->patch * w->patch;
  float* x = (float*)(ws + L.x);
  void *h = ws + L.h, *qkv = ws + L.qkv, *att = ws + L.att, *mlp = ws + L.mlp, *kv = ws + L.kv;

  float *stats = (float*)(ws + L.stats), *rowstats = (float*)(ws + L.rowstats);
  // deferred LayerNorm (hm_gemm HM_EPI_RESID_LN / HM_EPI_LN_*): `ln` = gamma of the next LayerNorm for a GEMM that
  // writes x, or the folded column sums for a GEMM that reads LN(x); `h` then holds x * gamma instead of LN(x)
  auto gemm = [&](const void* X, int ldx, const void* W, int K, int N, void* C, int ldc, const float* bias, int epi,
                  const float* resid, int ldr, int rmod, const float* ln = nullptr, float out_scale = 0.f) {
    hm_gemm_args g{};
    g.X = X; g.W = W; g.C = C; g.bias = bias; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.ldx = ldx; g.ldw = K; g.ldc = ldc; g.ldr = ldr; g.resid_mod = rmod;
    g.epilogue = epi; g.dtype = dt; g.out_scale = out_scale;
    if (epi == HM_EPI_RESID_LN) { g.ln_gamma = ln; g.ln_xg = h; g.ln_stats = stats; }
    if (epi == HM_EPI_LN_STORE || epi == HM_EPI_LN_GELU) { g.ln_colsum = ln; g.ln_stats = rowstats; }
    HM_TRY(hm_gemm(&g, stream));
    if (epi == HM_EPI_RESID_LN) return hm_ln_finalize(stats, rowstats, M, N, w->vit_eps, stream);
    return HM_OK;
  };
  // x += X.W^T + b followed by LayerNorm(x) -> h (or tok): one residual GEMM + LayerNorm, or, for a few hands, a
  // split-K GEMM into partial slabs and a LayerNorm that adds them (hm_layernorm_accum)
  float* partials = (float*)(ws + L.partials);
  auto resid_gemm_ln = [&](const void* X, int K, const void* W, const float* bias, int ksplit, const float* ln_g,
                           const float* ln_b, void* ln_out) {
    if (ksplit > 1) {
      hm_gemm_args g{};
      g.X = X; g.W = W; g.C = partials; g.M = M; g.N = D; g.K = K; g.ldx = K; g.ldw = K; g.ldc = D;
      g.epilogue = HM_EPI_F32; g.dtype = dt; g.k_split = ksplit;
      HM_TRY(hm_gemm(&g, stream));
      return hm_layernorm_accum(x, partials, ksplit, bias, ln_g, ln_b, ln_out, dt, M, D, w->vit_eps, stream);
    }
    HM_TRY(gemm(X, K, W, K, D, x, D, bias, HM_EPI_RESID_F32, x, D, 0));
    return hm_layernorm(x, ln_g, ln_b, ln_out, dt, M, D, w->vit_eps, stream);
  };
  // fp8 path (BASELINE configs[4]): qkv / fc1 / fc2 on hm_gemm_fp8 when every block carries e4m3 weights
  bool fp8 = dt == HM_DTYPE_BF16 && D % 128 == 0 && w->mlp_dim % 128 == 0 && w->depth > 0;
  for (int i = 0; i < w->depth && fp8; ++i) {
    const hm_vit_block& b = w->blocks[i];
    fp8 = b.qkv_w8 && b.qkv_ws && b.fc1_w8 && b.fc1_ws && b.fc2_w8 && b.fc2_ws;
  }
  void *hs = ws + L.hs, *mlps = ws + L.mlps, *atts = ws + L.atts;
  bool fp8_proj = fp8 && D / w->heads == 80;            // att (16-bit [M][D]) has room for the widened [M][heads*96] bytes
  for (int i = 0; i < w->depth && fp8_proj; ++i) fp8_proj = w->blocks[i].proj_w8 && w->blocks[i].proj_ws;
  auto gemm8 = [&](const void* X8, const void* xsc, int K, const void* W8, const float* wsc, int N, void* C, int ldc,
                   const float* bias, int epi, const float* resid, void* out_scales) {
    hm_gemm_fp8_args g{};
    g.X8 = X8; g.x_scales = xsc; g.W8 = W8; g.w_scale = wsc; g.C = C; g.bias = bias; g.resid = resid; g.out_scales = out_scales;
    g.M = M; g.N = N; g.K = K; g.ldx = K; g.ldw = K; g.ldc = ldc; g.ldr = ldc; g.epilogue = epi; g.out_dtype = HM_DTYPE_BF16;
    return hm_gemm_fp8(&g, stream);
  };
  bool fold = !fp8 && D % 64 == 0 && w->depth > 0;
  for (int i = 0; i < w->depth && fold; ++i) {
    const hm_vit_block& b = w->blocks[i];
    fold = b.qkv_colsum && b.qkv_bias_ln && b.fc1_colsum && b.fc1_bias_ln;
  }

  // range probe (load-time calibration, hm_hamer_weights.range_stats): slot layout in include/hamer_hip.h
  float* rs = w->range_stats;
  if (rs && (fp8 || fold || w->tome_r)) return hm_set_error(HM_ERR_ARG, "hm_hamer_forward: range_stats is for the dense 16-bit path");
  auto probe = [&](const void* p, int ld, int rows, int col0, int ncols, int slot) {
    return rs ? hm_absmax16(p, ld, rows, col0, ncols, dt, rs + slot, stream) : HM_OK;
  };
  for (int i = 0; i < w->depth; ++i)
    if ((w->blocks[i].attn_scale_mul != 0.f && w->blocks[i].attn_scale_mul != 1.f) || (w->blocks[i].gelu_out_scale != 0.f && w->blocks[i].gelu_out_scale != 1.f))
      if (fp8 || fold || w->tome_r) return hm_set_error(HM_ERR_ARG, "hm_hamer_forward: the range prescale exists on the dense 16-bit path only");

  // ---- ViT backbone (vit.py:320-339)
  HM_TRY(hm_patch_im2col(img, ws + L.patches, B, w->img_h, w->img_w_full, w->win_x0, w->win_w, w->patch, w->pad, dt, stream));
  const float scale = 1.0f / sqrtf((float)(D / w->heads));
  void* tok = out->tokens ? out->tokens : h;
  int ctx_tokens = tokens;                              // tokens per crop the decoder attends over
  if (w->tome_r) {
    // token-merging variant (selective_vit_adapter.py ToMeBlock.forward :210-235): the token count T shrinks after the
    // attention of every block, the same for every crop, so all buffers stay compact [B * T][.] and only M changes
    if (fp8 || D / w->heads != 80) return hm_set_error(HM_ERR_ARG, "hm_hamer_forward: token merging needs the 16-bit path and head_dim 80");
    int T = tokens;
    float* xc = x;                                      // current / spare residual stream
    float* xn = (float*)(ws + L.x2);
    float *szc = nullptr, *szn = (float*)(ws + L.tsize);
    auto gemm_m = [&](int Mi, const void* X, int ldx, const void* W, int K, int N, void* C, int ldc, const float* bias, int epi,
                      const float* resid, int ldr, int rmod) {
      hm_gemm_args g{};
      g.X = X; g.W = W; g.C = C; g.bias = bias; g.resid = resid;
      g.M = Mi; g.N = N; g.K = K; g.ldx = ldx; g.ldw = K; g.ldc = ldc; g.ldr = ldr; g.resid_mod = rmod;
      g.epilogue = epi; g.dtype = dt;
      return hm_gemm(&g, stream);
    };
    // xcur += X.W^T + bias for Mi rows: one residual GEMM, or -- once merging has left few rows -- K split into slabs that
    // hm_layernorm_accum adds in a fixed order (its LayerNorm output lands in h and is simply overwritten by the
    // LayerNorm the block issues next)
    auto resid_m = [&](int Mi, const void* X, int K, const void* W, const float* bias, float* xcur, const float* lg, const float* lb) {
      const int ks = hm_option(HM_OPT_TOME_NO_SPLITK) ? 1 : pick_split_k(Mi, D, K);     // (hm_set_option: tests compare the two routes)
      if (ks > 1 && Mi <= 2048) {
        hm_gemm_args g{};
        g.X = X; g.W = W; g.C = partials; g.M = Mi; g.N = D; g.K = K; g.ldx = K; g.ldw = K; g.ldc = D;
        g.epilogue = HM_EPI_F32; g.dtype = dt; g.k_split = ks;
        HM_TRY(hm_gemm(&g, stream));
        return hm_layernorm_accum(xcur, partials, ks, bias, lg, lb, h, dt, Mi, D, w->vit_eps, stream);
      }
      return gemm_m(Mi, X, K, W, K, D, xcur, D, bias, HM_EPI_RESID_F32, xcur, D, 0);
    };
    HM_TRY(gemm_m(M, ws + L.patches, kpe, w->patch_w, kpe, D, xc, D, w->patch_b, HM_EPI_RESID_F32, w->pos, D, tokens));
    for (int i = 0; i < w->depth; ++i) {
      const hm_vit_block& b = w->blocks[i];
      int Mi = B * T;
      HM_TRY(hm_layernorm(xc, b.ln1_g, b.ln1_b, h, dt, Mi, D, w->vit_eps, stream));
      HM_TRY(gemm_m(Mi, h, D, b.qkv_w, D, 3 * D, qkv, 3 * D, b.qkv_b, HM_EPI_STORE, nullptr, 0, 0));
      const int rr = w->tome_r[i] < T / 2 ? w->tome_r[i] : T / 2;
      // matching metric in fp32 (k.mean(heads) is linear in LN1(x)): fp32 LayerNorm output into the spare residual buffer (free
      // until this block's merge writes it), then one small fp32 linear; must read x BEFORE proj updates it
      if (b.kmean_w && rr > 0) {
        HM_TRY(hm_layernorm(xc, b.ln1_g, b.ln1_b, xn, HM_OUT_F32, Mi, D, w->vit_eps, stream));
        HM_TRY(hm_linear_f32(xn, D, b.kmean_w, D, b.kmean_b, nullptr, 0, (float*)(ws + L.tmetric), D / w->heads, Mi, D / w->heads, D, 0, stream));
      }
      HM_TRY(hm_tome_attention(qkv, szc, att, B, T, w->heads, D / w->heads, scale, dt, stream));
      HM_TRY(resid_m(Mi, att, D, b.proj_w, b.proj_b, xc, b.ln2_g, b.ln2_b));
      int r = w->tome_r[i] < T / 2 ? w->tome_r[i] : T / 2;          // r = min(r, t // 2) (:42)
      if (r > 0) {
        if (b.kmean_w)
          HM_TRY(hm_tome_merge_metric((const float*)(ws + L.tmetric), D / w->heads, 0, xc, szc, xn, szn, (int*)(ws + L.tindex), B, T, r, D, stream));
        else
          HM_TRY(hm_tome_merge(qkv, xc, szc, xn, szn, (float*)(ws + L.tmetric), (int*)(ws + L.tindex), B, T, r, w->heads,
                               D / w->heads, D, dt, stream));
        float* t0 = xc; xc = xn; xn = t0;
        float* s_old = szc;
        szc = szn;
        szn = s_old ? s_old : (float*)(ws + L.tsize) + M;            // the two halves of the size buffer alternate
        T -= r; Mi = B * T;
      }
      HM_TRY(hm_layernorm(xc, b.ln2_g, b.ln2_b, h, dt, Mi, D, w->vit_eps, stream));
      HM_TRY(gemm_m(Mi, h, D, b.fc1_w, D, w->mlp_dim, mlp, w->mlp_dim, b.fc1_b, HM_EPI_GELU, nullptr, 0, 0));
      HM_TRY(resid_m(Mi, mlp, w->mlp_dim, b.fc2_w, b.fc2_b, xc, b.ln2_g, b.ln2_b));
    }
    HM_TRY(hm_layernorm(xc, w->last_g, w->last_b, tok, dt, B * T, D, w->vit_eps, stream));
    ctx_tokens = T;
  } else {
  if (fold) HM_TRY(gemm(ws + L.patches, kpe, w->patch_w, kpe, D, x, D, w->patch_b, HM_EPI_RESID_LN, w->pos, D, tokens, w->blocks[0].ln1_g));
  else HM_TRY(gemm(ws + L.patches, kpe, w->patch_w, kpe, D, x, D, w->patch_b, HM_EPI_RESID_F32, w->pos, D, tokens));
  for (int i = 0; i < w->depth; ++i) {
    const hm_vit_block& b = w->blocks[i];
    if (fp8) {
      // h / mlp hold e4m3 bytes here (half of their 16-bit size), hs / mlps the block scales
      HM_TRY(hm_layernorm_mx8(x, b.ln1_g, b.ln1_b, h, hs, M, D, w->vit_eps, stream));
      HM_TRY(gemm8(h, hs, D, b.qkv_w8, b.qkv_ws, 3 * D, qkv, 3 * D, b.qkv_b, HM_EPI_STORE, nullptr, nullptr));
      if (fp8_proj) {
        HM_TRY(hm_vit_attention_mx8(qkv, att, atts, B, tokens, w->heads, D / w->heads, scale, stream));
        HM_TRY(gemm8(att, atts, w->heads * 96, b.proj_w8, b.proj_ws, D, x, D, b.proj_b, HM_EPI_RESID_F32, x, nullptr));
      } else {
        HM_TRY(hm_vit_attention(qkv, att, B, tokens, w->heads, D / w->heads, scale, dt, stream));
        HM_TRY(gemm(att, D, b.proj_w, D, D, x, D, b.proj_b, HM_EPI_RESID_F32, x, D, 0));
      }
      HM_TRY(hm_layernorm_mx8(x, b.ln2_g, b.ln2_b, h, hs, M, D, w->vit_eps, stream));
      HM_TRY(gemm8(h, hs, D, b.fc1_w8, b.fc1_ws, w->mlp_dim, mlp, w->mlp_dim, b.fc1_b, HM_EPI_GELU_MX8, nullptr, mlps));
      HM_TRY(gemm8(mlp, mlps, w->mlp_dim, b.fc2_w8, b.fc2_ws, D, x, D, b.fc2_b, HM_EPI_RESID_F32, x, nullptr));
    } else if (fold) {
      HM_TRY(gemm(h, D, b.qkv_w, D, 3 * D, qkv, 3 * D, b.qkv_bias_ln, HM_EPI_LN_STORE, nullptr, 0, 0, b.qkv_colsum));
      HM_TRY(hm_vit_attention(qkv, att, B, tokens, w->heads, D / w->heads, scale, dt, stream));
      HM_TRY(gemm(att, D, b.proj_w, D, D, x, D, b.proj_b, HM_EPI_RESID_LN, x, D, 0, b.ln2_g));
      HM_TRY(gemm(h, D, b.fc1_w, D, w->mlp_dim, mlp, w->mlp_dim, b.fc1_bias_ln, HM_EPI_LN_GELU, nullptr, 0, 0, b.fc1_colsum));
      if (i + 1 < w->depth) HM_TRY(gemm(mlp, w->mlp_dim, b.fc2_w, w->mlp_dim, D, x, D, b.fc2_b, HM_EPI_RESID_LN, x, D, 0, w->blocks[i + 1].ln1_g));
      else HM_TRY(gemm(mlp, w->mlp_dim, b.fc2_w, w->mlp_dim, D, x, D, b.fc2_b, HM_EPI_RESID_F32, x, D, 0));   // last_norm stays a kernel
    } else {
      // the LayerNorm that follows each residual GEMM is issued with it: LN1 of block i+1 (or last_norm) after fc2
      if (i == 0) {
        HM_TRY(hm_layernorm(x, b.ln1_g, b.ln1_b, h, dt, M, D, w->vit_eps, stream));
        HM_TRY(probe(h, D, M, 0, D, 0));
      }
      HM_TRY(gemm(h, D, b.qkv_w, D, 3 * D, qkv, 3 * D, b.qkv_b, HM_EPI_STORE, nullptr, 0, 0));
      for (int c = 0; c < 3; ++c) HM_TRY(probe(qkv, 3 * D, M, c * D, D, 6 * i + 1 + c));
      HM_TRY(hm_vit_attention(qkv, att, B, tokens, w->heads, D / w->heads, b.attn_scale_mul != 0.f ? scale * b.attn_scale_mul : scale, dt, stream));
      HM_TRY(resid_gemm_ln(att, D, b.proj_w, b.proj_b, L.ksplit_proj, b.ln2_g, b.ln2_b, h));
      HM_TRY(probe(h, D, M, 0, D, 6 * i + 4));
      HM_TRY(gemm(h, D, b.fc1_w, D, w->mlp_dim, mlp, w->mlp_dim, b.fc1_b, HM_EPI_GELU, nullptr, 0, 0, nullptr, b.gelu_out_scale));
      HM_TRY(probe(mlp, w->mlp_dim, M, 0, w->mlp_dim, 6 * i + 5));
      const bool last = i + 1 == w->depth;
      HM_TRY(resid_gemm_ln(mlp, w->mlp_dim, b.fc2_w, b.fc2_b, L.ksplit_fc2, last ? w->last_g : w->blocks[i + 1].ln1_g,
                           last ? w->last_b : w->blocks[i + 1].ln1_b, last ? tok : h));
      HM_TRY(probe(last ? tok : h, D, M, 0, D, 6 * (i + 1)));       // LN1 of block i + 1, or last_norm (slot 6 * depth)
    }
  }
  if (fold || fp8) HM_TRY(hm_layernorm(x, w->last_g, w->last_b, tok, dt, M, D, w->vit_eps, stream));
  }   // dense backbone

  // ---- decoder head (mano_head.py:61-95, pose_transformer.py:191-201)
  const int dim = w->dec_dim, inner = w->dec_heads * w->dec_dim_head, ldkv = w->dec_depth * 2 * inner;
  {
    hm_gemm_args g{};                                   // to_kv of all decoder layers over the B * ctx_tokens context rows
    g.X = tok; g.W = w->kv_w; g.C = kv; g.M = B * ctx_tokens; g.N = ldkv; g.K = D; g.ldx = D; g.ldw = D; g.ldc = ldkv;
    g.epilogue = HM_EPI_STORE; g.dtype = dt;
    HM_TRY(hm_gemm(&g, stream));
    for (int i = 0; i < w->dec_depth; ++i) {
      HM_TRY(probe(kv, ldkv, B * ctx_tokens, i * 2 * inner, inner, 6 * w->depth + 1 + 2 * i));
      HM_TRY(probe(kv, ldkv, B * ctx_tokens, i * 2 * inner + inner, inner, 6 * w->depth + 2 + 2 * i));
    }
  }
  float *xd = (float*)(ws + L.xd), *hd = (float*)(ws + L.hd), *t1 = (float*)(ws + L.t1), *t2 = (float*)(ws + L.t2);
  HM_TRY(hm_broadcast_rows(w->token0, xd, B, dim, stream));
  const float dscale = 1.0f / sqrtf((float)w->dec_dim_head);
  for (int i = 0; i < w->dec_depth; ++i) {
    const hm_dec_layer& l = w->layers[i];
    // self-attention over a single token: softmax == 1, so out = to_out(to_v(LN(x))) (pose_transformer.py:75-86); with the
    // folded matrix sa_w = to_out . to_v that is one linear.  (Folding the PreNorm LayerNorms into the linears as well was
    // tried: every workgroup redoing the row statistics costs what the 18 LayerNorm launches cost.)
    HM_TRY(hm_layernorm(xd, l.ln0_g, l.ln0_b, hd, HM_OUT_F32, B, dim, w->dec_eps, stream));
    if (l.sa_w) {
      HM_TRY(hm_linear_f32(hd, dim, l.sa_w, dim, l.sa_out_b, xd, dim, xd, dim, B, dim, dim, 0, stream));
    } else {
      HM_TRY(hm_linear_f32(hd, dim, l.sa_v_w, dim, nullptr, nullptr, 0, t1, inner, B, inner, dim, 0, stream));
      HM_TRY(hm_linear_f32(t1, inner, l.sa_out_w, inner, l.sa_out_b, xd, dim, xd, dim, B, dim, inner, 0, stream));
    }
    // cross-attention on the backbone tokens (pose_transformer.py:111-124)
    HM_TRY(hm_layernorm(xd, l.ln1_g, l.ln1_b, hd, HM_OUT_F32, B, dim, w->dec_eps, stream));
    HM_TRY(hm_linear_f32(hd, dim, l.ca_q_w, dim, nullptr, nullptr, 0, t1, inner, B, inner, dim, 0, stream));
    HM_TRY(hm_cross_attention(t1, kv, ldkv, i * 2 * inner, i * 2 * inner + inner, t2, B, ctx_tokens, w->dec_heads,
                              w->dec_dim_head, l.ca_scale_mul != 0.f ? dscale * l.ca_scale_mul : dscale, dt, stream));
    HM_TRY(hm_linear_f32(t2, inner, l.ca_out_w, inner, l.ca_out_b, xd, dim, xd, dim, B, dim, inner, 0, stream));
    // feed-forward (pose_transformer.py:40-52)
    HM_TRY(hm_layernorm(xd, l.ln2_g, l.ln2_b, hd, HM_OUT_F32, B, dim, w->dec_eps, stream));
    HM_TRY(hm_linear_f32(hd, dim, l.ff1_w, dim, l.ff1_b, nullptr, 0, t1, w->dec_mlp, B, w->dec_mlp, dim, 1, stream));
    HM_TRY(hm_linear_f32(t1, w->dec_mlp, l.ff2_w, w->dec_mlp, l.ff2_b, xd, dim, xd, dim, B, dim, w->dec_mlp, 0, stream));
  }
  float* head = (float*)(ws + L.head);
  HM_TRY(hm_linear_f32(xd, dim, w->head_w, dim, w->head_b, nullptr, 0, head, 112, B, 112, dim, 0, stream));
  HM_TRY(hm_split_head(head, 112, out->pose6d, out->betas, out->cam, B, (hipStream_t)stream));
  // ---- rot6d + MANO + projection (hamer.py:131-154)
  return hm_mano_forward(&w->mano, out->pose6d, out->betas, out->cam, out->rotmats, out->verts, out->joints, out->cam_t,
                         out->kp2d, B, w->focal_length, w->image_size, stream);
}

