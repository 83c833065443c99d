/* libhamer_hip: C ABI of the MI355X (gfx950) hand-mesh hot path.
 *
 * The reference (2646207530/hamer-yolo) has no FFI: its boundary is a set of Python call
 * signatures.  Each entry point below replaces the PyTorch/cv2/smplx
 * arithmetic behind one of those calls; the Python host layer (hamer_yolo_amd/) keeps the
 * reference's names and argument meaning and binds these symbols with ctypes
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - every pointer is DEVICE memory owned by the caller unless the name ends in _host;
 *    the library allocates nothing persistent and keeps no reference after return;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*), no internal sync;
 *  - return value: 0 on success, negative library code otherwise (HM_ERR_*); the message is
 *    available from hm_last_error_string() (thread-local); nothing throws across the ABI;
 *  - matrices are row-major; "ld*" are leading dimensions in ELEMENTS;
 *  - dtype selects the 16-bit GEMM operand type (bf16 or fp16, same MFMA rate).
 */
#ifndef HAMER_HIP_H
#define HAMER_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HM_VERSION 401   /* 401: hm_conv2d_stem_pair, HM_OP_CONV_PAIR, HM_OPT_CONV_STEM_PAIR; 400 = round 4: hm_option_count, hm_gemm_px_grid (302 = round 3: hm_set_option, hm_hamer_weights.tome_r) -- lib.load() checks it */

enum { HM_DTYPE_BF16 = 0, HM_DTYPE_F16 = 1 };

/* GEMM epilogues */
enum {
  HM_EPI_STORE = 0,     /* C(16-bit) = acc + bias                                  */
  HM_EPI_GELU = 1,      /* C(16-bit) = gelu_erf(acc + bias)        vit.py:82-87    */
  HM_EPI_RESID_F32 = 2, /* C(f32)    = acc + bias + resid[m % resid_mod][n]        */
  HM_EPI_F32 = 3,       /* C(f32)    = acc + bias                                  */
  HM_EPI_SILU = 4,      /* C(16-bit) = silu(acc + bias)      yolov7 common.py:114  */
  /* deferred LayerNorm (Block.forward vit.py:148-151: x += f(LN(x))): the GEMM that writes the residual
   * stream also emits the next LayerNorm's statistics and x*gamma, the GEMM that follows applies them:
   * LN(x).W^T + b = rstd * ((x*gamma).W^T - mean * colsum) + (b + W.beta), colsum[n] = sum_k W[n][k]*gamma[k] */
  HM_EPI_RESID_LN = 5,  /* RESID_F32, plus ln_xg = C*ln_gamma (16-bit) and ln_stats     */
  HM_EPI_LN_STORE = 6,  /* C(16-bit) = rstd*(acc - mean*ln_colsum) + bias               */
  HM_EPI_LN_GELU = 7,   /* C(16-bit) = gelu_erf(rstd*(acc - mean*ln_colsum) + bias)     */
  HM_EPI_GELU_MX8 = 8,  /* hm_gemm_fp8 only: C = MXFP8(gelu_erf(acc + bias)): e4m3 bytes + E8M0 scale per 32 columns */
  HM_EPI_RELU = 9,      /* hm_conv2d_nhwc only: C(16-bit) = relu(acc + bias)                 torchvision BasicBlock     */
  HM_EPI_ADD_RELU = 10  /* hm_conv2d_nhwc only: C(16-bit) = relu(acc + bias + resid16[m][n]) (out += identity; relu)    */
};

typedef struct hm_gemm_args {
  const void* X;      /* [M][ldx] 16-bit activations                                  */
  const void* W;      /* [N][ldw] 16-bit weights (nn.Linear layout)                   */
  void* C;            /* [M][ldc] 16-bit or f32 by epilogue                           */
  const float* bias;  /* [N] or NULL                                                  */
  const float* resid; /* [resid_mod or M][ldr] f32, HM_EPI_RESID_F32 only (may alias C) */
  int M, N, K;
  int ldx, ldw, ldc, ldr;
  int resid_mod;      /* >0: residual row = m % resid_mod (positional embedding)      */
  int epilogue;
  int dtype;
  /* deferred LayerNorm, NULL / 0 for the other epilogues */
  const float* ln_gamma;  /* RESID_LN: [N] gamma of the LayerNorm that reads C                            */
  void* ln_xg;            /* RESID_LN: out [M][N] 16-bit, C * gamma (the consuming GEMM's X)              */
  float* ln_stats;        /* RESID_LN: out [N/64][M][2] = (sum, sum of squares) of C per 64 columns;
                             LN_*: in [M][2] = (mean, rstd) per row, made from those by hm_ln_finalize     */
  const float* ln_colsum; /* LN_*: [N] sum_k W[n][k] * gamma[k] over the 16-bit W; bias = b + W.beta      */
  /* split-K (HM_EPI_F32, bias == NULL): K is cut into k_split ranges computed by separate workgroups; range s
   * writes its partial product to C + s*M*ldc.  For small M (few output tiles, long K); the consumer adds the
   * slabs in order (hm_layernorm_accum), so the result is deterministic.  0 or 1: off. */
  int k_split;
  /* HM_EPI_GELU only: C = gelu_erf(acc + bias) * out_scale, a power of two chosen at load so that the 16-bit activation stays
   * finite (the consumer's weights carry 1 / out_scale; HamerEngine prescale).  0 or 1: off. */
  float out_scale;
} hm_gemm_args;

/* nn.Linear forward on MFMA: C = epilogue(X . W^T).  Replaces the aten::addmm calls behind
 * Attention.qkv/.proj (vit.py:114,:124), Mlp.fc1/.fc2 (vit.py:83,:85), PatchEmbed.proj
 * (vit.py:172, after hm_patch_im2col) and CrossAttention.to_kv (pose_transformer.py:114). */
int hm_gemm(const hm_gemm_args* args, void* stream);
/* HM_EPI_RESID_LN partials [D/64][M][2] -> row_stats [M][2] = (mean, 1/sqrt(var + eps)) for HM_EPI_LN_*. */
int hm_ln_finalize(const float* partials, float* row_stats, int M, int D, float eps, void* stream);
/* Tuning hook: pin the GEMM tile configuration (see launch_gemm in gemm.hip: 0 = 128x128, 10 = 256x256 two-stage,
 * 24 = 256x256 with X two K-steps ahead, 26 = persistent; the experimental tiles and the wrong-result timing ablations
 * exist only in the -DHM_ABLATIONS build, libhamer_hip_abl.so); -1 restores the default
 * (also settable through the HM_GEMM_VARIANT environment variable).  Results do not depend on it
 * beyond fp32 summation order. */
int hm_gemm_set_variant(int variant);
/* Tuning hook: M-tiles per group in the XCD-aware tile walk (default 8). */
int hm_gemm_set_group_m(int group_m);
/* Process-wide test / tuning switches.  Launch paths read these, never the environment (HM_GEMM_VARIANT and HM_PX_GRID are
 * read ONCE, on first use, as start-up defaults for tuning runs).  Returns 0, or HM_ERR_ARG for an unknown key / bad value. */
enum {
  HM_OPT_PX_GRID = 0,               /* workgroups of the persistent 16-bit GEMM (0 = one per CU)                        */
  HM_OPT_FP8P_GRID = 1,             /* workgroups of the persistent fp8 GEMM (0 = one per CU; tests: few, many tiles each) */
  HM_OPT_FP8_ONE_TILE = 2,          /* 1: hm_gemm_fp8 always takes the one-tile kernel (tests compare the two)           */
  HM_OPT_FP8P_RESID = 3,            /* 1: persistent fp32-residual fp8 epilogue (bit-identical, measured not faster)     */
  HM_OPT_TOME_NO_SPLITK = 4,        /* 1: token-merging forward never splits proj / fc2 over K (tests compare the routes) */
  HM_OPT_TOME_SCALAR_ATTENTION = 5, /* 1: hm_tome_attention takes the fp32 lane-per-key kernel                           */
  HM_OPT_RESID_IN_EPILOGUE = 6,     /* 1: the fp32-residual GEMM fetches its residual rows in the epilogue (round-2 form) */
  HM_OPT_CONV_TILE = 7,             /* tuning: force convolution tile (1..9 = 128x128, 128x64, 128x32, 256x128, 256x256, 256x64, then the deep-ring 128x32, 128x64, 128x128; 10..15 = the two-K-group tiles; 16 = the persistent 256x256 GEMM kernel for the 1x1 layers it applies to); 0 = per-layer choice */
  HM_OPT_CONV_SPLITK = 8,           /* tuning: 1 = never split a convolution over K, n > 1 = ask for n ranges where splitting applies; 0 = automatic */
  HM_OPT_PX_LDS_EPILOGUE = 9,       /* persistent GEMM epilogue: 0 = per epilogue (GELU: lane swaps, store: through LDS), 1 = always LDS, 2 = always lane swaps */
  HM_OPT_CONV_DIRECT = 10,          /* direct kernels (3x3: 3(8) -> 32 stem, 64 -> 64 stride 1, 32 -> 64 stride 2; 1x1 with K, Cout in {128, 256}): 0 = all, each from its own tile count up, 1 = none (implicit GEMM everywhere), 2 = stem only, 3 = all at any size (experiments library only: 4 / 5 = the two stem layers without their activation / without their stores, WRONG results, bound diagnosis) */
  HM_OPT_GEMM_TILE_RULE = 11,       /* tuning: 1 = round 2's GEMM tile rule (256 x 256 only from 85 % full rounds), 0 = the rate model */
  HM_OPT_CONV_KGROUPS = 12,         /* tuning: 1 = no K groups inside a convolution workgroup (small maps), 0 = automatic */
  HM_OPT_CONV_GENERAL_LOADER = 13,  /* tuning / tests: 1 = the implicit-GEMM convolution takes its general loader (per-lane tap arithmetic every K-step) even where the lean one applies (Cin % 64 == 0); 0 = automatic.  Same bytes either way */
  HM_OPT_CONV_STEM_PAIR = 14,       /* tuning / tests: 1 = hm_conv2d_stem_pair (and HM_OP_CONV_PAIR of hm_yolo_run) always runs its two convolutions as two launches; 0 = one launch where the fused kernel applies.  Same bytes either way */
  HM_OPT_COUNT = 15
};
int hm_set_option(int key, int value);
int hm_get_option(int key);
/* HM_OPT_COUNT of the library as built: a binding checks its own key table against it (hamer_yolo_amd/lib.py load()). */
int hm_option_count(void);
/* Host-side query, no device work: the workgroup count the persistent 16-bit GEMM takes for `tiles` whole 256 x 256 tiles on a
 * chip of `cus` compute units (0: 256) under the current HM_OPT_PX_GRID -- tests check that the option is not sticky. */
int hm_gemm_px_grid(int tiles, int cus);

/* The fp8 flavour of hm_gemm for BASELINE configs[4] ("fp8 ViT-H weights on CDNA4 fp8 MFMA"): C = epilogue(X . W^T) on
 * v_mfma_scale_f32_16x16x128_f8f6f4 (2x the bf16 MFMA rate, half the operand bytes).
 *   X : MXFP8 -- e4m3 (OCP) bytes [M][ldx] plus one E8M0 scale per row and 32 K-elements, stored [K/32][M]
 *       (written by hm_layernorm_mx8 and by the HM_EPI_GELU_MX8 epilogue);
 *   W : e4m3 bytes [N][ldw] with one f32 scale per output channel (w_scale[n] = max|W[n]| / 448);
 *   epilogues HM_EPI_STORE (16-bit C, out_dtype), HM_EPI_RESID_F32, HM_EPI_GELU_MX8 (C bytes [M][ldc], out_scales [N/32][M]).
 * M % 16 == 0, N % 64 == 0, K % 128 == 0. */
typedef struct hm_gemm_fp8_args {
  const void* X8; const void* x_scales;
  const void* W8; const float* w_scale;
  void* C; const float* bias; const float* resid; void* out_scales;
  int M, N, K, ldx, ldw, ldc, ldr;
  int epilogue;
  int out_dtype;      /* HM_EPI_STORE: HM_DTYPE_BF16 or HM_DTYPE_F16 */
} hm_gemm_fp8_args;
int hm_gemm_fp8(const hm_gemm_fp8_args* args, void* stream);
/* nn.LayerNorm with MXFP8 output (the X operand of hm_gemm_fp8): out8 [M][D] e4m3, out_scales [D/32][M] E8M0.  D % 32 == 0. */
int hm_layernorm_mx8(const float* x, const float* gamma, const float* beta, void* out8, void* out_scales, int M, int D,
                     float eps, void* stream);

/* nn.LayerNorm over the last dim (vit.py:136,:144,:252 eps 1e-6; t_cond_mlp.py:51-52 eps 1e-5).
 * x [M][D] f32 -> out [M][D]; out_dtype: HM_DTYPE_BF16 / HM_DTYPE_F16 / HM_OUT_F32. */
#define HM_OUT_F32 2
int hm_layernorm(const float* x, const float* gamma, const float* beta, void* out, int out_dtype,
                 int M, int D, float eps, void* stream);
/* The residual add in front of a LayerNorm (Block.forward vit.py:148-151) for a split-K producer:
 * x[m] += bias + sum_s partials[s][m] (s ascending; partials [n_partials][M][D] f32 from hm_gemm k_split), then
 * out = LayerNorm(x) as hm_layernorm.  x is updated in place. */
int hm_layernorm_accum(float* x, const float* partials, int n_partials, const float* bias, const float* gamma,
                       const float* beta, void* out, int out_dtype, int M, int D, float eps, void* stream);
/* max |x[m][col0 + c]| over m < M, c < ncols of a 16-bit matrix with row pitch ld, folded into *slot (device, >= 0 on entry)
 * with an atomic max: the range probe of hm_hamer_weights.range_stats. */
int hm_absmax16(const void* x, int ld, int M, int col0, int ncols, int dtype, float* slot, void* stream);

/* Attention.forward core (vit.py:115-123): softmax(scale q k^T) v for `tokens`=192 keys.
 * qkv [B*tokens][3*heads*head_dim] 16-bit, column = which*H*d + head*d + i (reshape at
 * vit.py:112); out [B*tokens][heads*head_dim] 16-bit (head-major, vit.py:123). */
int hm_vit_attention(const void* qkv, void* out, int B, int tokens, int heads, int head_dim, float scale,
                     int dtype, void* stream);

/* Token merging (selective_vit_adapter.py).  hm_tome_attention: ToMeAttention.forward core (:166-195) for any
 * tokens <= 192 -- softmax(scale q k^T + log(size)) v, `size` [B*tokens] f32 or NULL (no merge yet).
 * hm_tome_merge: the matching metric (head-averaged keys, :198), bipartite_soft_matching (:17-66: alternate tokens form
 * the sets A / B, every A token proposes its most similar B token, the r best proposals merge) and merge_wavg (:98-113)
 * in one call: x [B*tokens][D] f32 -> x_out [B*(tokens-r)][D], size_out [B*(tokens-r)] ([unmerged A tokens in proposal
 * order | B tokens]).  metric_ws: B*tokens*80 floats, index_ws: hm_tome_index_bytes(B). */
size_t hm_tome_index_bytes(int B);
int hm_tome_attention(const void* qkv, const float* size, void* out, int B, int tokens, int heads, int head_dim, float scale,
                      int dtype, void* stream);
int hm_tome_merge(const void* qkv, const float* x, const float* size, float* x_out, float* size_out, float* metric_ws,
                  int* index_ws, int B, int tokens, int r, int heads, int head_dim, int D, int dtype, void* stream);
/* hm_tome_merge with a caller-supplied fp32 metric: metric[(b*tokens + t) * ld_metric + d] (+ the same at column lo_off + d when
 * lo_off > 0: a hi / lo pair), d < 80. */
int hm_tome_merge_metric(const float* metric, int ld_metric, int lo_off, const float* x, const float* size, float* x_out,
                         float* size_out, int* index_ws, int B, int tokens, int r, int D, void* stream);

/* Same attention with MXFP8 output for an fp8 proj GEMM (bf16 qkv in).  Heads are widened from 80 to 96 columns so that
 * scale blocks of 32 never straddle two heads: out8 [B*tokens][heads*96] e4m3 bytes (columns 80..95 of every head zero),
 * out_scales [heads*3][B*tokens] E8M0.  The proj weight must use the same K order (hm_vit_block.proj_w8). */
int hm_vit_attention_mx8(const void* qkv, void* out8, void* out_scales, int B, int tokens, int heads, int head_dim, float scale,
                         void* stream);

/* PatchEmbed.proj im2col (vit.py:168-176, called on x[:, :, :, 32:-32] at hamer.py:119):
 * img [B][3][img_h][img_w_full] f32, window columns [x0, x0+win_w), conv k=patch, s=patch,
 * zero pad `pad` -> patches [B*gh*gw][3*patch*patch] 16-bit, K order (c, ky, kx). */
int hm_patch_im2col(const float* img, void* patches, int B, int img_h, int img_w_full, int x0, int win_w,
                    int patch, int pad, int dtype, void* stream);

/* Small f32 nn.Linear on f32-input MFMA: out[M][N] = act(x[M][K] . W[N][K]^T + bias) (+ resid).
 * act: 0 none, 1 gelu_erf.  Decoder layers (pose_transformer.py:40-124) and the read-out
 * heads (mano_head.py:93-95).  K % 16 == 0. */
int hm_linear_f32(const float* x, int ldx, const float* W, int ldw, const float* bias, const float* resid, int ldr,
                  float* out, int ldo, int M, int N, int K, int act, void* stream);


/* out[b][:] = vec[:] for b < B (the zero-token embedding, pose_transformer.py:350-354). */
int hm_broadcast_rows(const float* vec, float* out, int B, int D, void* stream);

/* CrossAttention core for one query token (pose_transformer.py:117-123):
 * q [B][heads*dim_head] f32; k,v rows b*tokens+t of kv (16-bit, ld = ldkv), k at column
 * k_off + h*dim_head, v at v_off + h*dim_head; out [B][heads*dim_head] f32.  dim_head == 64. */
int hm_cross_attention(const float* q, const void* kv, int ldkv, int k_off, int v_off, float* out, int B, int tokens,
                       int heads, int dim_head, float scale, int dtype, void* stream);

/* MANO-shaped model parameters (smplx.MANOLayer buffers; mano_wrapper.py:12-30). */
typedef struct hm_mano_model {
  const float* v_template;  /* [V][3]                */
  const float* shapedirs;   /* [V][3][10]            */
  const float* posedirs;    /* [135][3V]             */
  const float* J_regressor; /* [16][V]               */
  const float* lbs_weights; /* [V][16]               */
  int n_verts;              /* V = 778               */
} hm_mano_model;

/* rot6d_to_rotmat (geometry.py:47-70) + MANO.forward (mano_wrapper.py:32-44 -> smplx lbs) +
 * cam_t and perspective_projection (hamer.py:131-154), one workgroup per hand.
 * pose6d [B][96], betas [B][10], cam [B][3] ->
 * rotmats [B][16][3][3], verts [B][V][3], joints [B][21][3], cam_t [B][3], kp2d [B][21][2]. */
int hm_mano_forward(const hm_mano_model* model, const float* pose6d, const float* betas, const float* cam,
                    float* rotmats, float* verts, float* joints, float* cam_t, float* kp2d, int B,
                    float focal_length, float image_size, void* stream);

/* Batched affine bilinear crop (prepare_batch_bbox, infer.py:154-259; generate_image_patch_cv2,
 * datasets/utils.py:318-376).  The per-box map is cv2.warpAffine's inverted matrix in its
 * fixed-point form (source x = (x0 + rint(m0 * dst_x * 1024)) / 1024, 1/32-px bilinear). */
typedef struct hm_crop_box {
  double m0, m4;    /* d src_x / d dst_x, d src_y / d dst_y                         */
  int32_t x0, y0;   /* rint(offset * 1024) + 16                                      */
  int32_t flip;     /* 1: left hand, patch mirrored after the crop (infer.py:229-230) */
  int32_t reserved;
} hm_crop_box;

/* HOST helper (no GPU work): box centre (cx, cy) and square side `size` in frame pixels ->
 * hm_crop_box for a P x P patch (gen_trans_from_patch_cv, datasets/utils.py:82-129, rot 0). */
int hm_crop_box_from_bbox(double cx, double cy, double size, int flip, int P, hm_crop_box* out);

/* frame [H][W][3] u8 BGR (device); boxes [B] (device); out [B][3][P][P] f32 RGB,
 * (x - mean_c) / std_c with mean/std in 0..255 units (infer.py:145-146,:235-238). */
int hm_crop_batch(const uint8_t* frame, int H, int W, const hm_crop_box* boxes, float* out, int B, int P,
                  const float* mean3_host, const float* std3_host, void* stream);

/* Whole HAMER.forward_step (hamer.py:99-156) as one enqueue: see hm_hamer_forward below. */
typedef struct hm_vit_block {
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  const void *qkv_w, *proj_w, *fc1_w, *fc2_w;       /* 16-bit [N][K]                 */
  const float *qkv_b, *proj_b, *fc1_b, *fc2_b;
  /* optional (all four, in every block, or none): deferred-LayerNorm operands, see HM_EPI_RESID_LN.
   * *_colsum[n] = sum_k W[n][k]*ln_g[k] over the 16-bit weights, *_bias_ln = b + W.ln_b.  When present
   * hm_hamer_forward runs no LayerNorm kernel inside the blocks. */
  const float *qkv_colsum, *qkv_bias_ln, *fc1_colsum, *fc1_bias_ln;
  /* optional (all six, in every block, or none; dtype must be HM_DTYPE_BF16): the fp8 path of BASELINE configs[4].
   * e4m3 bytes [N][K] and one f32 scale per output channel for qkv / fc1 / fc2; hm_hamer_forward then runs those three
   * GEMMs through hm_gemm_fp8 with MXFP8 activations (hm_layernorm_mx8, HM_EPI_GELU_MX8); proj stays 16-bit. */
  const void *qkv_w8, *fc1_w8, *fc2_w8;
  const float *qkv_ws, *fc1_ws, *fc2_ws;
  /* optional on top of those: proj in fp8 as well.  proj_w8 is [D][heads*96] e4m3, column h*96 + d = proj.weight[:, h*80 + d]
   * for d < 80 and zero for d >= 80 (the K order of hm_vit_attention_mx8), proj_ws its per-output-channel scale. */
  const void* proj_w8;
  const float* proj_ws;
  /* optional, token merging only (round 3): the matching metric k.mean(heads) = LN1(x) . Wbar^T + bbar with
   * Wbar = mean over heads of the key rows of qkv.weight (the metric is linear in the keys, selective_vit_adapter.py:198), so
   * hm_hamer_forward forms it entirely in fp32 -- an fp32 LayerNorm output and hm_linear_f32 -- instead of averaging 16-bit
   * keys: merge decisions then follow the fp32 reference up to what the residual stream itself differs by.
   * kmean_w: f32 [80][embed_dim]; kmean_b: f32 [80]. */
  const float* kmean_w;
  const float* kmean_b;
  /* range prescale (round 4; 0 = 1): powers of two folded into the weights at load so that every 16-bit activation of the
   * block stays inside fp16 on checkpoints whose activations overflow it.  attn_scale_mul multiplies the softmax scale (q and k
   * rows were scaled down by 2^-aq, 2^-ak: attn_scale_mul = 2^(aq+ak)); gelu_out_scale is hm_gemm's out_scale for fc1
   * (fc2.weight carries its inverse).  Everything else (LayerNorm gamma / beta against the next weight's columns, v rows
   * against proj) is weight folding only. */
  float attn_scale_mul, gelu_out_scale;
} hm_vit_block;

typedef struct hm_dec_layer {
  const float *ln0_g, *ln0_b, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  const float* sa_v_w;    /* rows [2*inner, 3*inner) of to_qkv.weight: [inner][dim]  */
  const float* sa_w;      /* optional: to_out.weight . sa_v_w, [dim][dim] -- self-attention over one token is linear
                             (softmax == 1), so the two projections fold into one                                    */
  const float *sa_out_w, *sa_out_b;
  const float* ca_q_w;
  const float *ca_out_w, *ca_out_b;
  const float *ff1_w, *ff1_b, *ff2_w, *ff2_b;
  float ca_scale_mul;     /* range prescale (0 = 1): this layer's key rows of kv_w were scaled by 2^-a, the cross-attention scale takes 2^a */
} hm_dec_layer;

typedef struct hm_hamer_weights {
  /* ViT (backbones/vit.py) */
  int img_h, img_w_full, win_x0, win_w, patch, pad, embed_dim, depth, heads, mlp_dim;
  float vit_eps;
  const void* patch_w;       /* 16-bit [D][3*p*p]                                    */
  const float* patch_b;
  const float* pos;          /* f32 [tokens][D] = pos_embed[1:] + pos_embed[0]       */
  const hm_vit_block* blocks; /* host array of `depth` entries                        */
  const float *last_g, *last_b;
  /* decoder (components/pose_transformer.py, heads/mano_head.py) */
  int dec_dim, dec_depth, dec_heads, dec_dim_head, dec_mlp;
  float dec_eps;
  const float* token0;       /* f32 [dec_dim] = to_token_embedding.bias + pos_embedding */
  const void* kv_w;          /* 16-bit [dec_depth*2*inner][embed_dim], layers stacked  */
  const hm_dec_layer* layers; /* host array of `dec_depth` entries                     */
  const float* head_w;       /* f32 [112][dec_dim]: decpose(96) | decshape(10) | deccam(3) | 3 zero rows */
  const float* head_b;       /* f32 [112]: bias + init_{hand_pose,betas,cam}            */
  hm_mano_model mano;
  float focal_length, image_size;
  int dtype;
  /* token merging (HAMER_INFER(token_merge=True), hamer.py:481-483): host array of `depth` ints, the tokens to merge away
   * after the attention of each block (parse_r of selective_vit_adapter.py:132-157), or NULL for the dense backbone */
  const int* tome_r;
  /* calibration (load time only): device array of 6 * depth + 1 + 2 * dec_depth floats, zeroed by the caller, or NULL.  When
   * set (dense 16-bit path only), the forward also records the largest magnitude of every 16-bit activation class:
   * per block [LN1 out, q, k, v, LN2 out, GELU out], then last_norm out, then per decoder layer [k, v] of the to_kv output. */
  float* range_stats;
} hm_hamer_weights;

typedef struct hm_hamer_outputs {
  float* pose6d;   /* [B][96]  */
  float* betas;    /* [B][10]  */
  float* cam;      /* [B][3]   */
  float* rotmats;  /* [B][16][3][3]: [:,0] = global_orient, [:,1:] = hand_pose */
  float* verts;    /* [B][V][3]  */
  float* joints;   /* [B][21][3] */
  float* cam_t;    /* [B][3]     */
  float* kp2d;     /* [B][21][2] */
  void* tokens;    /* optional [B*tokens][D] 16-bit copy of the backbone output, or NULL */
} hm_hamer_outputs;

/* Bytes of workspace hm_hamer_forward needs for a batch of B crops. */
size_t hm_hamer_workspace_bytes(const hm_hamer_weights* w, int B);

/* img [B][3][img_h][img_w_full] f32 normalised crops -> outputs.  Enqueues ~300 kernels. */
int hm_hamer_forward(const hm_hamer_weights* w, const float* img, int B, const hm_hamer_outputs* out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- YOLOv7 detector path
 * Activations are NHWC 16-bit tensors addressed as (base pointer, pixel stride in elements): a
 * producer can write straight into a channel slice of a concat buffer (Concat, common.py:60-66,
 * costs nothing), a consumer can read a slice the same way. */
typedef struct hm_conv_args {
  const void* X;      /* [N][H][W_in][ldx] 16-bit, pointer already offset to the first input channel   */
  const void* W;      /* [Cout][Kpad] 16-bit, K order (ky, kx, ci), zero padded to Kpad (% 64 == 0)   */
  void* Y;            /* [N][Hout][Wout][ldy] 16-bit (or f32 when out_f32), offset to the channel slice */
  const float* bias;  /* [Cout]                                                                        */
  const void* zeros;  /* >= 16 zero bytes on the device: source of padding taps                        */
  int N, H, W_in, Cin, Cout, ksize, stride, ldx, ldy, Kpad;
  int act;            /* 1: SiLU (Conv.fuseforward, common.py:114); 2: ReLU (ResNet-34 of the RootNet backbone) */
  int out_f32;        /* 1: f32 output, no activation (detect head, yolo.py:151)                       */
  int dtype;
  const void* resid;  /* optional with act == 2: [N][Hout][Wout][ldr] 16-bit added before the ReLU (BasicBlock identity) */
  int ldr;
  /* round 3, optional: scratch for split-K (few output tiles, long K: the 12x20 / 24x40 maps of the YOLOv7 neck).  When given
   * (16-byte aligned, at least hm_conv_splitk_bytes(args) bytes), the library cuts K into 2 or 4 ranges by a rule that looks at
   * ONE image's output only (so a frame's result does not depend on the batch it rides in), writes fp32 partial slabs
   * [ranges][N*Hout*Wout][Cout] here and adds them, in order, in a second small kernel. */
  void* splitk_ws;
  size_t splitk_ws_bytes;
} hm_conv_args;

/* Conv2d(k in {1,3,5,7}, stride in {1,2}, pad k/2) + bias (+ SiLU / ReLU / residual add + ReLU) as an implicit GEMM on MFMA.
 * Cin must be a power of two >= 8 (the 3-channel image is stored with 8 channels). */
int hm_conv2d_nhwc(const hm_conv_args* args, void* stream);
/* bytes of splitk_ws this convolution would use (0: it is never split) */
size_t hm_conv_splitk_bytes(const hm_conv_args* args);
/* Two consecutive convolutions of which the second is the ONLY reader of the first's output: Conv 0 and Conv 1 of yolov7.yaml
 * (yolo.py Model.forward_once walks them one after the other; 3 -> 32, k3 s1, then 32 -> 64, k3 s2, both SiLU).  Where the fused
 * kernel applies (first: Cin 8 (3 real), ldx 8, Cout 32, k3 s1, SiLU; second: X == first.Y, ldx == first.ldy == 32, Cout 64, k3 s2,
 * SiLU, ldy % 8 == 0, Y 16-byte aligned; one dtype) both run as ONE launch and first->Y is NOT written (the intermediate stays in
 * LDS; the result is bit-identical to the two launches); anywhere else, or with HM_OPT_CONV_STEM_PAIR = 1, this is
 * hm_conv2d_nhwc(first) followed by hm_conv2d_nhwc(second). */
int hm_conv2d_stem_pair(const hm_conv_args* first, const hm_conv_args* second, void* stream);

/* nn.MaxPool2d(k, stride, pad) on NHWC 16-bit (MP common.py:34-40: k=2,s=2; SPPCSPC common.py:275:
 * k=5/9/13, s=1, pad k/2 -- the 9 and 13 windows are cascades of the 5 window). C % 8 == 0. */
int hm_maxpool_nhwc(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, int k, int stride, int pad,
                    int dtype, void* stream);

/* (B,3,H,W) f32 planes -> NHWC 16-bit with 8 channels (3 real, 5 zero): the convolution input layout, for crops that
 * hm_crop_batch produced in HaMeR's layout (RootNet patch, rootnet/Model_RGB.py:596-610). */
int hm_nchw3_to_nhwc8(const float* x, void* y, int B, int H, int W, int dtype, void* stream);
/* ResRootNet.forward_coord (rootnet/Model_RGB.py:282-292): global average pool of feat [B][HW][C] (16-bit), 1x1 conv to
 * one channel (w [C], bias), times k_value[b] -> depth [B] f32. */
int hm_gap_linear(const void* feat, int HW, int C, const float* w, float bias, const float* k_value, float* depth, int B,
                  int dtype, void* stream);

/* nn.Upsample(scale_factor=2, mode='nearest') on NHWC 16-bit (yolov7.yaml:78,:92). C % 8 == 0. */
int hm_upsample2x_nhwc(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, int dtype, void* stream);

/* letterbox (utils/datasets.py:999-1029, auto=True) + LoadImage.process_img (:137-141) + /255
 * (detector.py:121-125).  The plan is host-side scalar work; the resize follows cv2's 8-bit
 * INTER_LINEAR fixed-point algorithm (11-bit coefficients). */
typedef struct hm_letterbox_plan {
  int src_h, src_w;     /* frame size                                  */
  int new_w, new_h;     /* resized (unpadded) size                      */
  int top, left;        /* padding before the resized image             */
  int out_h, out_w;     /* network input size (multiples of the stride) */
  float gain, pad_x, pad_y; /* scale_coords inputs (general.py:323-331) */
} hm_letterbox_plan;
int hm_letterbox_plan_make(int H, int W, int new_shape, int stride, hm_letterbox_plan* plan);      /* host */
/* host: tab[0:new_w]=x0, [new_w:2new_w]=ax0, [2new_w:3new_w]=ax1, then y0, ay0, ay1 (3*new_h)     */
int hm_letterbox_tables(const hm_letterbox_plan* plan, int32_t* tab_host);
/* frame [H][W][3] u8 BGR -> x8 [out_h][out_w][8] 16-bit RGB/255 (+5 zero channels), and optionally
 * u8_chw [3][out_h][out_w] RGB (the reference's uint8 network input, for parity checks). */
int hm_letterbox(const uint8_t* frame, const hm_letterbox_plan* plan, const int32_t* tab_dev, void* x8, int dtype,
                 uint8_t* u8_chw, void* stream);
/* `nb` equally sized frames, frame i at frames + i * frame_stride_bytes, into x8 [nb][out_h][out_w][8] in one launch (round 3:
 * the folder drivers upload a chunk's frames as one tensor). */
int hm_letterbox_batch(const uint8_t* frames, size_t frame_stride_bytes, int nb, const hm_letterbox_plan* plan,
                       const int32_t* tab_dev, void* x8, int dtype, void* stream);

/* Detect decode (IDetect.fuseforward, yolo.py:148-184): raw [ny*nx][3*(5+nc)] f32 of one level ->
 * rows [row0 + a*ny*nx + y*nx + x][5+nc] of pred: sigmoid, xy = (2s-0.5+grid)*stride, wh = (2s)^2*anchor. */
int hm_yolo_decode(const float* raw, int ldraw, float* pred, int row0, int ny, int nx, int nc, float stride,
                   const float* anchors6_host, void* stream);
/* The same for `nb` images of one batched pass in one launch: image i's raw map starts ny*nx*ldraw floats after image i-1's,
 * its rows of pred `pred_rows_per_image` rows after (round 3: 3 launches per pass instead of 3 per frame). */
int hm_yolo_decode_batch(const float* raw, int ldraw, float* pred, int row0, int ny, int nx, int nc, float stride,
                         const float* anchors, int nb, size_t pred_rows_per_image, void* stream);

/* non_max_suppression (utils/general.py:611-703, best-class branch) + scale_coords/clip/round
 * (general.py:323-344, detector.py:142).  pred [n][5+nc] f32.  class_mask: bit c set = class c kept.
 * Workspace: hm_nms_workspace_bytes(n).  dets [max_det][6] f32 = x1,y1,x2,y2,conf,cls (score order),
 * count[0] = number of rows.  When plan != NULL boxes are mapped to frame pixels and rounded. */
size_t hm_nms_workspace_bytes(int n);
int hm_yolo_nms(const float* pred, int n, int nc, float conf_thres, float iou_thres, unsigned class_mask, int agnostic,
                int max_det, const hm_letterbox_plan* plan, float* dets, int* count, void* workspace,
                size_t workspace_bytes, void* stream);

/* One enqueue for a whole planned graph (Model.forward_once, yolo.py:609-639): the host planner
 * (hamer_yolo_amd/yolo/engine.py) turns the layer list into this op array once per input size. */
enum { HM_OP_CONV = 0, HM_OP_MAXPOOL = 1, HM_OP_UPSAMPLE2X = 2,
       HM_OP_CONV_PAIR = 3 /* this op's conv and the NEXT op's (kind HM_OP_CONV) through hm_conv2d_stem_pair; the next op is consumed */ };
typedef struct hm_yolo_op {
  int kind;
  int pool_pad;       /* HM_OP_MAXPOOL: padding; ksize/stride/N/H/W_in/Cin(=C)/X/Y/ldx/ldy/dtype come from conv */
  hm_conv_args conv;
} hm_yolo_op;
int hm_yolo_run(const hm_yolo_op* ops_host, int n_ops, void* stream);

/* Optional per-launch timing (HIP events on the launch stream); kinds below. */
enum { HM_K_GEMM = 0, HM_K_LAYERNORM = 1, HM_K_ATTENTION = 2, HM_K_IM2COL = 3, HM_K_LINEAR_F32 = 4,
       HM_K_CROSS_ATTN = 5, HM_K_MANO = 6, HM_K_CROP = 7, HM_K_CONV = 8, HM_K_OTHER = 9 };
typedef struct hm_prof_record { int kind, epilogue, M, N, K; float ms; } hm_prof_record;
int hm_prof_begin(int capacity);                        /* allocate 2*capacity events, start logging */
int hm_prof_collect(hm_prof_record* out_host, int cap); /* sync, copy records, clear; returns count   */
int hm_prof_end(void);                                  /* stop logging, destroy the events           */

int hm_version(void);
const char* hm_last_error_string(void);

#ifdef __cplusplus
}
#endif
#endif /* HAMER_HIP_H */
