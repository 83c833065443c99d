"""HamerEngine: device-resident HaMeR weights + one-call forward through libhamer_hip.

Operand type.  The checkpoint holds fp32 weights (models/__init__.py:46) and the reference computes in fp32; the 1e-3
parity bar is against that.  Rounding the weights to bf16 (8 significant bits) alone moves pose / shape by 1.3-1.9e-3
on the ViT-H geometry, rounding to fp16 (11 bits) by 1.7e-4 (tools/precision_study.py), at the same
MFMA rate -- so the default operand type is fp16; ``dtype=torch.bfloat16`` stays selectable for checkpoints whose
activations need the wider exponent.

Load time (host, once): GEMM matrices are converted to the 16-bit operand type and laid out
[N][K] as nn.Linear stores them; the positional embedding is folded to ``pos[1:] + pos[0]``
(vit.py:327); the six decoder ``to_kv`` matrices are stacked into one [6*1024][1280] GEMM
operand; the V rows of the decoder's ``to_qkv`` are sliced out (one token: softmax == 1);
the three read-out heads are stacked into one [112][1024] matrix with ``bias + init_*``.
Run time: one ``hm_hamer_forward`` enqueue on the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import lib as L
from .ops import mano_model_struct
from .synth import HamerConfig


def parse_tome_r(num_layers: int, r):
    """Tokens to merge away per block from a constant, an (r, inflection) pair or a list -- the schedule forms of the
    reference's ToMe patch (selective_vit_adapter.py:132-157); (8, -1) is what HAMER_INFER sets."""
    if isinstance(r, (list,)):
        return [int(v) for v in r] + [0] * max(0, num_layers - len(r))
    inflect = 0.0
    if isinstance(r, tuple):
        r, inflect = r
    lo = int(r * (1.0 - inflect))
    hi = 2 * r - lo
    step = (hi - lo) / (num_layers - 1)
    return [int(lo + step * i) for i in range(num_layers)]


RANGE_LIMIT = 4096.0      # prescale target: calibrated magnitudes end up at or below this, 16x under fp16's 65504


def prescale_from_ranges(ranges, depth: int, dec_depth: int, limit: float = RANGE_LIMIT) -> Dict:
    """Power-of-two exponents per 16-bit activation class from the magnitudes hm_hamer_forward recorded (range_stats layout:
    per block [LN1 out, q, k, v, LN2 out, GELU out], last_norm out, per decoder layer [k, v]): a = max(0, ceil(log2(m / limit))),
    i.e. the class is stored as value * 2^-a.  Non-finite magnitudes (the probe itself overflowed) are an error."""
    import math
    r = [float(v) for v in ranges]
    if len(r) != 6 * depth + 1 + 2 * dec_depth or not all(math.isfinite(v) for v in r):
        raise L.HipLibraryError("prescale_from_ranges: bad or non-finite range statistics (measure them with bf16 operands)")
    ex = [max(0, math.ceil(math.log2(v / limit))) if v > limit else 0 for v in r]
    keys = ("ln1", "q", "k", "v", "ln2", "gelu")
    return {"blocks": [dict(zip(keys, ex[6 * i:6 * i + 6])) for i in range(depth)], "last": ex[6 * depth],
            "dec": [(ex[6 * depth + 1 + 2 * i], ex[6 * depth + 2 + 2 * i]) for i in range(dec_depth)]}


def prescale_is_identity(pre: Optional[Dict]) -> bool:
    return pre is None or (all(v == 0 for b in pre["blocks"] for v in b.values()) and pre["last"] == 0
                           and all(a == 0 and b == 0 for a, b in pre["dec"]))


class ForwardContext:
    """One batch in flight: its own HIP stream, workspace and output tensors (HamerEngine.contexts)."""

    def __init__(self, stream, workspace, out):
        self.stream, self.workspace, self.out = stream, workspace, out


class HamerEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], mano: Dict[str, torch.Tensor],
                 cfg: Optional[HamerConfig] = None, device="cuda", dtype=torch.float16, fold_ln: Optional[bool] = None,
                 fp8: Optional[bool] = None, token_merge=None, prescale: Optional[Dict] = None):
        """``prescale`` (round 4, `prescale_from_ranges`): per activation class a power of two folded into the weights at load --
        LayerNorm gamma / beta against the columns of the matrix that reads its output, q / k / v rows of qkv against the
        softmax scale and proj, the GELU output (an epilogue factor) against fc2, to_kv rows against the cross-attention scale
        and its to_out -- so that every 16-bit activation of a checkpoint that overflows fp16 lands inside it while the operands
        keep their 11 significant bits.  Exact arithmetic: powers of two commute with rounding (no value underflows on the way),
        so the forward computes the same function.  ``want_tokens`` then returns last_norm's output times 2^-prescale['last']."""
        if not torch.cuda.is_available():
            raise L.HipLibraryError("HamerEngine needs an MI355X (HIP device); there is no CPU fallback")
        self.lib = L.load()
        self.cfg = cfg or HamerConfig()
        self.device = torch.device(device)
        self.dtype = dtype
        self._keep = []          # device tensors referenced by raw pointers
        self._ws = None
        self._ws_B = 0
        v, d = self.cfg.vit, self.cfg.dec
        sd = state_dict

        def f32(t):
            t = t.detach().to(self.device, torch.float32).contiguous()
            self._keep.append(t)
            return t

        def w16(t):
            t = t.detach().to(self.device, torch.float32).to(dtype).contiguous()
            self._keep.append(t)
            return t

        # deferred LayerNorm (hm_gemm HM_EPI_RESID_LN / HM_EPI_LN_*): LN1/LN2 of every block are folded into the
        # neighbouring GEMMs; needs colsum = W16 . gamma and bias + W16 . beta of the 16-bit weights actually used
        if fold_ln is None:
            fold_ln = os.environ.get("HAMER_FOLD_LN", "0") == "1"    # measured equal to the LayerNorm kernel at B=64: off
        self.fold_ln = bool(fold_ln) and v.embed_dim % 64 == 0

        def ln_fold(W, bias, gamma, beta):
            Wd = W.detach().to(self.device, torch.float32).to(dtype).to(torch.float64)
            g64, b64 = gamma.detach().to(self.device, torch.float64), beta.detach().to(self.device, torch.float64)
            return f32(Wd @ g64), f32(bias.detach().to(self.device, torch.float64) + Wd @ b64)

        # fp8 path (BASELINE configs[4]): qkv / fc1 / fc2 weights in e4m3 with per-channel scales, MXFP8 activations
        if fp8 is None:
            fp8 = os.environ.get("HAMER_FP8", "0") == "1"
        self.fp8 = bool(fp8)
        if self.fp8:
            if v.embed_dim % 128 != 0:
                raise L.HipLibraryError("the fp8 path needs embed_dim % 128 == 0")
            dtype = self.dtype = torch.bfloat16       # the 16-bit side of the fp8 configuration (proj operand, kv) is bf16
            self.fold_ln = False
            from .quant import quantize_weight_e4m3

        def w8(t):
            q, sc = quantize_weight_e4m3(t.detach().to(self.device))
            self._keep += [q, sc]
            return L.ptr(q), L.ptr(sc)

        pre = None if prescale_is_identity(prescale) else prescale
        if pre is not None and (self.fp8 or self.fold_ln or token_merge):
            raise L.HipLibraryError("the range prescale exists on the dense 16-bit path only (not fp8, deferred LayerNorm or token merging)")
        self.prescale = pre
        limit16 = 65504.0 if dtype == torch.float16 else 3.0e38

        def scaled(t, e):
            """t * 2^e in fp32 (exact), refusing a matrix that would leave the operand type's range."""
            t = t.detach().to(self.device, torch.float32)
            if e == 0:
                return t
            t = t * (2.0 ** e)
            if float(t.abs().max()) >= limit16:
                raise L.HipLibraryError("range prescale: a compensated weight matrix leaves the fp16 range")
            return t

        D = v.embed_dim
        self.blocks = (L.VitBlock * v.depth)()
        for i in range(v.depth):
            p = f"backbone.blocks.{i}."
            b = self.blocks[i]
            if pre is None:
                b.ln1_g, b.ln1_b = L.ptr(f32(sd[p + "norm1.weight"])), L.ptr(f32(sd[p + "norm1.bias"]))
                b.ln2_g, b.ln2_b = L.ptr(f32(sd[p + "norm2.weight"])), L.ptr(f32(sd[p + "norm2.bias"]))
                b.qkv_w, b.qkv_b = L.ptr(w16(sd[p + "attn.qkv.weight"])), L.ptr(f32(sd[p + "attn.qkv.bias"]))
                b.proj_w, b.proj_b = L.ptr(w16(sd[p + "attn.proj.weight"])), L.ptr(f32(sd[p + "attn.proj.bias"]))
                b.fc1_w, b.fc1_b = L.ptr(w16(sd[p + "mlp.fc1.weight"])), L.ptr(f32(sd[p + "mlp.fc1.bias"]))
                b.fc2_w, b.fc2_b = L.ptr(w16(sd[p + "mlp.fc2.weight"])), L.ptr(f32(sd[p + "mlp.fc2.bias"]))
            else:
                e = pre["blocks"][i]
                b.ln1_g, b.ln1_b = L.ptr(f32(scaled(sd[p + "norm1.weight"], -e["ln1"]))), L.ptr(f32(scaled(sd[p + "norm1.bias"], -e["ln1"])))
                b.ln2_g, b.ln2_b = L.ptr(f32(scaled(sd[p + "norm2.weight"], -e["ln2"]))), L.ptr(f32(scaled(sd[p + "norm2.bias"], -e["ln2"])))
                wq, bq = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
                b.qkv_w = L.ptr(w16(torch.cat([scaled(wq[j * D:(j + 1) * D], e["ln1"] - e[c]) for j, c in enumerate("qkv")], 0)))
                b.qkv_b = L.ptr(f32(torch.cat([scaled(bq[j * D:(j + 1) * D], -e[c]) for j, c in enumerate("qkv")], 0)))
                b.attn_scale_mul = 2.0 ** (e["q"] + e["k"])
                b.proj_w, b.proj_b = L.ptr(w16(scaled(sd[p + "attn.proj.weight"], e["v"]))), L.ptr(f32(sd[p + "attn.proj.bias"]))
                b.fc1_w, b.fc1_b = L.ptr(w16(scaled(sd[p + "mlp.fc1.weight"], e["ln2"]))), L.ptr(f32(sd[p + "mlp.fc1.bias"]))
                b.gelu_out_scale = 2.0 ** -e["gelu"]
                b.fc2_w, b.fc2_b = L.ptr(w16(scaled(sd[p + "mlp.fc2.weight"], e["gelu"]))), L.ptr(f32(sd[p + "mlp.fc2.bias"]))
            if self.fp8:
                b.qkv_w8, b.qkv_ws = w8(sd[p + "attn.qkv.weight"])
                b.fc1_w8, b.fc1_ws = w8(sd[p + "mlp.fc1.weight"])
                b.fc2_w8, b.fc2_ws = w8(sd[p + "mlp.fc2.weight"])
                if v.head_dim == 80:                 # proj too: K reordered to 96 columns per head (hm_vit_attention_mx8)
                    wp = sd[p + "attn.proj.weight"].detach().to(self.device, torch.float32)
                    wpad = torch.zeros(D, v.heads, 96, device=self.device)
                    wpad[:, :, :80] = wp.reshape(D, v.heads, 80)
                    b.proj_w8, b.proj_ws = w8(wpad.reshape(D, v.heads * 96))
            if token_merge and v.head_dim == 80 and not self.fp8:
                # matching metric of ToMe in fp32 (hm_vit_block.kmean_w): mean over heads of the key rows and biases
                wk = sd[p + "attn.qkv.weight"].detach().to(self.device, torch.float64)[D:2 * D].reshape(v.heads, v.head_dim, D).mean(0)
                bk = sd[p + "attn.qkv.bias"].detach().to(self.device, torch.float64)[D:2 * D].reshape(v.heads, v.head_dim).mean(0)
                b.kmean_w, b.kmean_b = L.ptr(f32(wk.float())), L.ptr(f32(bk.float()))
            if self.fold_ln:
                cs, bl = ln_fold(sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"], sd[p + "norm1.weight"], sd[p + "norm1.bias"])
                b.qkv_colsum, b.qkv_bias_ln = L.ptr(cs), L.ptr(bl)
                cs, bl = ln_fold(sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], sd[p + "norm2.weight"], sd[p + "norm2.bias"])
                b.fc1_colsum, b.fc1_bias_ln = L.ptr(cs), L.ptr(bl)
        pos = sd["backbone.pos_embed"].to(torch.float32)
        pos = pos[0, 1:] + pos[0, :1]

        t = "mano_head.transformer."
        inner = d.inner
        self.layers = (L.DecLayer * d.depth)()
        kv_rows = []
        for i in range(d.depth):
            p = f"{t}transformer.layers.{i}."
            l = self.layers[i]
            for j in range(3):
                setattr(l, f"ln{j}_g", L.ptr(f32(sd[p + f"{j}.norm.weight"])))
                setattr(l, f"ln{j}_b", L.ptr(f32(sd[p + f"{j}.norm.bias"])))
            l.sa_v_w = L.ptr(f32(sd[p + "0.fn.to_qkv.weight"][2 * inner:3 * inner]))
            l.sa_out_w, l.sa_out_b = L.ptr(f32(sd[p + "0.fn.to_out.0.weight"])), L.ptr(f32(sd[p + "0.fn.to_out.0.bias"]))
            # one token: softmax == 1, so self-attention is to_out(to_v(.)) -- folded into one [dim][dim] matrix in fp64
            l.sa_w = L.ptr(f32((sd[p + "0.fn.to_out.0.weight"].to(self.device, torch.float64)
                                @ sd[p + "0.fn.to_qkv.weight"][2 * inner:3 * inner].to(self.device, torch.float64)).float()))
            l.ca_q_w = L.ptr(f32(sd[p + "1.fn.to_q.weight"]))
            ekk, ekv = pre["dec"][i] if pre is not None else (0, 0)
            l.ca_out_w, l.ca_out_b = L.ptr(f32(scaled(sd[p + "1.fn.to_out.0.weight"], ekv))), L.ptr(f32(sd[p + "1.fn.to_out.0.bias"]))
            if pre is not None:
                l.ca_scale_mul = 2.0 ** ekk
            l.ff1_w, l.ff1_b = L.ptr(f32(sd[p + "2.fn.net.0.weight"])), L.ptr(f32(sd[p + "2.fn.net.0.bias"]))
            l.ff2_w, l.ff2_b = L.ptr(f32(sd[p + "2.fn.net.3.weight"])), L.ptr(f32(sd[p + "2.fn.net.3.bias"]))
            wkv = sd[p + "1.fn.to_kv.weight"]
            if pre is None:
                kv_rows.append(wkv.to(torch.float32))
            else:             # [k rows | v rows] of this layer (pose_transformer.py:114-115 chunk(2)), against last_norm's prescale
                kv_rows.append(torch.cat([scaled(wkv[:inner], pre["last"] - ekk), scaled(wkv[inner:], pre["last"] - ekv)], 0))
        token0 = sd[t + "to_token_embedding.bias"].to(torch.float32) + sd[t + "pos_embedding"].to(torch.float32)[0, 0]
        head_w = torch.zeros(112, d.dim, dtype=torch.float32, device=sd["mano_head.decpose.weight"].device)
        head_b = torch.zeros(112, dtype=torch.float32, device=head_w.device)
        head_w[:96], head_w[96:106], head_w[106:109] = sd["mano_head.decpose.weight"], sd["mano_head.decshape.weight"], sd["mano_head.deccam.weight"]
        head_b[:96] = sd["mano_head.decpose.bias"] + sd["mano_head.init_hand_pose"][0]
        head_b[96:106] = sd["mano_head.decshape.bias"] + sd["mano_head.init_betas"][0]
        head_b[106:109] = sd["mano_head.deccam.bias"] + sd["mano_head.init_cam"][0]

        self.mano = {k: f32(mano[k]) for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights")}
        self.faces = mano.get("faces")

        w = L.HamerWeights()
        w.img_h, w.img_w_full = v.img_h, self.cfg.image_size
        w.win_x0, w.win_w = (self.cfg.image_size - v.img_w) // 2, v.img_w      # hamer.py:119 x[:,:,:,32:-32]
        w.patch, w.pad, w.embed_dim, w.depth, w.heads, w.mlp_dim = v.patch, v.pad, D, v.depth, v.heads, D * v.mlp_ratio
        w.vit_eps = v.ln_eps
        w.patch_w = L.ptr(w16(sd["backbone.patch_embed.proj.weight"].reshape(D, -1)))
        w.patch_b = L.ptr(f32(sd["backbone.patch_embed.proj.bias"]))
        w.pos = L.ptr(f32(pos))
        w.blocks = C.cast(self.blocks, C.POINTER(L.VitBlock))
        elast = pre["last"] if pre is not None else 0
        w.last_g, w.last_b = L.ptr(f32(scaled(sd["backbone.last_norm.weight"], -elast))), L.ptr(f32(scaled(sd["backbone.last_norm.bias"], -elast)))
        w.dec_dim, w.dec_depth, w.dec_heads, w.dec_dim_head, w.dec_mlp = d.dim, d.depth, d.heads, d.dim_head, d.mlp_dim
        w.dec_eps = d.ln_eps
        w.token0 = L.ptr(f32(token0))
        w.kv_w = L.ptr(w16(torch.cat(kv_rows, dim=0)))
        w.layers = C.cast(self.layers, C.POINTER(L.DecLayer))
        w.head_w, w.head_b = L.ptr(f32(head_w)), L.ptr(f32(head_b))
        w.mano = mano_model_struct(self.mano)
        w.focal_length, w.image_size = float(self.cfg.focal_length), float(self.cfg.image_size)
        w.dtype = L.HM_DTYPE_BF16 if dtype == torch.bfloat16 else L.HM_DTYPE_F16
        # token merging (HAMER_INFER(token_merge=True), hamer.py:481-483): True = the reference's schedule r = (8, -1);
        # an int, an (r, inflection) pair or a per-block list are parsed as selective_vit_adapter.py:132-157 does
        self.tome_r = None
        self.ctx_tokens = v.tokens
        if token_merge:
            if self.fp8:
                raise L.HipLibraryError("token merging runs on the 16-bit path")
            self.tome_r = parse_tome_r(v.depth, (8, -1) if token_merge is True else token_merge)
            self._tome_arr = (C.c_int * v.depth)(*self.tome_r)
            w.tome_r = C.cast(self._tome_arr, C.POINTER(C.c_int))
            self.fold_ln = False
            for ri in self.tome_r:
                self.ctx_tokens -= min(ri, self.ctx_tokens // 2)
        self.w = w
        self.tokens = v.tokens
        self.n_verts = int(self.mano["v_template"].shape[0])

    # ------------------------------------------------------------------ load-time check
    def calibration_is_finite(self, n: int = 4) -> bool:
        """One forward over `n` seeded crops (uniform u8 noise, ImageNet-normalised -- the full input range), checking that the
        backbone tokens and every output are finite.  With fp16 operands the 16-bit intermediates (LayerNorm output, q / k / v,
        the GELU output, to_kv) saturate at 65504 and an overflow turns into NaN poses without any other sign; the synthetic
        weights stay far below that, a trained checkpoint is checked with this at load (HAMER.to)."""
        from . import synth
        img = synth.normalize_crops(synth.crops_u8(n, seed0=0)).to(self.device)
        out = self.forward(img, want_tokens=self.tome_r is None)
        torch.cuda.synchronize(self.device)
        return all(bool(torch.isfinite(v.float()).all()) for v in out.values())

    def measure_ranges(self, n: int = 4) -> torch.Tensor:
        """Largest magnitude of every 16-bit activation class over one forward of `n` calibration crops (hm_hamer_weights.
        range_stats; layout in include/hamer_hip.h).  Meant for an engine with bfloat16 operands, whose exponent range cannot
        overflow: the result feeds prescale_from_ranges for the fp16 engine of the same checkpoint."""
        from . import synth
        v, d = self.cfg.vit, self.cfg.dec
        stats = torch.zeros(6 * v.depth + 1 + 2 * d.depth, dtype=torch.float32, device=self.device)
        img = synth.normalize_crops(synth.crops_u8(n, seed0=0)).to(self.device)
        self.w.range_stats = L.ptr(stats)
        try:
            self.forward(img)
            torch.cuda.synchronize(self.device)
        finally:
            self.w.range_stats = None
        return stats.cpu()

    # ------------------------------------------------------------------ run
    def workspace(self, B: int) -> torch.Tensor:
        if self._ws is None or self._ws_B < B:
            n = self.lib.hm_hamer_workspace_bytes(C.byref(self.w), B)
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws_B = B
        return self._ws

    def contexts(self, B: int, n: int = 2, want_tokens: bool = False):
        """n independent (stream, workspace, outputs) triples for keeping n batches in flight: consecutive forwards issued
        on alternating contexts overlap -- the HBM-bound phases of one batch (LayerNorm, attention, GEMM epilogues, decoder)
        run under the MFMA phases of the other (measured: +6 % hands/s at B=64 with 2, x1.85 at B=16 with 3)."""
        nbytes = self.lib.hm_hamer_workspace_bytes(C.byref(self.w), B)
        # the n streams come from the package's one shared set per device (streams.py: a second set of streams in the same
        # process may share a hardware queue and then does not overlap at all)
        from .streams import get_streams
        return [ForwardContext(st, torch.empty(nbytes, dtype=torch.uint8, device=self.device), self.alloc_outputs(B, want_tokens))
                for st in get_streams(self.device, n)]

    def forward_on(self, ctx: ForwardContext, img: torch.Tensor, want_tokens: bool = False) -> Dict[str, torch.Tensor]:
        """forward() enqueued on ctx.stream (after everything already queued on the caller's current stream, so `img` is
        ready); read ctx.out after ctx.stream.synchronize() or from a stream that waited for it."""
        ctx.stream.wait_stream(torch.cuda.current_stream(self.device))
        img.record_stream(ctx.stream)          # the caching allocator must not hand `img`'s memory out while ctx.stream still reads it
        with torch.cuda.stream(ctx.stream):
            return self.forward(img, ctx.out, want_tokens=want_tokens, workspace=ctx.workspace)

    def alloc_outputs(self, B: int, want_tokens: bool = False) -> Dict[str, torch.Tensor]:
        dev, V = self.device, self.n_verts
        o = {
            "pose6d": torch.empty(B, 96, device=dev), "betas": torch.empty(B, 10, device=dev),
            "pred_cam": torch.empty(B, 3, device=dev), "rotmats": torch.empty(B, 16, 3, 3, device=dev),
            "pred_vertices": torch.empty(B, V, 3, device=dev), "pred_keypoints_3d": torch.empty(B, 21, 3, device=dev),
            "pred_cam_t": torch.empty(B, 3, device=dev), "pred_keypoints_2d": torch.empty(B, 21, 2, device=dev),
        }
        if want_tokens:
            o["tokens"] = torch.empty(B * self.tokens, self.cfg.vit.embed_dim, device=dev, dtype=self.dtype)
        return o

    def forward(self, img: torch.Tensor, out: Optional[Dict[str, torch.Tensor]] = None,
                want_tokens: bool = False, workspace: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """img: (B, 3, 256, 256) fp32 normalised crops on the device; one hm_hamer_forward enqueue on the current stream."""
        if not img.is_cuda:
            raise L.HipLibraryError("HamerEngine.forward takes a device tensor")
        B = img.shape[0]
        assert img.shape[1:] == (3, self.cfg.vit.img_h, self.cfg.image_size) and img.dtype == torch.float32
        img = img.contiguous()
        if out is None:
            out = self.alloc_outputs(B, want_tokens)
        ws = workspace if workspace is not None else self.workspace(B)     # own workspace per in-flight batch when forwards overlap
        ho = L.HamerOutputs(L.ptr(out["pose6d"]), L.ptr(out["betas"]), L.ptr(out["pred_cam"]), L.ptr(out["rotmats"]),
                            L.ptr(out["pred_vertices"]), L.ptr(out["pred_keypoints_3d"]), L.ptr(out["pred_cam_t"]),
                            L.ptr(out["pred_keypoints_2d"]), L.ptr(out.get("tokens")))
        L.check(self.lib.hm_hamer_forward(C.byref(self.w), L.ptr(img), B, C.byref(ho), L.ptr(ws), ws.numel(),
                                          L.current_stream()), "hm_hamer_forward")
        return out
