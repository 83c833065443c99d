"""Thin tensor-level wrappers over the C ABI (one per exported kernel).

Tensors are PyTorch-ROCm tensors used as device buffers; every function checks shapes on the
host before handing raw pointers to the library, and raises on any library error.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import lib as L

_DT = {torch.bfloat16: L.HM_DTYPE_BF16, torch.float16: L.HM_DTYPE_F16}


def _dt(t: torch.Tensor) -> int:
    if t.dtype not in _DT:
        raise TypeError(f"16-bit operand expected (bfloat16/float16), got {t.dtype}")
    return _DT[t.dtype]


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.HipLibraryError("libhamer_hip kernels take device tensors; there is no CPU path")


def gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = L.HM_EPI_STORE,
         resid: Optional[torch.Tensor] = None, resid_mod: int = 0, out: Optional[torch.Tensor] = None,
         ln_gamma: Optional[torch.Tensor] = None, ln_xg: Optional[torch.Tensor] = None,
         ln_stats: Optional[torch.Tensor] = None, ln_colsum: Optional[torch.Tensor] = None, k_split: int = 0) -> torch.Tensor:
    """out = epilogue(x @ w.T + bias); x (M,K), w (N,K) 16-bit row-major.  Deferred LayerNorm: HM_EPI_RESID_LN also
    fills ln_xg (M,N) 16-bit = out * ln_gamma and ln_stats (N/64, M, 2); HM_EPI_LN_STORE / HM_EPI_LN_GELU read
    ln_stats (M, 2) = ln_finalize(partials) and ln_colsum (N,)."""
    _dev(x, w, bias, resid, out, ln_gamma, ln_xg, ln_stats, ln_colsum)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.stride(1) == 1 and w.stride(1) == 1 and x.dtype == w.dtype
    f32_out = epilogue in (L.HM_EPI_RESID_F32, L.HM_EPI_F32, L.HM_EPI_RESID_LN)
    if epilogue == L.HM_EPI_RESID_LN:
        assert ln_xg.shape == (M, N) and ln_xg.is_contiguous() and ln_xg.dtype == x.dtype
        assert ln_stats.shape == (N // 64, M, 2) and ln_stats.is_contiguous() and ln_gamma.shape == (N,)
    if epilogue in (L.HM_EPI_LN_STORE, L.HM_EPI_LN_GELU):
        assert ln_stats.shape == (M, 2) and ln_stats.is_contiguous() and ln_colsum.shape == (N,)
    if k_split > 1:                      # HM_EPI_F32 split-K: out is (k_split, M, N) partial products
        assert epilogue == L.HM_EPI_F32 and bias is None
        if out is None:
            out = torch.empty(k_split, M, N, device=x.device, dtype=torch.float32)
        assert out.shape == (k_split, M, N) and out.is_contiguous()
        a = L.GemmArgs(L.ptr(x), L.ptr(w), L.ptr(out), None, None, M, N, K, x.stride(0), w.stride(0), N, 0, 0, epilogue,
                       _dt(x), None, None, None, None, k_split)
        L.check(L.load().hm_gemm(C.byref(a), L.current_stream()), "hm_gemm")
        return out
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32 if f32_out else x.dtype)
    assert out.shape == (M, N) and out.stride(1) == 1
    a = L.GemmArgs(L.ptr(x), L.ptr(w), L.ptr(out), L.ptr(bias), L.ptr(resid), M, N, K, x.stride(0), w.stride(0),
                   out.stride(0), resid.stride(0) if resid is not None else 0, resid_mod, epilogue, _dt(x),
                   L.ptr(ln_gamma), L.ptr(ln_xg), L.ptr(ln_stats), L.ptr(ln_colsum), 0)
    L.check(L.load().hm_gemm(C.byref(a), L.current_stream()), "hm_gemm")
    return out


def layernorm_accum(x: torch.Tensor, partials: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor,
                    beta: torch.Tensor, eps: float, out_dtype=torch.bfloat16) -> torch.Tensor:
    """x (M,D) f32 += bias + partials.sum(0) in place (partials (S,M,D) from gemm(k_split=S)); returns LayerNorm(x)."""
    _dev(x, partials, bias, gamma, beta)
    M, D = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and partials.shape[1:] == (M, D) and partials.is_contiguous()
    out = torch.empty(M, D, device=x.device, dtype=out_dtype)
    code = L.HM_OUT_F32 if out_dtype == torch.float32 else _DT[out_dtype]
    L.check(L.load().hm_layernorm_accum(L.ptr(x), L.ptr(partials), partials.shape[0], L.ptr(bias), L.ptr(gamma), L.ptr(beta),
                                        L.ptr(out), code, M, D, eps, L.current_stream()), "hm_layernorm_accum")
    return out


def layernorm_mx8(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float):
    """LayerNorm -> MXFP8: (out8 (M,D) uint8 of e4m3 bytes, scales (D/32, M) uint8 E8M0)."""
    _dev(x, gamma, beta)
    M, D = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    out8 = torch.empty(M, D, device=x.device, dtype=torch.uint8)
    scales = torch.empty(D // 32, M, device=x.device, dtype=torch.uint8)
    L.check(L.load().hm_layernorm_mx8(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(out8), L.ptr(scales), M, D, eps,
                                      L.current_stream()), "hm_layernorm_mx8")
    return out8, scales


def gemm_fp8(x8: torch.Tensor, x_scales: torch.Tensor, w8: torch.Tensor, w_scale: torch.Tensor,
             bias: Optional[torch.Tensor] = None, epilogue: int = L.HM_EPI_STORE, resid: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None):
    """epilogue(dequant(x8, x_scales) @ (dequant(w8) * w_scale).T + bias).  x8 (M,K) / w8 (N,K) uint8 e4m3 bytes, x_scales
    (K/32, M) uint8 E8M0, w_scale (N,) f32.  HM_EPI_STORE -> bf16; HM_EPI_RESID_F32 -> f32 (+resid); HM_EPI_GELU_MX8 ->
    (uint8 (M,N), scales (N/32, M))."""
    _dev(x8, x_scales, w8, w_scale, bias, resid, out)
    M, K = x8.shape
    N = w8.shape[0]
    assert x8.dtype == torch.uint8 and w8.dtype == torch.uint8 and w8.shape[1] == K and x8.stride(1) == 1 and w8.stride(1) == 1
    assert x_scales.shape == (K // 32, M) and x_scales.is_contiguous() and w_scale.shape == (N,)
    out_scales = None
    if out is None:
        dt = {L.HM_EPI_STORE: torch.bfloat16, L.HM_EPI_RESID_F32: torch.float32, L.HM_EPI_GELU_MX8: torch.uint8}[epilogue]
        out = torch.empty(M, N, device=x8.device, dtype=dt)
    if epilogue == L.HM_EPI_GELU_MX8:
        out_scales = torch.empty(N // 32, M, device=x8.device, dtype=torch.uint8)
    a = L.GemmFp8Args(L.ptr(x8), L.ptr(x_scales), L.ptr(w8), L.ptr(w_scale), L.ptr(out), L.ptr(bias), L.ptr(resid),
                      L.ptr(out_scales), M, N, K, x8.stride(0), w8.stride(0), out.stride(0),
                      resid.stride(0) if resid is not None else 0, epilogue, L.HM_DTYPE_BF16)
    L.check(L.load().hm_gemm_fp8(C.byref(a), L.current_stream()), "hm_gemm_fp8")
    return (out, out_scales) if epilogue == L.HM_EPI_GELU_MX8 else out


def ln_finalize(partials: torch.Tensor, eps: float) -> torch.Tensor:
    """(D/64, M, 2) partial (sum, sum of squares) -> (M, 2) (mean, rstd)."""
    _dev(partials)
    P, M, _ = partials.shape
    out = torch.empty(M, 2, device=partials.device, dtype=torch.float32)
    L.check(L.load().hm_ln_finalize(L.ptr(partials), L.ptr(out), M, P * 64, eps, L.current_stream()), "hm_ln_finalize")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out_dtype=torch.bfloat16) -> torch.Tensor:
    _dev(x, gamma, beta)
    M, D = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(M, D, device=x.device, dtype=out_dtype)
    code = L.HM_OUT_F32 if out_dtype == torch.float32 else _DT[out_dtype]
    L.check(L.load().hm_layernorm(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(out), code, M, D, eps, L.current_stream()),
            "hm_layernorm")
    return out


def vit_attention(qkv: torch.Tensor, B: int, tokens: int, heads: int, head_dim: int, scale: float) -> torch.Tensor:
    _dev(qkv)
    assert qkv.shape == (B * tokens, 3 * heads * head_dim) and qkv.is_contiguous()
    out = torch.empty(B * tokens, heads * head_dim, device=qkv.device, dtype=qkv.dtype)
    L.check(L.load().hm_vit_attention(L.ptr(qkv), L.ptr(out), B, tokens, heads, head_dim, scale, _dt(qkv),
                                      L.current_stream()), "hm_vit_attention")
    return out


def vit_attention_mx8(qkv: torch.Tensor, B: int, tokens: int, heads: int, head_dim: int, scale: float):
    """Attention with MXFP8 output: (out8 (B*tokens, heads*96) uint8, scales (heads*3, B*tokens) uint8)."""
    _dev(qkv)
    assert qkv.shape == (B * tokens, 3 * heads * head_dim) and qkv.is_contiguous() and qkv.dtype == torch.bfloat16
    out8 = torch.empty(B * tokens, heads * 96, device=qkv.device, dtype=torch.uint8)
    scales = torch.empty(heads * 3, B * tokens, device=qkv.device, dtype=torch.uint8)
    L.check(L.load().hm_vit_attention_mx8(L.ptr(qkv), L.ptr(out8), L.ptr(scales), B, tokens, heads, head_dim, scale,
                                          L.current_stream()), "hm_vit_attention_mx8")
    return out8, scales


def tome_attention(qkv: torch.Tensor, size: Optional[torch.Tensor], B: int, tokens: int, heads: int, head_dim: int,
                   scale: float) -> torch.Tensor:
    """ToMeAttention core: softmax(scale q k^T + log(size)) v for any tokens <= 192; size (B*tokens,) f32 or None."""
    _dev(qkv, size)
    assert qkv.shape == (B * tokens, 3 * heads * head_dim) and qkv.is_contiguous()
    assert size is None or (size.shape == (B * tokens,) and size.dtype == torch.float32 and size.is_contiguous())
    out = torch.empty(B * tokens, heads * head_dim, device=qkv.device, dtype=qkv.dtype)
    L.check(L.load().hm_tome_attention(L.ptr(qkv), L.ptr(size), L.ptr(out), B, tokens, heads, head_dim, scale, _dt(qkv),
                                       L.current_stream()), "hm_tome_attention")
    return out


def tome_merge(qkv: torch.Tensor, x: torch.Tensor, size: Optional[torch.Tensor], B: int, tokens: int, r: int, heads: int,
               head_dim: int):
    """Matching on the head-averaged keys of `qkv` + size-weighted merge of the fp32 stream x (B*tokens, D):
    returns (x_out (B*(tokens-r), D), size_out (B*(tokens-r),), index (B, 3, 96) int32 = unm | src | dst)."""
    _dev(qkv, x, size)
    D = x.shape[1]
    assert x.shape[0] == B * tokens and x.dtype == torch.float32 and x.is_contiguous() and qkv.is_contiguous()
    xo = torch.empty(B * (tokens - r), D, device=x.device, dtype=torch.float32)
    so = torch.empty(B * (tokens - r), device=x.device, dtype=torch.float32)
    metric = torch.empty(B * tokens * head_dim, device=x.device, dtype=torch.float32)
    index = torch.zeros(L.load().hm_tome_index_bytes(B) // 4, device=x.device, dtype=torch.int32)
    L.check(L.load().hm_tome_merge(L.ptr(qkv), L.ptr(x), L.ptr(size), L.ptr(xo), L.ptr(so), L.ptr(metric), L.ptr(index), B, tokens, r,
                                   heads, head_dim, D, _dt(qkv), L.current_stream()), "hm_tome_merge")
    return xo, so, index.view(B, 3, -1), metric.view(B, tokens, head_dim)


def patch_im2col(img: torch.Tensor, x0: int, win_w: int, patch: int, pad: int, dtype=torch.bfloat16) -> torch.Tensor:
    _dev(img)
    B, Cc, H, Wf = img.shape
    assert Cc == 3 and img.dtype == torch.float32 and img.is_contiguous()
    gh, gw = (H + 2 * pad - patch) // patch + 1, (win_w + 2 * pad - patch) // patch + 1
    out = torch.empty(B * gh * gw, 3 * patch * patch, device=img.device, dtype=dtype)
    L.check(L.load().hm_patch_im2col(L.ptr(img), L.ptr(out), B, H, Wf, x0, win_w, patch, pad, _DT[dtype],
                                     L.current_stream()), "hm_patch_im2col")
    return out


def linear_f32(x: torch.Tensor, w: torch.Tensor, bias=None, resid=None, act: int = 0) -> torch.Tensor:
    _dev(x, w, bias, resid)
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    L.check(L.load().hm_linear_f32(L.ptr(x), x.stride(0), L.ptr(w), w.stride(0), L.ptr(bias), L.ptr(resid),
                                   resid.stride(0) if resid is not None else 0, L.ptr(out), N, M, N, K, act,
                                   L.current_stream()), "hm_linear_f32")
    return out


def cross_attention(q: torch.Tensor, kv: torch.Tensor, k_off: int, v_off: int, B: int, tokens: int, heads: int,
                    dim_head: int, scale: float) -> torch.Tensor:
    _dev(q, kv)
    out = torch.empty(B, heads * dim_head, device=q.device, dtype=torch.float32)
    L.check(L.load().hm_cross_attention(L.ptr(q), L.ptr(kv), kv.stride(0), k_off, v_off, L.ptr(out), B, tokens, heads,
                                        dim_head, scale, _dt(kv), L.current_stream()), "hm_cross_attention")
    return out


def mano_model_struct(mp: dict) -> L.ManoModel:
    return L.ManoModel(L.ptr(mp["v_template"]), L.ptr(mp["shapedirs"]), L.ptr(mp["posedirs"]), L.ptr(mp["J_regressor"]),
                       L.ptr(mp["lbs_weights"]), int(mp["v_template"].shape[0]))


def mano_forward(mp: dict, pose6d: torch.Tensor, betas: torch.Tensor, cam: torch.Tensor, focal: float = 5000.0,
                 image_size: float = 256.0) -> dict:
    """mp: device fp32 tensors keyed like synth.mano_params()."""
    _dev(pose6d, betas, cam, *[mp[k] for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights")])
    B, V, dev = pose6d.shape[0], mp["v_template"].shape[0], pose6d.device
    o = {
        "rotmats": torch.empty(B, 16, 3, 3, device=dev), "verts": torch.empty(B, V, 3, device=dev),
        "joints": torch.empty(B, 21, 3, device=dev), "cam_t": torch.empty(B, 3, device=dev),
        "kp2d": torch.empty(B, 21, 2, device=dev),
    }
    m = mano_model_struct(mp)
    L.check(L.load().hm_mano_forward(C.byref(m), L.ptr(pose6d), L.ptr(betas), L.ptr(cam), L.ptr(o["rotmats"]),
                                     L.ptr(o["verts"]), L.ptr(o["joints"]), L.ptr(o["cam_t"]), L.ptr(o["kp2d"]), B,
                                     focal, image_size, L.current_stream()), "hm_mano_forward")
    return o


def crop_boxes(boxes, P: int = 256) -> torch.Tensor:
    """boxes: iterable of (cx, cy, size, flip) -> uint8 host tensor holding hm_crop_box records."""
    lib = L.load()
    arr = (L.CropBox * len(boxes))()
    for i, (cx, cy, size, flip) in enumerate(boxes):
        L.check(lib.hm_crop_box_from_bbox(float(cx), float(cy), float(size), int(bool(flip)), P, C.byref(arr[i])),
                "hm_crop_box_from_bbox")
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()


def crop_batch(frame: torch.Tensor, boxes_rec: torch.Tensor, mean, std, P: int = 256, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """frame (H,W,3) uint8 BGR on device; boxes_rec: device uint8 tensor of hm_crop_box records.  ``out``: a contiguous
    (B,3,P,P) fp32 slice to fill (all hands of several frames land in one batch tensor)."""
    _dev(frame, boxes_rec, out)
    H, W, _ = frame.shape
    B = boxes_rec.numel() // C.sizeof(L.CropBox)
    if out is None:
        out = torch.empty(B, 3, P, P, device=frame.device, dtype=torch.float32)
    assert out.shape == (B, 3, P, P) and out.dtype == torch.float32 and out.is_contiguous() and frame.is_contiguous()
    m3 = (C.c_float * 3)(*[float(v) for v in mean])
    s3 = (C.c_float * 3)(*[float(v) for v in std])
    L.check(L.load().hm_crop_batch(L.ptr(frame), H, W, L.ptr(boxes_rec), L.ptr(out), B, P, m3, s3, L.current_stream()),
            "hm_crop_batch")
    return out
