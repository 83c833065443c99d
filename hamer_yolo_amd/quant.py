"""Load-time e4m3 quantisation of the ViT GEMM weights for the fp8 path (hm_gemm_fp8; BASELINE configs[4]).

OCP e4m3fn (torch.float8_e4m3fn, round-to-nearest-even), one fp32 scale per output channel:
``scale[n] = max|W[n]| / 448`` so the largest element of every row maps to the largest e4m3 value.
Activations are quantised on the GPU by the kernels that produce them (hm_layernorm_mx8, HM_EPI_GELU_MX8).
"""
import torch


def quantize_weight_e4m3(w: torch.Tensor):
    """w (N, K) -> (uint8 (N, K) of e4m3 bytes, fp32 (N,) scales)."""
    w = w.detach().to(torch.float32)
    scale = (w.abs().amax(1) / 448.0).clamp_min(1e-30)
    return (w / scale[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).contiguous(), scale.contiguous()
