"""CPU restatement of the MXFP8 / e4m3 quantisation used by BASELINE configs[4] ("fp8 ViT-H weights on CDNA4 fp8 MFMA").

TEST INFRASTRUCTURE ONLY (checker for tests/, smoke and bench's cpu_baseline): the product never imports it.
The reference repository has no fp8 path ("fp8 will need the bar stated against an
fp8-quantised oracle"), so this file defines that oracle: OCP e4m3fn elements (torch.float8_e4m3fn, round-to-nearest-even),
activations in MX blocks of 32 along K with a power-of-two E8M0 scale, weights with one fp32 scale per output channel.
Parity unpinned against any external implementation; pinned against the hardware by the exact-data probes in tools/probes/.
"""
import torch

_INV448 = torch.tensor(1.0 / 448.0, dtype=torch.float32)     # the fp32 constant the kernels multiply by


def mx8_scale_bytes(amax: torch.Tensor) -> torch.Tensor:
    """E8M0 byte = ceil(log2(amax / 448)) + 127, taken from the exponent field of fp32(amax * (1/448))."""
    bits = (amax.to(torch.float32) * _INV448).view(torch.int32)
    return (((bits >> 23) & 0xFF) + ((bits & 0x7FFFFF) != 0).to(torch.int32)).to(torch.uint8)


def mx8_quantize(x: torch.Tensor):
    """x (M, K) fp32 -> (q (M, K) uint8 e4m3 bytes, scales (K/32, M) uint8)."""
    M, K = x.shape
    xb = x.to(torch.float32).reshape(M, K // 32, 32)
    sb = mx8_scale_bytes(xb.abs().amax(-1))                               # (M, K/32)
    inv = torch.ldexp(torch.ones((), dtype=torch.float32), (127 - sb.to(torch.int32)))
    q = (xb * inv.unsqueeze(-1)).to(torch.float8_e4m3fn).view(torch.uint8).reshape(M, K)
    return q, sb.t().contiguous()


def mx8_dequantize(q: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    M, K = q.shape
    v = q.view(torch.float8_e4m3fn).to(torch.float32).reshape(M, K // 32, 32)
    s = torch.ldexp(torch.ones((), dtype=torch.float32), scales.t().to(torch.int32) - 127)
    return (v * s.unsqueeze(-1)).reshape(M, K)


def quantize_weight(w: torch.Tensor):
    """w (N, K) -> (w8 uint8 e4m3 bytes, scale (N,) fp32) with scale[n] = max|w[n]| / 448."""
    w = w.to(torch.float32)
    scale = (w.abs().amax(1) / 448.0).clamp_min(1e-30)
    return (w / scale[:, None]).to(torch.float8_e4m3fn).view(torch.uint8), scale


def dequantize_weight(w8: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return w8.view(torch.float8_e4m3fn).to(torch.float32) * scale[:, None]


def fake_quant_mx8(x: torch.Tensor) -> torch.Tensor:
    """Quantise-dequantise the last dimension in MX blocks of 32 (what the fp8 GEMM sees of its X operand)."""
    shp = x.shape
    q, s = mx8_quantize(x.reshape(-1, shp[-1]))
    return mx8_dequantize(q, s).reshape(shp)
