"""RootNetEngine: ResNet-34 backbone + ResRootNet depth head on libhamer_hip (NHWC 16-bit implicit-GEMM convolutions with
BatchNorm folded in, ReLU / residual-add epilogues, max-pool, fused global-average-pool + 1x1 conv)."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from .. import lib as L
from . import arch


def _kpad(k: int) -> int:
    return (k + 63) // 64 * 64


class RootNetEngine:
    def __init__(self, net_sd: Dict[str, torch.Tensor], root_sd: Dict[str, torch.Tensor], device="cuda", dtype=torch.float16):
        if not torch.cuda.is_available():
            raise L.HipLibraryError("RootNetEngine needs an MI355X (HIP device); there is no CPU fallback")
        self.lib = L.load()
        self.device, self.dtype = torch.device(device), dtype
        self.dt = L.HM_DTYPE_BF16 if dtype == torch.bfloat16 else L.HM_DTYPE_F16
        self.zeros = torch.zeros(64, dtype=torch.uint8, device=self.device)
        self.w: Dict[str, tuple] = {}

        def fold(conv_key, bn_key):
            w = net_sd[conv_key + ".weight"].float()
            g, b = net_sd[bn_key + ".weight"].float(), net_sd[bn_key + ".bias"].float()
            mu, var = net_sd[bn_key + ".running_mean"].float(), net_sd[bn_key + ".running_var"].float()
            sc = g / torch.sqrt(var + arch.BN_EPS)
            return w * sc.reshape(-1, 1, 1, 1), b - mu * sc

        def put(name, w, b):
            co, ci, k, _ = w.shape
            cin = max(8, ci)                                   # the 3-channel image travels with 8 channels
            wk = torch.zeros(co, k, k, cin)
            wk[:, :, :, :ci] = w.permute(0, 2, 3, 1)
            flat = torch.zeros(co, _kpad(k * k * cin))
            flat[:, :k * k * cin] = wk.reshape(co, -1)
            self.w[name] = (flat.to(self.device, dtype).contiguous(), b.to(self.device, torch.float32).contiguous(), cin, co, k)

        put("stem", *fold(arch.STEM_CONV, arch.STEM_BN))
        for pre, cin, cout, s, ds in arch.blocks():
            put(pre + "conv1", *fold(pre + "conv1", pre + "bn1"))
            put(pre + "conv2", *fold(pre + "conv2", pre + "bn2"))
            if ds:
                put(pre + "downsample", *fold(pre + "downsample.0", pre + "downsample.1"))
        self.depth_w = root_sd["depth_layer.weight"].reshape(-1).to(self.device, torch.float32).contiguous()
        self.depth_b = float(root_sd["depth_layer.bias"].reshape(-1)[0])

    def _conv(self, name, x, n, h, w, stride, act, resid=None):
        wt, bs, cin, co, k = self.w[name]
        ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
        y = torch.empty(n, ho, wo, co, device=self.device, dtype=self.dtype)
        a = L.ConvArgs(L.ptr(x), L.ptr(wt), L.ptr(y), L.ptr(bs), L.ptr(self.zeros), n, h, w, cin, co, k, stride, x.shape[-1], co,
                       wt.shape[1], act, 0, self.dt, L.ptr(resid), co if resid is not None else 0)
        L.check(self.lib.hm_conv2d_nhwc(C.byref(a), L.current_stream()), "hm_conv2d_nhwc")
        return y, ho, wo

    def features(self, img: torch.Tensor) -> torch.Tensor:
        """img (B, 3, 256, 256) fp32 normalised RGB planes (the layout hm_crop_batch writes) -> (B, 8, 8, 512) NHWC."""
        B, _, H, W = img.shape
        x = torch.empty(B, H, W, 8, device=self.device, dtype=self.dtype)
        L.check(self.lib.hm_nchw3_to_nhwc8(L.ptr(img.contiguous()), L.ptr(x), B, H, W, self.dt, L.current_stream()), "hm_nchw3_to_nhwc8")
        x, h, w = self._conv("stem", x, B, H, W, 2, 2)
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        y = torch.empty(B, ho, wo, 64, device=self.device, dtype=self.dtype)
        L.check(self.lib.hm_maxpool_nhwc(L.ptr(x), 64, L.ptr(y), 64, B, h, w, 64, 3, 2, 1, self.dt, L.current_stream()), "hm_maxpool_nhwc")
        x, h, w = y, ho, wo
        for pre, cin, cout, s, ds in arch.blocks():
            idn = x
            if ds:
                idn, _, _ = self._conv(pre + "downsample", x, B, h, w, s, 0)
            t, h2, w2 = self._conv(pre + "conv1", x, B, h, w, s, 2)
            x, h, w = self._conv(pre + "conv2", t, B, h2, w2, 1, 2, resid=idn)
        return x

    def forward(self, img: torch.Tensor, k_value: torch.Tensor) -> torch.Tensor:
        """depth (B,) = (GAP(features) . w + b) * k_value (ResRootNet.forward_coord, Model_RGB.py:282-292)."""
        kv = k_value.to(self.device, torch.float32).contiguous()      # (a pageable upload waits for the stream: before the backbone is queued)
        f = self.features(img)
        B, h, w, c = f.shape
        depth = torch.empty(B, device=self.device, dtype=torch.float32)
        L.check(self.lib.hm_gap_linear(L.ptr(f), h * w, c, L.ptr(self.depth_w), self.depth_b, L.ptr(kv), L.ptr(depth), B, self.dt,
                                       L.current_stream()), "hm_gap_linear")
        return depth
