"""GPU parity of the detector path through the C ABI: implicit-GEMM convolution, pooling,
letterbox (byte-exact), decode, NMS (index-exact) and the whole Detector.detect."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hamer_yolo_amd import lib as L
from hamer_yolo_amd import synth
from hamer_yolo_amd.yolo import arch, fuse
from hamer_yolo_amd.yolo.detector import Detector
from hamer_yolo_amd.yolo.engine import YoloEngine
from oracle import yolo_ref

DEV = "cuda"


def _conv_gpu(x_nchw, w, b, k, s, act, dt, out_f32=False, ld_extra=0, y_extra=0, splitk_ws=None):
    """Run hm_conv2d_nhwc on an NHWC copy (optionally inside wider buffers to exercise strides)."""
    lib = L.load()
    N, Ci, H, W = x_nchw.shape
    Co = w.shape[0]
    cin = 8 if Ci < 8 else Ci
    xb = torch.zeros(N, H, W, cin + ld_extra, dtype=dt)
    xb[..., ld_extra:ld_extra + Ci] = x_nchw.permute(0, 2, 3, 1).to(dt)
    wk = torch.zeros(Co, k, k, cin)
    wk[..., :Ci] = w.permute(0, 2, 3, 1)
    kp = (k * k * cin + 63) // 64 * 64
    wf = torch.zeros(Co, kp)
    wf[:, :k * k * cin] = wk.reshape(Co, -1)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    xd, wd, bd = xb.to(DEV), wf.to(dt).to(DEV), b.float().to(DEV)
    yd = torch.zeros(N, Ho, Wo, Co + y_extra, dtype=torch.float32 if out_f32 else dt, device=DEV)
    zeros = torch.zeros(64, dtype=torch.uint8, device=DEV)
    esz = 2
    a = L.ConvArgs(xd.data_ptr() + ld_extra * esz, wd.data_ptr(), yd.data_ptr() + y_extra * (4 if out_f32 else 2), bd.data_ptr(),
                   zeros.data_ptr(), N, H, W, cin, Co, k, s, cin + ld_extra, Co + y_extra, kp, int(act), int(out_f32),
                   L.HM_DTYPE_BF16 if dt == torch.bfloat16 else L.HM_DTYPE_F16, None, 0,
                   splitk_ws.data_ptr() if splitk_ws is not None else None, splitk_ws.numel() if splitk_ws is not None else 0)
    L.check(lib.hm_conv2d_nhwc(C.byref(a), L.current_stream()), "hm_conv2d_nhwc")
    torch.cuda.synchronize()
    y = yd.cpu().float()
    assert (y[..., :y_extra] == 0).all()                      # the neighbouring channel slice is untouched
    return y[..., y_extra:].permute(0, 3, 1, 2)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Ci,Co,k,s,H,W", [(3, 32, 3, 1, 40, 72), (32, 64, 3, 2, 38, 70), (64, 64, 1, 1, 24, 40), (128, 256, 3, 1, 12, 20),
                                            (256, 24, 1, 1, 12, 20), (64, 128, 3, 2, 31, 33), (512, 512, 3, 1, 6, 10)])
def test_conv2d_nhwc_vs_torch(Ci, Co, k, s, H, W, dt):
    x = synth.uniform("cx", (2, Ci, H, W), 1.0, seed=Ci).to(dt).float()
    w = synth.uniform("cw", (Co, Ci, k, k), (3.0 / (Ci * k * k)) ** 0.5, seed=Co).to(dt).float()
    b = synth.uniform("cb", (Co,), 0.3, seed=k)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=k // 2).float()
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    y = _conv_gpu(x, w, b, k, s, act=True, dt=dt, ld_extra=8 if Ci >= 8 else 0, y_extra=8)
    np.testing.assert_allclose(y.numpy(), F.silu(ref).numpy(), atol=2e-3, rtol=2 * ulp)
    y = _conv_gpu(x, w, b, k, s, act=False, dt=dt, out_f32=True)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), atol=2e-4, rtol=1e-5)


def test_conv_exact_integer_data():
    x = (torch.arange(2 * 16 * 9 * 11).reshape(2, 16, 9, 11) % 5 - 2).float()
    w = ((torch.arange(32 * 16 * 9).reshape(32, 16, 3, 3) * 7 + torch.arange(32)[:, None, None, None]) % 3 - 1).float()
    ref = F.conv2d(x, w, torch.zeros(32), stride=1, padding=1)
    y = _conv_gpu(x, w, torch.zeros(32), 3, 1, act=False, dt=torch.float16, out_f32=True)
    assert torch.equal(y, ref)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_conv_every_tile_exact_and_equal(tile):
    """Round 3: the convolution runs on nine tiles (128 x 128 / 64 / 32 with four waves and a two-stage ring, 256 x 128 / 256 / 64 with
    eight, the 128-row ones again with a three- / four-stage ring for lone workgroups), chosen
    per layer.  Every one of them forced in turn (HM_OPT_CONV_TILE) on exact-integer data -- 3x3 stride 1 and 2, 1x1, ragged
    M (not a multiple of any tile), Cout below and above the tile width, strided input and output slices -- must give the
    exact result, and on random data bit for bit what the default choice gives (same K order in every tile)."""
    with L.option(L.HM_OPT_CONV_TILE, tile):
        for (Ci, Co, k, s, H, W) in ((16, 32, 3, 1, 9, 11), (32, 264, 3, 2, 21, 19), (64, 72, 1, 1, 33, 35), (8, 256, 3, 1, 30, 34)):
            x = (torch.arange(2 * Ci * H * W).reshape(2, Ci, H, W) % 5 - 2).float()
            w = ((torch.arange(Co * Ci * k * k).reshape(Co, Ci, k, k) * 7 + torch.arange(Co)[:, None, None, None]) % 3 - 1).float()
            b = (torch.arange(Co) % 7 - 3).float()
            ref = F.conv2d(x, w, b, stride=s, padding=k // 2)
            assert torch.equal(_conv_gpu(x, w, b, k, s, act=False, dt=torch.float16, out_f32=True), ref), (tile, Ci, Co, k, s)
            y = _conv_gpu(x, w, b, k, s, act=False, dt=torch.float16, ld_extra=8, y_extra=8)       # 16-bit store path, strided slices
            assert torch.equal(y, ref.half().float()), (tile, Ci, Co, k, s)
    x = synth.uniform("tx", (2, 128, 24, 40), 1.0, seed=3).half().float()
    w = synth.uniform("tw", (256, 128, 3, 3), 0.05, seed=4).half().float()
    b = synth.uniform("tb", (256,), 0.3, seed=5)
    with L.option(L.HM_OPT_CONV_KGROUPS, 1):                 # (this map would take two K groups, whose tiles are tested below)
        base = _conv_gpu(x, w, b, 3, 1, act=True, dt=torch.float16)
        with L.option(L.HM_OPT_CONV_TILE, tile):
            assert torch.equal(_conv_gpu(x, w, b, 3, 1, act=True, dt=torch.float16), base)


@pytest.mark.parametrize("tile", [0, 10, 11, 12, 13, 14, 15])
def test_conv_k_groups_exact_and_close(tile):
    """Round 3: on the 12 x 20 / 24 x 40 maps (at most 1024 output pixels per image, >= 8 K tiles per K range) the workgroup
    carries TWO K groups of four waves, each with its own LDS ring, that take alternate K tiles and hand their accumulators to
    group 0 through LDS (gemm_tn_kernel<..., WK = 2>; tiles 128 x 32 / 64 / 128 `_K2`, forced as 10-12) when the launch has few
    workgroups -- and, when it has many, on ONE group with two accumulator sets for the even and the odd K tiles (`_P2`, 13-15):
    the same summation order, so the choice by launch size cannot change a byte.  Exact on integer data for the
    automatic choice and for each of the six tiles forced, 3x3 and 1x1, stride 1 and 2, alone and on top of split-K
    (K = 4608 on 12 x 20: four K ranges of 18 tiles, nine steps per group); the same bytes for every tile (one K order); on
    random data within fp32 summation-order distance of the kernel without K groups (HM_OPT_CONV_KGROUPS = 1), and a frame gets
    the same bytes alone and inside a batch of five (the rule looks at one image)."""
    with L.option(L.HM_OPT_CONV_TILE, tile):
        for (Ci, Co, k, s, H, W, ws) in ((128, 72, 3, 1, 12, 20, False), (128, 264, 3, 2, 47, 79, False), (512, 40, 1, 1, 24, 40, False),
                                         (512, 64, 3, 1, 12, 20, True)):
            x = (torch.arange(2 * Ci * H * W).reshape(2, Ci, H, W) % 5 - 2).float()
            w = ((torch.arange(Co * Ci * k * k).reshape(Co, Ci, k, k) * 7 + torch.arange(Co)[:, None, None, None]) % 3 - 1).float()
            b = (torch.arange(Co) % 7 - 3).float()
            ref = F.conv2d(x, w, b, stride=s, padding=k // 2)
            wsb = torch.empty(8 * 2 * H * W * Co * 4, dtype=torch.uint8, device=DEV) if ws else None
            y = _conv_gpu(x, w, b, k, s, act=False, dt=torch.float16, ld_extra=8, y_extra=8, splitk_ws=wsb)
            assert torch.equal(y, ref.half().float()), (tile, Ci, Co, k, s)
            if not ws:
                assert torch.equal(_conv_gpu(x, w, b, k, s, act=False, dt=torch.float16, out_f32=True), ref), (tile, Ci, Co, k, s)
    x = synth.uniform("kx", (5, 256, 24, 40), 1.0, seed=3).half().float()
    w = synth.uniform("kw", (256, 256, 3, 3), 0.03, seed=4).half().float()
    b = synth.uniform("kb", (256,), 0.3, seed=5)
    base = _conv_gpu(x, w, b, 3, 1, act=True, dt=torch.float16)
    with L.option(L.HM_OPT_CONV_TILE, tile):
        y = _conv_gpu(x, w, b, 3, 1, act=True, dt=torch.float16)
        one = _conv_gpu(x[3:4], w, b, 3, 1, act=True, dt=torch.float16)
    assert torch.equal(y, base) and torch.equal(one, y[3:4])
    with L.option(L.HM_OPT_CONV_KGROUPS, 1):
        plain = _conv_gpu(x, w, b, 3, 1, act=True, dt=torch.float16)
    assert not torch.equal(plain, base)                       # (another summation order: some fp16 roundings move)
    ref = F.silu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).float()
    assert (base - ref).abs().max() < 2e-3 and (plain - ref).abs().max() < 2e-3 and (base - plain).abs().max() < 2e-3


@pytest.mark.parametrize("Ci,Co,k,s,H,W", [(64, 64, 3, 1, 23, 37), (128, 72, 3, 2, 47, 79), (256, 128, 1, 1, 24, 40), (128, 96, 5, 1, 13, 13),
                                           (512, 264, 3, 1, 12, 20), (64, 40, 3, 2, 9, 11)])
def test_conv_lean_loader_equals_the_general_loader(Ci, Co, k, s, H, W):
    """Round 4: with Cin % 64 == 0 a 64-deep K-step of the implicit GEMM lies inside one tap, and the lean loader keeps the K
    position in scalar registers and the in-image test as one bit of a per-lane mask (one 64-bit add, a bit test and a select per
    piece and K-step instead of ~25 instructions).  Same copies, same order: the bytes must equal the general loader's
    (HM_OPT_CONV_GENERAL_LOADER = 1) -- odd map sizes (every border case of the mask), stride 2, 1x1 / 3x3 / 5x5, ragged Cout,
    channel slices of wider buffers, with and without split-K scratch; exact on integers."""
    x = synth.uniform("lx", (3, Ci, H, W), 1.0, seed=Ci + H).half().float()
    w = synth.uniform("lw", (Co, Ci, k, k), (3.0 / (Ci * k * k)) ** 0.5, seed=Co).half().float()
    b = synth.uniform("lb", (Co,), 0.3, seed=k)
    ws = torch.empty(8 * 3 * H * W * Co * 4, dtype=torch.uint8, device=DEV)
    for kw in (dict(ld_extra=64, y_extra=8), dict(splitk_ws=ws)):
        lean = _conv_gpu(x, w, b, k, s, act=True, dt=torch.float16, **kw)
        with L.option(L.HM_OPT_CONV_GENERAL_LOADER, 1):
            gen = _conv_gpu(x, w, b, k, s, act=True, dt=torch.float16, **kw)
        assert torch.equal(lean, gen), (Ci, Co, k, s, kw.keys())
    ref = F.silu(F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=k // 2)).float()
    assert (lean - ref).abs().max() < 3e-3
    xi = (torch.arange(3 * Ci * H * W).reshape(3, Ci, H, W) % 5 - 2).float()
    wi = ((torch.arange(Co * Ci * k * k).reshape(Co, Ci, k, k) * 7 + torch.arange(Co)[:, None, None, None]) % 3 - 1).float()
    bi = (torch.arange(Co) % 7 - 3).float()
    assert torch.equal(_conv_gpu(xi, wi, bi, k, s, act=False, dt=torch.float16), F.conv2d(xi, wi, bi, stride=s, padding=k // 2).half().float())


@pytest.mark.parametrize("Ci,Co,k,s,H,W,nb", [(1024, 512, 1, 1, 12, 20, 40),     # 1x1, K = 1024 -> 2 ranges (one chain each), 128 x 128 tile
                                              (256, 256, 3, 1, 12, 20, 72),      # 3x3, K = 2304 -> 4 ranges of 9 tiles: one chain each (odd count)
                                              (512, 264, 3, 1, 12, 20, 48),      # K = 4608 -> 4 ranges of 18 tiles: even / odd sets inside a range; ragged N
                                              (256, 128, 3, 1, 24, 40, 36),      # 24 x 40 map, K = 2304 -> 2 ranges of 18: two sets, 128 x 128 tile (3 accumulator sets)
                                              (512, 64, 3, 2, 24, 40, 140)])     # stride 2 onto 12 x 20, narrow layer: 128 x 64 tile
def test_conv_serial_k_ranges_give_the_split_k_bytes(Ci, Co, k, s, H, W, nb):
    """Round 4: split-K is a rule on ONE image (a frame must get the same bytes alone and in a batch), but its slabs and reduce
    launch cost 15-30 % once a detector pass carries 48 frames.  From 256 tiles on, launch_conv hands the layer to
    gemm_tn_kernel<..., KSER>: one workgroup sums the same K ranges one after the other -- each from zero, in the split kernel's
    order inside a range, the running total in the reduce kernel's order.  The bytes must equal (a) the parallel split forced on
    the same batch (HM_OPT_CONV_SPLITK = the rule's own count), (b) what a lone image gets (which takes the parallel split), on
    random data, with bias and SiLU; exact-integer data must come back exact."""
    lib = L.load()
    x = synth.uniform("sx", (nb, Ci, H, W), 1.0, seed=Ci + nb).half().float()
    w = synth.uniform("sw", (Co, Ci, k, k), (3.0 / (Ci * k * k)) ** 0.5, seed=Co).half().float()
    b = synth.uniform("sb", (Co,), 0.3, seed=k)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    probe = L.ConvArgs(1, 1, 1, 1, 1, nb, H, W, Ci, Co, k, s, Ci, Co, (k * k * Ci + 63) // 64 * 64, 1, 0, L.HM_DTYPE_F16)
    need = lib.hm_conv_splitk_bytes(C.byref(probe))
    ranges = need // (nb * Ho * Wo * Co * 4)
    assert need > 0 and ranges in (2, 4), (need, ranges)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    auto = _conv_gpu(x, w, b, k, s, act=True, dt=torch.float16, y_extra=8, splitk_ws=ws)            # many tiles: the serial route
    with L.option(L.HM_OPT_CONV_SPLITK, ranges):
        par = _conv_gpu(x, w, b, k, s, act=True, dt=torch.float16, y_extra=8, splitk_ws=ws)         # the parallel split, forced
    one = _conv_gpu(x[5:6], w, b, k, s, act=True, dt=torch.float16, y_extra=8, splitk_ws=ws)         # a lone image: parallel split
    assert torch.equal(auto, par) and torch.equal(auto[5:6], one)
    ref = F.silu(F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=k // 2)).float()
    assert (auto - ref).abs().max() < 3e-3
    xi = (torch.arange(nb * Ci * H * W).reshape(nb, Ci, H, W) % 5 - 2).float()
    wi = ((torch.arange(Co * Ci * k * k).reshape(Co, Ci, k, k) * 7 + torch.arange(Co)[:, None, None, None]) % 3 - 1).float()
    bi = (torch.arange(Co) % 7 - 3).float()
    yi = _conv_gpu(xi, wi, bi, k, s, act=False, dt=torch.float16, splitk_ws=ws)
    assert torch.equal(yi, F.conv2d(xi, wi, bi, stride=s, padding=k // 2).half().float())


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Ci,Co,s,H,W", [(3, 32, 1, 20, 72), (3, 32, 1, 9, 150), (3, 32, 1, 33, 64), (3, 32, 1, 5, 640)])
def test_conv_direct_stem_kernel_equals_the_implicit_gemm(Ci, Co, s, H, W, dt):
    """Round 3: the first convolution of the detector (3x3, 3(8) -> 32) runs on conv3x3_direct_kernel -- the whole
    weight matrix resident in LDS, activations loaded straight into MFMA fragments, 64 output pixels of a row per wave, lane-swap
    epilogue.  Same K order, MFMA and epilogue arithmetic as the implicit GEMM: the bytes must be equal to it
    (HM_OPT_CONV_DIRECT = 1 selects the implicit GEMM), for widths that are and are not multiples of the 64-pixel wave tile, odd
    heights under stride 2, inputs and outputs that are channel slices of wider buffers; and both match torch."""
    x = synth.uniform("dx", (2, Ci, H, W), 1.0, seed=Ci + W).to(dt).float()
    w = synth.uniform("dw", (Co, Ci, 3, 3), (3.0 / (Ci * 9)) ** 0.5, seed=Co).to(dt).float()
    b = synth.uniform("db", (Co,), 0.3, seed=3)
    ld_extra = 8 if Ci >= 8 else 0
    direct = _conv_gpu(x, w, b, 3, s, act=True, dt=dt, ld_extra=ld_extra, y_extra=8)
    with L.option(L.HM_OPT_CONV_DIRECT, 1):
        gemm = _conv_gpu(x, w, b, 3, s, act=True, dt=dt, ld_extra=ld_extra, y_extra=8)
    assert torch.equal(direct, gemm)
    ref = F.silu(F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=1)).float()
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    np.testing.assert_allclose(direct.numpy(), ref.numpy(), atol=2e-3, rtol=2 * ulp)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,H,W", [(2, 8, 16), (1, 24, 48), (3, 19, 37), (2, 5, 70), (1, 40, 16), (16, 96, 160)])
def test_conv_3x3_64_to_64_kernel_equals_the_implicit_gemm(n, H, W, dt):
    """Round 3: 3x3 stride-1 64 -> 64 convolutions (the second stem layer and the 3x3s of the first two E-ELAN stages) run on
    conv3x3_c64_kernel: weights resident in registers, the 10 x 18 pixel halo of an 8 x 16 tile staged once per tile by LDS-DMA
    (double-buffered across the tiles of a persistent workgroup), one barrier per tile, counted waits for LDS reads and stores.
    Same K order, MFMA and epilogue arithmetic as the implicit GEMM: the bytes must be equal to it (HM_OPT_CONV_DIRECT = 1
    selects the implicit GEMM) on maps that are whole tiles, maps that overhang them on either side, one tile, more tiles than
    workgroups (16 x 96 x 160: 1920 tiles for 512 workgroups), input and output as channel slices of wider buffers; and both
    match torch."""
    x = synth.uniform("cx", (n, 64, H, W), 1.0, seed=H + W).to(dt).float()
    w = synth.uniform("cw", (64, 64, 3, 3), (3.0 / (64 * 9)) ** 0.5, seed=7).to(dt).float()
    b = synth.uniform("cb", (64,), 0.3, seed=3)
    with L.option(L.HM_OPT_CONV_DIRECT, 3):                                      # 3: at any size (by default from 1024 tiles up)
        direct = _conv_gpu(x, w, b, 3, 1, act=True, dt=dt, ld_extra=64, y_extra=192)
    if n == 16:
        assert torch.equal(direct, _conv_gpu(x, w, b, 3, 1, act=True, dt=dt, ld_extra=64, y_extra=192))     # the default takes it here
    with L.option(L.HM_OPT_CONV_DIRECT, 1):
        gemm = _conv_gpu(x, w, b, 3, 1, act=True, dt=dt, ld_extra=64, y_extra=192)
    with L.option(L.HM_OPT_CONV_DIRECT, 2):                                      # 2: the stem kernel only
        gemm2 = _conv_gpu(x, w, b, 3, 1, act=True, dt=dt, ld_extra=64, y_extra=192)
    assert torch.equal(direct, gemm)
    assert torch.equal(gemm2, gemm)
    if n * H * W <= 4096:
        ref = F.silu(F.conv2d(x.double(), w.double(), b.double(), stride=1, padding=1)).float()
        ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        np.testing.assert_allclose(direct.numpy(), ref.numpy(), atol=2e-3, rtol=2 * ulp)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Ci,Co", [(32, 64), (64, 128)])
@pytest.mark.parametrize("n,H,W", [(2, 16, 32), (1, 48, 96), (3, 37, 75), (2, 10, 140), (1, 80, 32), (2, 33, 34), (16, 192, 320)])
def test_conv_3x3_stride2_32_to_64_kernel_equals_the_implicit_gemm(n, H, W, Ci, Co, dt):
    """Round 3: the 3x3 stride-2 32 -> 64 layer behind the stem runs on conv3x3_s2c32_kernel: weights in registers, the 17 x 33
    input halo of an 8 x 16 output tile staged once per tile as two planes (even / odd input columns) of 64-byte rows with the
    chunk swizzle c ^ 2 ((row >> 2) & 1), one barrier per tile, counted waits.  Same K order and epilogue arithmetic as the
    implicit GEMM: the bytes must be equal to it (HM_OPT_CONV_DIRECT = 1) on even and odd input sizes, maps smaller than a tile,
    tiles overhanging on either side, more tiles than workgroups, channel slices of wider buffers; and both match torch.
    (64, 128): conv3x3_s2c64_kernel, the same planes with 128-byte rows on eight waves, two passes of two pixel rows."""
    x = synth.uniform("sx", (n, Ci, H, W), 1.0, seed=H + W).to(dt).float()
    w = synth.uniform("sw", (Co, Ci, 3, 3), (3.0 / (Ci * 9)) ** 0.5, seed=7).to(dt).float()
    b = synth.uniform("sb", (Co,), 0.3, seed=3)
    with L.option(L.HM_OPT_CONV_DIRECT, 3):                                      # 3: at any size (by default from 1024 tiles up)
        direct = _conv_gpu(x, w, b, 3, 2, act=True, dt=dt, ld_extra=32, y_extra=192)
    if n == 16:
        assert torch.equal(direct, _conv_gpu(x, w, b, 3, 2, act=True, dt=dt, ld_extra=32, y_extra=192))     # the default takes it here
    with L.option(L.HM_OPT_CONV_DIRECT, 1):
        gemm = _conv_gpu(x, w, b, 3, 2, act=True, dt=dt, ld_extra=32, y_extra=192)
    assert torch.equal(direct, gemm)
    if n * H * W <= 16384:
        ref = F.silu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)).float()
        ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        np.testing.assert_allclose(direct.numpy(), ref.numpy(), atol=2e-3, rtol=2 * ulp)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Ci,Co,n,H,W", [(512, 512, 2, 48, 80), (1024, 256, 1, 48, 64), (256, 512, 3, 16, 64), (256, 1024, 48, 24, 40)])
def test_conv_1x1_persistent_kernel_equals_the_tile_kernel(Ci, Co, n, H, W, dt):
    """Round 4: 1x1 / stride 1 / SiLU layers with Cout % 256 == 0 and whole 256-row tiles are plain GEMMs and, from a round
    and a half of 256 x 256 tiles up, run on the ViT's persistent kernel (gemm_px_kernel with a SiLU epilogue).  Same K order,
    same epilogue arithmetic: the bytes equal the one-tile 256 x 256 kernel's and the 128 x 128 kernel's (HM_OPT_CONV_TILE = 16
    forces the persistent kernel at any size, 5 / 1 the tiles), input and output as channel slices of wider buffers; the last
    shape is one the default rule sends there by itself.  (Layers whose per-image rule sums two K groups -- at most 1024 pixels
    per image and K >= 512 -- keep their tile kernels: the persistent kernel has the one-group order only.)"""
    x = synth.uniform("px", (n, Ci, H, W), 1.0, seed=Ci + W).to(dt).float()
    w = synth.uniform("pw", (Co, Ci, 1, 1), (3.0 / Ci) ** 0.5, seed=Co).to(dt).float()
    b = synth.uniform("pb", (Co,), 0.3, seed=5)
    with L.option(L.HM_OPT_CONV_TILE, 16):
        pers = _conv_gpu(x, w, b, 1, 1, act=True, dt=dt, ld_extra=64, y_extra=192)
    for tile in (5, 1):
        with L.option(L.HM_OPT_CONV_TILE, tile):
            assert torch.equal(pers, _conv_gpu(x, w, b, 1, 1, act=True, dt=dt, ld_extra=64, y_extra=192)), tile
    if n == 48:
        assert torch.equal(pers, _conv_gpu(x, w, b, 1, 1, act=True, dt=dt, ld_extra=64, y_extra=192))          # the default rule
    if n * H * W <= 4096:
        ref = F.silu(F.conv2d(x.double(), w.double(), b.double())).float()
        ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        np.testing.assert_allclose(pers.numpy(), ref.numpy(), atol=2e-3, rtol=2 * ulp)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Ci,Co,n,H,W", [(256, 256, 1, 8, 8), (256, 128, 2, 16, 12), (128, 128, 3, 8, 24), (128, 256, 1, 16, 20), (256, 256, 16, 96, 160),
                                          (128, 128, 16, 96, 160)])
def test_conv_1x1_weights_in_registers_kernel_equals_the_implicit_gemm(Ci, Co, n, H, W, dt):
    """Round 3: 1x1 convolutions with K = 128 / 256 and Cout = 128 / 256 on the large maps run on conv1x1_wreg_kernel: the
    weights in registers for the life of a persistent workgroup, 64-pixel tiles of X streamed through a double-buffered LDS image
    (chunk index XOR row & 15), K/32 precomputed lane addresses, counted waits, lane-swap epilogue.  Same K order and epilogue as
    the implicit GEMM: the bytes must be equal to it (HM_OPT_CONV_DIRECT = 1), one tile, several, more tiles than workgroups,
    input and output as channel slices of wider buffers; and both match torch."""
    x = synth.uniform("ox", (n, Ci, H, W), 1.0, seed=Ci + W).to(dt).float()
    w = synth.uniform("ow", (Co, Ci, 1, 1), (3.0 / Ci) ** 0.5, seed=Co).to(dt).float()
    b = synth.uniform("ob", (Co,), 0.3, seed=3)
    with L.option(L.HM_OPT_CONV_DIRECT, 3):                                      # 3: at any size (by default from 200 tiles up)
        direct = _conv_gpu(x, w, b, 1, 1, act=True, dt=dt, ld_extra=64, y_extra=192)
    if n == 16:
        assert torch.equal(direct, _conv_gpu(x, w, b, 1, 1, act=True, dt=dt, ld_extra=64, y_extra=192))     # the default takes it here
    with L.option(L.HM_OPT_CONV_DIRECT, 1):
        gemm = _conv_gpu(x, w, b, 1, 1, act=True, dt=dt, ld_extra=64, y_extra=192)
    assert torch.equal(direct, gemm)
    if n * H * W <= 4096:
        ref = F.silu(F.conv2d(x.double(), w.double(), b.double())).float()
        ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        np.testing.assert_allclose(direct.numpy(), ref.numpy(), atol=2e-3, rtol=2 * ulp)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_conv_split_k_exact_and_close(dt):
    """Split-K convolution (hm_conv_args.splitk_ws: few output tiles, long K -- the 12x20 / 24x40 maps of the YOLOv7 neck):
    fp32 partial slabs per K range, added in order by the reduce kernel, then bias + SiLU.  Exact on integer data for 2, 4 and
    8 ranges and for the automatic choice (a rule on ONE image's output size, so a frame's result does not depend on the batch it
    rides in); on random data within fp32 summation-order distance of the unsplit kernel; a workspace smaller than
    hm_conv_splitk_bytes is an error."""
    Ci, Co, H, W = 256, 256, 12, 20                                   # K = 2304 = 36 K tiles; 2 x 240 rows -> 4 x 1 tiles of 128 x 256
    x = (torch.arange(2 * Ci * H * W).reshape(2, Ci, H, W) % 5 - 2).float()
    w = ((torch.arange(Co * Ci * 9).reshape(Co, Ci, 3, 3) * 7 + torch.arange(Co)[:, None, None, None]) % 3 - 1).float()
    b = (torch.arange(Co) % 7 - 3).float()
    ref = F.conv2d(x, w, b, stride=1, padding=1)
    ws = torch.full((8 * 2 * H * W * Co * 4 + 64,), 0xAB, dtype=torch.uint8, device=DEV)
    for ranges in (0, 2, 4, 8):
        with L.option(L.HM_OPT_CONV_SPLITK, ranges):
            y = _conv_gpu(x, w, b, 3, 1, act=False, dt=dt, ld_extra=8, y_extra=8, splitk_ws=ws[:-64])
        assert torch.equal(y, ref.to(dt).float()), ranges
        assert bool((ws[-64:] == 0xAB).all())                        # nothing written past the workspace
    tiny = torch.empty(2 * H * W * Co * 4 * 2, dtype=torch.uint8, device=DEV)      # room for two slabs only: an error, never a
    with pytest.raises(L.HipLibraryError, match="splitk_ws too small"):            # silent change of the summation order
        _conv_gpu(x, w, b, 3, 1, act=False, dt=dt, splitk_ws=tiny)
    xr = synth.uniform("sx", (2, Ci, H, W), 1.0, seed=1).to(dt).float()
    wr = synth.uniform("sw", (Co, Ci, 3, 3), 0.03, seed=2).to(dt).float()
    br = synth.uniform("sb", (Co,), 0.3, seed=3)
    one = _conv_gpu(xr, wr, br, 3, 1, act=True, dt=dt)
    with L.option(L.HM_OPT_CONV_SPLITK, 4):
        four = _conv_gpu(xr, wr, br, 3, 1, act=True, dt=dt, splitk_ws=ws[:-64])
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    np.testing.assert_allclose(four.numpy(), one.numpy(), atol=1e-4, rtol=2 * ulp)     # one 16-bit ulp: the fp32 sums differ in the last bits


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,H,W", [(2, 64, 96), (1, 50, 70), (3, 33, 47), (1, 384, 640), (5, 16, 32)])
def test_stem_pair_bit_identical_to_two_launches(n, H, W, dt):
    """hm_conv2d_stem_pair: Conv 0 (3(8) -> 32, k3 s1) and Conv 1 (32 -> 64, k3 s2) of the detector as ONE launch whose
    intermediate lives in LDS.  Both layers keep their K order, MFMA shape and epilogue arithmetic and the intermediate is
    rounded to 16 bits as the stored tensor would be, so the output must equal the two launches (HM_OPT_CONV_STEM_PAIR = 1)
    BIT FOR BIT -- interior tiles, tiles on every border, odd sizes whose last tiles overhang the map, several frames -- and
    the first layer's output buffer must stay untouched."""
    lib = L.load()
    g = torch.Generator().manual_seed(n * 1000 + H)
    x = torch.zeros(n, H, W, 8, dtype=dt)
    x[..., :3] = torch.rand(n, H, W, 3, generator=g).to(dt)
    w0 = torch.zeros(32, 128)
    w0[:, :72] = ((torch.rand(32, 3, 3, 8, generator=g) - 0.5) * 0.8 * (torch.arange(8) < 3)).reshape(32, 72)
    w1 = torch.zeros(64, 320)
    w1[:, :288] = ((torch.rand(64, 288, generator=g) - 0.5) * 0.25)
    b0, b1 = (torch.rand(32, generator=g) - 0.5), (torch.rand(64, generator=g) - 0.5)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    xd, w0d, w1d, b0d, b1d = x.to(DEV), w0.to(dt).to(DEV), w1.to(dt).to(DEV), b0.to(DEV), b1.to(DEV)
    zeros = torch.zeros(64, dtype=torch.uint8, device=DEV)
    code = L.HM_DTYPE_BF16 if dt == torch.bfloat16 else L.HM_DTYPE_F16
    outs = []
    try:
        for two in (1, 0):
            mid = torch.full((n, H, W, 32), 7.0, dtype=dt, device=DEV)
            y = torch.full((n, Ho, Wo, 64 + 8), -3.0, dtype=dt, device=DEV)       # (a wider buffer: ldy 72, channel slice at +8)
            a = L.ConvArgs(xd.data_ptr(), w0d.data_ptr(), mid.data_ptr(), b0d.data_ptr(), zeros.data_ptr(), n, H, W, 8, 32, 3, 1, 8, 32, 128, 1, 0, code, None, 0, None, 0)
            b = L.ConvArgs(mid.data_ptr(), w1d.data_ptr(), y.data_ptr() + 16, b1d.data_ptr(), zeros.data_ptr(), n, H, W, 32, 64, 3, 2, 32, 72, 320, 1, 0, code, None, 0, None, 0)
            L.check(lib.hm_set_option(L.HM_OPT_CONV_STEM_PAIR, two))
            L.check(lib.hm_conv2d_stem_pair(C.byref(a), C.byref(b), L.current_stream()), "hm_conv2d_stem_pair")
            torch.cuda.synchronize()
            outs.append((y.cpu(), mid.cpu()))
    finally:
        L.check(lib.hm_set_option(L.HM_OPT_CONV_STEM_PAIR, 0))
    (y2, mid2), (y1, mid1) = outs
    assert torch.equal(y1.view(torch.int16), y2.view(torch.int16))
    assert (y1[..., :8] == -3.0).all()                               # the neighbouring channel slice is untouched
    assert (mid1 == 7.0).all() and not (mid2 == 7.0).all()           # one launch: the intermediate is never written
    assert y1[..., 8:].float().abs().max() > 0.05


def test_stem_fusion_does_not_change_the_network():
    """The engine with Conv 0 + Conv 1 as one launch (HM_OP_CONV_PAIR, the default) against the engine that launches them
    apart: Conv 1's output and the head's raw logits are bit-identical, so every per-layer statement of
    test_every_layer_vs_oracle_on_the_gpus_own_inputs (which needs Conv 0's output in memory and therefore runs the two
    launches) holds for the fused network as well; two frames in one pass too."""
    sd = synth.yolo_state_dict(seed=0, nc=3)
    frames = [synth.frame_u8(540, 960, seed=5).to(DEV), synth.frame_u8(540, 960, seed=6).to(DEV)]
    outs = {}
    for fused_stem in (False, True):
        e = YoloEngine(sd, nc=3, device=DEV)
        e.fuse_stem = fused_stem
        p = e.forward(frames[0])
        torch.cuda.synchronize()
        one = (e.layer_output(p, 0), e.layer_output(p, 1), [r[0].clone().cpu() for r in p["raws"]], p["n_ops"], [int(o.kind) for o in p["ops"]][:2])
        p2 = e.forward(frames)
        torch.cuda.synchronize()
        outs[fused_stem] = one + ([r[0].clone().cpu() for r in p2["raws"]],)
    assert outs[True][4] == [3, 0] and outs[False][4] == [0, 0] and outs[True][3] == outs[False][3]
    assert float(outs[True][0].abs().max()) == 0.0 and float(outs[False][0].abs().max()) > 0.0      # Conv 0's map is not written
    assert torch.equal(outs[True][1], outs[False][1])
    for a, b_ in zip(outs[True][2] + outs[True][5], outs[False][2] + outs[False][5]):
        assert torch.equal(a, b_)


def test_yolo_pair_fusion_and_split_k_do_not_change_the_network():
    """E-ELAN cv1 / cv2 (two 1x1 convolutions of one input, adjacent slices of the block's Concat) run as one launch with
    stacked weight rows: every output channel is the same dot product in the same K order, so every layer output of the
    network must be bit-identical with and without the fusion.  Split-K changes the fp32 summation order of the split layers
    only: the head's raw logits stay within the fp16 reordering noise of the unsplit run."""
    sd = synth.yolo_state_dict(seed=0, nc=3)
    frame = synth.frame_u8(384, 640, seed=5).to(DEV)
    outs = {}
    for name, fuse_pairs, split in (("plain", False, False), ("fused", True, False), ("fused_split", True, True)):
        e = YoloEngine(sd, nc=3, device=DEV)
        e.fuse_pairs, e.split_k = fuse_pairs, split
        p = e.forward(frame)
        torch.cuda.synchronize()
        outs[name] = ([e.layer_output(p, i) for i in (4, 5, 11, 24, 37, 50, 63, 75, 88, 101)], [r[0].clone().cpu() for r in p["raws"]], p["n_ops"])
    assert outs["fused"][2] == outs["plain"][2] - 8                  # eight E-ELAN blocks
    for a, b_ in zip(outs["plain"][0], outs["fused"][0]):
        assert torch.equal(a, b_)
    for a, b_ in zip(outs["plain"][1], outs["fused"][1]):
        assert torch.equal(a, b_)
    for a, b_ in zip(outs["plain"][1], outs["fused_split"][1]):
        assert float((a - b_).abs().max()) < 0.15 and float((a - b_).abs().mean()) < 5e-3        # measured 0.07 max / 2.1e-3 mean on logits of |x| ~ 1


def test_maxpool_and_upsample():
    lib = L.load()
    x = synth.uniform("px", (1, 64, 12, 20), 2.0, seed=1).half()
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = torch.empty(1, 6, 10, 64, dtype=torch.float16, device=DEV)
    L.check(lib.hm_maxpool_nhwc(xd.data_ptr(), 64, y.data_ptr(), 64, 1, 12, 20, 64, 2, 2, 0, L.HM_DTYPE_F16, L.current_stream()))
    assert torch.equal(y.cpu().permute(0, 3, 1, 2), F.max_pool2d(x.float(), 2, 2).half())
    # MP (k 2, stride 2): the two-outputs-per-thread kernel (even maps, W % 4 == 0) and the generic one, several frames, input and
    # output as channel slices of wider buffers, both dtypes
    for n, Cc, Hh, Ww, dt in ((3, 128, 16, 24, torch.float16), (2, 64, 10, 14, torch.float16), (2, 256, 8, 12, torch.bfloat16), (1, 64, 7, 9, torch.bfloat16)):
        xs = synth.uniform("mp", (n, Cc, Hh, Ww), 2.0, seed=Hh).to(dt)
        xb = torch.zeros(n, Hh, Ww, Cc + 64, dtype=dt)
        xb[..., 32:32 + Cc] = xs.permute(0, 2, 3, 1)
        xbd = xb.to(DEV)
        yb = torch.full((n, Hh // 2, Ww // 2, Cc + 32), 3.0, dtype=dt, device=DEV)
        L.check(lib.hm_maxpool_nhwc(xbd.data_ptr() + 64, Cc + 64, yb.data_ptr() + 32, Cc + 32, n, Hh, Ww, Cc, 2, 2, 0,
                                    L.HM_DTYPE_BF16 if dt == torch.bfloat16 else L.HM_DTYPE_F16, L.current_stream()))
        got = yb.cpu()
        assert torch.equal(got[..., 16:16 + Cc].permute(0, 3, 1, 2).float(), F.max_pool2d(xs.float(), 2, 2)), (n, Cc, Hh, Ww)
        assert (got[..., :16] == 3.0).all() and (got[..., 16 + Cc:] == 3.0).all()
    # SPP cascade: 5, 9 = 5o5, 13 = 5o5o5 written into channel slices of one buffer
    cat = torch.zeros(1, 12, 20, 256, dtype=torch.float16, device=DEV)
    cat[..., :64] = xd
    for step in range(3):
        L.check(lib.hm_maxpool_nhwc(cat.data_ptr() + step * 128, 256, cat.data_ptr() + (step + 1) * 128, 256, 1, 12, 20, 64, 5, 1, 2,
                                    L.HM_DTYPE_F16, L.current_stream()))
    ref = torch.cat([x.float()] + [F.max_pool2d(x.float(), k, 1, k // 2) for k in (5, 9, 13)], 1)
    assert torch.equal(cat.cpu().permute(0, 3, 1, 2).float(), ref)
    up = torch.empty(1, 24, 40, 64, dtype=torch.float16, device=DEV)
    L.check(lib.hm_upsample2x_nhwc(xd.data_ptr(), 64, up.data_ptr(), 64, 1, 12, 20, 64, L.HM_DTYPE_F16, L.current_stream()))
    assert torch.equal(up.cpu().permute(0, 3, 1, 2).float(), F.interpolate(x.float(), scale_factor=2, mode="nearest"))


@pytest.fixture(scope="module")
def engine():
    return YoloEngine(synth.yolo_state_dict(seed=0, nc=3), nc=3, device=DEV)


@pytest.mark.parametrize("hw", [(1080, 1920), (565, 848), (384, 640), (700, 500)])
def test_letterbox_bit_exact(engine, hw):
    frame = synth.frame_u8(hw[0], hw[1], seed=hw[0])
    p = engine.letterbox(frame.to(DEV), want_u8=True)
    torch.cuda.synchronize()
    ref, g = yolo_ref.letterbox(frame.numpy())
    got = p["u8"].cpu().numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)
    lp = p["lp"]
    x8 = torch.empty(lp.out_h * lp.out_w * 8, dtype=torch.float16, device=DEV)
    # the network input tensor: RGB/255 in the first 3 of 8 channels
    img = torch.frombuffer(bytearray(8), dtype=torch.uint8)   # placeholder to keep flake quiet
    arena = p["arena"]
    off = p["img_ptr"] - arena.data_ptr()
    x8 = arena[off:off + lp.out_h * lp.out_w * 16].view(torch.float16).reshape(lp.out_h, lp.out_w, 8).cpu().float()
    np.testing.assert_allclose(x8[..., :3].permute(2, 0, 1).numpy(), ref.astype(np.float32) / 255.0, atol=5e-4)
    assert (x8[..., 3:] == 0).all()


def _fused():
    layers = arch.yolov7_layers()
    return layers, fuse.fuse_state_dict(synth.yolo_state_dict(seed=0, nc=3), arch.conv_specs(layers, 3, 3))


def test_every_layer_vs_oracle_on_the_gpus_own_inputs():
    """Teacher-forced, layer by layer: each of the 105 layers + the detect head is recomputed by the oracle (emu="fp16": half
    weights / activations, fp32 accumulate -- the reference's GPU branch, detector.py:110-112) FROM THE GPU'S OWN INPUT
    TENSORS of that layer and compared with the GPU's output of that layer.  What is left between the two is the fp32
    summation order inside ONE layer: a result differs by at most one or two half ulps, and almost every element is
    bit-equal.  Pooling, upsampling and the concat-by-addressing must be exact.  (End to end the same noise random-walks
    through 105 layers of |x| <= 15 activations to ~2e-3 mean / 6e-2 max on the logits -- the floor for ANY two half
    implementations with different summation orders -- so the whole-network bounds below are necessarily looser.)"""
    frame = synth.frame_u8(540, 960, seed=5)
    engine = YoloEngine(synth.yolo_state_dict(seed=0, nc=3), nc=3, device=DEV)
    engine.fuse_stem = False          # (every layer's output in memory; the fused stem is tied to this run bit for bit by
    p = engine.forward(frame.to(DEV))  #  test_stem_fusion_does_not_change_the_network)
    torch.cuda.synchronize()
    layers, fused = _fused()
    lp = p["lp"]
    off = p["img_ptr"] - p["arena"].data_ptr()
    x8 = p["arena"][off:off + lp.out_h * lp.out_w * 16].view(torch.float16).reshape(lp.out_h, lp.out_w, 8).cpu().float()
    img = x8[..., :3].permute(2, 0, 1)[None].contiguous()
    outs = {}
    get = lambda j: outs.setdefault(j, engine.layer_output(p, j)[None])
    n_conv = 0
    with torch.no_grad():
        for i, (srcs, kind, args) in enumerate(arch.resolve(layers)):
            inp = [img if s < 0 else get(s) for s in srcs]
            ref = yolo_ref.layer_forward(layers, fused, i, inp, 3, emu="fp16")
            if kind == "detect":
                for (raw, hh, ww), r in zip(p["raws"], ref):
                    mine = raw.cpu().reshape(hh, ww, 3, 8).permute(2, 0, 1, 3)
                    np.testing.assert_allclose(mine.numpy(), r[0].numpy(), rtol=1e-4, atol=2e-4, err_msg=f"detect level {hh}x{ww}")
                continue
            got = get(i)
            assert got.shape == ref.shape, (i, kind, got.shape, ref.shape)
            if kind in ("mp", "up", "concat"):
                assert torch.equal(got, ref), (i, kind)
                continue
            n_conv += 1
            k = 4.0 if kind == "sppcspc" else 1.0                       # 7 chained convolutions inside SPPCSPC
            bad = (got - ref).abs() > k * (2.0 ** -9 * ref.abs() + 1e-3)
            same = float((got == ref).float().mean())
            assert not bool(bad.any()), (i, kind, int(bad.sum()), float((got - ref).abs().max()))
            assert same > (0.4 if kind == "sppcspc" else 0.97), (i, kind, same)
    assert n_conv == 79 + 3 + 1


def test_forward_decode_vs_oracle_and_reference_golden(engine, golden_dir):
    """Whole network on the reference's golden input.  Against the oracle in the kernel's own arithmetic (emu="fp16") the
    raw head logits agree to the half-precision reordering floor (see the per-layer test: mean ~2e-3, max ~6e-2) and the
    scores to 1e-2; against the fp32 reference golden the bounds are those of half vs single precision."""
    g = np.load(os.path.join(golden_dir, "yolo_forward.npz"))
    # the golden input is this image taken as RGB; the detector takes BGR frames (cv2.imread order) and the
    # letterbox kernel swaps to RGB, so hand it the channel-reversed frame.  384x640: the resize is the identity.
    frame = synth.frame_u8(384, 640, seed=int(g["frame_seed"])).flip(-1).contiguous()
    p = engine.forward(frame.to(DEV))
    torch.cuda.synchronize()
    pred = p["pred"].cpu()
    assert pred.shape == (15120, 8) and torch.isfinite(pred).all()
    layers, fused = _fused()
    x = synth.frame_u8(384, 640, seed=int(g["frame_seed"])).permute(2, 0, 1).float()[None] / 255.0
    with torch.no_grad():
        epred, eraws = yolo_ref.yolo_forward(layers, fused, x, 3, arch.ANCHORS, arch.STRIDES, emu="fp16")
        _, raws = yolo_ref.yolo_forward(layers, fused, x, 3, arch.ANCHORS, arch.STRIDES)
    for (raw, hh, ww), r, r32 in zip(p["raws"], eraws, raws):                # r: (1, 3, ny, nx, 8)
        mine = raw.cpu().reshape(hh, ww, 3, 8).permute(2, 0, 1, 3)
        d = (mine - r[0]).abs()
        assert float(d.max()) < 0.1 and float(d.mean()) < 4e-3, (float(d.max()), float(d.mean()))
        d32 = (mine - r32[0]).abs()
        assert float(d32.max()) < 0.3 and float(d32.mean()) < 0.02
    assert float((pred[:, 4:] - epred[0, :, 4:]).abs().max()) < 2e-2
    size = epred[0, :, 2:4].abs().mean(1, keepdim=True).clamp_min(8.0)
    # wh = (2 sigma)^2 anchor doubles the relative logit error.  The oracle sums every convolution in ONE pass over K; the product
    # splits the small maps' convolutions over K ranges and K groups (another fp32 summation order: 0.113 measured).  So
    # the 1e-1 bound of the single-pass order is asserted on the single-pass route (no split, no K groups), and the product route
    # is tied to it by a bound that expresses nothing but the summation-order difference.
    with L.option(L.HM_OPT_CONV_SPLITK, 1), L.option(L.HM_OPT_CONV_KGROUPS, 1):
        p1 = engine.forward(frame.to(DEV))
        torch.cuda.synchronize()
        pred1 = p1["pred"].cpu().clone()
    p = engine.forward(frame.to(DEV))
    torch.cuda.synchronize()
    pred = p["pred"].cpu()
    assert float(((pred1[:, :4] - epred[0, :, :4]).abs() / size).max()) < 1e-1
    assert float((pred1[:, 4:] - epred[0, :, 4:]).abs().max()) < 2e-2
    assert float(((pred[:, :4] - pred1[:, :4]).abs() / size).max()) < 8e-2 and float((pred[:, 4:] - pred1[:, 4:]).abs().max()) < 1e-2
    assert float(((pred[:, :4] - epred[0, :, :4]).abs() / size).max()) < 1.5e-1
    # loose: the reference's own fp32 output rows
    ref_rows = torch.from_numpy(g["pred_rows"])
    got = pred[::9]
    assert float((got[:, 4:] - ref_rows[:, 4:]).abs().max()) < 3e-2
    np.testing.assert_allclose(pred.double().sum(0).numpy(), g["pred_sum"], rtol=5e-3)


def test_nms_exact_on_reference_prediction(engine, golden_dir):
    """Feed the reference's own prediction rows to the HIP NMS: the kept rows must equal the reference's output."""
    g = np.load(os.path.join(golden_dir, "yolo_forward.npz"))
    p = engine._plan(384, 640)
    cand = torch.from_numpy(g["pred_cand"])
    pred = torch.zeros(15120, 8)
    perm = (synth._hash_u32(torch.arange(15120, dtype=torch.int64), 5) % 15120).unique()[:len(cand)].sort().values
    pred[perm] = cand                                                     # scattered, original order preserved
    p["pred"].copy_(pred.to(DEV))
    det = engine.nms(p, 0.25, 0.35, [0, 1, 2], True, scale=False).cpu()
    assert torch.equal(det, torch.from_numpy(g["dets"]))
    det1 = engine.nms(p, 0.25, 0.35, [1], False, scale=False).cpu()
    assert torch.equal(det1, torch.from_numpy(g["dets_cls1"]))
    # scaled + rounded variant vs the oracle's scale_coords on a 1080p plan
    p2 = engine._plan(1080, 1920)
    p2["pred"].copy_(pred.to(DEV))
    det2 = engine.nms(p2, 0.25, 0.35, [0, 1, 2], True, scale=True).cpu()
    ref = torch.from_numpy(g["dets"]).clone()
    ref[:, :4] = yolo_ref.scale_coords((384, 640), ref[:, :4], (1080, 1920, 3)).round()
    assert torch.equal(det2, ref)
    # empty input
    p["pred"].zero_()
    assert engine.nms(p, 0.25, 0.35, None, True).shape == (0, 6)


class _Opt:
    # seed 2 with objectness / class biases that yield boxes of both labels at every frame size below
    weights = "synthetic:2:-1.9:0"; imgsz = 640; augment = True; conf_thres = 0.25; iou_thres = 0.35
    classes = [0, 1, 2]; agnostic_nms = True; device = "cuda"; save_path = "./output"


@pytest.fixture(scope="module")
def detector():
    return Detector(_Opt)


# 16:9 (384x640 -> 15120 rows), the reference's own sample hamer/example_data/test1.jpg (565x848 -> 448x640, 17640 rows),
# 4:3 (480x640, 18900), square (640x640, 25200), portrait 1080p (640x384)
@pytest.mark.parametrize("hw,rows", [((540, 960), 15120), ((565, 848), 17640), ((480, 640), 18900), ((800, 800), 25200),
                                     ((1920, 1080), 15120)])
def test_detector_detect_end_to_end(detector, hw, rows):
    """Detector.detect at every letterbox shape.  With random weights the kept set of a greedy NMS is chaotic under
    fp16-sized score perturbations, so the end-to-end check is (a) the network output against the oracle in the same
    arithmetic (emu="fp16") on the SAME letterboxed input, tight, and (b) the box list against the oracle's
    non_max_suppression + scale_coords applied to the GPU's own prediction, which must agree exactly."""
    det = detector
    H, W = hw
    frame = synth.frame_u8(H, W, seed=11 + H)
    pred, dets_list = det.detect(frame.numpy())
    got = pred[0].cpu()
    assert len(dets_list) == 1 and len(dets_list[0]) == len(got) and got.shape[1] == 6 and len(got) > 0
    assert [lbl for lbl, _ in dets_list[0]] == ['right' if c == 1 else 'left' for c in got[:, 5].tolist()]
    assert all(0 <= b[0] <= W and 0 <= b[1] <= H and b[0] == round(b[0]) for _, b in dets_list[0])
    p = det.engine._plan(H, W)
    assert p["n_pred"] == rows
    gpu_pred = p["pred"].cpu()[None]
    layers = arch.yolov7_layers()
    fused = fuse.fuse_state_dict(synth.yolo_state_dict(seed=2, nc=3, obj_bias=-1.9, cls_bias=0.0), arch.conv_specs(layers, 3, 3))
    with torch.no_grad():
        ref_dets, ref_list, ref_pred = yolo_ref.detect(layers, fused, frame.numpy(), 3, arch.ANCHORS, emu="fp16")
    assert float((gpu_pred[..., 4:] - ref_pred[..., 4:]).abs().max()) < 2e-2               # (a) (the half reordering floor)
    mine = yolo_ref.non_max_suppression(gpu_pred, 0.25, 0.35, [0, 1, 2], True)[0]          # (b)
    mine[:, :4] = yolo_ref.scale_coords((p["lp"].out_h, p["lp"].out_w), mine[:, :4], frame.shape).round()
    assert torch.equal(got, mine)
    assert 0.5 * len(ref_dets[0]) <= len(got) <= 2 * len(ref_dets[0]) + 3
    assert {lbl for lbl, _ in dets_list[0]} == {"left", "right"}


@pytest.mark.parametrize("n,frac", [(25200, 0.1), (25200, 0.9), (40000, 0.95)])
def test_nms_many_candidates_exact(n, frac):
    """hm_yolo_nms takes any row count (general.py:611-703 does): few survivors, more survivors than the in-LDS sort holds
    (22680 > 16384: sorted in the workspace) and more than max_nms = 30000 (38000: the best 30000 enter the suppression,
    general.py:679-680); duplicated scores check the tie rule (lower row first).  Exact against the oracle."""
    lib = L.load()
    rng = np.random.default_rng(n + int(frac * 100))
    pred = np.zeros((n, 8), dtype=np.float32)
    pred[:, 0] = rng.uniform(0, 640, n); pred[:, 1] = rng.uniform(0, 640, n)
    pred[:, 2] = rng.uniform(8, 60, n); pred[:, 3] = rng.uniform(8, 60, n)
    keep = rng.uniform(size=n) < frac
    pred[:, 4] = np.where(keep, rng.uniform(0.5, 1.0, n), rng.uniform(0.0, 0.2, n)).astype(np.float32)
    pred[:, 5:] = rng.uniform(0.55, 1.0, (n, 3)).astype(np.float32)
    pred[1::7, 4:] = pred[0:-1:7, 4:][:len(pred[1::7])]                 # equal scores in neighbouring rows
    pt = torch.from_numpy(pred)
    ws = torch.empty(lib.hm_nms_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    dets = torch.zeros(300, 6, device=DEV)
    count = torch.zeros(1, dtype=torch.int32, device=DEV)
    pd = pt.to(DEV)
    for agnostic, mask in ((1, 0b111), (0, 0b011)):
        L.check(lib.hm_yolo_nms(pd.data_ptr(), n, 3, 0.25, 0.35, mask, agnostic, 300, None, dets.data_ptr(), count.data_ptr(),
                                ws.data_ptr(), ws.numel(), L.current_stream()), "hm_yolo_nms")
        k = int(count.item())
        ref = yolo_ref.non_max_suppression(pt[None], 0.25, 0.35, [c for c in range(3) if (mask >> c) & 1], bool(agnostic))[0]
        assert k == len(ref) and torch.equal(dets[:k].cpu(), ref)
    with pytest.raises(L.HipLibraryError):
        L.check(lib.hm_yolo_nms(pd.data_ptr(), n, 3, 0.25, 0.35, 7, 1, 300, None, dets.data_ptr(), count.data_ptr(),
                                ws.data_ptr(), 1024, L.current_stream()), "hm_yolo_nms")          # workspace too small


def test_batched_frames_equal_single_frame_passes(engine):
    """Three 540x960 frames in one batched pass give the same predictions and detections as three single passes."""
    frames = [synth.frame_u8(540, 960, seed=30 + i).to(DEV) for i in range(3)]
    singles, dets1 = [], []
    for f in frames:
        p = engine.forward(f)
        singles.append(p["pred"].clone())
        dets1.append(engine.nms(p, 0.25, 0.35, [0, 1, 2], True).cpu())
    pb = engine.forward(frames)
    torch.cuda.synchronize()
    n = pb["n_pred"]
    for i in range(3):
        assert torch.equal(pb["pred"][i * n:(i + 1) * n], singles[i])
    detsb = engine.nms(pb, 0.25, 0.35, [0, 1, 2], True)
    assert len(detsb) == 3 and all(torch.equal(a.cpu(), b) for a, b in zip(detsb, dets1))
    # frames that are slices of ONE device tensor (how the folder drivers upload a chunk) take the one-launch letterbox
    # (hm_letterbox_batch): same predictions again
    stacked = torch.stack(frames)
    ps = engine.forward([stacked[i] for i in range(3)])
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(ps["pred"][i * n:(i + 1) * n], singles[i])
