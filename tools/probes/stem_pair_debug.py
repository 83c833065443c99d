"""Where hm_conv2d_stem_pair (one launch) differs from the two launches, if anywhere: counts by tile position, pixel and channel."""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch, numpy as np
from hamer_yolo_amd import lib as L
DEV = "cuda"
lib = L.load()


def run(n, H, W, dt=torch.float16):
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
    for two in (1, 0):
        mid = torch.full((n, H, W, 32), 7.0, dtype=dt, device=DEV)
        y = torch.full((n, Ho, Wo, 64), -3.0, dtype=dt, device=DEV)
        a = L.ConvArgs(xd.data_ptr(), w0d.data_ptr(), mid.data_ptr(), b0d.data_ptr(), zeros.data_ptr(), n, H, W, 8, 32, 3, 1, 8, 32, 128, 1, 0, code, None, 0, None, 0)
        b = L.ConvArgs(mid.data_ptr(), w1d.data_ptr(), y.data_ptr(), b1d.data_ptr(), zeros.data_ptr(), n, H, W, 32, 64, 3, 2, 32, 64, 320, 1, 0, code, None, 0, None, 0)
        L.check(lib.hm_set_option(L.HM_OPT_CONV_STEM_PAIR, two))
        L.check(lib.hm_conv2d_stem_pair(C.byref(a), C.byref(b), L.current_stream()), "pair")
        torch.cuda.synchronize()
        outs.append(y.cpu().float())
    L.check(lib.hm_set_option(L.HM_OPT_CONV_STEM_PAIR, 0))
    y2, y1 = outs
    bad = (y1 != y2)
    print(f"n={n} H={H} W={W} {dt}: {int(bad.sum())} of {bad.numel()} differ, max |d| {float((y1 - y2).abs().max()):.4g}")
    if bad.any():
        idx = bad.nonzero()
        print("  images", sorted(set(idx[:, 0].tolist())))
        print("  oy % 8", np.bincount(idx[:, 1].numpy() % 8, minlength=8).tolist(), " oy", sorted(set(idx[:, 1].tolist()))[:40])
        print("  ox % 16", np.bincount(idx[:, 2].numpy() % 16, minlength=16).tolist(), " ox", sorted(set(idx[:, 2].tolist()))[:40])
        print("  channel", np.bincount(idx[:, 3].numpy(), minlength=64).tolist())
        for r in idx[:6].tolist():
            print("   ", r, float(y1[tuple(r)]), float(y2[tuple(r)]))


for shape in ((1, 16, 32), (1, 32, 64), (2, 64, 96), (1, 50, 70)):
    run(*shape)
run(1, 32, 64, torch.bfloat16)
