"""A few launches of hm_conv2d_stem_pair at the folder driver's size (48 frames of 384 x 640), for rocprofv3 --pmc passes.
Usage: python tools/probes/stem_pair_run.py [frames=48] [launches=6]"""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch
from hamer_yolo_amd import lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
H, W, dt, DEV = 384, 640, torch.float16, "cuda"
lib = L.load()
g = torch.Generator().manual_seed(1)
x = torch.zeros(n, H, W, 8, dtype=dt)
x[..., :3] = torch.rand(n, H, W, 3, generator=g).to(dt)
w0 = torch.zeros(32, 128); w0[:, :72] = ((torch.rand(32, 3, 3, 8, generator=g) - 0.5) * 0.8 * (torch.arange(8) < 3)).reshape(32, 72)
w1 = torch.zeros(64, 320); w1[:, :288] = (torch.rand(64, 288, generator=g) - 0.5) * 0.25
b0, b1 = torch.rand(32, generator=g) - 0.5, torch.rand(64, generator=g) - 0.5
xd, w0d, w1d, b0d, b1d = x.to(DEV), w0.to(dt).to(DEV), w1.to(dt).to(DEV), b0.to(DEV), b1.to(DEV)
zeros = torch.zeros(64, dtype=torch.uint8, device=DEV)
mid = torch.empty((n, H, W, 32), dtype=dt, device=DEV)
y = torch.empty((n, H // 2, W // 2, 64), dtype=dt, device=DEV)
a = L.ConvArgs(xd.data_ptr(), w0d.data_ptr(), mid.data_ptr(), b0d.data_ptr(), zeros.data_ptr(), n, H, W, 8, 32, 3, 1, 8, 32, 128, 1, 0, L.HM_DTYPE_F16, None, 0, None, 0)
b = L.ConvArgs(mid.data_ptr(), w1d.data_ptr(), y.data_ptr(), b1d.data_ptr(), zeros.data_ptr(), n, H, W, 32, 64, 3, 2, 32, 64, 320, 1, 0, L.HM_DTYPE_F16, None, 0, None, 0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for i in range(reps):
    L.check(lib.hm_conv2d_stem_pair(C.byref(a), C.byref(b), L.current_stream()), "pair")
    ev[i + 1].record()
torch.cuda.synchronize()
print("us per launch:", [round(ev[i].elapsed_time(ev[i + 1]) * 1e3, 1) for i in range(reps)])
