"""How fast do N decoder threads turn 1080p .bmp files into frame slots -- page-locked (torch pin_memory) against ordinary host
memory, and how does the rate scale with the thread count?  Usage: python tools/probes/decode_rate.py"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, ".")
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from PIL import Image
from hamer_yolo_amd import infer, synth
torch.cuda.init()
root = tempfile.mkdtemp(dir="/dev/shm")
try:
    for i in range(8):
        Image.fromarray(synth.frame_u8(1080, 1920, seed=i).numpy()[:, :, ::-1]).save(os.path.join(root, f"f{i}.bmp"))
    paths = [os.path.join(root, f"f{i % 8}.bmp") for i in range(64)]
    pinned = torch.empty((64, 1080, 1920, 3), dtype=torch.uint8, pin_memory=True).numpy()
    plain = np.empty((64, 1080, 1920, 3), dtype=np.uint8)
    plain[:] = 0
    for name, dst in (("page-locked", pinned), ("ordinary", plain)):
        for nt in (1, 4, 8, 16):
            with ThreadPoolExecutor(nt) as ex:
                list(ex.map(lambda i: infer._read_bmp24(paths[i], lambda shp: dst[i]), range(16)))          # warm (scratch buffers)
                t0 = time.perf_counter()
                list(ex.map(lambda i: infer._read_bmp24(paths[i], lambda shp: dst[i]), range(16)))
                t16 = (time.perf_counter() - t0) * 1e3
                t0 = time.perf_counter()
                list(ex.map(lambda i: infer._read_bmp24(paths[i], lambda shp: dst[i]), range(64)))
                t64 = (time.perf_counter() - t0) * 1e3
            print(f"{name:12s} {nt:2d} threads: 16 frames {t16:6.1f} ms, 64 frames {t64:6.1f} ms ({t64 / 64:.2f} ms per frame)")
finally:
    shutil.rmtree(root, ignore_errors=True)

# round 4, second question: ONE readinto of the pixel block straight into the slot (no scratch, no row flip: the flip would move to
# the device), page-locked against ordinary memory
def raw_read(path, dst):
    with open(path, "rb", buffering=0) as f:
        f.seek(54)
        view = memoryview(dst.reshape(-1))
        got = 0
        while got < len(view):
            k = f.readinto(view[got:])
            if not k:
                break
            got += k
    return got

root = tempfile.mkdtemp(dir="/dev/shm")
try:
    for i in range(8):
        Image.fromarray(synth.frame_u8(1080, 1920, seed=i).numpy()[:, :, ::-1]).save(os.path.join(root, f"f{i}.bmp"))
    paths = [os.path.join(root, f"f{i % 8}.bmp") for i in range(64)]
    for name, dst in (("page-locked", pinned), ("ordinary", plain)):
        for nt in (1, 8, 16):
            with ThreadPoolExecutor(nt) as ex:
                list(ex.map(lambda i: raw_read(paths[i], dst[i]), range(16)))
                t0 = time.perf_counter()
                list(ex.map(lambda i: raw_read(paths[i], dst[i]), range(16)))
                t16 = (time.perf_counter() - t0) * 1e3
                t0 = time.perf_counter()
                list(ex.map(lambda i: raw_read(paths[i], dst[i]), range(64)))
                t64 = (time.perf_counter() - t0) * 1e3
            print(f"single readinto, {name:12s} {nt:2d} threads: 16 frames {t16:6.1f} ms, 64 frames {t64:6.1f} ms ({t64 / 64:.2f} ms per frame)")
finally:
    shutil.rmtree(root, ignore_errors=True)
