"""CPU tests of the folder driver's host logic (hamer_yolo_amd/infer.py ``iter_folder_results`` /
``process_batch_manopara``; reference: hamer/infer.py:1246-1318, the serial per-file / per-hand loop): detector passes
and HaMeR batches that no longer share a chunk, exact batch sizes across frame boundaries, rank sharding of the folder,
the page-locked ring's reuse rule.  The models are stubs whose outputs are functions of the frame's pixels and the box,
so any hand attributed to the wrong file, frame or batch position changes a number.  No GPU, no HIP call."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest
import torch

from hamer_yolo_amd import infer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_bmp(path, arr):
    h, w, _ = arr.shape
    stride = (w * 3 + 3) & ~3
    rows = np.zeros((h, stride), np.uint8)
    rows[:, :w * 3] = arr[::-1].reshape(h, w * 3)
    head = b"BM" + struct.pack("<IHHI", 54 + stride * h, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, stride * h, 0, 0, 0, 0)
    with open(path, "wb") as f:
        f.write(head + rows.tobytes())


def _frame(i, h=24, w=32):
    rng = np.random.default_rng(1000 + i)
    fr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    fr[0, 0, 0] = (i * 5) % 7                      # number of hands the stub detector reports (0..6)
    return fr


def _rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)


class StubDetector:
    """detect_frames over host tensors: frame[0,0,0] boxes, coordinates from the first pixels; label from the parity."""

    def __init__(self):
        self.pass_sizes = []

    def detect_frames(self, frames):
        self.pass_sizes.append(len(frames))
        out = []
        for fr in frames:
            a = fr.numpy()
            n = int(a[0, 0, 0])
            dets = []
            for j in range(n):
                x1, y1 = float(a[1, j, 0] % 8), float(a[1, j, 1] % 8)
                dets.append(['right' if (int(a[1, j, 2]) + j) % 2 else 'left', [x1, y1, x1 + 4.0 + j, y1 + 3.0]])
            if n == 6:
                dets[2][1] = [5.0, 5.0, 5.0, 5.0]                  # a box clipped to nothing: the driver must drop it
            out.append(dets)
        return None, out


class StubHamer:
    device = torch.device("cpu")

    def __init__(self):
        self.batch_sizes = []

    def estimate_from_frames(self, frames, dets_lists, k_real=None, depth_refine=None):
        rows = []
        for fr, dl in zip(frames, dets_lists):
            key = float(fr.numpy().astype(np.float64).mean())
            for label, (x1, y1, x2, y2) in dl:
                rows.append((key, x1, y1, x2, y2, label))
        n = len(rows)
        self.batch_sizes.append(n)
        betas = torch.tensor([[r[0] + k * r[1] + r[3] for k in range(10)] for r in rows], dtype=torch.float32)
        go = torch.tensor(np.stack([_rz(0.01 * (r[0] % 50) + 0.1 * r[2])[None] for r in rows]))
        hp = torch.tensor(np.stack([np.stack([_rz(0.02 * (k + 1) + 0.003 * r[4]) for k in range(15)]) for r in rows]))
        return {"pred_mano_params": {"betas": betas, "global_orient": go, "hand_pose": hp},
                "pred_cam_t_full": torch.tensor([[r[1], r[2], r[0]] for r in rows], dtype=torch.float32),
                "do_flip": torch.tensor([0.0 if r[5] == 'right' else 1.0 for r in rows])}, None


@pytest.fixture(scope="module")
def folder(tmp_path_factory):
    d = tmp_path_factory.mktemp("rgb")
    for i in range(23):
        _write_bmp(str(d / f"f{i:03d}.bmp"), _frame(i) if i not in (9, 10, 17) else _frame(i, 20, 28))    # another frame size in between
    (d / "f011.bmp").write_bytes(b"BMnot a bitmap at all")                                              # unreadable
    return str(d)


def _collect(folder, **kw):
    det, ham = StubDetector(), StubHamer()
    st = {}
    res = {}
    for path, dets, hands in infer.iter_folder_results(infer._list_images(folder), ham, det, stats=st, **kw):
        assert path not in res
        assert len(dets) == hands["betas"].shape[0] == hands["pose_hand"].shape[0]
        res[path] = (dets, hands)
    return res, st, det, ham


def _same(a, b):
    assert sorted(a) == sorted(b)
    for k in a:
        assert a[k][0] == b[k][0], k
        for name in a[k][1]:
            np.testing.assert_array_equal(a[k][1][name], b[k][1][name], err_msg=f"{k} {name}")


def test_results_are_per_file_whatever_the_pass_and_batch_sizes(folder):
    base, st, det, ham = _collect(folder)                          # defaults: everything fits one batch
    files = infer._list_images(folder)
    want = {}
    for i, p in enumerate(files):
        if i == 11:
            continue
        n = int(((i * 5) % 7))
        n_ok = n - 1 if n == 6 else n
        if n_ok:
            want[p] = n_ok
    assert {p: len(v[0]) for p, v in base.items()} == want and list(base) == sorted(base)        # path order, empty files skipped
    assert st["images"] == 23 and st["frames"] == len(want) and st["hands"] == sum(want.values()) == sum(ham.batch_sizes)
    for kw in (dict(hands_per_forward=5, frames_per_step=3, det_frames=7), dict(hands_per_forward=1, frames_per_step=1),
               dict(hands_per_forward=16, frames_per_step=2, det_frames=2, in_flight=3, decode_threads=2),
               dict(hands_per_forward=7, frames_per_step=16, in_flight=1)):
        got, st2, det2, ham2 = _collect(folder, balance_tail=False, **kw)
        _same(base, got)
        H = kw["hands_per_forward"]
        # every forward but the last is EXACTLY hands_per_forward hands, across frame and pass boundaries; the last takes the
        # remainder (up to H + H/4 rather than a full batch and a sliver)
        assert all(b == H for b in ham2.batch_sizes[:-1]) and 0 < ham2.batch_sizes[-1] <= H + H // 4, ham2.batch_sizes
        assert sum(ham2.batch_sizes) == st2["hands"] == st["hands"]
        # the default: the same forwards of exactly H hands, then -- once the folder's last hands are queued -- the rest in
        # EQUAL parts (within one hand) of at most 1.5 H, one or two per stream, so that the streams finish together
        got3, st3, det3, ham3 = _collect(folder, **kw)
        _same(base, got3)
        sizes, n_str = ham3.batch_sizes, kw.get("in_flight", 2)
        k = 0
        while k < len(sizes) and sizes[k] == H:
            k += 1
        tail = sizes[k:]
        assert sum(sizes) == st3["hands"] == st["hands"] and len(tail) <= 2 * n_str, sizes
        assert not tail or (max(tail) - min(tail) <= 1 and 0 < max(tail) <= H + H // 2) or (n_str == 1 and len(tail) == 1 and tail[0] <= H + H // 4), sizes
        assert det3.pass_sizes == det2.pass_sizes
        # a detector pass holds frames of ONE size: first pass <= frames_per_step, later ones <= det_frames
        assert det2.pass_sizes[0] <= kw["frames_per_step"] and max(det2.pass_sizes) <= max(kw.get("det_frames", kw["frames_per_step"]), kw["frames_per_step"])
        assert sum(det2.pass_sizes) == 22


@pytest.mark.parametrize("world", [2, 3, 5])
def test_rank_shards_partition_the_folder(folder, world):
    """SURVEY 8e / VERDICT r3 item 1: rank r takes files r, r + N, ...; union of the ranks' outputs == the single-rank output,
    no file twice, counts add up."""
    base, st, _, _ = _collect(folder, hands_per_forward=8)
    files = infer._list_images(folder)
    assert sorted(sum((infer.shard_paths(files, r, world) for r in range(world)), [])) == files
    union, hands, images = {}, 0, 0
    for r in range(world):
        got, s, _, _ = _collect(folder, hands_per_forward=8, rank=r, world=world)
        assert set(got) <= set(infer.shard_paths(files, r, world)) and not (set(got) & set(union))
        union.update(got)
        hands += s["hands"]; images += s["images"]
    _same(base, union)
    assert hands == st["hands"] and images == 23
    with pytest.raises(ValueError):
        infer.shard_paths(files, world, world)


def test_ring_slot_is_not_rewritten_before_its_upload(monkeypatch):
    """a decoder must not overwrite a page-locked slot whose asynchronous upload may still be pending.
    The ring keeps the event recorded behind a slot's last upload and waits for it before handing the slot out again."""
    waited = []

    class Ev:
        def __init__(self, name): self.name = name
        def synchronize(self): waited.append(self.name)

    ring = infer._FrameRing(4, pin=False)
    t0, v0 = ring.slot(1, (2, 3, 3))
    assert waited == [] and t0.shape == (2, 3, 3) and v0.shape == (2, 3, 3)
    ring.events[1] = Ev("upload of file 1")
    ring.slot(2, (2, 3, 3))
    assert waited == []                                        # another slot: nothing to wait for
    t1, _ = ring.slot(5, (2, 3, 3))                            # file 5 -> slot 1 again: waits for file 1's upload first
    assert waited == ["upload of file 1"] and t1.data_ptr() == t0.data_ptr()
    assert ring.slot(0, (4, 4, 3)) is not None and ring.slot(0, (5, 5, 3)) is None      # a third frame size: no more slots
    # and two generators never share a ring (nothing cached at class level)
    assert not hasattr(infer._FrameRing, "_cache") and infer._FrameRing(4, pin=False).bufs == {}


_WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
from hamer_yolo_amd import infer, shard
from test_folder_pipeline import StubDetector, StubHamer
rank, local, world = shard.init_distributed("gloo")
st = infer.process_batch_manopara(sys.argv[2], sys.argv[3], None, hamer=StubHamer(), detector=StubDetector(), hands_per_forward=6)
infer.gather_job_stats(st)                       # rank / world came from the process group
assert len(st["per_rank"]) == world and st["per_rank"][rank]["hands"] == st["hands"]
assert st["global_images"] == 23 and st["global_hands"] == sum(r["hands"] for r in st["per_rank"])
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", st["global_frames"], st["global_hands"])
'''


def test_two_ranks_write_one_folder_on_gloo(folder, tmp_path):
    """process_batch_manopara under torch.distributed (gloo, world 2): every rank writes its own share, the union of the
    .npy files is what one process writes, bit for bit; one all_gather of counts after the job."""
    single = tmp_path / "single"
    st = infer.process_batch_manopara(folder, str(single), None, hamer=StubHamer(), detector=StubDetector(), hands_per_forward=6)
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    out = tmp_path / "sharded"
    port = 31000 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, folder, str(out)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert outs[0].strip().endswith(f"ok {st['frames']} {st['hands']}")
    assert sorted(os.listdir(out)) == sorted(os.listdir(single)) and len(os.listdir(single)) == st["frames"]
    for f in os.listdir(single):
        a, b = np.load(single / f, allow_pickle=True).item(), np.load(out / f, allow_pickle=True).item()
        for label in ("left", "right"):
            assert (a[label] is None) == (b[label] is None)
            if a[label] is not None:
                for k in a[label]:
                    np.testing.assert_array_equal(a[label][k], b[label][k])


def test_one_bad_frame_or_hand_costs_only_itself(folder, capsys):
    """The reference wraps every file and every hand in try / except (hamer/infer.py:1306-1316).  Here a detector pass that raises
    is redone file by file and a HaMeR batch that raises frame by frame: the bad file / the bad frame's hands are dropped with a
    message, everything else comes out exactly as in a clean run, nothing twice."""
    base, _, _, _ = _collect(folder, hands_per_forward=8, frames_per_step=4, det_frames=6)
    files = infer._list_images(folder)

    class BadPassDetector(StubDetector):
        def detect_frames(self, frames):                  # file 5's pixels poison whatever pass they ride in
            if any(int(fr.numpy()[0, 0, 0]) == (5 * 5) % 7 and np.array_equal(fr.numpy(), _frame(5)) for fr in frames):
                raise RuntimeError("corrupt frame")
            return super().detect_frames(frames)

    det, ham, got = BadPassDetector(), StubHamer(), {}
    for p, d, h in infer.iter_folder_results(files, ham, det, hands_per_forward=8, frames_per_step=4, det_frames=6):
        assert p not in got
        got[p] = (d, h)
    assert "corrupt frame" in capsys.readouterr().out
    want = {k: v for k, v in base.items() if k != files[5]}
    _same(want, got)

    class BadHandHamer(StubHamer):
        def estimate_from_frames(self, frames, dets_lists, k_real=None, depth_refine=None):
            if any(np.array_equal(fr.numpy(), _frame(8)) for fr in frames):
                raise ValueError("Invalid bbox format")
            return super().estimate_from_frames(frames, dets_lists, k_real, depth_refine)

    got = {}
    for p, d, h in infer.iter_folder_results(files, BadHandHamer(), StubDetector(), hands_per_forward=8, frames_per_step=4, det_frames=6):
        assert p not in got
        got[p] = (d, h)
    assert "Invalid bbox format" in capsys.readouterr().out
    _same({k: v for k, v in base.items() if k != files[8]}, got)
