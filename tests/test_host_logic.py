"""CPU tests: host-side mirror of the reference interface, C-ABI export surface, crop oracle KATs,
and the world_size-2 shard path on gloo.  No GPU compute is launched here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from hamer_yolo_amd import lib as L
from hamer_yolo_amd import shard, synth
from hamer_yolo_amd.hamer.configs import get_config
from hamer_yolo_amd.hamer.datasets.utils import expand_to_aspect_ratio, gen_trans_from_patch_cv
from hamer_yolo_amd.hamer.utils.renderer import cam_crop_to_full, custom_cam_crop_to_full
from oracle import crop_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C ABI surface
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "hamer_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(hm_[a-z0-9_]+)\s*\(", hdr)))
    assert "hm_gemm" in declared and "hm_hamer_forward" in declared
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/hamer_hip.h but not exported"
    assert lib.hm_version() == L.HM_VERSION == int(re.search(r"#define HM_VERSION (\d+)", hdr).group(1))


def test_round3_entry_points_reject_bad_arguments_without_gpu():
    lib = L.load()
    assert lib.hm_letterbox_batch(None, 0, 1, None, None, None, 0, None) != 0
    assert lib.hm_yolo_decode_batch(None, 24, None, 0, 12, 20, 3, 32.0, None, 1, 15120, None) != 0
    assert lib.hm_tome_merge_metric(None, 80, 0, None, None, None, None, None, 1, 192, 8, 1280, None) != 0
    assert lib.hm_conv_splitk_bytes(None) == 0
    a = L.ConvArgs(1, 1, 1, 1, 1, 16, 12, 20, 256, 256, 3, 1, 256, 256, 2304, 1, 0, L.HM_DTYPE_F16)     # a 12x20 map, K = 2304: 4 ranges
    assert lib.hm_conv_splitk_bytes(C.byref(a)) == 4 * 16 * 12 * 20 * 256 * 4
    b = L.ConvArgs(1, 1, 1, 1, 1, 16, 96, 160, 64, 64, 3, 1, 64, 64, 576, 1, 0, L.HM_DTYPE_F16)         # a large map: never split
    assert lib.hm_conv_splitk_bytes(C.byref(b)) == 0
    one = L.ConvArgs(1, 1, 1, 1, 1, 1, 12, 20, 256, 256, 3, 1, 256, 256, 2304, 1, 0, L.HM_DTYPE_F16)    # the rule looks at ONE image: the same ranges for 1 frame
    assert lib.hm_conv_splitk_bytes(C.byref(one)) == 4 * 1 * 12 * 20 * 256 * 4


def test_options_are_explicit_setters_not_environment_reads(monkeypatch):
    """launch paths must not call getenv (a stray variable or a test's setenv would change which kernel runs).
    The switches live behind hm_set_option / hm_get_option; only the two start-up tuning defaults are read from the
    environment, once."""
    lib = L.load()
    n_opt = lib.hm_option_count()
    for key in range(n_opt):                              # HM_OPT_COUNT
        assert lib.hm_get_option(key) == 0
    assert lib.hm_get_option(n_opt) == 0 and lib.hm_set_option(n_opt, 1) != 0
    assert lib.hm_set_option(99, 1) != 0 and lib.hm_set_option(L.HM_OPT_FP8P_GRID, 12) != 0 and lib.hm_set_option(0, -1) != 0
    with L.option(L.HM_OPT_FP8P_GRID, 40):
        assert lib.hm_get_option(L.HM_OPT_FP8P_GRID) == 40
    assert lib.hm_get_option(L.HM_OPT_FP8P_GRID) == 0
    src = "".join(open(os.path.join(ROOT, "hamer_yolo_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "hamer_yolo_amd", "csrc")))
    assert sorted(set(re.findall(r'getenv\("(\w+)"\)', src))) == ["HM_GEMM_VARIANT", "HM_PX_GRID"]


def test_option_table_of_the_binding_is_the_headers_enum():
    """lib.py names the HM_OPT_* keys itself; the header's enum is the truth.  Names, values and the count must agree
    (load() also checks the count against the built library)."""
    hdr = open(os.path.join(ROOT, "include", "hamer_hip.h")).read()
    keys = re.findall(r"^\s*(HM_OPT_[A-Z0-9_]+)\s*=\s*(\d+)", hdr, flags=re.M)
    count = int(dict(keys).pop("HM_OPT_COUNT"))
    named = [(k, int(v)) for k, v in keys if k != "HM_OPT_COUNT"]
    assert named == [(n, i) for i, n in enumerate(L.OPTION_NAMES)]
    assert count == len(L.OPTION_NAMES) == L.load().hm_option_count()
    for i, n in enumerate(L.OPTION_NAMES):
        assert getattr(L, n) == i


def test_persistent_grid_option_is_not_sticky():
    """HM_OPT_PX_GRID used to be copied into the start-up default on first use, so an A/B arm that set it back to 0
    kept the previous arm's grid.  hm_gemm_px_grid reports the grid launch_px takes (host arithmetic only)."""
    lib = L.load()
    if os.environ.get("HM_PX_GRID"):
        pytest.skip("HM_PX_GRID is set: the start-up default is not the built-in one")
    assert lib.hm_gemm_px_grid(720, 256) == 248 and lib.hm_gemm_px_grid(960, 256) == 248      # qkv / fc1 at 64 hands
    assert lib.hm_gemm_px_grid(1020, 256) == 256                                              # all CUs when that saves a round
    assert lib.hm_gemm_px_grid(180, 256) == 184                                               # fewer tiles than workgroups: rounded UP to the XCDs
    with L.option(L.HM_OPT_PX_GRID, 16):
        assert lib.hm_gemm_px_grid(720, 256) == 16
    assert lib.hm_get_option(L.HM_OPT_PX_GRID) == 0 and lib.hm_gemm_px_grid(720, 256) == 248


def test_library_is_loaded_behind_torchs_hip_runtime():
    """Round 4: loaded BEFORE torch, libhamer_hip.so maps the system's libamdhip64 and the process ends up with two HIP runtimes
    (torch sees the GPU, the kernels here report "no ROCm-capable device": __graft_entry__.build() followed by smoke() in one
    process).  lib.load() therefore imports torch first; checked in a fresh interpreter."""
    code = "import sys; from hamer_yolo_amd import lib as L; assert 'torch' not in sys.modules; L.load(); assert 'torch' in sys.modules; print('ok')"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_abi_rejects_bad_arguments_without_gpu():
    lib = L.load()
    assert lib.hm_gemm(None, None) != 0
    assert b"null" in lib.hm_last_error_string()
    a = L.GemmArgs(1, 1, 1, None, None, 16, 16, 96, 96, 96, 16, 0, 0, 0, 0)
    assert lib.hm_gemm(C.byref(a), None) != 0 and b"multiple of 64" in lib.hm_last_error_string()
    assert lib.hm_vit_attention(16, 16, 1, 100, 4, 80, 0.1, 0, None) != 0   # tokens != 192
    assert lib.hm_layernorm(16, 16, 16, 16, 0, 4, 4098, 1e-6, None) != 0
    assert lib.hm_hamer_workspace_bytes(None, 4) == 0


def test_host_tensor_is_refused():
    from hamer_yolo_amd import ops
    with pytest.raises(L.HipLibraryError):
        ops.layernorm(torch.zeros(4, 256), torch.ones(256), torch.zeros(256), 1e-5)


def test_crop_box_helper_known_answers():
    lib = L.load()
    b = L.CropBox()
    assert lib.hm_crop_box_from_bbox(250.0, 200.0, 256.0, 0, 256, C.byref(b)) == 0
    assert (b.m0, b.m4, b.x0, b.y0, b.flip) == (1.0, 1.0, 122 * 1024 + 16, 72 * 1024 + 16, 0)
    assert lib.hm_crop_box_from_bbox(100.0, 50.0, 512.0, 1, 256, C.byref(b)) == 0
    assert (b.m0, b.m4, b.flip) == (2.0, 2.0, 1) and b.x0 == (100 - 256) * 1024 + 16
    assert lib.hm_crop_box_from_bbox(1.0, 1.0, 0.0, 0, 256, C.byref(b)) != 0


# ------------------------------------------------------------------ reference helper KATs
def test_expand_to_aspect_ratio_and_crop_size_rule():
    np.testing.assert_allclose(expand_to_aspect_ratio(np.array([300.0, 300.0]), [192, 256]), [300.0, 400.0])
    np.testing.assert_allclose(expand_to_aspect_ratio(np.array([100.0, 400.0]), [192, 256]), [300.0, 400.0])
    # square box of side s: S = 2.5 * s * 4/3 = 10 s / 3
    cx, cy, S = crop_ref.bbox_to_center_size(100.0, 200.0, 160.0, 260.0)
    assert (cx, cy) == (130.0, 230.0) and abs(S - 200.0) < 1e-9
    np.testing.assert_allclose(crop_ref.expand_to_aspect_ratio(np.array([300.0, 300.0]), [192, 256]), [300.0, 400.0])


def test_affine_matches_closed_form_and_oracle():
    M = gen_trans_from_patch_cv(130.0, 230.0, 200.0, 200.0, 256, 256, 1.0, 0)
    a = 256 / 200.0
    np.testing.assert_allclose(M, [[a, 0, 128 - a * 130.0], [0, a, 128 - a * 230.0]], rtol=1e-12)
    np.testing.assert_allclose(crop_ref.gen_trans_from_patch(130.0, 230.0, 200.0, 200.0, 256, 256), M, rtol=1e-9, atol=1e-9)
    # float32 control points: a non-representable centre moves the matrix by ~1e-6, identically in both
    M1 = gen_trans_from_patch_cv(130.123456789, 230.987654321, 123.456, 123.456, 256, 256)
    M2 = crop_ref.gen_trans_from_patch(130.123456789, 230.987654321, 123.456, 123.456, 256, 256)
    np.testing.assert_allclose(M1, M2, rtol=1e-9, atol=1e-7)


def test_oracle_warp_identity_and_border():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(300, 400, 3), dtype=np.uint8)
    M = crop_ref.gen_trans_from_patch(200.0, 150.0, 256.0, 256.0, 256, 256)
    out = crop_ref.warp_affine_u8(img, M, 256, 256)
    assert np.array_equal(out, img[150 - 128:150 + 128, 200 - 128:200 + 128])
    M = crop_ref.gen_trans_from_patch(10.0, 10.0, 256.0, 256.0, 256, 256)          # mostly outside: zeros
    out = crop_ref.warp_affine_u8(img, M, 256, 256)
    assert out[:117, :].max() == 0 and out[:, :117].max() == 0 and np.array_equal(out[118:, 118:], img[:138, :138])
    # half-pixel shift: exact average of two neighbours with cv2's rounding (sum + 512) >> 10
    M = crop_ref.gen_trans_from_patch(200.5, 150.0, 256.0, 256.0, 256, 256)
    out = crop_ref.warp_affine_u8(img, M, 256, 256).astype(np.int64)
    a = img[22:278, 72:328].astype(np.int64); b = img[22:278, 73:329].astype(np.int64)
    assert np.array_equal(out, (512 * a + 512 * b + 512) >> 10)


def test_cam_crop_to_full_known_answers():
    cam = torch.tensor([[1.0, 0.0, 0.0]])
    t = custom_cam_crop_to_full(cam, torch.tensor([[320.0, 240.0]]), torch.tensor([256.0]), torch.tensor([[640.0, 480.0]]),
                                5000.0, 5000.0, 320.0, 240.0)
    np.testing.assert_allclose(t.numpy(), [[0.0, 0.0, 2 * 5000.0 / 256.0]], rtol=1e-6)
    t2 = cam_crop_to_full(cam, torch.tensor([[320.0, 240.0]]), torch.tensor([256.0]), torch.tensor([[640.0, 480.0]]), 5000.0)
    np.testing.assert_allclose(t2.numpy(), t.numpy(), rtol=1e-6)
    t3 = custom_cam_crop_to_full(cam, torch.tensor([[420.0, 240.0]]), torch.tensor([256.0]), torch.tensor([[640.0, 480.0]]),
                                 1000.0, 500.0, 320.0, 240.0, depth_refine=2.0)
    np.testing.assert_allclose(t3.numpy(), [[2 * 100.0 / 1000.0, 0.0, 2.0]], rtol=1e-5, atol=1e-7)


def test_rodrigues_roundtrip():
    from hamer_yolo_amd.infer import axis_angle_to_rotation_matrix_torch, matrix_to_axis_angle
    aa = synth.uniform("aa", (64, 3), 1.7, seed=4)   # |aa| <= 2.95 < pi: the log map is unique
    aa[0] = 0.0
    aa[1] = torch.tensor([3.14159, 0.0, 0.0])
    R = axis_angle_to_rotation_matrix_torch(aa).numpy()
    back = matrix_to_axis_angle(R).reshape(-1, 3)
    R2 = axis_angle_to_rotation_matrix_torch(torch.from_numpy(back)).numpy()
    np.testing.assert_allclose(R2, R, atol=2e-6)
    np.testing.assert_allclose(back[2:], aa[2:].numpy(), atol=2e-5)


def test_cfg_node_surface():
    cfg = get_config(None)
    assert cfg.MODEL.IMAGE_SIZE == 256 and cfg.EXTRA.FOCAL_LENGTH == 5000 and cfg.MANO.NUM_HAND_JOINTS == 15
    assert cfg.MODEL.get("BBOX_SHAPE", None) is None and "BBOX_SHAPE" not in cfg.MODEL
    assert {k.lower() for k in dict(cfg.MANO)} >= {"model_path", "mean_params"}


def test_hamer_refuses_cpu():
    from hamer_yolo_amd.hamer.models.hamer import HAMER
    from hamer_yolo_amd.hamer.models.mano_wrapper import MANO
    m = HAMER(get_config(None), synth.hamer_state_dict(synth.tiny_config(), seed=0), MANO.synthetic(0), hamer_cfg=synth.tiny_config())
    with pytest.raises(L.HipLibraryError):
        m.to("cpu")
    with pytest.raises(L.HipLibraryError):
        m({"img": torch.zeros(1, 3, 256, 256)})


def test_synth_is_deterministic_and_bf16_representable():
    a = synth.uniform("x", (1000,), 0.3, seed=5)
    b = synth.uniform("x", (1000,), 0.3, seed=5, chunk=77)
    assert torch.equal(a, b) and float(a.abs().max()) <= 0.3 and a.std() > 0.15
    sd = synth.hamer_state_dict(synth.tiny_config(), seed=2, bf16_representable=True)
    w = sd["backbone.blocks.0.mlp.fc1.weight"]
    assert torch.equal(w, w.bfloat16().float())
    assert not torch.equal(sd["backbone.blocks.0.mlp.fc1.bias"], sd["backbone.blocks.0.mlp.fc1.bias"].bfloat16().float())


# ------------------------------------------------------------------ shard path (gloo, world size 2)
def test_shard_range():
    assert [shard.shard_range(1024, r, 8) for r in (0, 7)] == [(0, 128), (896, 1024)]
    assert [shard.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard.shard_range(3, 3, 4) == (3, 3)


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from hamer_yolo_amd import shard, synth
rank, local, world = shard.init_distributed("gloo")
cfg = synth.tiny_config()
sd0 = synth.hamer_state_dict(cfg, seed=3) if rank == 0 else None
import torch.distributed as dist
sd = shard.broadcast_state_dict(sd0, "cpu", src=0, half_dtype=torch.float16)
ref = synth.hamer_state_dict(cfg, seed=3)
assert list(sd.keys()) == list(ref.keys())
n16 = 0
for k in ref:
    if k.endswith(shard.GEMM_WEIGHT_SUFFIXES):      # GEMM matrices travel rounded to the operand type
        assert sd[k].dtype == torch.float16 and torch.equal(sd[k], ref[k].half()), k
        n16 += 1
    else:
        assert sd[k].dtype == torch.float32 and torch.equal(sd[k], ref[k]), k
assert n16 == 1 + 4 * cfg.vit.depth + cfg.dec.depth
# even split: 6 hands over `world` ranks; uneven: 7 hands (BASELINE configs[3] is 1024 over N, e.g. 1024 over 3)
for n_total in (6, 7, 1):
    lo, hi = shard.shard_range(n_total, rank, world)
    B = hi - lo
    out = {"rotmats": torch.full((B, 16, 3, 3), float(rank)), "betas": torch.arange(lo, hi).float()[:, None].repeat(1, 10),
           "pred_cam": torch.zeros(B, 3)}
    full = shard.gather_mano(shard.pack_mano(out), dst=0, n_total=n_total)
    if rank == 0:
        assert full.shape == (n_total, 157)
        assert torch.equal(full[:, 144], torch.arange(n_total).float())
        per = (n_total + world - 1) // world
        assert torch.equal(full[:, 0], (torch.arange(n_total) // per).float())
    else:
        assert full is None
eq = shard.gather_mano(torch.full((3, 157), float(rank)), dst=0)            # equal shards, no n_total
assert (eq.shape == (3 * world, 157) and eq[-1, 0] == world - 1) if rank == 0 else eq is None
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_broadcast_and_gather_on_gloo(tmp_path, world):
    """The N > 1 path on CPU (gloo): flat-buffer weight broadcast and the gather of MANO parameters for even and uneven
    shards, at world size 2 and 3."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = 29500 + (os.getpid() % 2000) + world
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_reference_pickled_yolo_checkpoint_loads_without_its_classes(golden_dir):
    """attempt_load (experimental.py:260-283) on a checkpoint pickled by the reference's own Model / Conv / RepConv /
    SPPCSPC / IDetect classes (tools/gen_golden_yolo_ckpt.py): none of those packages exists here, the class-free
    unpickler must still recover exactly Model.state_dict(), names and nc -- and run no foreign code."""
    import os
    import sys
    import numpy as np
    import torch
    from hamer_yolo_amd.utils.checkpoint import Stub, load_checkpoint
    from hamer_yolo_amd.yolo.detector import checkpoint_state_dict
    assert "yolo.yolov7.models.yolo" not in sys.modules and "models.yolo" not in sys.modules
    g = np.load(os.path.join(golden_dir, "yolo_tiny_ckpt.npz"))
    ck = load_checkpoint(os.path.join(golden_dir, "yolo_tiny_ckpt.pt"))
    assert isinstance(ck["model"], Stub) and ck["epoch"] == 7 and ck["ema"] is None
    assert list(ck["model"].names) == list(g["names"]) and ck["model"].yaml["nc"] == int(g["nc"])
    sd = checkpoint_state_dict(ck)
    keys = list(g["keys"])
    assert sorted(sd) == sorted(keys)
    for i, k in enumerate(keys):
        assert sd[k].dtype == torch.float32
        np.testing.assert_array_equal(sd[k].numpy(), g[f"t{i}"], err_msg=k)     # fp16 -> fp32 is exact
    det = [k for k in sd if k.endswith(".m.0.weight")]
    assert sd[det[0]].shape[0] // 3 - 5 == 3
    # wrappers around plain state dicts still work
    assert checkpoint_state_dict({"model": dict(sd)}).keys() == sd.keys()
    with pytest.raises(TypeError):
        checkpoint_state_dict({"model": {"a": 1}})


def test_lightning_checkpoint_with_foreign_objects(tmp_path):
    """A Lightning-style .ckpt whose hyper-parameters hold objects of packages that are not installed."""
    import sys
    import types
    import torch
    from hamer_yolo_amd.hamer.models import _read_state_dict
    mod = types.ModuleType("yacs_like.config")
    sys.modules["yacs_like"] = types.ModuleType("yacs_like")
    sys.modules["yacs_like.config"] = mod

    class CfgNode(dict):
        pass
    CfgNode.__module__ = "yacs_like.config"
    CfgNode.__qualname__ = "CfgNode"
    mod.CfgNode = CfgNode
    sd = {"backbone.pos_embed": torch.arange(6.0).reshape(1, 2, 3), "mano_head.decpose.weight": torch.ones(2, 2)}
    path = str(tmp_path / "hamer.ckpt")
    try:
        torch.save({"state_dict": sd, "hyper_parameters": {"cfg": CfgNode(MODEL=CfgNode(IMAGE_SIZE=256))}, "epoch": 1}, path)
    finally:
        del sys.modules["yacs_like.config"], sys.modules["yacs_like"]
    out = _read_state_dict(path)
    assert out.keys() == sd.keys() and all(torch.equal(out[k], sd[k]) for k in sd)


def test_rootnet_bbox_and_k_known_answers():
    """process_bbox (rootnet/preprocessing.py:166-188) and calculate_k (Model_RGB.py:494-498): product mirror and oracle
    against hand-derived values.  Box 100x60 at (200, 150) in a 640x480 image: sanitize shrinks w, h by one (x2 = x1 + w - 1),
    the square side is max(w, h) * 1.5."""
    import numpy as np
    from hamer_yolo_amd.rootnet.preprocessing import process_bbox, sanitize_bbox
    from oracle import rootnet_ref as RR
    b = process_bbox([200.0, 150.0, 100.0, 60.0], 640, 480, (256, 256), 1.5)
    np.testing.assert_allclose(b, [200 + 49.5 - 74.25, 150 + 29.5 - 74.25, 148.5, 148.5], rtol=0, atol=1e-4)
    np.testing.assert_allclose(RR.process_bbox([200.0, 150.0, 100.0, 60.0], 640, 480, (256, 256), 1.5), b, atol=1e-4)
    assert sanitize_bbox([10.0, 10.0, 0.0, 50.0], 640, 480) is None and RR.sanitize_bbox([10.0, 10.0, 0.0, 50.0], 640, 480) is None
    np.testing.assert_allclose(sanitize_bbox([-20.0, 400.0, 100.0, 200.0], 640, 480), [0, 400, 99, 79])
    # k = sqrt(0.3 * 0.3 * fx * fy / (w * h)): fx = fy = 1000, 150 px square -> 0.3 * 1000 / 150 = 2
    assert abs(RR.calculate_k([0, 0, 150.0, 150.0], 1000.0, 1000.0) - 2.0) < 1e-6


def test_unpicklers_never_resolve_code_execution_gadgets(tmp_path):
    """Both checkpoint readers work from exact (module, name) allow-lists.  Pickles that reach for os.system, builtins.eval,
    torch._utils._import_dotted_name (an import-by-name helper inside an otherwise trusted module) or numpy's runstring must
    load as inert stubs / be refused, and nothing they name may run."""
    import io
    import pickle
    import zipfile
    import torch
    from hamer_yolo_amd.hamer.models.mano_wrapper import _RestrictedUnpickler
    from hamer_yolo_amd.utils.checkpoint import Stub, _Unpickler, load_checkpoint
    marker = tmp_path / "pwned"

    def gadget(module, name, *args):
        """GLOBAL module name; args...; REDUCE -- what ``__reduce__`` returning (callable, args) pickles to."""
        out = io.BytesIO()
        out.write(b"\x80\x02c" + module.encode() + b"\n" + name.encode() + b"\n")
        out.write(pickle.dumps(tuple(args), protocol=2)[2:-1])      # the args tuple without PROTO / STOP
        out.write(b"R.")
        return out.getvalue()

    payloads = [
        gadget("os", "system", f"touch {marker}"),
        gadget("posix", "system", f"touch {marker}"),
        gadget("builtins", "eval", f"open({str(marker)!r}, 'w').close()"),
        gadget("builtins", "exec", f"open({str(marker)!r}, 'w').close()"),
        gadget("torch._utils", "_import_dotted_name", "os.system"),
        gadget("numpy.testing._private.utils", "runstring", f"open({str(marker)!r}, 'w').close()", {}),
        gadget("numpy_x", "anything", 1),
        gadget("collectionsfoo", "OrderedDict"),
        gadget("torch.storage", "_load_from_bytes", b"\x80\x02N."),
    ]
    for raw in payloads:
        obj = _Unpickler(io.BytesIO(raw)).load()
        assert isinstance(obj, Stub), raw
        with pytest.raises(pickle.UnpicklingError):
            _RestrictedUnpickler(io.BytesIO(raw), encoding="latin1").load()
    # two-step gadget: resolve os.system through the import helper, then call it
    two = (b"\x80\x02ctorch._utils\n_import_dotted_name\n" + pickle.dumps(("os.system",), protocol=2)[2:-1] + b"R"
           + pickle.dumps((f"touch {marker}",), protocol=2)[2:-1] + b"R.")
    assert isinstance(_Unpickler(io.BytesIO(two)).load(), Stub)
    # the same through torch.load's zip container
    path = str(tmp_path / "evil.pt")
    torch.save({"state_dict": {"w": torch.ones(2)}}, path)
    with zipfile.ZipFile(path) as zin, zipfile.ZipFile(str(tmp_path / "evil2.pt"), "w") as zout:
        for item in zin.infolist():
            data = zin.read(item.filename)
            if item.filename.endswith("data.pkl"):
                data = payloads[0]
            zout.writestr(item, data)
    assert isinstance(load_checkpoint(str(tmp_path / "evil2.pt")), Stub)
    assert not marker.exists()


def test_engine_config_follows_checkpoint_and_rejects_unsupported_heads():
    """HAMER derives the engine geometry from model_config.yaml + the checkpoint's tensor shapes: a config the HIP
    forward does not implement must raise, not produce silently wrong outputs (mano_head.py:28-31,:81,:86)."""
    from hamer_yolo_amd.hamer.configs import get_config
    from hamer_yolo_amd.hamer.models.hamer import engine_config
    tiny = synth.tiny_config()
    sd = synth.hamer_state_dict(tiny, seed=0)
    cfg = get_config(None)
    with pytest.raises(ValueError, match="does not match the checkpoint"):
        engine_config(cfg, sd)                                   # default yaml says depth 6 / dim 1024, the tensors say 2 / 256
    cfg.MODEL.MANO_HEAD.TRANSFORMER_DECODER.merge({"depth": 2, "heads": 4, "mlp_dim": 256, "dim_head": 64, "context_dim": 320, "dim": 256})
    assert engine_config(cfg, sd) == tiny
    for key, val in (("TRANSFORMER_INPUT", "mean_shape"), ("IEF_ITERS", 3), ("JOINT_REP", "aa"), ("TYPE", "mlp")):
        bad = get_config(None)
        bad.MODEL.MANO_HEAD.TRANSFORMER_DECODER.merge({"depth": 2, "heads": 4, "mlp_dim": 256, "context_dim": 320, "dim": 256})
        bad.MODEL.MANO_HEAD[key] = val
        with pytest.raises(ValueError, match="unsupported"):
            engine_config(bad, sd)
    bad = get_config(None)
    bad.MODEL.BACKBONE.TYPE = "resnet"
    with pytest.raises(ValueError, match="BACKBONE"):
        engine_config(bad, sd)


def test_detect_anchors_come_from_the_checkpoint(golden_dir):
    """IDetect decodes with its anchor_grid buffer (yolo.py:164); autoanchor may have rewritten it for a custom detector."""
    import os
    from hamer_yolo_amd.utils.checkpoint import load_checkpoint
    from hamer_yolo_amd.yolo import arch
    from hamer_yolo_amd.yolo.detector import attempt_load, checkpoint_state_dict
    from hamer_yolo_amd.yolo.engine import detect_anchors
    assert detect_anchors({}) == [[float(v) for v in l] for l in arch.ANCHORS]
    grid = torch.tensor([[10., 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]])
    assert detect_anchors({"model.105.anchor_grid": grid.view(3, 1, 3, 1, 1, 2)}) == grid.tolist()
    unit = grid.view(3, 3, 2) / torch.tensor([8., 16, 32]).view(3, 1, 1)
    assert detect_anchors({"model.105.anchors": unit}) == grid.tolist()
    with pytest.raises(ValueError):
        detect_anchors({"model.105.anchor_grid": torch.zeros(2, 1, 3, 1, 1, 2)})
    sd, nc, names = attempt_load(os.path.join(golden_dir, "yolo_tiny_ckpt.pt"))
    assert nc == 3 and names == ["left", "right", "other"]
    assert detect_anchors(sd) == [[12.0, 16.0, 19.0, 36.0, 40.0, 28.0], [36.0, 75.0, 76.0, 55.0, 72.0, 146.0], [142.0, 110.0, 192.0, 243.0, 459.0, 401.0]]


def test_reference_import_names_resolve_to_this_package():
    """the import lines of hamer/infer.py:15-44 and d_infer.py:21 work unchanged after ``compat.install()`` and give
    the same module objects as the package's own names (checked in a fresh interpreter so the aliases do not leak into this
    test session).  The finder defers to the path finder: a caller's own ``config.py`` and ``model/`` package keep resolving to the
    caller's code, and ``model.rootnet`` still falls through to this package."""
    import subprocess
    import sys
    code = r"""
import sys
import hamer_yolo_amd.compat as compat
assert "yolo" not in sys.modules and not any(type(f).__name__ == "_AliasFinder" for f in sys.meta_path)   # importing installs nothing
compat.install()
from yolo.detector import Detector
from hamer.models import load_hamer
from hamer.models.mano_wrapper import MANO
from hamer.utils.renderer import cam_crop_to_full, custom_cam_crop_to_full
from hamer.utils.geometry import perspective_projection
from hamer.datasets.utils import convert_cvimg_to_tensor, expand_to_aspect_ratio
from config.yolo_config import yolo_opt
from config.hamer_config import hamer_opt
from model.rootnet.Model_RGB import get_model
import hamer_yolo_amd.yolo.detector as D, hamer_yolo_amd.hamer.models as M, hamer_yolo_amd.config.yolo_config as Y
import yolo.detector, hamer.models, config.yolo_config
assert yolo.detector is D and hamer.models is M and config.yolo_config is Y
assert Detector is D.Detector and load_hamer is M.load_hamer and yolo_opt is Y.yolo_opt
assert yolo_opt.conf_thres == 0.25 and yolo_opt.iou_thres == 0.35 and yolo_opt.imgsz == 640
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
    # a caller with its own config.py and model/ package: those win, model.rootnet still resolves to this package
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "config.py"), "w").write("MINE = 1\n")
        os.makedirs(os.path.join(td, "model"))
        open(os.path.join(td, "model", "__init__.py"), "w").write("")
        open(os.path.join(td, "model", "other.py"), "w").write("X = 2\n")
        code2 = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import hamer_yolo_amd.compat as compat
compat.install()
import config, model.other
assert config.MINE == 1 and model.other.X == 2 and not config.__name__.startswith("hamer_yolo_amd")
from model.rootnet.Model_RGB import get_model
import hamer_yolo_amd.rootnet.Model_RGB as R
assert get_model is R.get_model
from yolo.detector import Detector
print("ok")
""" % (root, td)
        r = subprocess.run([sys.executable, "-c", code2], cwd=td, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]

    # the documented caller has the REFERENCE checkout on sys.path, whose root holds hamer/, yolo/ and config/
    # ({hamer,yolo}_config.py inside).  install() must not be a silent no-op there: the aliases win, with a warning.
    with tempfile.TemporaryDirectory() as td:
        for pkg, files in (("yolo", {"detector.py": "Detector = 'REFERENCE'\n"}), ("hamer", {"models.py": "load_hamer = 'REFERENCE'\n"}),
                           ("config", {"yolo_config.py": "yolo_opt = 'REFERENCE'\n", "hamer_config.py": "hamer_opt = 'REFERENCE'\n"})):
            os.makedirs(os.path.join(td, pkg))
            open(os.path.join(td, pkg, "__init__.py"), "w").write("")
            for fn, body in files.items():
                open(os.path.join(td, pkg, fn), "w").write(body)
        code3 = r"""
import sys, warnings
sys.path.insert(0, %r); sys.path.insert(0, %r)
import hamer_yolo_amd.compat as compat
compat.install()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    from yolo.detector import Detector
    from hamer.models import load_hamer
    from config.yolo_config import yolo_opt
    from config.hamer_config import hamer_opt
import hamer_yolo_amd.yolo.detector as D, hamer_yolo_amd.hamer.models as M, hamer_yolo_amd.config.yolo_config as Y
assert Detector is D.Detector and load_hamer is M.load_hamer and yolo_opt is Y.yolo_opt and hamer_opt != 'REFERENCE'
shadowed = sorted(str(x.message).split("'")[1] for x in w if issubclass(x.category, ImportWarning) and "shadowed" in str(x.message))
assert shadowed == ["config", "hamer", "yolo"], shadowed
print("ok")
""" % (root, td)
        r = subprocess.run([sys.executable, "-c", code3], cwd=td, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_tome_schedule_parser_matches_oracle():
    from hamer_yolo_amd.engine import parse_tome_r
    from oracle import tome_ref as T
    for n, r in ((32, (8, -1)), (6, (8, -1)), (12, 5), (4, [7, 2]), (24, (6, 0.5)), (8, (3, 1))):
        assert parse_tome_r(n, r) == T.parse_r(n, r), (n, r)
    assert sum(parse_tome_r(32, (8, -1))) == 241
