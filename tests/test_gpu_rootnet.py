"""GPU parity of the RootNet root-depth path (reference rootnet/Model_RGB.py, d_infer.py:1275-1276)
against oracle/rootnet_ref.py (restated ResNet-34; parity unpinned against torchvision, see the oracle's header)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hamer_yolo_amd import lib as L
from hamer_yolo_amd import synth
from hamer_yolo_amd.rootnet.engine import RootNetEngine
from oracle import rootnet_ref as RR

DEV = "cuda"


@pytest.mark.parametrize("k,stride,cin,cout,resid", [(7, 2, 8, 64, False), (3, 1, 64, 64, True), (3, 2, 64, 128, False),
                                                     (1, 2, 128, 256, False), (3, 1, 512, 512, True)])
def test_conv_relu_and_residual_epilogues(k, stride, cin, cout, resid):
    """hm_conv2d_nhwc with act = 2 (ReLU), an optional 16-bit identity added before it, and the 7x7 stem."""
    import ctypes as C
    N, H, W = 2, 20, 24
    x = synth.uniform("rx", (N, cin, H, W), 1.0, 0.0, seed=k).half()
    w = synth.uniform("rw", (cout, cin, k, k), (3.0 / (cin * k * k)) ** 0.5, 0.0, seed=cin).half()
    b = synth.uniform("rb", (cout,), 0.2, 0.0, seed=3)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    idn = synth.uniform("ri", (N, cout, Ho, Wo), 1.0, 0.0, seed=9).half()
    ref = F.conv2d(x.float(), w.float(), b, stride, k // 2)
    ref = F.relu(ref + idn.float()) if resid else F.relu(ref)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    kp = (k * k * cin + 63) // 64 * 64
    wk = torch.zeros(cout, kp, dtype=torch.float16)
    wk[:, :k * k * cin] = w.permute(0, 2, 3, 1).reshape(cout, -1)
    wd, bd = wk.to(DEV), b.to(DEV)
    idd = idn.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = torch.empty(N, Ho, Wo, cout, device=DEV, dtype=torch.float16)
    zeros = torch.zeros(64, dtype=torch.uint8, device=DEV)
    a = L.ConvArgs(L.ptr(xd), L.ptr(wd), L.ptr(y), L.ptr(bd), L.ptr(zeros), N, H, W, cin, cout, k, stride, cin, cout, kp, 2, 0,
                   L.HM_DTYPE_F16, L.ptr(idd) if resid else None, cout if resid else 0)
    L.check(L.load().hm_conv2d_nhwc(C.byref(a), L.current_stream()), "hm_conv2d_nhwc")
    got = y.permute(0, 3, 1, 2).float().cpu()
    assert (got >= 0).all()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=4e-3 * max(1.0, ref.abs().max().item()), rtol=2e-3)


def test_backbone_and_depth_vs_oracle():
    net, root = synth.rootnet_state_dict(seed=0)
    eng = RootNetEngine(net, root)
    img = synth.normalize_crops(synth.crops_u8(3, seed0=70))
    kv = torch.tensor([1.7, 0.9, 2.4])
    with torch.no_grad():
        feat_ref = RR.backbone(net, img)
        d_ref = RR.root_depth(root, feat_ref, kv).reshape(-1)
    feat = eng.features(img.to(DEV)).permute(0, 3, 1, 2).float().cpu()
    assert feat.shape == feat_ref.shape == (3, 512, 8, 8)
    scale = feat_ref.abs().max().item()
    assert scale > 0.1                                    # the synthetic net keeps its activations alive through 33 convs
    assert (feat - feat_ref).abs().max().item() < 0.03 * scale and (feat - feat_ref).abs().mean().item() < 0.004 * scale
    d = eng.forward(img.to(DEV), kv).cpu()
    np.testing.assert_allclose(d.numpy(), d_ref.numpy(), rtol=5e-3, atol=1e-3)


def test_estimate_root_depth_custom_and_d_infer_hook():
    """The d_infer.py flow (:1275-1276): depth = sar.estimate_root_depth_custom(image, K, bbox) -> estimate_from_rgb(...,
    depth_refine=depth): the depth against the oracle pipeline on the same frame, and the camera translation it implies."""
    from hamer_yolo_amd.rootnet.Model_RGB import get_model
    sar = get_model()
    frame = synth.frame_u8(720, 1280, seed=33).numpy()
    K = np.array([[900.0, 0, 640], [0, 880.0, 360], [0, 0, 1]], np.float32)
    bbox = [500.0, 260.0, 690.0, 470.0]
    depth = sar.estimate_root_depth_custom(frame, K, bbox)
    net, root = synth.rootnet_state_dict(seed=0)
    ref, _ = RR.estimate_root_depth(net, root, frame, K, bbox)
    assert abs(depth - ref) < 5e-3 * abs(ref) + 1e-3, (depth, ref)
    # box clipped by the image border still works, an empty one raises
    assert np.isfinite(sar.estimate_root_depth_custom(frame, K, [1200.0, 650.0, 1400.0, 800.0]))
    with pytest.raises(ValueError):
        sar.estimate_root_depth_custom(frame, K, [100.0, 100.0, 100.0, 300.0])
    from hamer_yolo_amd.d_infer import hamer_inference

    class _Cfg:
        ckpt_path = "synthetic:0"; model_cfg = None; use_onnx = False; onnx_path = None
    hi = hamer_inference(_Cfg)
    out, _ = hi.estimate_from_rgb(frame, [["right", bbox]], K, depth_refine=depth)
    np.testing.assert_allclose(out["pred_cam_t_full"][0, 2].item(), depth, rtol=1e-5)


def test_d_infer_batch_driver(tmp_path):
    """d_infer.process_batch_manopara (d_infer.py:1223-1318) on a folder: every saved hand carries the RootNet depth as its
    camera z, and refuses to run without intrinsics."""
    from PIL import Image
    from hamer_yolo_amd import d_infer
    from hamer_yolo_amd.rootnet.Model_RGB import get_model

    class _Cfg:
        ckpt_path = "synthetic:0"; model_cfg = None; use_onnx = False; onnx_path = None

    class _Det:
        def __init__(self, dets): self.dets = dets
        def detect(self, image): return [None], [self.dets]
    frame = synth.frame_u8(480, 640, seed=8).numpy()
    (tmp_path / "rgb").mkdir()
    Image.fromarray(frame[:, :, ::-1]).save(tmp_path / "rgb" / "a.png")
    dets = [["right", [100.0, 120.0, 260.0, 300.0]], ["left", [380.0, 200.0, 520.0, 330.0]]]
    K = np.array([[600.0, 0, 320], [0, 610.0, 240], [0, 0, 1]], np.float32)
    sar = get_model()
    hi = d_infer.hamer_inference(_Cfg)
    d_infer.process_batch_manopara(str(tmp_path / "rgb"), str(tmp_path / "out"), K, hamer=hi, detector=_Det(dets), sar=sar)
    rec = np.load(tmp_path / "out" / "a.npy", allow_pickle=True).item()
    for label, box in dets:
        depth = sar.estimate_root_depth_custom(frame, K, box)
        assert rec[label]["is_right"] == (label == "right")
        np.testing.assert_allclose(rec[label]["cam_t"][2], depth, rtol=1e-5)
    with pytest.raises(ValueError):
        d_infer.process_batch_manopara(str(tmp_path / "rgb"), str(tmp_path / "out2"), None, hamer=hi, detector=_Det(dets), sar=sar)


def test_d_infer_chunked_driver_equals_one_hand_calls(tmp_path):
    """The folder driver of d_infer runs ONE RootNet forward and ONE HaMeR forward per chunk of frames; per hand it must
    save what the reference's flow (one estimate_root_depth_custom + one estimate_from_rgb(depth_refine=...) per hand,
    d_infer.py:1268-1304) produces: three frames of two sizes, several hands each, one box that has no RootNet patch (dropped,
    as the reference's per-hand try/except drops it), one frame without detections (no file)."""
    from PIL import Image
    from hamer_yolo_amd import d_infer
    from hamer_yolo_amd.infer import hand_record
    from hamer_yolo_amd.rootnet.Model_RGB import get_model

    class _Cfg:
        ckpt_path = "synthetic:0"; model_cfg = None; use_onnx = False; onnx_path = None

    frames = {"a": synth.frame_u8(480, 640, seed=8).numpy(), "b": synth.frame_u8(480, 640, seed=9).numpy(),
              "c": synth.frame_u8(720, 1280, seed=10).numpy(), "d": synth.frame_u8(480, 640, seed=11).numpy()}
    dets = {"a": [["right", [100.0, 120.0, 260.0, 300.0]], ["left", [380.0, 200.0, 520.0, 330.0]]],
            "b": [["left", [30.0, 40.0, 200.0, 260.0]], ["right", [600.0, 100.0, 640.0, 101.0]], ["right", [300.0, 220.0, 420.0, 400.0]]],
            "c": [["right", [900.0, 300.0, 1150.0, 560.0]]], "d": []}
    # (b's second box is 1 pixel high: it has area for HaMeR's crop but clips to nothing in RootNet's box arithmetic)

    class _Det:
        def detect(self, image):
            for k, f in frames.items():
                if f.shape == image.shape and np.array_equal(f, image):
                    return [None], [dets[k]]
            raise AssertionError("unknown frame")
    (tmp_path / "rgb").mkdir()
    for k, f in frames.items():
        Image.fromarray(f[:, :, ::-1]).save(tmp_path / "rgb" / f"{k}.png")
    K = np.array([[600.0, 0, 320], [0, 610.0, 240], [0, 0, 1]], np.float32)
    sar = get_model()
    hi = d_infer.hamer_inference(_Cfg)
    assert not sar.valid_boxes(dets["b"], 640, 480)[1] and sar.valid_boxes(dets["b"], 640, 480)[[0, 2]].all()
    d_infer.process_batch_manopara(str(tmp_path / "rgb"), str(tmp_path / "out"), K, hamer=hi, detector=_Det(), sar=sar)
    assert sorted(p.name for p in (tmp_path / "out").iterdir()) == ["a.npy", "b.npy", "c.npy"]
    for k in "abc":
        rec = np.load(tmp_path / "out" / f"{k}.npy", allow_pickle=True).item()
        want = {"left": None, "right": None}
        for det in dets[k]:
            try:
                depth = sar.estimate_root_depth_custom(frames[k], K, det[1])
            except ValueError:
                continue
            out, _ = hi.estimate_from_rgb(frames[k], [det], K, depth_refine=depth)
            want[det[0]] = hand_record(out, det[0] == "right", 0)
        for label in ("left", "right"):
            assert (rec[label] is None) == (want[label] is None), (k, label)
            if want[label] is None:
                continue
            assert rec[label]["is_right"] == want[label]["is_right"]
            for key in ("betas", "theta", "cam_t"):
                np.testing.assert_allclose(rec[label][key], want[label][key], rtol=1e-4, atol=2e-4, err_msg=f"{k} {label} {key}")   # (B = 1 and B = n forwards differ in fp32 summation order)
