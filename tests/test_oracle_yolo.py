"""CPU: detector-path oracle against fixtures from the reference's own YOLOv7 code
(tools/gen_golden_yolo.py) and the hand-derived known answers of SURVEY.md 8a."""
import ctypes as C
import os

import numpy as np
import torch

from hamer_yolo_amd import lib as L
from hamer_yolo_amd import synth
from hamer_yolo_amd.yolo import arch, fuse
from oracle import yolo_ref


def test_graph_matches_reference_inventory():
    layers = arch.yolov7_layers()
    kinds = [k for _, k, _ in layers]
    # 79 Conv, 15 Concat, 5 MP, 2 Upsample, 1 SPPCSPC, 3 RepConv (+ detect)
    assert (kinds.count("conv"), kinds.count("concat"), kinds.count("mp"), kinds.count("up"), kinds.count("sppcspc"),
            kinds.count("repconv"), kinds.count("detect")) == (79, 15, 5, 2, 1, 3, 1)
    specs = arch.conv_specs(layers, 3, 3)
    assert len(specs) == 92                                    # 92 Conv2d after fusion
    ch = arch.channels(layers)
    assert (ch[102], ch[103], ch[104], ch[105]) == (256, 512, 1024, 24)
    flops = 0
    hw = {0: (384, 640)}
    assert sum(co * ci * k * k for co, ci, k, s in specs.values()) + sum(co for co, *_ in specs.values()) > 36e6


def test_fuse_matches_reference_model_fuse(golden_dir):
    g = np.load(os.path.join(golden_dir, "yolo_fuse.npz"))
    sd = synth.yolo_state_dict(seed=int(g["seed"]), nc=int(g["nc"]))
    specs = arch.conv_specs(arch.yolov7_layers(), 3, int(g["nc"]))
    fused = fuse.fuse_state_dict(sd, specs)
    for name in ("model.0.conv", "model.37.conv", "model.50.conv", "model.103.rbr_reparam", "model.105.m.1"):
        w, b = fused[name]
        np.testing.assert_allclose(b.numpy(), g[name + ".bias"], atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(w.flatten()[::97].numpy(), g[name + ".wsub"], atol=1e-6, rtol=1e-6)


def test_forward_nms_scale_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "yolo_forward.npz"))
    nc = int(g["nc"])
    layers = arch.yolov7_layers()
    fused = fuse.fuse_state_dict(synth.yolo_state_dict(seed=int(g["seed"]), nc=nc), arch.conv_specs(layers, 3, nc))
    x = synth.frame_u8(384, 640, seed=int(g["frame_seed"])).permute(2, 0, 1).float()[None] / 255.0
    with torch.no_grad():
        pred, _ = yolo_ref.yolo_forward(layers, fused, x, nc, arch.ANCHORS, arch.STRIDES)
    assert pred.shape == (1, 15120, 8)
    np.testing.assert_allclose(pred[0, ::9].numpy(), g["pred_rows"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(pred[0].double().sum(0).numpy(), g["pred_sum"], rtol=1e-5)
    # non_max_suppression of the reference on ITS prediction vs the oracle on the same candidates
    cand = torch.from_numpy(g["pred_cand"])[None]
    dets = yolo_ref.non_max_suppression(cand, 0.25, 0.35, classes=[0, 1, 2], agnostic=True)[0]
    np.testing.assert_array_equal(dets.numpy(), g["dets"])
    dets1 = yolo_ref.non_max_suppression(cand, 0.25, 0.35, classes=[1], agnostic=False)[0]
    np.testing.assert_array_equal(dets1.numpy(), g["dets_cls1"])
    assert len(g["dets"]) == 300 and 0 < len(g["dets_cls1"]) < 300 and (g["dets_cls1"][:, 5] == 1).all()
    b = torch.from_numpy(g["boxes"])
    np.testing.assert_array_equal(yolo_ref.scale_coords((384, 640), b.clone(), (1080, 1920, 3)).numpy(), g["scaled_1080"])
    np.testing.assert_array_equal(yolo_ref.scale_coords((448, 640), b.clone(), (565, 848, 3)).numpy(), g["scaled_565"])


def test_letterbox_known_answers_and_host_plan():
    g = yolo_ref.letterbox_geometry(1080, 1920)
    assert (g["nw"], g["nh"], g["top"], g["left"], g["out_h"], g["out_w"]) == (640, 360, 12, 0, 384, 640)
    g2 = yolo_ref.letterbox_geometry(565, 848)
    assert (g2["nw"], g2["nh"], g2["top"], g2["out_h"], g2["out_w"]) == (640, 426, 11, 448, 640)
    lib = L.load()
    for (h, w) in ((1080, 1920), (565, 848), (640, 640), (480, 640), (333, 517), (2160, 3840), (64, 1000)):
        p = L.LetterboxPlan()
        assert lib.hm_letterbox_plan_make(h, w, 640, 32, C.byref(p)) == 0
        o = yolo_ref.letterbox_geometry(h, w)
        assert (p.new_w, p.new_h, p.top, p.left, p.out_h, p.out_w) == (o["nw"], o["nh"], o["top"], o["left"], o["out_h"], o["out_w"])
        assert p.out_h % 32 == 0 and p.out_w % 32 == 0
        gain = min(p.out_h / h, p.out_w / w)
        assert abs(p.gain - gain) < 1e-6 and abs(p.pad_y - (p.out_h - h * gain) / 2) < 1e-4


def test_resize_identity_and_letterbox_padding():
    img = synth.frame_u8(96, 160, seed=3).numpy()
    assert np.array_equal(yolo_ref.resize_linear_u8(img, 160, 96), img)
    up = yolo_ref.resize_linear_u8(img, 320, 192)
    assert up.shape == (192, 320, 3) and abs(int(up.mean()) - int(img.mean())) <= 1
    chw, g = yolo_ref.letterbox(synth.frame_u8(270, 480, seed=4).numpy())
    assert chw.shape == (3, 384, 640) and (chw[:, :g["top"]] == 114).all() and (chw[:, g["top"] + g["nh"]:] == 114).all()


def test_greedy_nms_basic_properties():
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.0]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95])
    keep = yolo_ref.nms_greedy(boxes, scores, 0.5)
    assert keep.tolist() == [3, 2]
    assert yolo_ref.nms_greedy(boxes, scores, 0.99).tolist() == [3, 1, 2]      # IoU == 1 > 0.99 suppresses the duplicate only
    assert yolo_ref.non_max_suppression(torch.zeros(1, 10, 8))[0].shape == (0, 6)
