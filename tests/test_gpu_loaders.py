"""GPU: the real-checkpoint branches of the loaders, fed with files of the reference's layouts (synthetic arrays inside):
``load_hamer(<...>/checkpoints/hamer.ckpt)`` -- Lightning checkpoint + ``model_config.yaml`` two levels up +
``MANO_RIGHT.pkl`` (chumpy pickle) + ``mano_mean_params.npz`` (hamer/hamer/models/__init__.py:32-47, heads/mano_head.py:53-59,
mano_wrapper.py:12-30) -- and ``Detector(weights=<file.pt>)`` (models/experimental.py:260-266).  Engines built from the
files must give, bit for bit, what the ``synthetic:`` route gives on the same arrays."""
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hamer_yolo_amd import synth
from hamer_yolo_amd.hamer.configs import get_config
from hamer_yolo_amd.hamer.models import load_hamer
from hamer_yolo_amd.hamer.models.hamer import HAMER
from hamer_yolo_amd.hamer.models.mano_wrapper import MANO
from hamer_yolo_amd.yolo.detector import Detector

DECODER_YAML = {"depth": 2, "heads": 4, "mlp_dim": 256, "dim_head": 64, "dropout": 0.0, "emb_dropout": 0.0, "norm": "layer",
                "context_dim": 320, "dim": 256}


def _write_mano_pkl(path, mp):
    """A MANO_RIGHT.pkl in the layout of the licensed file (py2-style pickle, chumpy wrappers around some arrays, the joint
    regressor as a scipy CSC matrix, uint32 kinematic tree with 2^32-1 as the root's parent) holding synthetic arrays."""
    import scipy.sparse
    mods = {n: types.ModuleType(n) for n in ("chumpy", "chumpy.ch", "chumpy.reordering")}

    class Ch(object):
        def __init__(self, x):
            self.x = x
    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    mods["chumpy.ch"].Ch = Ch
    V = mp["v_template"].shape[0]
    kt = np.stack([np.array([4294967295] + synth.MANO_PARENTS[1:], dtype=np.uint32), np.arange(16, dtype=np.uint32)])
    data = {
        "v_template": mp["v_template"].numpy().astype(np.float64),
        "shapedirs": Ch(mp["shapedirs"].numpy().astype(np.float64)),
        "posedirs": mp["posedirs"].numpy().T.reshape(V, 3, 135).astype(np.float64),
        "J_regressor": scipy.sparse.csc_matrix(mp["J_regressor"].numpy().astype(np.float64)),
        "weights": mp["lbs_weights"].numpy().astype(np.float64),
        "kintree_table": kt, "f": mp["faces"].numpy().astype(np.uint32),
        "hands_mean": np.zeros(45), "hands_components": np.eye(45), "bs_style": "lbs", "bs_type": "lrotmin",
    }
    sys.modules.update(mods)
    try:
        with open(path, "wb") as f:
            pickle.dump(data, f, protocol=2)
    finally:
        for n in mods:
            del sys.modules[n]


def test_load_hamer_from_checkpoint_files_equals_synthetic_route(tmp_path):
    import yaml
    cfgh = synth.tiny_config()
    sd = synth.hamer_state_dict(cfgh, seed=4)
    mp = synth.mano_params(seed=4)
    root = tmp_path / "_DATA"
    (root / "hamer_ckpts" / "checkpoints").mkdir(parents=True)
    (root / "data" / "mano").mkdir(parents=True)
    _write_mano_pkl(str(root / "data" / "mano" / "MANO_RIGHT.pkl"), mp)
    np.savez(root / "data" / "mano_mean_params.npz", pose=sd["mano_head.init_hand_pose"][0].numpy(),
             shape=sd["mano_head.init_betas"][0].numpy(), cam=sd["mano_head.init_cam"][0].numpy())
    with open(root / "hamer_ckpts" / "model_config.yaml", "w") as f:
        yaml.safe_dump({"MODEL": {"IMAGE_SIZE": 256, "IMAGE_MEAN": [0.485, 0.456, 0.406], "IMAGE_STD": [0.229, 0.224, 0.225],
                                  "BACKBONE": {"TYPE": "vit"},
                                  "MANO_HEAD": {"TYPE": "transformer_decoder", "IN_CHANNELS": 2048, "TRANSFORMER_DECODER": DECODER_YAML}},
                        "EXTRA": {"FOCAL_LENGTH": 5000},
                        "MANO": {"MODEL_PATH": str(root / "data" / "mano"), "MEAN_PARAMS": str(root / "data" / "mano_mean_params.npz"),
                                 "NUM_HAND_JOINTS": 15, "GENDER": "neutral", "CREATE_BODY_POSE": False}}, f)
    # a Lightning checkpoint: state_dict without the mean-parameter buffers (the loader then reads mano_mean_params.npz),
    # plus hyper-parameters of a package that is not installed
    ck_sd = {k: v for k, v in sd.items() if not k.startswith("mano_head.init_")}
    ck_sd["discriminator.fc.weight"] = torch.zeros(3, 3)             # training-only entries ride along in hamer.ckpt
    path = root / "hamer_ckpts" / "checkpoints" / "hamer.ckpt"
    torch.save({"state_dict": ck_sd, "epoch": 3, "global_step": 10, "pytorch-lightning_version": "1.9"}, str(path))

    model, model_cfg = load_hamer(str(path))
    assert model_cfg.MODEL.BBOX_SHAPE == [192, 256] and model_cfg.EXTRA.FOCAL_LENGTH == 5000
    assert model.mano.faces.shape == (1538, 3)
    for k, v in mp.items():
        if v.dtype == torch.float32:
            np.testing.assert_array_equal(model.mano.params[k].numpy(), v.numpy(), err_msg=k)   # float64 round trip is exact
    model = model.to("cuda").eval()
    cfg2 = get_config(None)
    cfg2.MODEL.MANO_HEAD.TRANSFORMER_DECODER.merge(DECODER_YAML)
    ref = HAMER(cfg2, sd, MANO(mp)).to("cuda").eval()
    img = synth.normalize_crops(synth.crops_u8(3, seed0=40)).cuda()
    out, params = model({"img": img})
    out_ref, params_ref = ref({"img": img})
    torch.cuda.synchronize()
    for k in ("pred_cam", "pred_cam_t", "pred_vertices", "pred_keypoints_3d", "pred_keypoints_2d"):
        assert torch.equal(out[k], out_ref[k]), k
    for k in ("global_orient", "hand_pose", "betas"):
        assert torch.equal(out["pred_mano_params"][k], out_ref["pred_mano_params"][k]), k
    assert torch.isfinite(out["pred_vertices"]).all() and out["pred_vertices"].shape == (3, 778, 3)
    with pytest.raises(FileNotFoundError):
        load_hamer(str(root / "hamer_ckpts" / "checkpoints" / "missing.ckpt"))


def test_detector_from_checkpoint_file_equals_synthetic_route(tmp_path):
    class Opt:
        weights = "synthetic:2:-2.2:0"; imgsz = 640; augment = True; conf_thres = 0.25; iou_thres = 0.35
        classes = [0, 1, 2]; agnostic_nms = True; device = "cuda"; save_path = "./output"
    sd = synth.yolo_state_dict(seed=2, nc=3, obj_bias=-2.2, cls_bias=0.0)
    from hamer_yolo_amd.yolo import arch
    grid = torch.tensor(arch.ANCHORS, dtype=torch.float32)
    sd_file = dict(sd)
    sd_file["model.105.anchor_grid"] = grid.view(3, 1, 3, 1, 1, 2).clone()
    sd_file["model.105.anchors"] = grid.view(3, 3, 2) / torch.tensor([8.0, 16.0, 32.0]).view(3, 1, 1)
    path = str(tmp_path / "yolov7_best.pt")
    torch.save({"model": {k: v.half() if v.is_floating_point() else v for k, v in sd_file.items()}, "epoch": -1, "ema": None}, path)

    class OptF(Opt):
        weights = path
    frame = synth.frame_u8(1080, 1920, seed=1).numpy()
    # a checkpoint stores half weights (yolov7 saves .half()): compare with the direct route on the same half-rounded values
    det_f = Detector(OptF)
    b, lb = det_f.detect(frame)
    assert len(lb[0]) >= 4
    from hamer_yolo_amd.yolo.engine import YoloEngine
    eng_h = YoloEngine({k: v.half().float() for k, v in sd.items()}, nc=3, device="cuda", new_shape=640)
    ph = eng_h.forward(torch.from_numpy(frame).cuda())
    assert torch.equal(ph["pred"], det_f.engine._plan(1080, 1920)["pred"])            # file route == direct route, bit for bit
    # a checkpoint whose anchors were rewritten (autoanchor): box sizes follow the checkpoint, not yolov7.yaml
    sd_big = dict(sd_file)
    sd_big["model.105.anchor_grid"] = 1.5 * sd_file["model.105.anchor_grid"]
    path2 = str(tmp_path / "yolov7_anchors.pt")
    torch.save({"model": {k: v.half() if v.is_floating_point() else v for k, v in sd_big.items()}}, path2)

    class OptA(Opt):
        weights = path2
    det_a = Detector(OptA)
    det_a.detect(frame)
    pa, pf = det_a.engine._plan(1080, 1920)["pred"], det_f.engine._plan(1080, 1920)["pred"]
    assert torch.equal(pa[:, :2], pf[:, :2]) and torch.equal(pa[:, 4:], pf[:, 4:])
    np.testing.assert_allclose(pa[:, 2:4].cpu().numpy(), 1.5 * pf[:, 2:4].cpu().numpy(), rtol=1e-6)


def test_fp16_overflow_is_detected_at_load_and_rescaled_or_falls_back_to_bf16():
    """ADVICE r2 / VERDICT r3 item 4: the default operand type is fp16 (65504 ceiling, no overflow detection in the kernels).
    HAMER.to() runs one calibration forward; a checkpoint whose activations do not fit (here: a GELU input pushed to 1e5 by the
    fc1 bias of block 0) makes it measure every activation class with a bf16 engine and rebuild the fp16 engine with powers of
    two folded into the weights -- fp16 operands are kept.  Token merging has no prescale path: there the old fallback to
    bfloat16 operands remains."""
    from hamer_yolo_amd.hamer.configs import get_config
    from hamer_yolo_amd.hamer.models.mano_wrapper import MANO
    cfg = synth.tiny_config()
    sd = synth.hamer_state_dict(cfg, seed=0)
    ok = HAMER(get_config(None), sd, MANO.synthetic(0), hamer_cfg=cfg).to("cuda")
    assert ok.dtype == torch.float16 and ok._engine.prescale is None      # the synthetic weights fit
    bad = dict(sd)
    bad["backbone.blocks.0.mlp.fc1.bias"] = sd["backbone.blocks.0.mlp.fc1.bias"] + 1e5
    m = HAMER(get_config(None), bad, MANO.synthetic(0), hamer_cfg=cfg)
    with pytest.warns(UserWarning, match="rescaled by powers of two"):
        m.to("cuda")
    assert m.dtype == torch.float16 and m._engine.dtype == torch.float16
    pre = m._engine.prescale
    assert pre["blocks"][0]["gelu"] >= 4 and all(b["gelu"] == 0 for b in pre["blocks"][1:])     # 1e5 -> <= 4096: 2^-5; nothing else moved
    img = synth.normalize_crops(synth.crops_u8(2, seed0=1)).cuda()
    out, _ = m({"img": img})
    assert torch.isfinite(out["pred_vertices"]).all()
    mt = HAMER(get_config(None), bad, MANO.synthetic(0), hamer_cfg=cfg, token_merge=[1] * cfg.vit.depth)
    with pytest.warns(UserWarning, match="falling back to bfloat16"):
        mt.to("cuda")
    assert mt.dtype == torch.bfloat16
    out, _ = mt({"img": img})
    assert torch.isfinite(out["pred_vertices"]).all()
