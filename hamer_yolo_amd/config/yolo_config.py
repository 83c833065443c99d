"""YOLOv7 detector options (reference: config/yolo_config.py:4-15; same attribute names)."""
import os

from .config import opt


class Config:
    weights = os.environ.get("YOLO_WEIGHTS", os.path.join(opt.root_dir, "checkpoints/yolov7_best.pt"))
    imgsz = 640
    augment = True        # ignored, as in the reference: TracedModel.forward drops it (torch_utils.py:371-374)
    conf_thres = 0.25
    iou_thres = 0.35
    classes = [0, 1, 2]
    agnostic_nms = True
    device = "cuda"
    save_path = "./output"


yolo_opt = Config()
