"""``d_infer.py`` of the reference is ``infer.py`` plus a root depth from RootNet fed to the camera maths as
``depth_refine`` (reference: hamer/d_infer.py:21,:355,:438-440,:1223-1318 -> renderer.py:47-52): per detected hand
``depth = sar.estimate_root_depth_custom(image, k_real, bbox)`` then ``estimate_from_rgb(image, [bbox], k_real,
depth_refine=depth)``.  Everything else (``hamer_inference``, savers, CLI) is infer.py's."""
import argparse
import os

import numpy as np

from .infer import *  # noqa: F401,F403
from .infer import (FRAMES_PER_STEP, _default_models, _detection_list, _imread_bgr, _list_images, _record_from, hamer_inference,  # noqa: F401
                    hamer_opt, hand_record, iter_folder_results, load_intrinsics, reconstruct_and_save_obj_with_wrapper)
from .rootnet.Model_RGB import get_model  # noqa: F401


def process_batch_manopara(input_folder, output_folder, k_real=None, hamer=None, detector=None, sar=None,
                           frames_per_step: int = FRAMES_PER_STEP, rank=None, world=None, **pipeline):
    """d_infer.py:1223-1318: as infer.py's, with the RootNet depth per hand.  ``k_real`` is required here (the depth is
    metric only with real intrinsics; the reference passes its camera file).  The reference runs one RootNet and one HaMeR
    forward per hand; here the folder goes through infer.py's pipeline with ONE RootNet forward and ONE HaMeR forward per
    batch of hands -- each hand's camera translation still uses its own depth (``depth_refine`` is a per-hand vector in the
    camera step), and the numbers are those of the one-hand calls.  Rank sharding as infer.process_batch_manopara."""
    from .infer import _rank_world
    os.makedirs(output_folder, exist_ok=True)
    if k_real is None:
        raise ValueError("d_infer needs camera intrinsics (k_real)")
    hamer, detector = _default_models(hamer, detector)
    if sar is None:
        sar = get_model()
    rank, world = _rank_world(rank, world)
    stats = {}
    for img_path, detection_list, hands in iter_folder_results(_list_images(input_folder), hamer, detector, k_real, frames_per_step,
                                                               depth_model=sar, rank=rank, world=world, stats=stats, **pipeline):
        file_name = os.path.splitext(os.path.basename(img_path))[0]
        image_results = {'left': None, 'right': None}
        for i, bbox in enumerate(detection_list):
            image_results[bbox[0]] = _record_from(hands, i, bbox[0] == 'right')
        np.save(os.path.join(output_folder, f"{file_name}.npy"), image_results)
    return stats


def main(argv=None):
    """``python -m hamer_yolo_amd.d_infer --input <RGB_dir> --output <out_dir> --intrinsics <cam_K.txt>``."""
    ap = argparse.ArgumentParser(description="YOLOv7 -> RootNet depth + HaMeR -> MANO parameters (.npy per image)")
    ap.add_argument('--input', type=str, required=True)
    ap.add_argument('--output', type=str, required=True)
    ap.add_argument('--intrinsics', type=str, required=True, help="3x3 camera matrix txt")
    ap.add_argument('--obj', type=str, default=None, help="also reconstruct OBJ meshes into this folder")
    args = ap.parse_args(argv)
    k_real = load_intrinsics(args.intrinsics)
    hamer = hamer_inference(hamer_opt)
    process_batch_manopara(args.input, args.output, k_real, hamer=hamer)
    if args.obj:
        reconstruct_and_save_obj_with_wrapper(args.output, args.obj, hamer)


if __name__ == '__main__':
    main()
