"""CPU restatement of the RootNet root-depth path (reference: rootnet/Model_RGB.py:179-196 SARresnet34, :240-292
ResRootNet, :494-498 calculate_k, :572-639 estimate_root_depth_custom; rootnet/preprocessing.py:152-188 process_bbox).

TEST INFRASTRUCTURE ONLY.  The backbone is torchvision's resnet34, which is not installed in this image and not part of
the reference tree: its published architecture (stem 7x7/2 - bn - relu - maxpool 3x3/2; BasicBlocks [3, 4, 6, 3] of
conv3x3-bn-relu-conv3x3-bn + identity / (conv1x1-bn)(x), relu; eval-mode BatchNorm eps 1e-5) is restated with
torch.nn.functional on the checkpoint's own state-dict keys.  PARITY UNPINNED against torchvision itself for the backbone;
ResRootNet.forward, process_bbox and calculate_k are pinned against the reference's own code (tests/golden/rootnet_head.npz,
tools/gen_golden_rootnet.py), the patch by oracle/crop_ref.py (same affine and cv2 restatement as HaMeR's crop).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import crop_ref

LAYERS = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]
PREFIX = ["backbone.extract_mid.4.", "backbone.extract_mid.5.", "backbone.extract_high.0.0.", "backbone.extract_high.0.1."]


def _bn(x, sd, key):
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"], False, 0.0, 1e-5)


def backbone(sd, x):
    """(B,3,256,256) -> (B,512,8,8): SARresnet34.forward."""
    x = F.relu(_bn(F.conv2d(x, sd["backbone.extract_mid.0.weight"], None, 2, 3), sd, "backbone.extract_mid.1"))
    x = F.max_pool2d(x, 3, 2, 1)
    cin = 64
    for (width, n, stride), pre in zip(LAYERS, PREFIX):
        for i in range(n):
            s = stride if i == 0 else 1
            p = f"{pre}{i}."
            idn = x
            out = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"], None, s, 1), sd, p + "bn1"))
            out = _bn(F.conv2d(out, sd[p + "conv2.weight"], None, 1, 1), sd, p + "bn2")
            if s != 1 or cin != width:
                idn = _bn(F.conv2d(x, sd[p + "downsample.0.weight"], None, s, 0), sd, p + "downsample.1")
            x = F.relu(out + idn)
            cin = width
    return x


def root_depth(root_sd, feat, k_value):
    """ResRootNet.forward_coord (:282-292): GAP -> 1x1 conv -> gamma * k."""
    img_feat = feat.reshape(feat.size(0), feat.size(1), -1).mean(2)[:, :, None, None]
    gamma = F.conv2d(img_feat, root_sd["depth_layer.weight"], root_sd["depth_layer.bias"]).view(-1, 1)
    return gamma * k_value.view(-1, 1)


def sanitize_bbox(bbox, img_width, img_height):
    x, y, w, h = bbox
    x1, y1 = max(0, x), max(0, y)
    x2 = min(img_width - 1, x1 + max(0, w - 1))
    y2 = min(img_height - 1, y1 + max(0, h - 1))
    return np.array([x1, y1, x2 - x1, y2 - y1], dtype=np.float64) if (w * h > 0 and x2 > x1 and y2 > y1) else None


def process_bbox(bbox, img_width, img_height, input_img_shape=(256, 256), ratio=1.25):
    b = sanitize_bbox(bbox, img_width, img_height)
    if b is None:
        return None
    w, h = b[2], b[3]
    cx, cy = b[0] + w / 2.0, b[1] + h / 2.0
    ar = input_img_shape[1] / input_img_shape[0]
    if w > ar * h:
        h = w / ar
    elif w < ar * h:
        w = h * ar
    return np.array([cx - w * ratio / 2.0, cy - h * ratio / 2.0, w * ratio, h * ratio], dtype=np.float64).astype(np.float32)


def calculate_k(bbox, fx, fy, bbox_real=(0.3, 0.3)):
    return float(np.sqrt(bbox_real[0] * bbox_real[1] * fx * fy / (float(bbox[2]) * float(bbox[3]))))


def estimate_root_depth(net_sd, root_sd, img_bgr_u8, K, bbox_xyxy):
    """estimate_root_depth_custom (:572-639) on the CPU."""
    x1, y1, x2, y2 = bbox_xyxy
    H, W = img_bgr_u8.shape[:2]
    bp = process_bbox([x1, y1, x2 - x1, y2 - y1], W, H, (256, 256), 1.5)
    cx, cy, S = float(bp[0] + 0.5 * bp[2]), float(bp[1] + 0.5 * bp[3]), float(bp[2])
    patch = crop_ref.warp_affine_u8(img_bgr_u8, crop_ref.gen_trans_from_patch(cx, cy, S, S, 256, 256), 256, 256)[:, :, ::-1]   # RGB
    mean = 255.0 * np.array([0.485, 0.456, 0.406]); std = 255.0 * np.array([0.229, 0.224, 0.225])
    t = np.transpose(patch, (2, 0, 1)).astype(np.float32)
    for c in range(3):
        t[c] = (t[c] - np.float32(mean[c])) / np.float32(std[c])
    x = torch.from_numpy(t)[None]
    k = calculate_k(bp, float(K[0][0]), float(K[1][1]))
    with torch.no_grad():
        return float(root_depth(root_sd, backbone(net_sd, x), torch.tensor([k]))[0, 0]), x
