"""Deterministic synthetic weights for the HaMeR / MANO / YOLOv7 hot path.

The reference ships no weights (every checkpoint is fetched from
outside the tree), so benchmarks, golden fixtures and parity tests all run on
random-init weights of the reference architecture.  The generator is counter
based: element ``i`` of tensor ``name`` is a pure function of ``(seed, name, i)``
built from exact integer arithmetic plus one IEEE multiply-add, so torch-CPU,
torch-ROCm and numpy produce bit-identical tensors.  Nothing here depends on a
library RNG stream.

Tensor names follow the reference ``state_dict`` keys
(hamer/hamer/models/hamer.py:34-52: ``backbone.*``, ``mano_head.*``, ``mano.*``)
so a converted real checkpoint can be dropped in unchanged.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

_M32 = 0xFFFFFFFF


def _hash_u32(idx: torch.Tensor, seed: int) -> torch.Tensor:
    """lowbias32 integer hash of (idx + seed) on int64 tensors, result in [0, 2^32)."""
    x = (idx + (seed & _M32)) & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def name_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) * 0x9E3779B1 + seed * 0x85EBCA6B + 0x1234567) & _M32


def uniform(name: str, shape, half_width: float, center: float = 0.0, seed: int = 0,
            device="cpu", chunk: int = 1 << 24) -> torch.Tensor:
    """fp32 tensor with entries ``center + half_width * v``, v uniform on the 2^-23 grid of [-1, 1).

    v is exact in fp32 (a 24-bit integer scaled by a power of two); the final
    multiply-add is a single fp32 mul followed by a single fp32 add on every backend.
    """
    n = 1
    for s in shape:
        n *= int(s)
    out = torch.empty(n, dtype=torch.float32, device=device)
    ns = name_seed(name, seed)
    for start in range(0, n, chunk):
        stop = min(n, start + chunk)
        idx = torch.arange(start, stop, dtype=torch.int64, device=device)
        h = _hash_u32(idx, ns)
        k = (h >> 8) - (1 << 23)                     # integer in [-2^23, 2^23)
        v = k.to(torch.float32) * (1.0 / (1 << 23))  # exact
        v = v * float(half_width)
        if center != 0.0:
            v = v + float(center)
        out[start:stop] = v
    return out.reshape(*shape)


# --------------------------------------------------------------------------- configs
@dataclass
class ViTConfig:
    """Geometry of backbones/vit.py:12-24 (ViT-H/16 on a 256x192 window)."""
    img_h: int = 256
    img_w: int = 192
    patch: int = 16
    pad: int = 2           # vit.py:168: padding = 4 + 2*(ratio//2 - 1) = 2 for ratio 1
    embed_dim: int = 1280
    depth: int = 32
    heads: int = 16
    mlp_ratio: int = 4
    ln_eps: float = 1e-6   # vit.py:222

    @property
    def grid_h(self):
        return (self.img_h + 2 * self.pad - self.patch) // self.patch + 1

    @property
    def grid_w(self):
        return (self.img_w + 2 * self.pad - self.patch) // self.patch + 1

    @property
    def tokens(self):
        return self.grid_h * self.grid_w

    @property
    def head_dim(self):
        return self.embed_dim // self.heads


@dataclass
class DecoderConfig:
    """MANO transformer-decoder head (configs_hydra/experiment/hamer_vit_transformer.yaml:35-42,
    heads/mano_head.py:33-46)."""
    dim: int = 1024
    depth: int = 6
    heads: int = 8
    dim_head: int = 64
    mlp_dim: int = 1024
    context_dim: int = 1280
    npose: int = 96        # 6 * (15 + 1), mano_head.py:30
    ln_eps: float = 1e-5   # torch.nn.LayerNorm default (t_cond_mlp.py:51-52)

    @property
    def inner(self):
        return self.heads * self.dim_head


@dataclass
class HamerConfig:
    vit: ViTConfig = field(default_factory=ViTConfig)
    dec: DecoderConfig = field(default_factory=DecoderConfig)
    image_size: int = 256
    focal_length: float = 5000.0


def tiny_config() -> HamerConfig:
    """Small geometry with the same kernel shapes (head_dim 80, 192 tokens) for fast tests."""
    return HamerConfig(
        vit=ViTConfig(embed_dim=320, depth=2, heads=4),
        dec=DecoderConfig(dim=256, depth=2, heads=4, dim_head=64, mlp_dim=256, context_dim=320),
    )


def tome_tiny_config() -> HamerConfig:
    """tiny_config with 6 blocks: the ToMe schedule r = (8, -1) then removes [16, 12, 9, 6, 3, 0] tokens (192 -> 146)."""
    return HamerConfig(
        vit=ViTConfig(embed_dim=320, depth=6, heads=4),
        dec=DecoderConfig(dim=256, depth=2, heads=4, dim_head=64, mlp_dim=256, context_dim=320),
    )


# --------------------------------------------------------------------------- HaMeR weights
def hamer_state_dict(cfg: Optional[HamerConfig] = None, seed: int = 0, device="cpu",
                     bf16_representable: bool = False) -> Dict[str, torch.Tensor]:
    """Random-init HaMeR weights keyed like the reference checkpoint's ``state_dict``.

    Widths follow the reference initialisers (vit.py:255,:300-312 trunc_normal std .02;
    nn.Linear default U(+-1/sqrt(fan_in)) for the decoder) but every bias and LayerNorm
    affine term is non-trivial so no term of the forward is dead in tests.
    ``bf16_representable`` rounds the GEMM weight matrices to bf16 values (kept as fp32).
    """
    cfg = cfg or HamerConfig()
    v, d = cfg.vit, cfg.dec
    sd: Dict[str, torch.Tensor] = {}
    w02 = 0.02 * 3 ** 0.5

    def U(name, shape, hw, center=0.0):
        sd[name] = uniform(name, shape, hw, center, seed=seed, device=device)

    def R(name):
        if bf16_representable:
            sd[name] = sd[name].to(torch.bfloat16).to(torch.float32)

    D, H = v.embed_dim, v.embed_dim * v.mlp_ratio
    kpe = 3 * v.patch * v.patch
    U("backbone.patch_embed.proj.weight", (D, 3, v.patch, v.patch), kpe ** -0.5); R("backbone.patch_embed.proj.weight")
    U("backbone.patch_embed.proj.bias", (D,), kpe ** -0.5)
    U("backbone.pos_embed", (1, v.tokens + 1, D), w02)
    for i in range(v.depth):
        p = f"backbone.blocks.{i}."
        U(p + "norm1.weight", (D,), 0.1, 1.0); U(p + "norm1.bias", (D,), 0.02)
        U(p + "attn.qkv.weight", (3 * D, D), w02); R(p + "attn.qkv.weight")
        U(p + "attn.qkv.bias", (3 * D,), 0.02)
        U(p + "attn.proj.weight", (D, D), w02); R(p + "attn.proj.weight")
        U(p + "attn.proj.bias", (D,), 0.02)
        U(p + "norm2.weight", (D,), 0.1, 1.0); U(p + "norm2.bias", (D,), 0.02)
        U(p + "mlp.fc1.weight", (H, D), w02); R(p + "mlp.fc1.weight")
        U(p + "mlp.fc1.bias", (H,), 0.02)
        U(p + "mlp.fc2.weight", (D, H), w02); R(p + "mlp.fc2.weight")
        U(p + "mlp.fc2.bias", (D,), 0.02)
    U("backbone.last_norm.weight", (D,), 0.1, 1.0); U("backbone.last_norm.bias", (D,), 0.02)

    t = "mano_head.transformer."
    U(t + "to_token_embedding.weight", (d.dim, 1), 1.0)
    U(t + "to_token_embedding.bias", (d.dim,), 1.0)
    U(t + "pos_embedding", (1, 1, d.dim), 3 ** 0.5)
    for i in range(d.depth):
        p = f"{t}transformer.layers.{i}."
        for j in range(3):
            U(p + f"{j}.norm.weight", (d.dim,), 0.1, 1.0); U(p + f"{j}.norm.bias", (d.dim,), 0.02)
        U(p + "0.fn.to_qkv.weight", (3 * d.inner, d.dim), d.dim ** -0.5)
        U(p + "0.fn.to_out.0.weight", (d.dim, d.inner), d.inner ** -0.5)
        U(p + "0.fn.to_out.0.bias", (d.dim,), d.inner ** -0.5)
        U(p + "1.fn.to_kv.weight", (2 * d.inner, d.context_dim), d.context_dim ** -0.5); R(p + "1.fn.to_kv.weight")
        U(p + "1.fn.to_q.weight", (d.inner, d.dim), d.dim ** -0.5)
        U(p + "1.fn.to_out.0.weight", (d.dim, d.inner), d.inner ** -0.5)
        U(p + "1.fn.to_out.0.bias", (d.dim,), d.inner ** -0.5)
        U(p + "2.fn.net.0.weight", (d.mlp_dim, d.dim), d.dim ** -0.5)
        U(p + "2.fn.net.0.bias", (d.mlp_dim,), d.dim ** -0.5)
        U(p + "2.fn.net.3.weight", (d.dim, d.mlp_dim), d.mlp_dim ** -0.5)
        U(p + "2.fn.net.3.bias", (d.dim,), d.mlp_dim ** -0.5)
    # read-out heads: a quarter of the nn.Linear default width, so the 6-D pose stays within
    # ~+-0.35 rms of the mean pose (a trained head predicts corrections to the mean pose, not
    # O(1) noise that would make the Gram-Schmidt step of rot6d_to_rotmat ill-conditioned)
    for nm, n in (("decpose", d.npose), ("decshape", 10), ("deccam", 3)):
        U(f"mano_head.{nm}.weight", (n, d.dim), 0.25 * d.dim ** -0.5)
        U(f"mano_head.{nm}.bias", (n,), 0.25 * d.dim ** -0.5)
    # mean parameters (mano_head.py:53-59): identity rotations in the 6-D representation
    # (geometry.py:56-58: a1 = x[0:3], a2 = x[3:6]) plus noise; scale ~0.9 for the camera.
    ident6 = torch.tensor([1.0, 0, 0, 0, 1.0, 0], device=device).repeat(d.npose // 6)
    sd["mano_head.init_hand_pose"] = (ident6 + uniform("mano_head.init_hand_pose", (d.npose,), 0.2, seed=seed, device=device)).reshape(1, -1)
    U("mano_head.init_betas", (1, 10), 0.5)
    sd["mano_head.init_cam"] = (torch.tensor([0.9, 0.0, 0.0], device=device)
                                + uniform("mano_head.init_cam", (3,), 0.1, seed=seed, device=device)).reshape(1, 3)
    return sd


# --------------------------------------------------------------------------- MANO-shaped parameters
MANO_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
MANO_TIP_VERTS = [744, 320, 443, 554, 671]      # smplx vertex_ids['mano'] (mano_wrapper.py:23)
MANO_JOINT_MAP = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]  # mano_wrapper.py:24


def mano_params(seed: int = 0, n_verts: int = 778) -> Dict[str, torch.Tensor]:
    """MANO-shaped random parameters (the licensed MANO arrays are never committed).

    Always built on the CPU (the row normalisations are reductions, whose order is
    backend dependent) and copied to the device by the caller.
    Keys follow smplx.MANOLayer buffers: v_template (V,3), shapedirs (V,3,10),
    posedirs (135, 3V), J_regressor (16,V), lbs_weights (V,16), parents (16,), faces (1538,3).
    """
    V = n_verts
    p: Dict[str, torch.Tensor] = {}
    p["v_template"] = uniform("mano.v_template", (V, 3), 0.08, seed=seed)
    p["shapedirs"] = uniform("mano.shapedirs", (V, 3, 10), 0.01, seed=seed)
    p["posedirs"] = uniform("mano.posedirs", (135, V * 3), 0.005, seed=seed)
    jr = uniform("mano.J_regressor", (16, V), 0.5, 0.5, seed=seed).double() ** 12
    p["J_regressor"] = (jr / jr.sum(1, keepdim=True)).float()
    w = uniform("mano.lbs_weights", (V, 16), 0.5, 0.5, seed=seed).double() ** 6
    p["lbs_weights"] = (w / w.sum(1, keepdim=True)).float()
    p["parents"] = torch.tensor(MANO_PARENTS, dtype=torch.int64)
    f = _hash_u32(torch.arange(1538 * 3, dtype=torch.int64), name_seed("mano.faces", seed)) % V
    p["faces"] = f.reshape(1538, 3)
    return p


def crops_u8(batch: int, seed0: int = 0, size: int = 256) -> torch.Tensor:
    """(B, size, size, 3) uint8 noise, crop b drawn with seed ``seed0 + b``."""
    out = torch.empty(batch, size, size, 3, dtype=torch.uint8)
    n = size * size * 3
    idx = torch.arange(n, dtype=torch.int64)
    for b in range(batch):
        out[b] = (_hash_u32(idx, name_seed("crop", seed0 + b)) >> 24).to(torch.uint8).reshape(size, size, 3)
    return out


def normalize_crops(u8: torch.Tensor) -> torch.Tensor:
    """uint8 RGB HWC -> (B,3,H,W) fp32 ``(x - 255 mean) / (255 std)`` (infer.py:145-146,:235-238)."""
    mean = 255.0 * torch.tensor([0.485, 0.456, 0.406], dtype=torch.float64)
    std = 255.0 * torch.tensor([0.229, 0.224, 0.225], dtype=torch.float64)
    x = u8.permute(0, 3, 1, 2).to(torch.float32)
    # the reference does the arithmetic on float32 arrays with python-float (double) scalars
    return ((x - mean.float().view(1, 3, 1, 1)) / std.float().view(1, 3, 1, 1)).contiguous()


# --------------------------------------------------------------------------- YOLOv7 weights
# Per-layer weight-width factors that keep every convolution's pre-activation rms near 1 on a
# seeded frame (a random-init CNN without them either explodes or dies within 100 layers).
# Produced by tools/calibrate_yolo_synth.py and frozen here as constants.
YOLO_CALIB: Dict[str, float] = {
    "model.0.conv": 1.369, "model.1.conv": 1.653, "model.2.conv": 1.063, "model.3.conv": 1.058,
    "model.4.conv": 1.231, "model.5.conv": 1.209, "model.6.conv": 1.178, "model.7.conv": 1.062,
    "model.8.conv": 1.182, "model.9.conv": 1.136, "model.11.conv": 1.109, "model.13.conv": 0.864,
    "model.14.conv": 0.991, "model.15.conv": 0.946, "model.17.conv": 1.227, "model.18.conv": 1.142,
    "model.19.conv": 1.347, "model.20.conv": 1.105, "model.21.conv": 1.224, "model.22.conv": 1.008,
    "model.24.conv": 1.239, "model.26.conv": 0.818, "model.27.conv": 1.264, "model.28.conv": 1.131,
    "model.30.conv": 1.076, "model.31.conv": 1.133, "model.32.conv": 1.106, "model.33.conv": 1.001,
    "model.34.conv": 1.163, "model.35.conv": 1.103, "model.37.conv": 1.157, "model.39.conv": 0.757,
    "model.40.conv": 1.138, "model.41.conv": 1.078, "model.43.conv": 1.082, "model.44.conv": 1.082,
    "model.45.conv": 1.135, "model.46.conv": 1.209, "model.47.conv": 1.181, "model.48.conv": 1.172,
    "model.50.conv": 1.105, "model.51.cv1.conv": 1.111, "model.51.cv3.conv": 1.089, "model.51.cv4.conv": 1.134,
    "model.51.cv5.conv": 0.372, "model.51.cv6.conv": 1.178, "model.51.cv2.conv": 1.082, "model.51.cv7.conv": 1.133,
    "model.52.conv": 1.085, "model.54.conv": 1.178, "model.56.conv": 1.064, "model.57.conv": 1.027,
    "model.58.conv": 1.055, "model.59.conv": 1.087, "model.60.conv": 1.064, "model.61.conv": 1.088,
    "model.63.conv": 1.07, "model.64.conv": 1.042, "model.66.conv": 1.216, "model.68.conv": 1.14,
    "model.69.conv": 1.094, "model.70.conv": 1.08, "model.71.conv": 1.166, "model.72.conv": 1.22,
    "model.73.conv": 1.131, "model.75.conv": 1.041, "model.77.conv": 0.748, "model.78.conv": 0.987,
    "model.79.conv": 1.103, "model.81.conv": 0.939, "model.82.conv": 0.991, "model.83.conv": 1.072,
    "model.84.conv": 0.96, "model.85.conv": 0.964, "model.86.conv": 1.026, "model.88.conv": 1.04,
    "model.90.conv": 0.68, "model.91.conv": 1.023, "model.92.conv": 1.093, "model.94.conv": 1.023,
    "model.95.conv": 1.056, "model.96.conv": 1.234, "model.97.conv": 1.151, "model.98.conv": 1.066,
    "model.99.conv": 1.188, "model.101.conv": 1.116, "model.102.rbr_reparam": 1.068, "model.103.rbr_reparam": 1.082,
    "model.104.rbr_reparam": 1.152, "model.105.m.0": 2.945, "model.105.m.1": 3.242, "model.105.m.2": 2.914,
}
def yolo_state_dict(seed: int = 0, nc: int = 3, obj_bias: float = -4.0, cls_bias: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """UNFUSED random-init YOLOv7 weights keyed like the reference ``Model(cfg/training/yolov7.yaml)``
    state dict (Conv+BN pairs, RepConv branches, IDetect with ImplicitA/M).  Conv widths are
    He-style so activations stay O(1) through the 105 layers; BatchNorm statistics are non-trivial.
    ``obj_bias`` shifts the objectness logit so a small fraction of the 15120 candidates passes the
    0.25 confidence threshold; ``cls_bias`` (when given) replaces the random class biases by one value, so
    that the winning class is decided by the features (both 'left' and 'right' boxes appear).  Seed 2 with obj_bias -2.2 and
    cls_bias 0 yields ~10 boxes of both labels on seeded 1080p frames (the chained driver test)."""
    from .yolo import arch
    layers = arch.yolov7_layers()
    specs = arch.conv_specs(layers, 3, nc)
    sd: Dict[str, torch.Tensor] = {}

    def U(name, shape, hw, center=0.0):
        sd[name] = uniform("yolo." + name, shape, hw, center, seed=seed)

    def bn(prefix, c):
        U(prefix + ".weight", (c,), 0.2, 1.0); U(prefix + ".bias", (c,), 0.1)
        U(prefix + ".running_mean", (c,), 0.1); U(prefix + ".running_var", (c,), 0.3, 1.0)

    for name, (co, ci, k, s) in specs.items():
        hw = (6.0 / (ci * k * k)) ** 0.5 * YOLO_CALIB.get(name, 1.0)
        if name.endswith(".conv"):
            base = name[:-5]
            U(name + ".weight", (co, ci, k, k), hw); bn(base + ".bn", co)
        elif name.endswith(".rbr_reparam"):
            base = name[:-len(".rbr_reparam")]
            U(base + ".rbr_dense.0.weight", (co, ci, 3, 3), hw * 0.8); bn(base + ".rbr_dense.1", co)
            U(base + ".rbr_1x1.0.weight", (co, ci, 1, 1), (6.0 / ci) ** 0.5 * 0.5 * YOLO_CALIB.get(name, 1.0)); bn(base + ".rbr_1x1.1", co)
        else:
            base, l = name.rsplit(".m.", 1)
            U(name + ".weight", (co, ci, 1, 1), hw)
            b = uniform("yolo." + name + ".bias", (co,), 0.5, seed=seed).view(3, -1)
            b[:, 4] += obj_bias
            if cls_bias is not None:
                b[:, 5:] = float(cls_bias)
            sd[name + ".bias"] = b.reshape(-1)
            U(f"{base}.ia.{l}.implicit", (1, ci, 1, 1), 0.02)
            U(f"{base}.im.{l}.implicit", (1, co, 1, 1), 0.02, 1.0)
    return sd


def frame_u8(h: int = 1080, w: int = 1920, seed: int = 0, smooth: int = 8) -> torch.Tensor:
    """(h, w, 3) uint8 BGR synthetic frame: seeded noise on a coarse grid, bilinearly upsampled, plus
    fine noise -- integer arithmetic only, so it is identical on every machine (pure white noise
    would be a degenerate input for a resize)."""
    s = smooth
    gh, gw = h // s + 2, w // s + 2
    coarse = (_hash_u32(torch.arange(gh * gw * 3, dtype=torch.int64), name_seed("frame.coarse", seed)) >> 24).reshape(gh, gw, 3)
    ys, xs = torch.arange(h, dtype=torch.int64), torch.arange(w, dtype=torch.int64)
    y0, fy, x0, fx = ys // s, (ys % s)[:, None, None], xs // s, (xs % s)[None, :, None]
    c00, c01 = coarse[y0][:, x0], coarse[y0][:, x0 + 1]
    c10, c11 = coarse[y0 + 1][:, x0], coarse[y0 + 1][:, x0 + 1]
    img = (c00 * (s - fy) * (s - fx) + c01 * (s - fy) * fx + c10 * fy * (s - fx) + c11 * fy * fx) // (s * s)
    fine = (_hash_u32(torch.arange(h * w * 3, dtype=torch.int64), name_seed("frame.fine", seed)) >> 27).reshape(h, w, 3) - 16
    return (img + fine).clamp(0, 255).to(torch.uint8).contiguous()


def rootnet_state_dict(seed: int = 0):
    """Seeded random-init weights with the reference's RootNet checkpoint keys: ``net`` = the SAR state dict's backbone part
    (``backbone.extract_mid.*`` / ``backbone.extract_high.0.*``, torchvision resnet34 wrapped as in rootnet/Model_RGB.py:179-196)
    and ``rootnet`` = ``depth_layer.{weight,bias}`` (:253-259).  He-style widths keep the ReLU activations O(1) through the
    16 residual blocks; the residual branch's second BatchNorm is scaled down as zero-init-residual training does."""
    from .rootnet import arch
    net: Dict[str, torch.Tensor] = {}

    def conv(key, co, ci, k):
        net[key + ".weight"] = uniform(key + ".weight", (co, ci, k, k), (6.0 / (ci * k * k)) ** 0.5, 0.0, seed=seed)

    def bn(key, c, gain=1.0):
        net[key + ".weight"] = uniform(key + ".weight", (c,), 0.1 * gain, gain, seed=seed)
        net[key + ".bias"] = uniform(key + ".bias", (c,), 0.1, 0.0, seed=seed)
        net[key + ".running_mean"] = uniform(key + ".running_mean", (c,), 0.1, 0.0, seed=seed)
        net[key + ".running_var"] = uniform(key + ".running_var", (c,), 0.2, 1.0, seed=seed)

    conv(arch.STEM_CONV, 64, 3, 7)
    bn(arch.STEM_BN, 64)
    for pre, cin, cout, s, ds in arch.blocks():
        conv(pre + "conv1", cout, cin, 3); bn(pre + "bn1", cout)
        conv(pre + "conv2", cout, cout, 3); bn(pre + "bn2", cout, gain=0.5)
        if ds:
            conv(pre + "downsample.0", cout, cin, 1); bn(pre + "downsample.1", cout)
    root = {"depth_layer.weight": uniform("depth_layer.weight", (1, 512, 1, 1), 0.05, 0.02, seed=seed),
            "depth_layer.bias": uniform("depth_layer.bias", (1,), 0.05, 0.3, seed=seed)}
    return net, root
