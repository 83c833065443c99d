"""ctypes binding of libhamer_hip.so (the C ABI declared in include/hamer_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails the caller gets
an exception (``HipLibraryError``).  PyTorch is used by the callers only to own device
memory and the current HIP stream; only raw pointers and sizes cross this boundary.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libhamer_hip.so")

HM_DTYPE_BF16, HM_DTYPE_F16, HM_OUT_F32 = 0, 1, 2
HM_EPI_STORE, HM_EPI_GELU, HM_EPI_RESID_F32, HM_EPI_F32, HM_EPI_SILU = 0, 1, 2, 3, 4
HM_EPI_RESID_LN, HM_EPI_LN_STORE, HM_EPI_LN_GELU, HM_EPI_GELU_MX8 = 5, 6, 7, 8
HM_VERSION = 401      # include/hamer_hip.h: load() refuses a library built from another header
# the HM_OPT_* keys of include/hamer_hip.h, in enum order: load() checks the count against the library's hm_option_count(), and
# tests/test_host_logic.py parses the header's enum and compares names and values with this table
OPTION_NAMES = ("HM_OPT_PX_GRID", "HM_OPT_FP8P_GRID", "HM_OPT_FP8_ONE_TILE", "HM_OPT_FP8P_RESID", "HM_OPT_TOME_NO_SPLITK",
                "HM_OPT_TOME_SCALAR_ATTENTION", "HM_OPT_RESID_IN_EPILOGUE", "HM_OPT_CONV_TILE", "HM_OPT_CONV_SPLITK",
                "HM_OPT_PX_LDS_EPILOGUE", "HM_OPT_CONV_DIRECT", "HM_OPT_GEMM_TILE_RULE", "HM_OPT_CONV_KGROUPS",
                "HM_OPT_CONV_GENERAL_LOADER", "HM_OPT_CONV_STEM_PAIR")
globals().update({_n: _i for _i, _n in enumerate(OPTION_NAMES)})

EXPORTS = [
    "hm_version", "hm_last_error_string", "hm_gemm", "hm_layernorm", "hm_vit_attention", "hm_patch_im2col",
    "hm_linear_f32", "hm_broadcast_rows", "hm_cross_attention", "hm_mano_forward", "hm_crop_box_from_bbox",
    "hm_crop_batch", "hm_hamer_workspace_bytes", "hm_hamer_forward", "hm_prof_begin", "hm_prof_collect", "hm_prof_end",
    "hm_conv2d_nhwc", "hm_conv2d_stem_pair", "hm_maxpool_nhwc", "hm_upsample2x_nhwc", "hm_letterbox_plan_make", "hm_letterbox_tables",
    "hm_letterbox", "hm_yolo_decode", "hm_nms_workspace_bytes", "hm_yolo_nms", "hm_yolo_run", "hm_gemm_set_variant", "hm_gemm_set_group_m", "hm_ln_finalize", "hm_layernorm_accum", "hm_gemm_fp8", "hm_layernorm_mx8", "hm_vit_attention_mx8", "hm_nchw3_to_nhwc8", "hm_gap_linear",
    "hm_tome_index_bytes", "hm_tome_attention", "hm_tome_merge", "hm_set_option", "hm_get_option", "hm_tome_merge_metric", "hm_conv_splitk_bytes", "hm_yolo_decode_batch", "hm_letterbox_batch",
    "hm_option_count", "hm_gemm_px_grid", "hm_absmax16",
]
KIND_NAMES = ["gemm", "layernorm", "attention", "im2col", "linear_f32", "cross_attn", "mano", "crop", "conv", "other"]


class HipLibraryError(RuntimeError):
    pass


vp, fp = C.c_void_p, C.c_void_p  # every device pointer travels as void*


class GemmArgs(C.Structure):
    _fields_ = [("X", vp), ("W", vp), ("C", vp), ("bias", vp), ("resid", vp),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("ldx", C.c_int), ("ldw", C.c_int), ("ldc", C.c_int), ("ldr", C.c_int),
                ("resid_mod", C.c_int), ("epilogue", C.c_int), ("dtype", C.c_int),
                ("ln_gamma", vp), ("ln_xg", vp), ("ln_stats", vp), ("ln_colsum", vp), ("k_split", C.c_int), ("out_scale", C.c_float)]


class GemmFp8Args(C.Structure):
    _fields_ = [("X8", vp), ("x_scales", vp), ("W8", vp), ("w_scale", vp), ("C", vp), ("bias", vp), ("resid", vp),
                ("out_scales", vp), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("ldx", C.c_int), ("ldw", C.c_int),
                ("ldc", C.c_int), ("ldr", C.c_int), ("epilogue", C.c_int), ("out_dtype", C.c_int)]


class ManoModel(C.Structure):
    _fields_ = [("v_template", vp), ("shapedirs", vp), ("posedirs", vp), ("J_regressor", vp), ("lbs_weights", vp),
                ("n_verts", C.c_int)]


class CropBox(C.Structure):
    _fields_ = [("m0", C.c_double), ("m4", C.c_double), ("x0", C.c_int32), ("y0", C.c_int32),
                ("flip", C.c_int32), ("reserved", C.c_int32)]


class VitBlock(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1_g", "ln1_b", "ln2_g", "ln2_b", "qkv_w", "proj_w", "fc1_w", "fc2_w",
                                  "qkv_b", "proj_b", "fc1_b", "fc2_b",
                                  "qkv_colsum", "qkv_bias_ln", "fc1_colsum", "fc1_bias_ln",
                                  "qkv_w8", "fc1_w8", "fc2_w8", "qkv_ws", "fc1_ws", "fc2_ws", "proj_w8", "proj_ws", "kmean_w", "kmean_b")] + \
               [("attn_scale_mul", C.c_float), ("gelu_out_scale", C.c_float)]


class DecLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("ln0_g", "ln0_b", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "sa_v_w", "sa_w", "sa_out_w",
                                  "sa_out_b", "ca_q_w", "ca_out_w", "ca_out_b", "ff1_w", "ff1_b", "ff2_w", "ff2_b")] + \
               [("ca_scale_mul", C.c_float)]


class HamerWeights(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("img_h", "img_w_full", "win_x0", "win_w", "patch", "pad", "embed_dim", "depth",
                                       "heads", "mlp_dim")] + \
               [("vit_eps", C.c_float), ("patch_w", vp), ("patch_b", vp), ("pos", vp),
                ("blocks", C.POINTER(VitBlock)), ("last_g", vp), ("last_b", vp)] + \
               [(n, C.c_int) for n in ("dec_dim", "dec_depth", "dec_heads", "dec_dim_head", "dec_mlp")] + \
               [("dec_eps", C.c_float), ("token0", vp), ("kv_w", vp), ("layers", C.POINTER(DecLayer)),
                ("head_w", vp), ("head_b", vp), ("mano", ManoModel),
                ("focal_length", C.c_float), ("image_size", C.c_float), ("dtype", C.c_int), ("tome_r", C.POINTER(C.c_int)),
                ("range_stats", vp)]


class ConvArgs(C.Structure):
    _fields_ = [("X", vp), ("W", vp), ("Y", vp), ("bias", vp), ("zeros", vp)] + \
               [(n, C.c_int) for n in ("N", "H", "W_in", "Cin", "Cout", "ksize", "stride", "ldx", "ldy", "Kpad", "act",
                                       "out_f32", "dtype")] + [("resid", vp), ("ldr", C.c_int), ("splitk_ws", vp), ("splitk_ws_bytes", C.c_size_t)]


class YoloOp(C.Structure):
    _fields_ = [("kind", C.c_int), ("pool_pad", C.c_int), ("conv", ConvArgs)]


class LetterboxPlan(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("src_h", "src_w", "new_w", "new_h", "top", "left", "out_h", "out_w")] + \
               [("gain", C.c_float), ("pad_x", C.c_float), ("pad_y", C.c_float)]


class ProfRecord(C.Structure):
    _fields_ = [("kind", C.c_int), ("epilogue", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("ms", C.c_float)]


class HamerOutputs(C.Structure):
    _fields_ = [(n, vp) for n in ("pose6d", "betas", "cam", "rotmats", "verts", "joints", "cam_t", "kp2d", "tokens")]


_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raise HipLibraryError (never fall back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -m hamer_yolo_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the hot path.")
    # torch FIRST: it ships its own HIP runtime (libamdhip64 under torch/lib).  Loaded before it, this library pulls in the
    # system's copy and the process ends up with two runtimes -- torch sees the GPU, the kernels here fail with "no ROCm-capable
    # device is detected" (seen when __graft_entry__.build() and smoke() ran in one process).  With torch's copy already mapped
    # the loader resolves this library's HIP symbols to it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    i, f, d = C.c_int, C.c_float, C.c_double
    lib.hm_version.restype = i
    lib.hm_last_error_string.restype = C.c_char_p
    lib.hm_gemm.argtypes = [C.POINTER(GemmArgs), vp]
    lib.hm_layernorm.argtypes = [vp, vp, vp, vp, i, i, i, f, vp]
    lib.hm_vit_attention.argtypes = [vp, vp, i, i, i, i, f, i, vp]
    lib.hm_patch_im2col.argtypes = [vp, vp, i, i, i, i, i, i, i, i, vp]
    lib.hm_linear_f32.argtypes = [vp, i, vp, i, vp, vp, i, vp, i, i, i, i, i, vp]
    lib.hm_broadcast_rows.argtypes = [vp, vp, i, i, vp]
    lib.hm_cross_attention.argtypes = [vp, vp, i, i, i, vp, i, i, i, i, f, i, vp]
    lib.hm_mano_forward.argtypes = [C.POINTER(ManoModel), vp, vp, vp, vp, vp, vp, vp, vp, i, f, f, vp]
    lib.hm_crop_box_from_bbox.argtypes = [d, d, d, i, i, C.POINTER(CropBox)]
    lib.hm_crop_batch.argtypes = [vp, i, i, vp, vp, i, i, C.POINTER(C.c_float), C.POINTER(C.c_float), vp]
    lib.hm_hamer_workspace_bytes.argtypes = [C.POINTER(HamerWeights), i]
    lib.hm_hamer_workspace_bytes.restype = C.c_size_t
    lib.hm_hamer_forward.argtypes = [C.POINTER(HamerWeights), vp, i, C.POINTER(HamerOutputs), vp, C.c_size_t, vp]
    lib.hm_ln_finalize.argtypes = [vp, vp, i, i, C.c_float, vp]
    lib.hm_gemm_fp8.argtypes = [C.POINTER(GemmFp8Args), vp]
    lib.hm_layernorm_mx8.argtypes = [vp, vp, vp, vp, vp, i, i, C.c_float, vp]
    lib.hm_vit_attention_mx8.argtypes = [vp, vp, vp, i, i, i, i, C.c_float, vp]
    lib.hm_nchw3_to_nhwc8.argtypes = [vp, vp, i, i, i, i, vp]
    lib.hm_gap_linear.argtypes = [vp, i, i, vp, C.c_float, vp, vp, i, i, vp]
    lib.hm_layernorm_accum.argtypes = [vp, vp, i, vp, vp, vp, vp, i, i, i, C.c_float, vp]
    lib.hm_absmax16.argtypes = [vp, i, i, i, i, i, vp, vp]
    lib.hm_conv2d_nhwc.argtypes = [C.POINTER(ConvArgs), vp]
    lib.hm_conv_splitk_bytes.argtypes = [C.POINTER(ConvArgs)]
    lib.hm_conv2d_stem_pair.argtypes = [C.POINTER(ConvArgs), C.POINTER(ConvArgs), vp]
    lib.hm_conv_splitk_bytes.restype = C.c_size_t
    lib.hm_maxpool_nhwc.argtypes = [vp, i, vp, i, i, i, i, i, i, i, i, i, vp]
    lib.hm_upsample2x_nhwc.argtypes = [vp, i, vp, i, i, i, i, i, i, vp]
    lib.hm_letterbox_plan_make.argtypes = [i, i, i, i, C.POINTER(LetterboxPlan)]
    lib.hm_letterbox_tables.argtypes = [C.POINTER(LetterboxPlan), C.POINTER(C.c_int32)]
    lib.hm_letterbox.argtypes = [vp, C.POINTER(LetterboxPlan), vp, vp, i, vp, vp]
    lib.hm_letterbox_batch.argtypes = [vp, C.c_size_t, i, C.POINTER(LetterboxPlan), vp, vp, i, vp]
    lib.hm_yolo_decode.argtypes = [vp, i, vp, i, i, i, i, f, C.POINTER(C.c_float), vp]
    lib.hm_yolo_decode_batch.argtypes = [vp, i, vp, i, i, i, i, f, C.POINTER(C.c_float), i, C.c_size_t, vp]
    lib.hm_nms_workspace_bytes.argtypes = [i]
    lib.hm_nms_workspace_bytes.restype = C.c_size_t
    lib.hm_yolo_nms.argtypes = [vp, i, i, f, f, C.c_uint, i, i, C.POINTER(LetterboxPlan), vp, vp, vp, C.c_size_t, vp]
    lib.hm_yolo_run.argtypes = [C.POINTER(YoloOp), i, vp]
    lib.hm_tome_index_bytes.argtypes = [i]
    lib.hm_tome_index_bytes.restype = C.c_size_t
    lib.hm_tome_attention.argtypes = [vp, vp, vp, i, i, i, i, f, i, vp]
    lib.hm_tome_merge.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, vp]
    lib.hm_tome_merge_metric.argtypes = [vp, i, i, vp, vp, vp, vp, vp, i, i, i, i, vp]
    lib.hm_gemm_set_variant.argtypes = [i]
    lib.hm_gemm_set_group_m.argtypes = [i]
    lib.hm_set_option.argtypes = [i, i]
    lib.hm_get_option.argtypes = [i]
    lib.hm_option_count.argtypes = []
    lib.hm_gemm_px_grid.argtypes = [i, i]
    lib.hm_prof_begin.argtypes = [i]
    lib.hm_prof_collect.argtypes = [C.POINTER(ProfRecord), i]
    lib.hm_prof_end.argtypes = []
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise HipLibraryError(f"{LIB_PATH} does not export {name}")
        fn = getattr(lib, name)
        if name not in ("hm_version", "hm_last_error_string", "hm_hamer_workspace_bytes", "hm_nms_workspace_bytes", "hm_tome_index_bytes", "hm_conv_splitk_bytes"):
            fn.restype = i
    if lib.hm_version() != HM_VERSION:
        raise HipLibraryError(f"{LIB_PATH} reports HM_VERSION {lib.hm_version()}, this binding is written for {HM_VERSION}: "
                              "rebuild it (python -m hamer_yolo_amd.build --force)")
    if lib.hm_option_count() != len(OPTION_NAMES):
        raise HipLibraryError(f"{LIB_PATH} has {lib.hm_option_count()} HM_OPT_* keys, this binding names {len(OPTION_NAMES)}: "
                              "include/hamer_hip.h and lib.OPTION_NAMES have drifted apart")
    _lib = lib
    return lib


class option:
    """Context manager over hm_set_option: `with L.option(L.HM_OPT_FP8_ONE_TILE, 1): ...` (tests and tuning tools; the
    launch paths read these switches, never the environment)."""

    def __init__(self, key: int, value: int):
        self.key, self.value = key, value

    def __enter__(self):
        self.old = load().hm_get_option(self.key)
        check(load().hm_set_option(self.key, self.value), "hm_set_option")
        return self

    def __exit__(self, *exc):
        load().hm_set_option(self.key, self.old)
        return False


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().hm_last_error_string().decode(errors="replace")
        raise HipLibraryError(f"{what or 'libhamer_hip'} failed (code {rc}): {msg}")


def ptr(t) -> int:
    """Device/host address of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


class profile:
    """Context manager: per-launch HIP-event timings of every library kernel launched inside."""

    def __init__(self, capacity: int = 1 << 16):
        self.capacity = capacity
        self.records = []

    def __enter__(self):
        check(load().hm_prof_begin(self.capacity), "hm_prof_begin")
        return self

    def collect(self):
        buf = (ProfRecord * self.capacity)()
        n = load().hm_prof_collect(buf, self.capacity)
        if n < 0:
            check(n, "hm_prof_collect")
        self.records += [(KIND_NAMES[r.kind], r.epilogue, r.M, r.N, r.K, r.ms) for r in buf[:n]]
        return self.records

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                self.collect()
        finally:
            load().hm_prof_end()
        return False
