"""ctypes binding of libresshift_hip.so (C ABI: include/resshift_hip.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded this module
raises, loudly.  `import torch` happens first on purpose so that the engine binds to the same
libamdhip64 (soname libamdhip64.so.7) that PyTorch-ROCm already mapped — one HIP runtime per process,
so torch device pointers and streams are valid inside the engine.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL call, see module docstring)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RESSHIFT_HIP_LIB") or os.path.join(HERE, "libresshift_hip.so")   # (override: A/B of two builds on one box)

RS_PREC_F16 = 0
RS_PREC_F32 = 1
RS_PREC_SPLIT = 2
RS_MAX_LEVELS = 8
RS_MAX_STEPS = 64


class UNetConfig(C.Structure):
    _fields_ = [
        ("image_size", C.c_int), ("in_channels", C.c_int), ("model_channels", C.c_int), ("out_channels", C.c_int),
        ("n_levels", C.c_int),
        ("channel_mult", C.c_int * RS_MAX_LEVELS),
        ("num_res_blocks", C.c_int * RS_MAX_LEVELS),
        ("n_attn_res", C.c_int),
        ("attention_resolutions", C.c_int * RS_MAX_LEVELS),
        ("swin_depth", C.c_int), ("swin_embed_dim", C.c_int), ("window_size", C.c_int), ("num_heads", C.c_int),
        ("mlp_ratio", C.c_float),
        ("cond_lq", C.c_int), ("cond_mask", C.c_int), ("lq_size", C.c_int),
    ]


class AEConfig(C.Structure):
    _fields_ = [
        ("ch", C.c_int), ("n_levels", C.c_int),
        ("ch_mult", C.c_int * RS_MAX_LEVELS),
        ("num_res_blocks", C.c_int * RS_MAX_LEVELS),
        ("in_channels", C.c_int), ("out_ch", C.c_int), ("z_channels", C.c_int), ("embed_dim", C.c_int),
        ("n_embed", C.c_int), ("resolution", C.c_int),
        ("n_attn_res", C.c_int),
        ("attn_resolutions", C.c_int * RS_MAX_LEVELS),
    ]


class Config(C.Structure):
    _fields_ = [("unet", UNetConfig), ("ae", AEConfig), ("has_unet", C.c_int), ("has_ae", C.c_int), ("enable_f16", C.c_int),
                ("enable_f32", C.c_int), ("enable_split", C.c_int)]


class SampleArgs(C.Structure):
    _fields_ = [
        ("y", C.c_void_p), ("mask", C.c_void_p), ("noise", C.c_void_p), ("out", C.c_void_p), ("z_out", C.c_void_p),
        ("idx_out", C.c_void_p),
        ("B", C.c_int), ("h", C.c_int), ("w", C.c_int), ("sf", C.c_int), ("steps", C.c_int),
        ("inv_std", C.c_float * RS_MAX_STEPS), ("coef1", C.c_float * RS_MAX_STEPS), ("coef2", C.c_float * RS_MAX_STEPS),
        ("sigma", C.c_float * RS_MAX_STEPS), ("tmap", C.c_int * RS_MAX_STEPS),
        ("prior_scale", C.c_float), ("scale_factor", C.c_float),
        ("prec_encode", C.c_int), ("prec_decode", C.c_int), ("prec_unet", C.c_int * RS_MAX_STEPS),
        ("stream", C.c_void_p),
    ]


_P, _I, _F, _LL, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t

# name -> (restype, argtypes); every symbol declared in include/resshift_hip.h is listed here.
SIGNATURES = {
    "rs_create": (_P, [C.POINTER(Config)]),
    "rs_destroy": (None, [_P]),
    "rs_last_error": (C.c_char_p, []),
    "rs_load_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "rs_weight_bytes": (_SZ, [_P]),
    "rs_bind_weight_blob": (_I, [_P, _P, _SZ]),
    "rs_pack_weights": (_I, [_P]),
    "rs_bcast_weights": (_I, [_P, _P, _I, _P]),
    "rs_weights_ready": (_I, [_P]),
    "rs_unet_forward": (_I, [_P, _P, C.POINTER(C.c_int), _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "rs_vq_encode": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "rs_vq_decode": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rs_bicubic": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rs_sample": (_I, [_P, C.POINTER(SampleArgs)]),
    "rs_axpbypcz": (_I, [_P, _P, _P, _P, _F, _F, _F, _LL, _P]),
    "rs_tile_accumulate": (_I, [_P, _P, _P] + [_I] * 8 + [_P]),
    "rs_tile_finalize": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "rs_window_copy": (_I, [_P, _P, _LL, _I, _I, _I, _I, _I, _I, _F, _P]),
    "rs_u8_to_input": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "rs_output_to_u8": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rs_arena_bytes": (_SZ, [_P]),
    "rs_last_launch_count": (_LL, [_P]),
    "rs_profile_enable": (_I, [_P, _I]),
    "rs_profile_get": (_I, [_P, C.POINTER(C.c_double)]),
    "rs_profile_families": (_I, [_P, C.POINTER(C.c_double), _I]),
    "rs_profile_shapes": (_I, [_P, C.c_char_p, _I]),
    "rs_debug_enable": (_I, [_P, _I]),
    "rs_debug_count": (_I, [_P]),
    "rs_debug_info": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(C.c_int)]),
    "rs_debug_fetch": (_I, [_P, _I, _P, _P]),
    "rs_op_conv2d": (_I, [_P, _P, _P, _P, _P, _P] + [_I] * 18 + [_P]),
    "rs_op_conv2d_bench": (_I, [_P, _P, _P, _P, _P] + [_I] * 16 + [C.POINTER(C.c_float), _P]),
    "rs_op_conv3x3_halo_stats_px": (_I, [_I, _I, _I, _I, _I, _I]),
    "rs_op_conv3x3_halo": (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "rs_op_conv3x3_wino": (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, C.POINTER(C.c_float), _P]),
    "rs_op_gemm_nt": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P]),
    "rs_op_groupnorm": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P]),
    "rs_op_window_attention": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "rs_op_window_attention_qkv": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rs_op_window_attention_qkv_split": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rs_op_swin_mlp": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rs_op_ae_flash_attention": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rs_op_ae_flash_attention_split": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rs_op_swin_mlp_split": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rs_op_swin_mlp_split_unembed": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rs_op_softmax_rows": (_I, [_P, _P, _LL, _I, _I, _P]),
    "rs_op_vq": (_I, [_P, _P, _P, _P, _LL, _I, _I, _P]),
    "rs_op_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "rs_op_nhwc_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "rs_op_convert": (_I, [_P, _I, _P, _I, _I, _LL, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP engine; raises RuntimeError (never falls back) when it is not available."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m resshift_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the ResShift hot path."
        )
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().rs_last_error().decode()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"resshift_hip {what} failed (rc={rc}): {last_error()}")


def current_stream_ptr() -> int:
    """hipStream_t of torch's current stream (0 = null stream when torch has no GPU)."""
    if torch.cuda.is_available():
        return int(torch.cuda.current_stream().cuda_stream)
    return 0


def window_copy(x, h0=0, w0=0, ho=None, wo=None, scale=1.0, out=None):
    """fp32 device tensor [..., H, W] -> [..., ho, wo] window starting at (h0, w0), reflected past the bottom / right edge and
    multiplied by `scale` (rs_window_copy: reflect padding, tile crops and the latent scaling of the host mirror).  The result is
    always float32 (the engine's user-facing tensors are fp32 like the reference's inference path; F.pad / slicing in the
    reference keep the input dtype, which is float32 there too)."""
    import torch

    if not x.is_cuda:
        raise RuntimeError("window_copy needs a device tensor: the host mirror does not fall back to CPU arithmetic")
    x = x.to(torch.float32).contiguous()
    H, W = x.shape[-2:]
    ho = H if ho is None else ho
    wo = W if wo is None else wo
    planes = x.numel() // (H * W)
    want = (*x.shape[:-2], ho, wo)
    if out is None:
        out = torch.empty(*want, device=x.device, dtype=torch.float32)
    elif out.dtype != torch.float32 or tuple(out.shape) != want or not out.is_contiguous() or out.device != x.device:
        raise ValueError(f"window_copy: `out` must be a contiguous float32 tensor of shape {want} on {x.device} "
                         f"(got {out.dtype}, {tuple(out.shape)}, contiguous={out.is_contiguous()}, {out.device})")
    check(load().rs_window_copy(x.data_ptr(), out.data_ptr(), planes, H, W, h0, w0, ho, wo, float(scale), current_stream_ptr()), "rs_window_copy")
    return out
