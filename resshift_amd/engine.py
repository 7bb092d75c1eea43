"""Python owner of one native `rs_engine` (include/resshift_hip.h): config marshalling, weight hand-over,
and typed wrappers over the network-level C-ABI calls.  PyTorch tensors are used for device memory
and streams only; all compute happens inside libresshift_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .spec import _listify, unet_heads

F16, F32, SPLIT = _lib.RS_PREC_F16, _lib.RS_PREC_F32, _lib.RS_PREC_SPLIT
# "split": (hi, lo) fp16 pair storage, three fp16 MFMAs per product (fp32-class results at 1/3 of the fp16 matrix rate)
PRECISIONS = {"fp16": F16, "f16": F16, "half": F16, "fp32": F32, "f32": F32, "float": F32, "exact": F32, "split": SPLIT,
              "fp16x3": SPLIT}


def parse_precision(p) -> int:
    if isinstance(p, str):
        return PRECISIONS[p.lower()]
    return int(p)


def _fill_unet(cu: _lib.UNetConfig, p: Mapping) -> None:
    mult = [int(m) for m in p.get("channel_mult", (1, 2, 4, 8))]
    nrb = _listify(p["num_res_blocks"], len(mult))
    if not p.get("use_scale_shift_norm", False):
        raise NotImplementedError("engine implements the use_scale_shift_norm=True ResBlock (all shipped configs)")
    if p.get("resblock_updown", False) or not p.get("conv_resample", True) or int(p.get("dims", 2)) != 2:
        raise NotImplementedError("engine implements conv_resample=True, resblock_updown=False, dims=2 (all shipped configs)")
    if p.get("patch_norm", False):
        raise NotImplementedError("patch_norm=True is not used by any shipped config")
    cu.image_size, cu.in_channels = int(p["image_size"]), int(p["in_channels"])
    cu.model_channels, cu.out_channels = int(p["model_channels"]), int(p["out_channels"])
    cu.n_levels = len(mult)
    for i, (m, r) in enumerate(zip(mult, nrb)):
        cu.channel_mult[i] = m
        cu.num_res_blocks[i] = r
    ar = [int(a) for a in p["attention_resolutions"]]
    cu.n_attn_res = len(ar)
    for i, a in enumerate(ar):
        cu.attention_resolutions[i] = a
    cu.swin_depth = int(p.get("swin_depth", 2))
    cu.swin_embed_dim = int(p.get("swin_embed_dim", 96))
    cu.window_size = int(p.get("window_size", 8))
    cu.num_heads = unet_heads(p)
    cu.mlp_ratio = float(p.get("mlp_ratio", 2.0))
    cu.cond_lq, cu.cond_mask = int(bool(p.get("cond_lq", True))), int(bool(p.get("cond_mask", False)))
    cu.lq_size = int(p.get("lq_size", 256))


def _fill_ae(ca: _lib.AEConfig, p: Mapping) -> None:
    dd = p["ddconfig"]
    mult = [int(m) for m in dd["ch_mult"]]
    nrb = _listify(dd["num_res_blocks"], len(mult))
    if dd.get("double_z", True):
        raise NotImplementedError("VQ autoencoders use double_z=False")
    ca.ch, ca.n_levels = int(dd["ch"]), len(mult)
    for i, (m, r) in enumerate(zip(mult, nrb)):
        ca.ch_mult[i] = m
        ca.num_res_blocks[i] = r
    ca.in_channels, ca.out_ch = int(dd["in_channels"]), int(dd["out_ch"])
    ca.z_channels, ca.embed_dim, ca.n_embed = int(dd["z_channels"]), int(p["embed_dim"]), int(p["n_embed"])
    ca.resolution = int(dd["resolution"])
    ar = list(dd.get("attn_resolutions", []))
    ca.n_attn_res = len(ar)


class Engine:
    """One native engine on the current device.  `unet_params` / `ae_params` are the YAML `params` blocks."""

    def __init__(self, unet_params: Optional[Mapping] = None, ae_params: Optional[Mapping] = None, enable_f16: bool = True,
                 enable_f32: bool = True, device: Optional[torch.device] = None, enable_split: bool = True):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("resshift_amd.Engine needs a HIP device; there is no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        cfg = _lib.Config()
        if unet_params is not None:
            _fill_unet(cfg.unet, unet_params)
            cfg.has_unet = 1
        if ae_params is not None:
            _fill_ae(cfg.ae, ae_params)
            cfg.has_ae = 1
        cfg.enable_f16, cfg.enable_f32, cfg.enable_split = int(enable_f16), int(enable_f32), int(enable_split)
        self.cfg = cfg
        self.unet_params, self.ae_params = unet_params, ae_params
        with torch.cuda.device(self.device):
            self._h = self.lib.rs_create(C.byref(cfg))
        if not self._h:
            raise RuntimeError("rs_create failed: " + _lib.last_error())
        nbytes = int(self.lib.rs_weight_bytes(self._h))
        # caller-owned blob so that it can be RCCL-broadcast as one message (sharding.broadcast_weights)
        self.blob = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        off = (-self.blob.data_ptr()) % 256
        self._blob_view = self.blob[off: off + nbytes]
        self._chk(self.lib.rs_bind_weight_blob(self._h, self._blob_view.data_ptr(), nbytes), "rs_bind_weight_blob")
        self.weights_loaded = False

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self.lib.rs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        _lib.check(rc, what)

    def _stream(self) -> int:
        return int(torch.cuda.current_stream(self.device).cuda_stream)

    # -- weights
    def load_state_dicts(self, unet_sd: Optional[Mapping[str, torch.Tensor]] = None, ae_sd: Optional[Mapping[str, torch.Tensor]] = None):
        """Hand reference-named tensors to the engine and pack them into the device blob (rank-0 side of a broadcast)."""
        for sd in (unet_sd, ae_sd):
            if sd is None:
                continue
            for k, v in sd.items():
                if not torch.is_floating_point(v):
                    continue  # relative_position_index: deterministic buffer, recomputed by the engine
                if k.endswith(".attn_mask"):
                    continue  # shift mask: recomputed on the fly by the window kernel
                t = v.detach().to("cpu", torch.float32).contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                self._chk(self.lib.rs_load_tensor(self._h, k.encode(), t.data_ptr(), shape, t.dim()), f"rs_load_tensor({k})")
        with torch.cuda.device(self.device):
            self._chk(self.lib.rs_pack_weights(self._h), "rs_pack_weights")
        self.weights_loaded = True

    def weight_blob(self) -> torch.Tensor:
        """The packed weights as one flat uint8 device tensor (for torch.distributed.broadcast)."""
        return self._blob_view

    def mark_weights_ready(self):
        self._chk(self.lib.rs_weights_ready(self._h), "rs_weights_ready")
        self.weights_loaded = True

    # -- network calls (NCHW fp32 tensors, like the reference)
    @staticmethod
    def _f32c(t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(torch.float32).contiguous()

    def unet_forward(self, x, timesteps: Sequence[int], lq=None, mask=None, prec=F16):
        x = self._f32c(x)
        B, _, H, W = x.shape
        lq_t = self._f32c(lq) if lq is not None else None
        mk_t = self._f32c(mask) if mask is not None else None
        Hl, Wl = (lq_t.shape[2], lq_t.shape[3]) if lq_t is not None else (H, W)
        out = torch.empty(B, int(self.cfg.unet.out_channels), H, W, device=x.device, dtype=torch.float32)
        ts = (C.c_int * B)(*[int(t) for t in timesteps])
        with torch.cuda.device(self.device):
            rc = self.lib.rs_unet_forward(self._h, x.data_ptr(), ts, lq_t.data_ptr() if lq_t is not None else None,
                                          mk_t.data_ptr() if mk_t is not None else None, out.data_ptr(), B, H, W, Hl, Wl, int(prec),
                                          self._stream())
        self._chk(rc, "rs_unet_forward")
        return out

    def vq_encode(self, img, prec=F16):
        img = self._f32c(img)
        B, _, H, W = img.shape
        f = 2 ** (int(self.cfg.ae.n_levels) - 1)
        z = torch.empty(B, int(self.cfg.ae.embed_dim), H // f, W // f, device=img.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            rc = self.lib.rs_vq_encode(self._h, img.data_ptr(), z.data_ptr(), B, H, W, int(prec), self._stream())
        self._chk(rc, "rs_vq_encode")
        return z

    def vq_decode(self, z, force_not_quantize=False, prec=F16, return_indices=False):
        z = self._f32c(z)
        B, _, h, w = z.shape
        f = 2 ** (int(self.cfg.ae.n_levels) - 1)
        img = torch.empty(B, int(self.cfg.ae.out_ch), h * f, w * f, device=z.device, dtype=torch.float32)
        idx = torch.empty(B * h * w, device=z.device, dtype=torch.int32) if return_indices else None
        with torch.cuda.device(self.device):
            rc = self.lib.rs_vq_decode(self._h, z.data_ptr(), img.data_ptr(), idx.data_ptr() if idx is not None else None, B, h, w,
                                       int(force_not_quantize), int(prec), self._stream())
        self._chk(rc, "rs_vq_decode")
        return (img, idx) if return_indices else img

    def bicubic(self, y, sf: int):
        y = self._f32c(y)
        B, Cc, H, W = y.shape
        out = torch.empty(B, Cc, H * sf, W * sf, device=y.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            rc = self.lib.rs_bicubic(self._h, y.data_ptr(), out.data_ptr(), B, Cc, H, W, sf, self._stream())
        self._chk(rc, "rs_bicubic")
        return out

    def axpbypcz(self, x, z, n, a, b, c, out=None):
        """out = a*x + b*z + c*n elementwise on fp32 tensors of identical layout (z, n optional)."""
        x = self._f32c(x)
        out = torch.empty_like(x) if out is None else out
        zt = self._f32c(z) if z is not None else None
        nt = self._f32c(n) if n is not None else None
        rc = self.lib.rs_axpbypcz(x.data_ptr(), zt.data_ptr() if zt is not None else None, nt.data_ptr() if nt is not None else None,
                                  out.data_ptr(), float(a), float(b), float(c), x.numel(), self._stream())
        self._chk(rc, "rs_axpbypcz")
        return out

    def u8_to_input(self, img_u8):
        """uint8 [B,H,W,C] device tensor -> fp32 [B,C,H,W] in [-1,1] (datapipe/datasets.py:59-63 on the device)."""
        assert img_u8.dtype == torch.uint8 and img_u8.dim() == 4 and img_u8.is_cuda
        img_u8 = img_u8.contiguous()
        B, H, W, Cc = img_u8.shape
        out = torch.empty(B, Cc, H, W, device=img_u8.device, dtype=torch.float32)
        self._chk(self.lib.rs_u8_to_input(img_u8.data_ptr(), out.data_ptr(), B, H, W, Cc, self._stream()), "rs_u8_to_input")
        return out

    def output_to_u8(self, sr, lq=None, mask=None, bgr=False):
        """fp32 [B,C,H,W] in [-1,1] -> uint8 [B,H,W,C]; with `lq` and `mask` ([B,1,H,W] in [-1,1]) the inpainting blend
        of sampler.py:218-222 is applied first; rounding as utils/util_image.py:245-269 (tensor2img)."""
        sr = self._f32c(sr)
        B, Cc, H, W = sr.shape
        lq_t = self._f32c(lq) if lq is not None else None
        mk_t = self._f32c(mask) if mask is not None else None
        out = torch.empty(B, H, W, Cc, device=sr.device, dtype=torch.uint8)
        rc = self.lib.rs_output_to_u8(sr.data_ptr(), lq_t.data_ptr() if lq_t is not None else None, mk_t.data_ptr() if mk_t is not None else None,
                                      out.data_ptr(), B, H, W, Cc, int(bool(bgr)), self._stream())
        self._chk(rc, "rs_output_to_u8")
        return out

    def sample(self, y, noise, tables: Dict[str, np.ndarray], sf: int, scale_factor: float, mask=None, prec_unet=F16, prec_encode=F16,
               prec_decode=F16, return_aux=False):
        """The whole p_sample_loop in one native call.  noise: [steps+1,B,Cz,hz,wz] fp32 in draw order."""
        y = self._f32c(y)
        noise = self._f32c(noise)
        B, _, h, w = y.shape
        steps = int(len(tables["coef1"]))
        f = 2 ** (int(self.cfg.ae.n_levels) - 1)
        hz, wz, cz = h * sf // f, w * sf // f, int(self.cfg.ae.embed_dim)
        assert tuple(noise.shape) == (steps + 1, B, cz, hz, wz), (tuple(noise.shape), (steps + 1, B, cz, hz, wz))
        out = torch.empty(B, int(self.cfg.ae.out_ch), h * sf, w * sf, device=y.device, dtype=torch.float32)
        a = _lib.SampleArgs()
        mk = self._f32c(mask) if mask is not None else None
        z_out = torch.empty(B, cz, hz, wz, device=y.device, dtype=torch.float32) if return_aux else None
        idx = torch.empty(B * hz * wz, device=y.device, dtype=torch.int32) if return_aux else None
        a.y, a.noise, a.out = y.data_ptr(), noise.data_ptr(), out.data_ptr()
        a.mask = mk.data_ptr() if mk is not None else None
        a.z_out = z_out.data_ptr() if z_out is not None else None
        a.idx_out = idx.data_ptr() if idx is not None else None
        a.B, a.h, a.w, a.sf, a.steps = B, h, w, int(sf), steps
        pu = [prec_unet] * steps if isinstance(prec_unet, (int, str)) else list(prec_unet)
        for t in range(steps):
            a.inv_std[t] = float(tables["inv_std"][t])
            a.coef1[t] = float(tables["coef1"][t])
            a.coef2[t] = float(tables["coef2"][t])
            a.sigma[t] = float(tables["sigma"][t])
            a.tmap[t] = int(tables["tmap"][t])
            a.prec_unet[t] = parse_precision(pu[t])
        a.prior_scale = float(tables["prior_scale"])
        a.scale_factor = float(scale_factor)
        a.prec_encode, a.prec_decode = parse_precision(prec_encode), parse_precision(prec_decode)
        a.stream = self._stream()
        with torch.cuda.device(self.device):
            rc = self.lib.rs_sample(self._h, C.byref(a))
        self._chk(rc, "rs_sample")
        if return_aux:
            return out, {"z_final": z_out, "indices": idx}
        return out

    def profile_enable(self, on: bool = True):
        self.lib.rs_profile_enable(self._h, int(on))

    def profile_get(self) -> Dict[str, float]:
        """MFMA implicit-GEMM statistics of the last native call (see rs_profile_get)."""
        out = (C.c_double * 9)()
        self.lib.rs_profile_get(self._h, out)
        return {"flops_f16": out[0], "flops_f32": out[1], "igemm_ms": out[2], "igemm_launches": int(out[3]), "igemm_bytes": out[4],
                "flops_split": out[5], "gn_ms": out[6], "gn_launches": int(out[7]), "gn_bytes": out[8]}

    FAMILIES = ("igemm4_kernel<*, false> (halo 3x3 conv, fp16)", "igemm4_kernel<*, true> (halo 3x3 conv, split storage)",
                "igemm2 / igemm3 / igemm_kernel (implicit GEMM, fp16)", "igemm_split_kernel (implicit GEMM, split storage)",
                "igemm2 / igemm_kernel<float> (implicit GEMM, exact fp32)", "win_attn_qkv_kernel (fused qkv + window attention + proj)",
                "swin_mlp_kernel (fused fc1 + GELU + fc2)", "win_attn_qkv_split_kernel (fused qkv + window attention + proj, split storage)",
                "swin_mlp_split_kernel (fused fc1 + GELU + fc2, split storage)", "ae_flash_attn_kernel (streaming AE mid-block attention, fp16)",
                "ae_flash_attn_split_kernel (streaming AE mid-block attention, split storage)",
                "wino_kernel (Winograd F(2x2,3x3) 3x3 conv, split storage; RS_WINO=1)")

    def profile_families(self):
        """per kernel family of the MFMA path: [(name, algorithmic FLOPs, kernel ms, launches)] of the last native call"""
        out = (C.c_double * (3 * len(self.FAMILIES)))()
        n = self.lib.rs_profile_families(self._h, out, 3 * len(self.FAMILIES))
        return [(self.FAMILIES[f], out[3 * f], out[3 * f + 1], int(out[3 * f + 2])) for f in range(max(0, n))]

    def profile_shapes(self):
        """(shapes, parts) of the last profiled native call: shapes = [{part, family, M, N, K, z, launches, ms, flops}] per distinct launch
        shape of the MFMA family, parts = {encoder / unet / decoder: wall ms}"""
        need = self.lib.rs_profile_shapes(self._h, None, 0)
        if need <= 1:
            return [], {}
        buf = C.create_string_buffer(need)
        self.lib.rs_profile_shapes(self._h, buf, need)
        shapes, parts = [], {}
        for line in buf.value.decode().splitlines():
            f = line.split()
            if f and f[0] == "shape" and len(f) >= 10:
                kv = dict(x.split("=") for x in f[3:])
                shapes.append({"part": f[1], "family": int(f[2][1:]), "M": int(kv["M"]), "N": int(kv["N"]), "K": int(kv["K"]), "z": int(kv["z"]),
                               "launches": int(float(kv["n"])), "ms": float(kv["ms"]), "flops": float(kv["flops"])})
            elif f and f[0] == "part":
                parts[f[1]] = float(f[2].split("=")[1])
        return shapes, parts

    def debug_enable(self, on: bool = True):
        self.lib.rs_debug_enable(self._h, int(on))

    def debug_trace(self):
        """name -> NCHW fp32 tensor for every activation recorded by the last call (debug_enable(True) first)."""
        out = {}
        for i in range(self.lib.rs_debug_count(self._h)):
            name = C.create_string_buffer(128)
            dims = (C.c_int * 4)()
            self.lib.rs_debug_info(self._h, i, name, 128, dims)
            t = torch.empty(dims[0], dims[1], dims[2], dims[3], device=self.device, dtype=torch.float32)
            self._chk(self.lib.rs_debug_fetch(self._h, i, t.data_ptr(), self._stream()), "rs_debug_fetch")
            out[name.value.decode()] = t
        torch.cuda.synchronize(self.device)
        return out

    def arena_bytes(self) -> int:
        return int(self.lib.rs_arena_bytes(self._h))

    def last_launch_count(self) -> int:
        return int(self.lib.rs_last_launch_count(self._h))
