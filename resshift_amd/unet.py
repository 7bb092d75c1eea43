"""Drop-in `UNetModelSwin` (reference: models/unet.py:603-912) backed by the HIP engine.

The object keeps the reference's constructor signature, `state_dict()` key names/shapes (so
`utils/util_net.reload_model` / `load_state_dict` work unchanged) and `forward(x, timesteps, lq, mask)`
contract, but holds no torch compute: `forward` hands device pointers to libresshift_hip.so.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .engine import F16, F32, Engine, parse_precision
from .spec import relative_position_index, shift_attn_mask, swin_shift, unet_param_spec


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted parameter names."""


def build_param_tree(root: nn.Module, spec, buffers, fill=None) -> None:
    """Register every `a.b.c.weight` of `spec` on nested sub-modules of `root` (order preserved)."""
    for name, shape in spec.items():
        parts = name.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Node())
            mod = mod._modules[p]
        if name in buffers:
            mod.register_buffer(parts[-1], fill(name, shape) if fill else torch.zeros(shape))
        else:
            mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape), requires_grad=False))


def params_version(module: nn.Module) -> int:
    """Cheap change detector: in-place writes (copy_, load_state_dict, .cuda()) bump tensor versions / ids."""
    v = 0
    for t in list(module.parameters()) + list(module.buffers()):
        v = (v * 1000003 + t._version + t.data_ptr()) & 0xFFFFFFFFFFFF
    return v


class UNetModelSwin(nn.Module):
    """Swin-UNet denoiser; constructor arguments identical to models/unet.py:632-657."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, use_fp16=False, num_heads=1, num_head_channels=-1,
                 use_scale_shift_norm=False, resblock_updown=False, swin_depth=2, swin_embed_dim=96, window_size=8, mlp_ratio=2.0,
                 patch_norm=False, cond_lq=True, cond_mask=False, lq_size=256):
        super().__init__()
        if dropout:
            raise NotImplementedError("inference engine: dropout must be 0")
        self.params = dict(image_size=image_size, in_channels=in_channels, model_channels=model_channels, out_channels=out_channels,
                           num_res_blocks=num_res_blocks, attention_resolutions=list(attention_resolutions), dropout=dropout,
                           channel_mult=list(channel_mult), conv_resample=conv_resample, dims=dims, use_fp16=use_fp16,
                           num_heads=num_heads, num_head_channels=num_head_channels, use_scale_shift_norm=use_scale_shift_norm,
                           resblock_updown=resblock_updown, swin_depth=swin_depth, swin_embed_dim=swin_embed_dim,
                           window_size=window_size, mlp_ratio=mlp_ratio, patch_norm=patch_norm, cond_lq=cond_lq, cond_mask=cond_mask,
                           lq_size=lq_size)
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.cond_lq, self.cond_mask = cond_lq, cond_mask
        self.dtype = torch.float16 if use_fp16 else torch.float32
        spec, buffers = unet_param_spec(self.params)
        self._spec, self._buffer_names = spec, buffers

        def fill(name, shape):
            if name.endswith("relative_position_index"):
                return relative_position_index(window_size)
            n_w = shape[0]
            side = int(round(n_w ** 0.5)) * window_size
            return shift_attn_mask(side, side, window_size, window_size // 2)

        build_param_tree(self, spec, buffers, fill)
        self.precision = F16 if use_fp16 else None  # None -> follow autocast / caller
        self._engine: Optional[Engine] = None
        self._engine_version = None

    # -- engine plumbing
    def engine(self) -> Engine:
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("UNetModelSwin (HIP engine) must be on a GPU: call .cuda() first; there is no CPU fallback")
        ver = params_version(self)
        if self._engine is None or self._engine.device != dev:
            self._engine = Engine(unet_params=self.params, device=dev)
            self._engine_version = None
        if self._engine_version != ver:
            self._engine.load_state_dicts(unet_sd=self.state_dict())
            self._engine_version = ver
        return self._engine

    def resolve_precision(self, prec=None) -> int:
        if prec is not None:
            return parse_precision(prec)
        if self.precision is not None:
            return self.precision
        return F16 if torch.is_autocast_enabled() else F32

    def forward(self, x, timesteps, lq=None, mask=None, prec=None):
        """x [N,C,H,W], timesteps [N] (int tensor or sequence), lq / mask as in models/unet.py:865-895."""
        if lq is not None:
            assert self.cond_lq
            if mask is not None:
                assert self.cond_mask
        ts = timesteps.tolist() if torch.is_tensor(timesteps) else list(timesteps)
        out = self.engine().unet_forward(x, ts, lq=lq, mask=mask, prec=self.resolve_precision(prec))
        return out.to(x.dtype)

    def convert_to_fp16(self):
        self.precision = F16

    def convert_to_fp32(self):
        self.precision = F32
