"""Drop-in `VQModelTorch` (reference: ldm/models/autoencoder.py:12-50) backed by the HIP engine.

Same constructor (`ddconfig`, `n_embed`, `embed_dim`, ...), same `state_dict()` names/shapes, same
`encode(x)` / `decode(h, force_not_quantize=False)` contract; no torch compute inside.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .engine import F16, F32, Engine, parse_precision
from .spec import ae_param_spec
from .unet import build_param_tree, params_version


class VQModelTorch(nn.Module):
    def __init__(self, ddconfig, n_embed, embed_dim, remap=None, sane_index_shape=False):
        super().__init__()
        if remap is not None:
            raise NotImplementedError("remap is not used by any shipped config")
        self.params = dict(ddconfig=dict(ddconfig), n_embed=int(n_embed), embed_dim=int(embed_dim))
        self.embed_dim, self.n_embed = int(embed_dim), int(n_embed)
        self.sane_index_shape = sane_index_shape
        spec = ae_param_spec(self.params)
        self._spec = spec
        build_param_tree(self, spec, set())
        self.precision: Optional[int] = None  # None -> follow autocast / parameter dtype
        self._engine: Optional[Engine] = None
        self._engine_version = None

    def engine(self) -> Engine:
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("VQModelTorch (HIP engine) must be on a GPU: call .cuda() first; there is no CPU fallback")
        ver = params_version(self)
        if self._engine is None or self._engine.device != dev:
            self._engine = Engine(ae_params=self.params, device=dev)
            self._engine_version = None
        if self._engine_version != ver:
            self._engine.load_state_dicts(ae_sd=self.state_dict())
            self._engine_version = ver
        return self._engine

    def resolve_precision(self, prec=None) -> int:
        if prec is not None:
            return parse_precision(prec)
        if self.precision is not None:
            return self.precision
        if next(self.parameters()).dtype == torch.float16 or torch.is_autocast_enabled():
            return F16
        return F32

    def encode(self, x, prec=None):
        """autoencoder.py:28-31: quant_conv(Encoder(x)); x [B,3,H,W] -> [B,embed_dim,H/f,W/f]."""
        return self.engine().vq_encode(x, prec=self.resolve_precision(prec)).to(x.dtype)

    def decode(self, h, force_not_quantize=False, prec=None, return_indices=False):
        """autoencoder.py:33-40: Decoder(post_quant_conv(VQ(h))); the VQ re-quantisation is always on by default."""
        out = self.engine().vq_decode(h, force_not_quantize=force_not_quantize, prec=self.resolve_precision(prec),
                                      return_indices=return_indices)
        if return_indices:
            return out[0].to(h.dtype), out[1]
        return out.to(h.dtype)

    def forward(self, input, force_not_quantize=False):
        return self.decode(self.encode(input), force_not_quantize)
