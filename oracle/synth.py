"""ORACLE support — test infrastructure only (see resshift_oracle.py header).

Deterministic synthetic weights and inputs.  The released ResShift checkpoints cannot be fetched
(no network) and the reference zero-initialises 167 tensors (models/unet.py:172-174,
models/basic_ops.py:64-70), so parity runs re-randomise EVERY tensor.  Values come from numpy's
PCG64 seeded per tensor name, hence are identical on any machine and independent of key order.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

from . import resshift_oracle as oc


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def synthetic_state_dict(spec: "Dict[str, Tuple[int, ...]]", seed: int, image_size: int = 64, window: int = 8) -> "OrderedDict[str, torch.Tensor]":
    """spec: key -> shape in the reference's state_dict naming (resshift_amd.spec builds it; tests pass it in)."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in spec.items():
        g = _rng(seed, name)
        if name.endswith("relative_position_index"):
            sd[name] = oc._rel_index(window)
            continue
        if name.endswith("attn_mask"):
            n_w = shape[0]
            side = int(round(n_w ** 0.5)) * window
            sd[name] = oc._shift_mask(side, side, window, window // 2)
            continue
        x = g.standard_normal(shape, dtype=np.float32)
        if name.endswith("relative_position_bias_table"):
            x *= 0.5
        elif name == "quantize.embedding.weight":
            x *= 0.6
        elif len(shape) == 1 and name.endswith(".weight"):  # GroupNorm gains
            x = 1.0 + 0.1 * x
        elif len(shape) == 1:  # biases (conv / linear / GroupNorm)
            x *= 0.05
        else:  # conv / linear weights: fan-in scaled
            fan_in = int(np.prod(shape[1:]))
            x *= 1.0 / np.sqrt(fan_in)
            if ".emb_layers." in name:
                x *= 0.5
            if name == "decoder.conv_out.weight":
                x *= 0.3  # keep most decoded pixels inside [-1,1] so that PSNR on the clamped image is meaningful
        sd[name] = torch.from_numpy(np.ascontiguousarray(x))
    return sd


def synthetic_inputs(seed: int, B: int, h: int, w: int, cz: int, hz: int, wz: int, steps: int, with_mask: bool = False):
    """LR batch in [-1,1], the (steps+1) noise tensors in draw order, and an optional inpainting mask."""
    g = _rng(seed, "inputs")
    y = torch.from_numpy(g.random((B, 3, h, w), dtype=np.float32) * 2 - 1)
    noises = [torch.from_numpy(g.standard_normal((B, cz, hz, wz), dtype=np.float32)) for _ in range(steps + 1)]
    mask = None
    if with_mask:
        mask = torch.from_numpy(((g.random((B, 1, h, w), dtype=np.float32) > 0.7).astype(np.float32)) * 2 - 1)
    return y, noises, mask
