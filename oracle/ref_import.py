"""ORACLE support — only usable where /root/reference exists (the build container).

Imports the UNMODIFIED reference modules so that oracle/make_golden.py and tests/test_oracle.py can pin
the restatement in resshift_oracle.py against the real thing.  The reference needs `timm` for three
init helpers (models/swin_transformer.py:13); a stub package providing them is injected.
Nothing on the GPU box may import this module.
"""
from __future__ import annotations

import os
import sys
import types

REF = os.environ.get("RESSHIFT_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "models"))


def load():
    """returns (UNetModelSwin, VQModelTorch, create_gaussian_diffusion) from the reference tree."""
    import torch

    if not available():
        raise RuntimeError("reference tree not present")
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")

        class DropPath(torch.nn.Identity):
            def __init__(self, *a, **k):
                super().__init__()

        layers.DropPath = DropPath
        layers.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
        layers.trunc_normal_ = torch.nn.init.trunc_normal_
        timm.models = models
        models.layers = layers
        sys.modules["timm"] = timm
        sys.modules["timm.models"] = models
        sys.modules["timm.models.layers"] = layers
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from ldm.models.autoencoder import VQModelTorch
    from models.script_util import create_gaussian_diffusion
    from models.unet import UNetModelSwin

    return UNetModelSwin, VQModelTorch, create_gaussian_diffusion
