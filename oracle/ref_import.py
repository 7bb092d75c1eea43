"""ORACLE support — imports the UNMODIFIED reference modules so that oracle/make_golden.py and tests/test_oracle.py can pin
the restatement in resshift_oracle.py against the real thing, and so that bench.py's baseline legs can time the reference itself.

Where the modules come from, in this order: $RESSHIFT_REFERENCE, /root/reference (the build container), oracle/_ref/reference_modules.zip
(the git-ignored archive oracle/make_ref_copy.py packs the hot-path modules into, byte for byte; it is what exists on the GPU box - the
archive's and every member's sha256 are verified against the manifest before anything is imported from it; Python imports from the zip).  The reference needs `timm` for three init helpers (models/swin_transformer.py:13); a stub
package providing them is injected.  Test / measurement infrastructure: nothing under resshift_amd/ imports this module.
"""
from __future__ import annotations

import os
import sys
import types

COPY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
ARCHIVE = "reference_modules.zip"


def _copy_ok() -> bool:
    """oracle/_ref/reference_modules.zip exists and it and every member match the manifest make_ref_copy.py wrote (an edited copy is not
    "the reference")"""
    import hashlib
    import json
    import zipfile

    mpath, zpath = os.path.join(COPY, "MANIFEST.json"), os.path.join(COPY, ARCHIVE)
    if not (os.path.exists(mpath) and os.path.exists(zpath)):
        return False
    try:
        with open(mpath) as fh:
            man = json.load(fh)
        with open(zpath, "rb") as fh:
            if hashlib.sha256(fh.read()).hexdigest() != man["archive_sha256"]:
                return False
        with zipfile.ZipFile(zpath) as z:
            names = {n for n in z.namelist() if not n.endswith("/")}
            if names != set(man["sha256"]):
                return False
            for rel, dig in man["sha256"].items():
                if hashlib.sha256(z.read(rel)).hexdigest() != dig:
                    return False
    except (OSError, KeyError, ValueError, zipfile.BadZipFile):
        return False
    return "models/unet.py" in man["sha256"]


def _copy_path() -> str:
    return os.path.join(COPY, ARCHIVE)


def where() -> "str | None":
    """directory the reference modules would be imported from, or None"""
    for cand in (os.environ.get("RESSHIFT_REFERENCE"), "/root/reference"):
        if cand and (os.path.isdir(os.path.join(cand, "models")) or (cand.endswith(".zip") and os.path.isfile(cand))):
            return cand
    return _copy_path() if _copy_ok() else None


REF = where() or "/root/reference"


def available() -> bool:
    return where() is not None


def is_copy() -> bool:
    return where() == _copy_path()


def load(prefer: "str | None" = None):
    """returns (UNetModelSwin, VQModelTorch, create_gaussian_diffusion) from the reference tree (or its verified copy)."""
    import torch

    global REF
    REF = prefer or where()
    if REF is None:
        raise RuntimeError("reference modules not present (neither /root/reference nor a verified oracle/_ref copy)")
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
