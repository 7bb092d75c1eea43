"""ORACLE support — test infrastructure only.  Bundles the reference's own low-resolution validation images
(testdata/Val_SR/lq/*.png: 32 RGB images of 64x64, SURVEY.md §8(d) input set (i)) as a small uint8 fixture, because
/root/reference does not exist on the GPU box.

    python -m oracle.make_real_inputs          # build container only -> tests/golden/val_sr_lq.npz
"""
from __future__ import annotations

import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/testdata/Val_SR/lq"
OUT = os.path.join(ROOT, "tests", "golden", "val_sr_lq.npz")


def main():
    from PIL import Image

    files = sorted(glob.glob(os.path.join(SRC, "*.png")))
    assert len(files) == 32, files
    ims = np.stack([np.asarray(Image.open(f).convert("RGB")) for f in files])
    assert ims.shape == (32, 64, 64, 3) and ims.dtype == np.uint8
    np.savez_compressed(OUT, lq=ims, names=np.array([os.path.basename(f) for f in files]))
    print(f"wrote {OUT}: {ims.shape} uint8, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
