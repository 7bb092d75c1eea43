"""ORACLE support — test infrastructure only.  Bundles the reference's own test inputs as small uint8 fixtures, because
/root/reference does not exist on the GPU box:
  * testdata/Val_SR/lq/*.png: 32 RGB images of 64x64 (SURVEY.md §8(d) input set (i))      -> tests/golden/val_sr_lq.npz
  * testdata/faceir/cropped_faces/lq/*.png: the first 8 of the 20 aligned 512x512 faces (faceir_gfpgan512_lpips.yaml's inputs,
    inference_resshift.py --task faceir)                                                     -> tests/golden/faceir_lq.npz
  * testdata/inpainting/imagenet/{lq,mask}/*.JPEG: all 6 masked 256x256 images and their masks (inpaint_lama256_imagenet.yaml;
    datapipe/datasets.py:428-473: lq and the gray mask are both mapped to [-1, 1] by (x / 255 - 0.5) / 0.5)   -> tests/golden/inpaint_imagenet.npz

    python -m oracle.make_real_inputs          # build container only
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


def main_faceir(n=8):
    from PIL import Image

    src = "/root/reference/testdata/faceir/cropped_faces/lq"
    files = sorted(glob.glob(os.path.join(src, "*.png")))[:n]
    ims = np.stack([np.asarray(Image.open(f).convert("RGB")) for f in files])
    assert ims.shape == (n, 512, 512, 3) and ims.dtype == np.uint8
    out = os.path.join(ROOT, "tests", "golden", "faceir_lq.npz")
    np.savez_compressed(out, lq=ims, names=np.array([os.path.basename(f) for f in files]))
    print(f"wrote {out}: {ims.shape} uint8, {os.path.getsize(out)} bytes")


def main_inpaint():
    from PIL import Image

    src = "/root/reference/testdata/inpainting/imagenet"
    files = sorted(glob.glob(os.path.join(src, "lq", "*.JPEG")))
    ims = np.stack([np.asarray(Image.open(f).convert("RGB")) for f in files])
    masks = np.stack([np.asarray(Image.open(os.path.join(src, "mask", os.path.basename(f))).convert("L")) for f in files])
    assert ims.shape == (len(files), 256, 256, 3) and masks.shape == (len(files), 256, 256)
    out = os.path.join(ROOT, "tests", "golden", "inpaint_imagenet.npz")
    np.savez_compressed(out, lq=ims, mask=masks, names=np.array([os.path.basename(f) for f in files]))
    print(f"wrote {out}: {ims.shape} + {masks.shape} uint8, {os.path.getsize(out)} bytes")


if __name__ == "__main__":
    main()
    main_faceir()
    main_inpaint()
