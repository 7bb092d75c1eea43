"""Golden vector for the tiled large-image path, produced by the UNMODIFIED reference pieces: the reference's
`ImageSpliterTh` class (utils/util_image.py:889-979; the module itself cannot be imported - cv2 / skimage - so the class
source is exec'd as is) driving the reference UNet / VQ-AE / diffusion loop exactly like sampler.py:186-208 does, with
sampler.py:130-165's pad / crop / clamp around every tile batch.  Pins oracle.sample_tiled in the same run.

    python -m oracle.make_golden_tiled          (build container only: needs /root/reference)
"""
from __future__ import annotations

import math
import os
import re
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cases, ref_import, resshift_oracle as oc, synth  # noqa: E402
from oracle.make_golden import SEED_W, ref_sample  # noqa: E402
from resshift_amd.spec import ae_param_spec, unet_param_spec  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CHOP_SIZE, CHOP_STRIDE, CHOP_BS, PAD_OFFSET, SEED = 16, 12, 2, 16, 3


def tiled_inputs(steps):
    """Input image and the per-call noise lists, shared with tests/ (same generator, same draw order)."""
    gen = torch.Generator().manual_seed(SEED)
    y = torch.rand(1, 3, 40, 28, generator=gen) * 2 - 1
    n_tiles = len(oc.tile_starts(40, CHOP_SIZE, CHOP_STRIDE)) * len(oc.tile_starts(28, CHOP_SIZE, CHOP_STRIDE))
    calls = []
    for k in range((n_tiles + CHOP_BS - 1) // CHOP_BS):
        nb = min(CHOP_BS, n_tiles - CHOP_BS * k)
        calls.append([torch.randn(nb, 3, 16, 16, generator=gen) for _ in range(steps + 1)])
    return y, calls


def main():
    torch.set_grad_enabled(False)
    U, V, create = ref_import.load()
    up, ap, dp = cases.TINY_UNET, cases.TINY_AE, cases.TINY_DIFFUSION
    usd = synth.synthetic_state_dict(unet_param_spec(up)[0], SEED_W, image_size=up["image_size"])
    asd = synth.synthetic_state_dict(ae_param_spec(ap), SEED_W)
    um = U(**up).eval(); um.load_state_dict(usd, strict=True)
    am = V(**ap).eval(); am.load_state_dict(asd, strict=True)
    d = create(**dp)
    sf = dp["sf"]
    src = open(os.path.join(ref_import.REF, "utils", "util_image.py")).read()
    ns = {"torch": torch}
    exec(re.search(r"^class ImageSpliterTh:.*?(?=^class |\Z)", src, re.S | re.M).group(0), ns)
    y, calls = tiled_inputs(dp["steps"])

    def ref_sample_func(y0, noises):   # sampler.py:130-165 around the reference loop
        ori_h, ori_w = y0.shape[2:]
        if not (ori_h % PAD_OFFSET == 0 and ori_w % PAD_OFFSET == 0):
            pad_h = (math.ceil(ori_h / PAD_OFFSET)) * PAD_OFFSET - ori_h
            pad_w = (math.ceil(ori_w / PAD_OFFSET)) * PAD_OFFSET - ori_w
            y0 = F.pad(y0, pad=(0, pad_w, 0, pad_h), mode="reflect")
        img, _, _ = ref_sample(d, um, am, y0, noises)
        return img[:, :, : ori_h * sf, : ori_w * sf].clamp_(-1.0, 1.0)

    spliter = ns["ImageSpliterTh"](y, CHOP_SIZE, stride=CHOP_STRIDE, sf=sf, extra_bs=CHOP_BS)   # sampler.py:187-193
    for k, (pch, index_infos) in enumerate(spliter):
        spliter.update(ref_sample_func(pch, calls[k]), index_infos)                               # sampler.py:194-207
    ref = spliter.gather()
    got = oc.sample_tiled(usd, up, asd, ap, dp, y, calls, chop_size=CHOP_SIZE, chop_stride=CHOP_STRIDE, chop_bs=CHOP_BS,
                          padding_offset=PAD_OFFSET)
    dmax = (got - ref).abs().max().item()
    print(f"pin tiled/sample: max|oracle-ref| = {dmax:.3e}, shape {tuple(ref.shape)}, {len(calls)} sampler calls")
    assert dmax <= 2e-5
    path = os.path.join(GOLD, "reference_tiled.npz")
    np.savez_compressed(path, sample=ref.numpy(), meta=np.array([CHOP_SIZE, CHOP_STRIDE, CHOP_BS, PAD_OFFSET, SEED]))
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
