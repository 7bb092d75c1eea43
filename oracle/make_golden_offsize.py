"""Golden vectors for the RESOLUTION-GENERIC path: the reference networks run at a latent size other than the one they
were constructed for (what the tiled large-image mode does on every tile whose latent is not `image_size`:
sampler.py:186-208, inference_resshift.py:149-161).  What changes with the size, and what these vectors pin:

  * `SwinTransformerBlock.forward` recomputes the SW-MSA mask from the runtime size (`calculate_mask(x_size)`,
    models/swin_transformer.py:214-236,256-262) while `shift_size` / `window_size` stay what the CONSTRUCTED resolution made them
    (:189-194): the level that was a single unshifted 8x8 window at construction stays unshifted on a larger map;
  * non-square maps (window rows != window columns);
  * the autoencoder's global attention over T = h*w tokens (ldm/modules/diffusionmodules/model.py:179-203).

Produced by the UNMODIFIED reference modules (imported from /root/reference); the oracle restatement is asserted against every
output in the same run.  Stored: tests/golden/reference_offsize.npz.

    python -m oracle.make_golden_offsize          (build container only: needs /root/reference)
"""
from __future__ import annotations

import math
import os
import re
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cases, ref_import, resshift_oracle as oc, synth  # noqa: E402
from oracle.make_golden import SEED_W, check, ref_sample  # noqa: E402
from resshift_amd.config import load_config, to_plain  # noqa: E402
from resshift_amd.spec import ae_param_spec, unet_param_spec  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED_X = 321

# tag -> (unet params, ae params, diffusion params, with_mask, B, latent H, latent W)
TINY_CASES = {
    "tiny@48x32": (cases.TINY_UNET, cases.TINY_AE, cases.TINY_DIFFUSION, False, 2, 48, 32),       # 6x4 / 3x2 windows
    "tiny@32x16": (cases.TINY_UNET, cases.TINY_AE, cases.TINY_DIFFUSION, False, 1, 32, 16),       # 16x8 at the second level: one window column
    "tiny_fe@32x48": (cases.TINY_UNET_FE, cases.TINY_AE, cases.TINY_DIFFUSION_SF1, True, 1, 32, 48),  # feature extractor + mask, sf = 1
}
REALSR_SIDE = 128   # LR 128 x 128 -> latent 128 x 128 (constructed: 64), AE attention T = 16 384, image 512 x 512

# tiled path with tiles whose latent is NOT image_size: 32-pixel tiles of the tiny net (constructed for 16)
TILED = dict(chop_size=32, chop_stride=24, chop_bs=1, padding_offset=16, seed=5, H=56, W=40)


def case_inputs(tag):
    up, ap, dp, with_mask, B, hz, wz = TINY_CASES[tag]
    sf = dp["sf"]
    return synth.synthetic_inputs(SEED_X, B, hz * 4 // sf, wz * 4 // sf, ap["embed_dim"], hz, wz, dp["steps"], with_mask=with_mask)


def realsr_inputs(steps):
    return synth.synthetic_inputs(SEED_X, 1, REALSR_SIDE, REALSR_SIDE, 3, REALSR_SIDE, REALSR_SIDE, steps)


def tiled_inputs(steps):
    gen = torch.Generator().manual_seed(TILED["seed"])
    y = torch.rand(1, 3, TILED["H"], TILED["W"], generator=gen) * 2 - 1
    cs, st = TILED["chop_size"], TILED["chop_stride"]
    n_tiles = len(oc.tile_starts(TILED["H"], cs, st)) * len(oc.tile_starts(TILED["W"], cs, st))
    calls = [[torch.randn(1, 3, cs, cs, generator=gen) for _ in range(steps + 1)] for _ in range(n_tiles)]
    return y, calls


def build(U, V, unet_p, ae_p):
    usd = synth.synthetic_state_dict(unet_param_spec(unet_p)[0], SEED_W, image_size=unet_p["image_size"])
    asd = synth.synthetic_state_dict(ae_param_spec(ae_p), SEED_W)
    um = U(**unet_p).eval()
    um.load_state_dict(usd, strict=True)
    am = V(**ae_p).eval()
    am.load_state_dict(asd, strict=True)
    return usd, asd, um, am


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    U, V, create = ref_import.load()
    out = {}
    for tag, (up, ap, dp, with_mask, B, hz, wz) in TINY_CASES.items():
        print(f"[{tag}]")
        usd, asd, um, am = build(U, V, up, ap)
        y, noises, mask = case_inputs(tag)
        x, t = noises[1] * 1.3, torch.tensor([2] * B)
        kw = {"lq": y}
        if with_mask:
            kw["mask"] = mask
        ref_u = um(x, t, **kw)
        check(f"{tag}/unet", oc.unet_forward(usd, up, x, t, **kw), ref_u, 2e-5)
        out[f"{tag}/unet"] = ref_u.numpy()
        d = create(**dp)
        ref_img, ref_zf, ref_idx = ref_sample(d, um, am, y, noises, mask)
        o_img, aux = oc.sample_loop(usd, up, asd, ap, dp, y, noises, mask=mask, return_aux=True)
        check(f"{tag}/sample z_final", aux["z_final"], ref_zf, 2e-5)
        assert (aux["indices"] == ref_idx).float().mean().item() == 1.0
        check(f"{tag}/sample image", o_img, ref_img, 2e-5)
        out[f"{tag}/sample"] = ref_img.numpy()
        out[f"{tag}/sample_z"] = ref_zf.numpy()
        out[f"{tag}/sample_idx"] = ref_idx.numpy().astype(np.int32)

    # ---- tiled path, tile latent 32 x 32 on a network constructed for 16 x 16: the reference's own ImageSpliterTh
    print("[tiled, 32-pixel tiles of the tiny net]")
    up, ap, dp = cases.TINY_UNET, cases.TINY_AE, cases.TINY_DIFFUSION
    usd, asd, um, am = build(U, V, up, ap)
    d = create(**dp)
    sf, po = dp["sf"], TILED["padding_offset"]
    src = open(os.path.join(ref_import.REF, "utils", "util_image.py")).read()
    ns = {"torch": torch}
    exec(re.search(r"^class ImageSpliterTh:.*?(?=^class |\Z)", src, re.S | re.M).group(0), ns)
    y, calls = tiled_inputs(dp["steps"])

    def ref_sample_func(y0, noises):   # sampler.py:130-165 around the reference loop
        ori_h, ori_w = y0.shape[2:]
        if not (ori_h % po == 0 and ori_w % po == 0):
            y0 = F.pad(y0, pad=(0, math.ceil(ori_w / po) * po - ori_w, 0, math.ceil(ori_h / po) * po - ori_h), mode="reflect")
        img, _, _ = ref_sample(d, um, am, y0, noises)
        return img[:, :, : ori_h * sf, : ori_w * sf].clamp_(-1.0, 1.0)

    spliter = ns["ImageSpliterTh"](y, TILED["chop_size"], stride=TILED["chop_stride"], sf=sf, extra_bs=TILED["chop_bs"])
    for k, (pch, index_infos) in enumerate(spliter):
        spliter.update(ref_sample_func(pch, calls[k]), index_infos)
    ref = spliter.gather()
    got = oc.sample_tiled(usd, up, asd, ap, dp, y, calls, chop_size=TILED["chop_size"], chop_stride=TILED["chop_stride"],
                          chop_bs=TILED["chop_bs"], padding_offset=po)
    dmax = (got - ref).abs().max().item()
    print(f"  pin tiled32/sample: max|oracle-ref| = {dmax:.3e}, shape {tuple(ref.shape)}, {len(calls)} sampler calls")
    assert dmax <= 2e-5
    out["tiled32/sample"] = ref.numpy()

    # ---- the headline network (constructed for 64 x 64) on ONE 128 x 128 tile, B = 1, 15 steps
    print(f"[realsr @ {REALSR_SIDE} x {REALSR_SIDE}]")
    cfg = to_plain(load_config("realsr_swinunet_realesrgan256"))
    up, ap, dp = cfg["model"]["params"], cfg["autoencoder"]["params"], cfg["diffusion"]["params"]
    usd, asd, um, am = build(U, V, up, ap)
    y, noises, _ = realsr_inputs(dp["steps"])
    x = noises[1] * 1.3
    ref_u = um(x, torch.tensor([7]), lq=y)
    check("realsr128/unet", oc.unet_forward(usd, up, x, torch.tensor([7]), lq=y), ref_u, 2e-5)
    out["realsr128/unet"] = ref_u.numpy()
    d = create(**dp)
    t0 = time.time()
    ref_img, ref_zf, ref_idx = ref_sample(d, um, am, y, noises)
    print(f"  reference loop: {time.time() - t0:.1f} s")
    t0 = time.time()
    o_img, aux = oc.sample_loop(usd, up, asd, ap, dp, y, noises, return_aux=True)
    print(f"  oracle loop:    {time.time() - t0:.1f} s")
    check("realsr128/sample z_final", aux["z_final"], ref_zf, 2e-5)
    agree = (aux["indices"] == ref_idx).float().mean().item()
    print(f"  pin realsr128/sample indices agreement {agree:.5f}")
    assert agree >= 0.9999
    check("realsr128/sample image", o_img, ref_img, 5e-5)
    out["realsr128/sample"] = ref_img.numpy().astype(np.float16)
    out["realsr128/sample_z"] = ref_zf.numpy()
    out["realsr128/sample_idx"] = ref_idx.numpy().astype(np.int16)
    out["meta/seeds"] = np.array([SEED_W, SEED_X])
    path = os.path.join(GOLD, "reference_offsize.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
