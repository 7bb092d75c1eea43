"""Shared test helpers: synthetic weights keyed by the product's specs, golden fixtures, metrics."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import cases, synth  # noqa: E402
from resshift_amd.config import load_config, to_plain  # noqa: E402
from resshift_amd.spec import ae_param_spec, unet_param_spec  # noqa: E402

SEED_W, SEED_X = 1, 123
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_outputs.npz")

CASES = {
    "tiny": (cases.TINY_UNET, cases.TINY_AE, cases.TINY_DIFFUSION, False),
    "tiny_fe": (cases.TINY_UNET_FE, cases.TINY_AE, cases.TINY_DIFFUSION_SF1, True),
    "tiny_fe8": (cases.TINY_UNET_FE8, cases.TINY_AE8, cases.TINY_DIFFUSION_SF1, False),
}


def golden():
    return np.load(GOLDEN)


def realsr_params():
    cfg = to_plain(load_config("realsr_swinunet_realesrgan256"))
    return cfg["model"]["params"], cfg["autoencoder"]["params"], cfg["diffusion"]["params"]


def weights(unet_p, ae_p):
    uspec, _ = unet_param_spec(unet_p)
    usd = synth.synthetic_state_dict(uspec, SEED_W, image_size=unet_p["image_size"])
    asd = synth.synthetic_state_dict(ae_param_spec(ae_p), SEED_W)
    return usd, asd


def case_inputs(unet_p, ae_p, dp, with_mask, B=2):
    sf, steps = dp["sf"], dp["steps"]
    hz = unet_p["image_size"]
    h = hz * 4 // sf
    return synth.synthetic_inputs(SEED_X, B, h, h, ae_p["embed_dim"], hz, hz, steps, with_mask=with_mask)


def psnr(a, b, peak_to_peak=2.0):
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * math.log10(peak_to_peak ** 2 / mse)


def rel_err(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
