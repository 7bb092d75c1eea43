"""Generate tests/golden/*.npz by running the UNMODIFIED reference modules (imported from /root/reference)
on seeded synthetic weights/inputs, and pin oracle/resshift_oracle.py against them in the same run.

    python -m oracle.make_golden            (build container only: needs /root/reference)

Only outputs are stored; weights and inputs are regenerated from their seeds (oracle/synth.py) on both
sides.  The script asserts that the oracle restatement reproduces every reference output before writing.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cases, ref_import, resshift_oracle as oc, synth  # noqa: E402
from resshift_amd.config import load_config, to_plain  # noqa: E402
from resshift_amd.spec import ae_param_spec, unet_param_spec  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED_W, SEED_X = 1, 123


def ref_sample(diffusion, unet, ae, y, noises, mask=None):
    """Drive the reference's own p_sample_loop with injected noise: torch.randn_like is patched to pop from `noises`
    (the loop draws prior noise first, then once per step — gaussian_diffusion.py:446,358)."""
    import models.gaussian_diffusion as gd

    queue = list(noises)
    orig = gd.th.randn_like
    gd.th.randn_like = lambda t, *a, **k: queue.pop(0)
    try:
        kwargs = {"lq": y}
        if mask is not None:
            kwargs["mask"] = mask
        finals = []
        for out in diffusion.p_sample_loop_progressive(y, unet, first_stage_model=ae, noise=None, clip_denoised=False,
                                                       model_kwargs=kwargs):
            finals.append(out["sample"])
        with torch.no_grad():
            img = diffusion.decode_first_stage(finals[-1], first_stage_model=ae)
            _, _, (_, _, idx) = ae.quantize(finals[-1] / diffusion.scale_factor)
    finally:
        gd.th.randn_like = orig
    assert not queue
    return img, finals[-1], idx


def check(name, a, b, tol):
    d = (a - b).abs().max().item()
    s = b.abs().max().item()
    print(f"  pin {name:34s} max|oracle-ref| = {d:.3e} (scale {s:.3e})")
    assert d <= tol * max(1.0, s), f"oracle does not reproduce the reference for {name}"


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    U, V, create = ref_import.load()
    os.makedirs(GOLD, exist_ok=True)
    out = {}

    # ---- 1. schedule known-answer vectors straight from the reference's schedule code
    for cname in ("realsr_swinunet_realesrgan256", "realsr_swinunet_realesrgan256_journal"):
        dp = to_plain(load_config(cname))["diffusion"]["params"]
        d = create(**dp)
        s = oc.Schedule(dp)
        for k in ("sqrt_etas", "posterior_mean_coef1", "posterior_mean_coef2", "posterior_variance", "posterior_log_variance_clipped"):
            ref = np.asarray(getattr(d, k))
            assert np.array_equal(ref, getattr(s, k)), (cname, k)
            out[f"sched/{cname}/{k}"] = ref
        assert list(d.timestep_map) == list(s.timestep_map)

    # ---- 2. tiny networks: UNet forward (3 conditioning variants), AE encode/decode, full loops
    def build(unet_p, ae_p):
        uspec, _ = unet_param_spec(unet_p)
        usd = synth.synthetic_state_dict(uspec, SEED_W, image_size=unet_p["image_size"])
        asd = synth.synthetic_state_dict(ae_param_spec(ae_p), SEED_W)
        um = U(**unet_p).eval()
        um.load_state_dict(usd, strict=True)
        am = V(**ae_p).eval()
        am.load_state_dict(asd, strict=True)
        return usd, asd, um, am

    for tag, up, ap, dp, with_mask in (
        ("tiny", cases.TINY_UNET, cases.TINY_AE, cases.TINY_DIFFUSION, False),
        ("tiny_fe", cases.TINY_UNET_FE, cases.TINY_AE, cases.TINY_DIFFUSION_SF1, True),
        ("tiny_fe8", cases.TINY_UNET_FE8, cases.TINY_AE8, cases.TINY_DIFFUSION_SF1, False),
    ):
        print(f"[{tag}]")
        usd, asd, um, am = build(up, ap)
        sf, steps = dp["sf"], dp["steps"]
        cz = ap["embed_dim"]
        hz = up["image_size"]
        h = hz * 4 // sf  # LR size such that the latent is hz x hz with the f4 autoencoder
        y, noises, mask = synth.synthetic_inputs(SEED_X, 2, h, h, cz, hz, hz, steps, with_mask=with_mask)
        # UNet forward at t = 2 on a latent-like input
        x = noises[1] * 1.3
        t = torch.tensor([2, 2])
        kw = {"lq": y}
        if with_mask:
            kw["mask"] = mask
        ref_u = um(x, t, **kw)
        check(f"{tag}/unet", oc.unet_forward(usd, up, x, t, **kw), ref_u, 2e-5)
        out[f"{tag}/unet"] = ref_u.numpy()
        img = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).random((2, 3, 64, 64), dtype=np.float32) * 2 - 1)
        ref_z = am.encode(img)
        check(f"{tag}/encode", oc.vq_encode(asd, ap, img), ref_z, 2e-5)
        out[f"{tag}/encode"] = ref_z.numpy()
        zin = noises[2] * 0.8
        ref_d = am.decode(zin)
        o_d, o_idx = oc.vq_decode(asd, ap, zin, return_indices=True)
        check(f"{tag}/decode", o_d, ref_d, 2e-5)
        _, _, (_, _, ref_idx) = am.quantize(zin)
        assert torch.equal(o_idx, ref_idx)
        out[f"{tag}/decode"] = ref_d.numpy()
        out[f"{tag}/decode_idx"] = ref_idx.numpy().astype(np.int32)
        d = create(**dp)
        ref_img, ref_zf, ref_idx = ref_sample(d, um, am, y, noises, mask)
        o_img, aux = oc.sample_loop(usd, up, asd, ap, dp, y, noises, mask=mask, return_aux=True)
        check(f"{tag}/sample z_final", aux["z_final"], ref_zf, 2e-5)
        agree = (aux["indices"] == ref_idx).float().mean().item()
        print(f"  pin {tag}/sample indices agreement {agree:.4f}")
        assert agree == 1.0
        check(f"{tag}/sample image", o_img, ref_img, 2e-5)
        out[f"{tag}/sample"] = ref_img.numpy()
        out[f"{tag}/sample_z"] = ref_zf.numpy()
        out[f"{tag}/sample_idx"] = ref_idx.numpy().astype(np.int32)

    # ---- 3. the headline config at full size, B = 1 (64x64 -> 256x256, 15 steps)
    cfg = to_plain(load_config("realsr_swinunet_realesrgan256"))
    up, ap, dp = cfg["model"]["params"], cfg["autoencoder"]["params"], cfg["diffusion"]["params"]
    print("[realsr full size]")
    usd, asd, um, am = build(up, ap)
    y, noises, _ = synth.synthetic_inputs(SEED_X, 1, 64, 64, 3, 64, 64, dp["steps"])
    d = create(**dp)
    t0 = time.time()
    ref_img, ref_zf, ref_idx = ref_sample(d, um, am, y, noises)
    print(f"  reference loop: {time.time() - t0:.1f} s")
    t0 = time.time()
    o_img, aux = oc.sample_loop(usd, up, asd, ap, dp, y, noises, return_aux=True)
    print(f"  oracle loop:    {time.time() - t0:.1f} s")
    check("realsr/sample z_final", aux["z_final"], ref_zf, 2e-5)
    agree = (aux["indices"] == ref_idx).float().mean().item()
    print(f"  pin realsr/sample indices agreement {agree:.4f}")
    assert agree == 1.0
    check("realsr/sample image", o_img, ref_img, 2e-5)
    x = noises[1] * 1.3
    ref_u = um(x, torch.tensor([7]), lq=y)
    check("realsr/unet", oc.unet_forward(usd, up, x, torch.tensor([7]), lq=y), ref_u, 2e-5)
    out["realsr/unet"] = ref_u.numpy()
    out["realsr/sample"] = ref_img.numpy().astype(np.float16)  # 393 KB; fp16 rounding (2.4e-4) is far below the test tolerances
    out["realsr/sample_z"] = ref_zf.numpy()
    out["realsr/sample_idx"] = ref_idx.numpy().astype(np.int16)
    out["realsr/z_y"] = aux["z_y"].numpy()
    out["meta/seeds"] = np.array([SEED_W, SEED_X])
    out["meta/torch"] = np.array(torch.__version__)
    path = os.path.join(GOLD, "reference_outputs.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
