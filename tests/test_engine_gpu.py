"""Network-level parity on the GPU: the HIP engine (called through the drop-in classes / C ABI) against the CPU
oracle on identical seeded weights, inputs and injected noise, and against the committed reference outputs.

Tolerances (floating point path, per north_star):
  * fp32 mode (exact fp32 MFMA): relative max error <= 2e-4 per network call; image PSNR >= 60 dB;
  * fp16 mode (fp16 storage / fp32 accumulate): relative max error <= 3e-2 per network call; latent PSNR and
    VQ index agreement are reported, image PSNR with the reference's indices forced must be >= 60 dB
    (the VQ argmin discontinuity is the only thing separating the two — SURVEY.md fact 5).
"""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import resshift_oracle as oc

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

TOL_NET = {"fp32": 2e-4, "fp16": 3e-2, "split": 2e-4}   # split: (hi, lo) fp16 pairs + 3 MFMAs per product = fp32-class


def _shells(up, ap, usd, asd, dev):
    from resshift_amd import UNetModelSwin, VQModelTorch

    um = UNetModelSwin(**up).to(dev)
    um.load_state_dict(usd, strict=True)
    am = VQModelTorch(**ap).to(dev)
    am.load_state_dict(asd, strict=True)
    return um.eval(), am.eval()


@pytest.mark.parametrize("prec", ["fp32", "fp16", "split"])
@pytest.mark.parametrize("tag", list(H.CASES))
def test_unet_forward_vs_oracle(gpu, tag, prec):
    up, ap, dp, with_mask = H.CASES[tag]
    usd, asd = H.weights(up, ap)
    um, _ = _shells(up, ap, usd, asd, gpu)
    y, noises, mask = H.case_inputs(up, ap, dp, with_mask)
    x, t = noises[1] * 1.3, torch.tensor([2, 2])
    kw = {"lq": y}
    if with_mask:
        kw["mask"] = mask
    ref = oc.unet_forward(usd, up, x, t, **kw)
    got = um(x.to(gpu), t.to(gpu), prec=prec, **{k: v.to(gpu) for k, v in kw.items()})
    torch.cuda.synchronize()
    err = H.rel_err(got, ref)
    print(f"unet {tag} {prec}: rel err {err:.3e}")
    assert err < TOL_NET[prec]
    if prec == "fp32":  # also against the reference's own output
        assert H.rel_err(got, torch.from_numpy(H.golden()[f"{tag}/unet"])) < TOL_NET[prec]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "split"])
@pytest.mark.parametrize("tag", ["tiny", "tiny_fe8"])
def test_autoencoder_vs_oracle(gpu, tag, prec):
    up, ap, dp, _ = H.CASES[tag]
    usd, asd = H.weights(up, ap)
    _, am = _shells(up, ap, usd, asd, gpu)
    img = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).random((2, 3, 64, 64), dtype=np.float32) * 2 - 1)
    ref_z = oc.vq_encode(asd, ap, img)
    z = am.encode(img.to(gpu), prec=prec)
    torch.cuda.synchronize()
    err = H.rel_err(z, ref_z)
    print(f"encode {tag} {prec}: rel err {err:.3e}")
    assert err < TOL_NET[prec]
    _, noises, _ = H.case_inputs(up, ap, dp, False)
    zin = noises[2] * 0.8
    ref_d, ref_idx = oc.vq_decode(asd, ap, zin, return_indices=True)
    d, idx = am.decode(zin.to(gpu), prec=prec, return_indices=True)
    torch.cuda.synchronize()
    assert (idx.cpu().long() == ref_idx).float().mean().item() >= 0.995  # identical fp32 latent in: VQ must agree
    err = H.rel_err(d, ref_d)
    print(f"decode {tag} {prec}: rel err {err:.3e}")
    assert err < TOL_NET[prec]
    # force_not_quantize path
    ref_nq = oc.vq_decode(asd, ap, zin, force_not_quantize=True)
    assert H.rel_err(am.decode(zin.to(gpu), force_not_quantize=True, prec=prec), ref_nq) < TOL_NET[prec]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "split"])
@pytest.mark.parametrize("tag", list(H.CASES))
def test_sample_loop_vs_oracle(gpu, tag, prec):
    """The fused native loop (rs_sample) and the step-wise API against the oracle loop with injected noise."""
    from resshift_amd import create_gaussian_diffusion

    up, ap, dp, with_mask = H.CASES[tag]
    usd, asd = H.weights(up, ap)
    um, am = _shells(up, ap, usd, asd, gpu)
    y, noises, mask = H.case_inputs(up, ap, dp, with_mask)
    ref, aux = oc.sample_loop(usd, up, asd, ap, dp, y, noises, mask=mask, return_aux=True)
    d = create_gaussian_diffusion(**dp)
    d.set_precision(prec, prec, prec)
    kw = {"lq": y.to(gpu)}
    if with_mask:
        kw["mask"] = mask.to(gpu)
    out, gaux = d.p_sample_loop(y.to(gpu), um, first_stage_model=am, noise=noises[0].to(gpu), clip_denoised=False, model_kwargs=kw,
                                step_noises=[n.to(gpu) for n in noises[1:]], return_aux=True)
    torch.cuda.synchronize()
    zerr = H.rel_err(gaux["z_final"], aux["z_final"])
    agree = (gaux["indices"].cpu().long() == aux["indices"]).float().mean().item()
    p = H.psnr(out.cpu().clamp(-1, 1), ref.clamp(-1, 1))
    print(f"sample {tag} {prec}: latent rel err {zerr:.3e}, VQ agreement {agree:.4f}, image PSNR {p:.1f} dB")
    assert zerr < (6e-2 if prec == "fp16" else 5e-4)
    if prec == "split":
        assert agree >= 0.99 and p >= 60.0
    if prec == "fp32":
        assert agree >= 0.99 and p >= 60.0
        assert H.psnr(out.cpu().clamp(-1, 1), torch.from_numpy(H.golden()[f"{tag}/sample"]).clamp(-1, 1)) >= 60.0
        # the step-wise (generator) API must agree with the fused native loop
        finals = [o["sample"] for o in d.p_sample_loop_progressive(y.to(gpu), um, first_stage_model=am, noise=noises[0].to(gpu),
                                                                   clip_denoised=False, model_kwargs=kw,
                                                                   step_noises=[n.to(gpu) for n in noises[1:]])]
        assert len(finals) == dp["steps"]
        assert H.rel_err(finals[-1], gaux["z_final"]) < 1e-5


def test_realsr_full_size_vs_reference_output(gpu):
    """Headline config, 64->256, 15 steps, B=1, against the stored output of the reference itself."""
    from resshift_amd import create_gaussian_diffusion

    up, ap, dp = H.realsr_params()
    g = H.golden()
    usd, asd = H.weights(up, ap)
    um, am = _shells(up, ap, usd, asd, gpu)
    y, noises, _ = H.synth.synthetic_inputs(H.SEED_X, 1, 64, 64, 3, 64, 64, dp["steps"])
    ref_img = torch.from_numpy(g["realsr/sample"].astype(np.float32)).clamp(-1, 1)
    ref_z = torch.from_numpy(g["realsr/sample_z"])
    ref_idx = torch.from_numpy(g["realsr/sample_idx"].astype(np.int64))
    d = create_gaussian_diffusion(**dp)
    results = {}
    for prec in ("fp32", "fp16"):
        d.set_precision(prec, prec, prec)
        out, aux = d.p_sample_loop(y.to(gpu), um, first_stage_model=am, noise=noises[0].to(gpu), clip_denoised=False,
                                   model_kwargs={"lq": y.to(gpu)}, step_noises=[n.to(gpu) for n in noises[1:]], return_aux=True)
        torch.cuda.synchronize()
        agree = (aux["indices"].cpu().long() == ref_idx).float().mean().item()
        results[prec] = (H.psnr(out.cpu().clamp(-1, 1), ref_img), H.psnr(aux["z_final"].cpu(), ref_z, peak_to_peak=ref_z.max().item() - ref_z.min().item()), agree)
        print(f"realsr full {prec}: image PSNR {results[prec][0]:.1f} dB, latent PSNR {results[prec][1]:.1f} dB, VQ agreement {agree:.4f}")
    # UNet single forward against the reference's own output
    got = um(noises[1].to(gpu) * 1.3, [7], lq=y.to(gpu), prec="fp32")
    assert H.rel_err(got, torch.from_numpy(g["realsr/unet"])) < 2e-4
    assert results["fp32"][0] >= 60.0 and results["fp32"][2] >= 0.995
    assert results["fp16"][1] >= 40.0  # fp16 latent stays within fp16 tolerance of the fp32 trajectory


def test_sampler_drop_in_sample_func(gpu):
    """ResShiftSampler(configs, ...).sample_func on a non-multiple-of-64 input: pad / crop / clamp (sampler.py:119-165)."""
    from resshift_amd import ResShiftSampler
    from resshift_amd.config import ConfigNode

    up, ap, dp, _ = H.CASES["tiny"]
    usd, asd = H.weights(up, ap)
    cfg = ConfigNode(model=ConfigNode(target="models.unet.UNetModelSwin", ckpt_path=None, params=up),
                     diffusion=ConfigNode(target="models.script_util.create_gaussian_diffusion", params=dp),
                     autoencoder=ConfigNode(target="ldm.models.autoencoder.VQModelTorch", ckpt_path=None, params=ap))
    s = ResShiftSampler(cfg, sf=4, use_amp=False, padding_offset=16, seed=1, state_dicts={"model": usd, "autoencoder": asd})
    y, noises, _ = H.synth.synthetic_inputs(5, 2, 13, 10, 3, 16, 16, dp["steps"])
    ref = oc.sample_func(usd, up, asd, ap, dp, y, noises, padding_offset=16)
    out = s.sample_func(y.to(gpu), noise=noises[0].to(gpu), step_noises=[n.to(gpu) for n in noises[1:]])
    torch.cuda.synchronize()
    assert tuple(out.shape) == (2, 3, 52, 40) and out.abs().max().item() <= 1.0
    assert H.psnr(out.cpu(), ref) >= 60.0


@pytest.mark.parametrize("cname,hw,with_mask", [("realsr_swinunet_realesrgan256_journal", 64, False),
                                                ("faceir_gfpgan512_lpips", 512, False),
                                                ("inpaint_lama256_imagenet", 256, True)])
def test_other_baseline_configs_full_size_vs_oracle(gpu, cname, hw, with_mask):
    """BASELINE.json configs[2..4] at full size, B=1, exact kernels vs the CPU oracle on the same seeded weights / inputs /
    noise: 4-step journal SR, 512x512 face restoration (f8 autoencoder, 3-stage feature extractor, 8-channel latent) and
    256x256 inpainting (mask concatenated to the conditioning)."""
    from resshift_amd import create_gaussian_diffusion

    cfg = H.to_plain(H.load_config(cname))
    up, ap, dp = cfg["model"]["params"], cfg["autoencoder"]["params"], cfg["diffusion"]["params"]
    usd, asd = H.weights(up, ap)
    um, am = _shells(up, ap, usd, asd, gpu)
    f = 2 ** (len(ap["ddconfig"]["ch_mult"]) - 1)
    hz = hw * dp["sf"] // f
    y, noises, mask = H.synth.synthetic_inputs(H.SEED_X, 1, hw, hw, ap["embed_dim"], hz, hz, dp["steps"], with_mask=with_mask)
    ref, aux = oc.sample_loop(usd, up, asd, ap, dp, y, noises, mask=mask, return_aux=True)
    d = create_gaussian_diffusion(**dp)
    d.set_precision("fp32", "fp32", "fp32")
    kw = {"lq": y.to(gpu)}
    if with_mask:
        kw["mask"] = mask.to(gpu)
    out, gaux = d.p_sample_loop(y.to(gpu), um, first_stage_model=am, noise=noises[0].to(gpu), clip_denoised=False, model_kwargs=kw,
                                step_noises=[n.to(gpu) for n in noises[1:]], return_aux=True)
    torch.cuda.synchronize()
    agree = (gaux["indices"].cpu().long() == aux["indices"]).float().mean().item()
    p = H.psnr(out.cpu().clamp(-1, 1), ref.clamp(-1, 1))
    print(f"{cname} fp32: image PSNR {p:.1f} dB, VQ agreement {agree:.4f}, latent rel err {H.rel_err(gaux['z_final'], aux['z_final']):.2e}")
    assert agree >= 0.995 and p >= 60.0
    d.set_precision("fp16", "fp16", "fp16")
    out16, g16 = d.p_sample_loop(y.to(gpu), um, first_stage_model=am, noise=noises[0].to(gpu), clip_denoised=False, model_kwargs=kw,
                                 step_noises=[n.to(gpu) for n in noises[1:]], return_aux=True)
    torch.cuda.synchronize()
    zr = aux["z_final"]
    lat = H.psnr(g16["z_final"].cpu(), zr, peak_to_peak=(zr.max() - zr.min()).item())
    print(f"{cname} fp16: latent PSNR {lat:.1f} dB, image PSNR {H.psnr(out16.cpu().clamp(-1, 1), ref.clamp(-1, 1)):.1f} dB")
    assert lat >= 40.0


def test_tiled_large_image_path(gpu):
    """sampler.py:186-208 with ImageSpliterTh semantics: a 40x28 LR input, 16x16 tiles with stride 12, two tiles per
    sampler call, overlap-averaged on the GPU; injected per-call noise; exact kernels vs the CPU oracle."""
    from resshift_amd import ResShiftSampler
    from resshift_amd.config import ConfigNode

    up, ap, dp, _ = H.CASES["tiny"]
    usd, asd = H.weights(up, ap)
    cfg = ConfigNode(model=ConfigNode(target="models.unet.UNetModelSwin", ckpt_path=None, params=up),
                     diffusion=ConfigNode(target="models.script_util.create_gaussian_diffusion", params=dp),
                     autoencoder=ConfigNode(target="ldm.models.autoencoder.VQModelTorch", ckpt_path=None, params=ap))
    s = ResShiftSampler(cfg, sf=4, use_amp=False, chop_size=16, chop_stride=12, chop_bs=2, padding_offset=16, seed=1,
                        state_dicts={"model": usd, "autoencoder": asd})
    from oracle import make_golden_tiled as mt   # the very inputs of tests/golden/reference_tiled.npz

    y, calls = mt.tiled_inputs(dp["steps"])
    n_calls = len(calls)
    n_tiles = sum(c[0].shape[0] for c in calls)
    ref = oc.sample_tiled(usd, up, asd, ap, dp, y, calls, chop_size=16, chop_stride=12, chop_bs=2, padding_offset=16)
    out = s.sample_tiled(y.to(gpu), tile_noises=[(c[0].to(gpu), [n.to(gpu) for n in c[1:]]) for c in calls])
    torch.cuda.synchronize()
    assert tuple(out.shape) == (1, 3, 160, 112)
    p = H.psnr(out.cpu(), ref)
    import os

    gold = torch.from_numpy(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_tiled.npz"))["sample"])
    pg = H.psnr(out.cpu(), gold)   # the reference's own ImageSpliterTh + modules (oracle/make_golden_tiled.py)
    print(f"tiled path: {n_tiles} tiles in {n_calls} calls, PSNR {p:.1f} dB vs oracle, {pg:.1f} dB vs the reference output")
    assert p >= 60.0 and pg >= 60.0


def test_u8_pre_and_post_processing_on_device(gpu):
    """rs_u8_to_input / rs_output_to_u8 against the reference's host arithmetic (datapipe/datasets.py:59-63,
    sampler.py:218-222, utils/util_image.py:245-269): bit-exact, incl. the inpainting blend and the BGR order."""
    from resshift_amd.engine import Engine

    up, ap, _, _ = H.CASES["tiny"]
    eng = Engine(unet_params=up, ae_params=ap, device=gpu)
    g = torch.Generator().manual_seed(11)
    im = torch.randint(0, 256, (3, 20, 28, 3), generator=g, dtype=torch.uint8)
    x = eng.u8_to_input(im.to(gpu)).cpu()
    ref = ((im.float() / 255.0) - 0.5) / 0.5
    assert torch.equal(x, ref.permute(0, 3, 1, 2).contiguous())
    sr = torch.randn(3, 3, 20, 28, generator=g) * 0.8
    mask = ((torch.rand(3, 1, 20, 28, generator=g) > 0.6).float() - 0.5) / 0.5

    def host_post(sr, lq=None, mask=None):
        t = sr * 0.5 + 0.5
        if mask is not None:
            m = mask * 0.5 + 0.5
            t = t * m + (lq * 0.5 + 0.5) * (1 - m)
        return torch.from_numpy(np.round(t.clamp(0, 1).permute(0, 2, 3, 1).numpy() * 255.0).astype(np.uint8))

    assert torch.equal(eng.output_to_u8(sr.to(gpu)).cpu(), host_post(sr))
    assert torch.equal(eng.output_to_u8(sr.to(gpu), lq=x.to(gpu), mask=mask.to(gpu)).cpu(), host_post(sr, x, mask))
    assert torch.equal(eng.output_to_u8(sr.to(gpu), bgr=True).cpu(), host_post(sr).flip(-1))
    gray = torch.rand(2, 1, 9, 7, generator=g) * 2 - 1
    assert torch.equal(eng.output_to_u8(gray.to(gpu)).cpu(), host_post(gray))


def _tiny_cfg(up, ap, dp, unet_ckpt=None, ae_ckpt=None):
    from resshift_amd.config import ConfigNode

    return ConfigNode(model=ConfigNode(target="models.unet.UNetModelSwin", ckpt_path=unet_ckpt, params=up),
                      diffusion=ConfigNode(target="models.script_util.create_gaussian_diffusion", params=dp),
                      autoencoder=ConfigNode(target="ldm.models.autoencoder.VQModelTorch", ckpt_path=ae_ckpt, params=ap))


def test_checkpoint_ingestion_and_blob_cache(gpu, tmp_path):
    """.pth files as the reference ships them ({"state_dict": ...}, DDP `module.` prefix; sampler.py:106-117,
    utils/util_net.py:86-98) -> same samples as in-memory state_dicts; a second start from the packed-blob cache never
    opens the checkpoints."""
    from resshift_amd import ResShiftSampler

    up, ap, dp, _ = H.CASES["tiny"]
    usd, asd = H.weights(up, ap)
    torch.save({"state_dict": {"module." + k: v for k, v in usd.items()}}, tmp_path / "unet.pth")
    torch.save(dict(asd), tmp_path / "ae.pth")
    y, noises, _ = H.synth.synthetic_inputs(5, 2, 16, 16, 3, 16, 16, dp["steps"])
    kw = dict(sf=4, use_amp=False, padding_offset=16, seed=1)

    def run(s):
        o = s.sample_func(y.to(gpu), noise=noises[0].to(gpu), step_noises=[n.to(gpu) for n in noises[1:]])
        torch.cuda.synchronize()
        return o.cpu()

    ref = run(ResShiftSampler(_tiny_cfg(up, ap, dp), state_dicts={"model": usd, "autoencoder": asd}, **kw))
    cache = tmp_path / "weights.rsblob"
    got = run(ResShiftSampler(_tiny_cfg(up, ap, dp, str(tmp_path / "unet.pth"), str(tmp_path / "ae.pth")), blob_cache=str(cache), **kw))
    assert torch.equal(got, ref)
    assert cache.exists() and cache.stat().st_size > 1000
    again = run(ResShiftSampler(_tiny_cfg(up, ap, dp, "/nonexistent/unet.pth", "/nonexistent/ae.pth"), blob_cache=str(cache), **kw))
    assert torch.equal(again, ref)
    cache.write_bytes(cache.read_bytes()[:-7])  # truncated cache -> ignored, falls back to the checkpoints
    with pytest.raises((FileNotFoundError, OSError)):
        ResShiftSampler(_tiny_cfg(up, ap, dp, "/nonexistent/unet.pth", "/nonexistent/ae.pth"), blob_cache=str(cache), **kw)


def test_inference_files_on_device_uint8(gpu, tmp_path):
    """ResShiftSampler.inference (sampler.py:167-308): PNG in -> PNG out, one input larger than chop_size (tiled path)."""
    from PIL import Image
    from resshift_amd import ResShiftSampler

    up, ap, dp, _ = H.CASES["tiny"]
    usd, asd = H.weights(up, ap)
    s = ResShiftSampler(_tiny_cfg(up, ap, dp), sf=4, use_amp=False, chop_size=16, chop_stride=12, chop_bs=2, padding_offset=16, seed=3,
                        state_dicts={"model": usd, "autoencoder": asd})
    g = torch.Generator().manual_seed(2)
    (tmp_path / "in").mkdir()
    ims = {"a": torch.randint(0, 256, (16, 16, 3), generator=g, dtype=torch.uint8),
           "b": torch.randint(0, 256, (24, 20, 3), generator=g, dtype=torch.uint8)}
    for k, v in ims.items():
        Image.fromarray(v.numpy()).save(tmp_path / "in" / f"{k}.png")
    s.inference(tmp_path / "in", tmp_path / "out", bs=1, noise_repeat=True)
    for k, v in ims.items():
        out = torch.from_numpy(np.asarray(Image.open(tmp_path / "out" / f"{k}.png")))
        assert tuple(out.shape) == (v.shape[0] * 4, v.shape[1] * 4, 3)
        lq = s.engine.u8_to_input(v[None].to(gpu))
        exp = s.engine.output_to_u8(s.sample_tiled(lq, noise_repeat=True)).cpu()[0]
        assert torch.equal(out, exp)


def test_fused_swin_paths_match_unfused(gpu, tmp_path):
    """The fused Swin kernels (qkv + attention + projection, MLP) and the GroupNorm fold against the one-kernel-per-op path
    on a full-size fp16 UNet forward.  The knobs are read once per process, hence the subprocesses.  Folding GroupNorm into
    the consumers must not change a bit; fusing the GEMMs only reorders fp32 accumulation."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))

    def run(tag, **env):
        out = tmp_path / f"{tag}.pt"
        e = dict(os.environ, **{k: str(v) for k, v in env.items()})
        subprocess.run([sys.executable, os.path.join(here, "_unet_once.py"), str(out)], check=True, env=e, timeout=300)
        return torch.load(out)

    fused = run("fused")
    nofold = run("nofold", RS_GN_FOLD=0)
    plain = run("plain", RS_GN_FOLD=0, RS_ATTN_FUSED=0, RS_MLP_FUSED=0)
    assert torch.isfinite(fused).all()
    assert torch.equal(fused, nofold), "folding GroupNorm into the fused kernels changed the result"
    err = H.rel_err(fused, plain)
    print(f"fused vs unfused Swin path: rel err {err:.2e}")
    assert err < 5e-3
