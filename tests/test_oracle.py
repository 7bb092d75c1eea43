"""The oracle (oracle/resshift_oracle.py) against the committed golden outputs of the reference itself
(tests/golden/reference_outputs.npz, produced by oracle/make_golden.py from /root/reference), plus — where the
reference tree is present — a live re-check against the unmodified reference modules."""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import ref_import, resshift_oracle as oc

torch.set_grad_enabled(False)


def test_schedule_known_answers():
    """SURVEY.md §8c known-answer vector + the reference's own tables stored in the golden file."""
    _, _, dp = H.realsr_params()
    s = oc.Schedule(dp)
    kat = [0.02, 0.11716508, 0.17630456, 0.23362892, 0.29157413, 0.35100928, 0.41235614, 0.47585773, 0.54167168, 0.60990993,
           0.68065802, 0.75398526, 0.82995066, 0.90860639, 0.99]
    assert np.allclose(s.sqrt_etas, kat, atol=1e-8)
    assert np.allclose(s.posterior_mean_coef1[1:4], [0.02913826, 0.44164095, 0.56947397], atol=1e-8)
    assert np.allclose(s.posterior_variance[1:3], [0.00155338, 0.03065984], atol=1e-8)
    assert s.timestep_map == list(range(15))
    g = H.golden()
    for k in ("sqrt_etas", "posterior_mean_coef1", "posterior_mean_coef2", "posterior_variance", "posterior_log_variance_clipped"):
        assert np.array_equal(getattr(s, k), g[f"sched/realsr_swinunet_realesrgan256/{k}"])


def test_product_schedule_matches_reference_tables():
    from resshift_amd.gaussian_diffusion import create_gaussian_diffusion

    g = H.golden()
    for cname in ("realsr_swinunet_realesrgan256", "realsr_swinunet_realesrgan256_journal"):
        dp = H.to_plain(H.load_config(cname))["diffusion"]["params"]
        d = create_gaussian_diffusion(**dp)
        for k in ("sqrt_etas", "posterior_mean_coef1", "posterior_mean_coef2", "posterior_variance", "posterior_log_variance_clipped"):
            assert np.array_equal(getattr(d, k), g[f"sched/{cname}/{k}"]), (cname, k)


@pytest.mark.parametrize("tag", list(H.CASES))
def test_oracle_tiny_cases_match_reference_outputs(tag):
    up, ap, dp, with_mask = H.CASES[tag]
    g = H.golden()
    usd, asd = H.weights(up, ap)
    y, noises, mask = H.case_inputs(up, ap, dp, with_mask)
    x, t = noises[1] * 1.3, torch.tensor([2, 2])
    kw = {"lq": y}
    if with_mask:
        kw["mask"] = mask
    assert H.rel_err(oc.unet_forward(usd, up, x, t, **kw), torch.from_numpy(g[f"{tag}/unet"])) < 2e-5
    img = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).random((2, 3, 64, 64), dtype=np.float32) * 2 - 1)
    assert H.rel_err(oc.vq_encode(asd, ap, img), torch.from_numpy(g[f"{tag}/encode"])) < 2e-5
    d, idx = oc.vq_decode(asd, ap, noises[2] * 0.8, return_indices=True)
    assert H.rel_err(d, torch.from_numpy(g[f"{tag}/decode"])) < 2e-5
    assert np.array_equal(idx.numpy().astype(np.int32), g[f"{tag}/decode_idx"])
    out, aux = oc.sample_loop(usd, up, asd, ap, dp, y, noises, mask=mask, return_aux=True)
    assert H.rel_err(aux["z_final"], torch.from_numpy(g[f"{tag}/sample_z"])) < 5e-5
    assert (aux["indices"].numpy() == g[f"{tag}/sample_idx"]).mean() >= 0.995
    assert H.psnr(out.clamp(-1, 1), torch.from_numpy(g[f"{tag}/sample"]).clamp(-1, 1)) > 70.0


def test_oracle_full_size_realsr_matches_reference_output():
    """64x64 -> 256x256, 15 steps, B=1: the headline configuration at full size."""
    up, ap, dp = H.realsr_params()
    g = H.golden()
    usd, asd = H.weights(up, ap)
    y, noises, _ = H.synth.synthetic_inputs(H.SEED_X, 1, 64, 64, 3, 64, 64, dp["steps"])
    assert H.rel_err(oc.unet_forward(usd, up, noises[1] * 1.3, torch.tensor([7]), lq=y), torch.from_numpy(g["realsr/unet"])) < 2e-5
    out, aux = oc.sample_loop(usd, up, asd, ap, dp, y, noises, return_aux=True)
    assert H.rel_err(aux["z_final"], torch.from_numpy(g["realsr/sample_z"])) < 1e-4
    agree = (aux["indices"].numpy() == g["realsr/sample_idx"].astype(np.int64)).mean()
    assert agree >= 0.995, agree
    assert H.psnr(out.clamp(-1, 1), torch.from_numpy(g["realsr/sample"].astype(np.float32)).clamp(-1, 1)) > 60.0


@pytest.mark.parametrize("tag", ["tiny@48x32", "tiny@32x16", "tiny_fe@32x48"])
def test_oracle_offsize_matches_reference_outputs(tag):
    """The resolution-generic path: the networks run at a latent size other than the constructed one -
    per-size SW-MSA masks with the construction-time shift (models/swin_transformer.py:189-194,214-262), non-square maps.
    tests/golden/reference_offsize.npz holds the unmodified reference modules' outputs (oracle/make_golden_offsize.py)."""
    from oracle import make_golden_offsize as mo

    g = np.load(mo.os.path.join(mo.GOLD, "reference_offsize.npz"))
    up, ap, dp, with_mask, B, hz, wz = mo.TINY_CASES[tag]
    usd, asd = H.weights(up, ap)
    y, noises, mask = mo.case_inputs(tag)
    kw = {"lq": y}
    if with_mask:
        kw["mask"] = mask
    assert H.rel_err(oc.unet_forward(usd, up, noises[1] * 1.3, torch.tensor([2] * B), **kw), torch.from_numpy(g[f"{tag}/unet"])) < 2e-5
    out, aux = oc.sample_loop(usd, up, asd, ap, dp, y, noises, mask=mask, return_aux=True)
    assert H.rel_err(aux["z_final"], torch.from_numpy(g[f"{tag}/sample_z"])) < 5e-5
    assert (aux["indices"].numpy() == g[f"{tag}/sample_idx"]).mean() >= 0.995
    assert H.psnr(out.clamp(-1, 1), torch.from_numpy(g[f"{tag}/sample"]).clamp(-1, 1)) > 70.0


def test_oracle_offsize_full_network_and_tiles():
    """The headline network (constructed for 64 x 64 latents) on a 128 x 128 latent: one UNet forward against the reference's
    output (the 15-step loop at this size is pinned when the fixture is generated: 57 s of reference time); and the tiled path
    with 32-pixel tiles of the tiny network (tile latent 32 x 32, constructed 16 x 16) against the reference's ImageSpliterTh."""
    from oracle import make_golden_offsize as mo

    g = np.load(mo.os.path.join(mo.GOLD, "reference_offsize.npz"))
    up, ap, dp = H.realsr_params()
    usd, _ = H.weights(up, ap)
    y, noises, _ = mo.realsr_inputs(dp["steps"])
    assert H.rel_err(oc.unet_forward(usd, up, noises[1] * 1.3, torch.tensor([7]), lq=y), torch.from_numpy(g["realsr128/unet"])) < 2e-5
    up, ap, dp, _ = H.CASES["tiny"]
    usd, asd = H.weights(up, ap)
    y, calls = mo.tiled_inputs(dp["steps"])
    T = mo.TILED
    got = oc.sample_tiled(usd, up, asd, ap, dp, y, calls, chop_size=T["chop_size"], chop_stride=T["chop_stride"], chop_bs=T["chop_bs"],
                          padding_offset=T["padding_offset"])
    assert (got - torch.from_numpy(g["tiled32/sample"])).abs().max().item() <= 2e-5


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")
def test_oracle_offsize_live_against_reference_modules():
    """live: the reference UNet at 48 x 32 and 32 x 16 on a network constructed for 16 x 16 (bit-exact)"""
    U, _, _ = ref_import.load()
    up, ap, dp, _ = H.CASES["tiny"]
    usd, _ = H.weights(up, ap)
    um = U(**up).eval()
    um.load_state_dict(usd, strict=True)
    g = torch.Generator().manual_seed(11)
    for (h, w) in ((48, 32), (32, 16), (16, 48)):
        x, y = torch.randn(2, 3, h, w, generator=g), torch.rand(2, 3, h, w, generator=g) * 2 - 1
        assert torch.equal(oc.unet_forward(usd, up, x, torch.tensor([1, 1]), lq=y), um(x, torch.tensor([1, 1]), lq=y))


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")
def test_oracle_live_against_reference_modules():
    U, V, create = ref_import.load()
    up, ap, dp, _ = H.CASES["tiny"]
    usd, asd = H.weights(up, ap)
    um = U(**up).eval()
    um.load_state_dict(usd, strict=True)
    am = V(**ap).eval()
    am.load_state_dict(asd, strict=True)
    y, noises, _ = H.case_inputs(up, ap, dp, False)
    x, t = noises[1] * 1.3, torch.tensor([3, 3])
    assert torch.equal(oc.unet_forward(usd, up, x, t, lq=y), um(x, t, lq=y))
    z = am.encode(torch.nn.functional.interpolate(y, scale_factor=4, mode="bicubic"))
    assert H.rel_err(oc.vq_encode(asd, ap, torch.nn.functional.interpolate(y, scale_factor=4, mode="bicubic")), z) < 2e-5


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("cname", ["realsr_swinunet_realesrgan256", "faceir_gfpgan512_lpips", "inpaint_lama256_imagenet"])
def test_spec_equals_reference_state_dict(cname):
    """resshift_amd.spec must list exactly the reference's state_dict keys, shapes and order."""
    import os

    U, V, _ = ref_import.load()
    ref_cfg = H.to_plain(H.load_config(os.path.join(ref_import.REF, "configs", cname + ".yaml")))
    mine = H.to_plain(H.load_config(cname))
    for sec in ("model", "diffusion", "autoencoder"):
        assert ref_cfg[sec] == mine[sec], f"config digest drifted from the reference YAML: {cname}/{sec}"
    m = U(**ref_cfg["model"]["params"])
    spec, _ = H.unet_param_spec(mine["model"]["params"])
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == list(spec.items())
    a = V(**ref_cfg["autoencoder"]["params"])
    assert [(k, tuple(v.shape)) for k, v in a.state_dict().items()] == list(H.ae_param_spec(mine["autoencoder"]["params"]).items())


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")
def test_tiling_restatements_against_reference_ImageSpliterTh():
    """The reference tiler cannot be imported (utils/util_image.py pulls cv2 / skimage), so its class source is exec'd as
    is; the oracle's and the product's tile origin lists, and the oracle's sum/count averaging, must reproduce it."""
    import os
    import re

    from resshift_amd.tiling import extract_starts

    src = open(os.path.join(ref_import.REF, "utils", "util_image.py")).read()
    m = re.search(r"^class ImageSpliterTh:.*?(?=^class |\Z)", src, re.S | re.M)
    ns = {"torch": torch}
    exec(m.group(0), ns)
    Ref = ns["ImageSpliterTh"]
    g = torch.Generator().manual_seed(0)
    for (H, W, ps, st, sf, ebs) in [(96, 80, 64, 48, 4, 1), (130, 64, 64, 64, 2, 3), (64, 64, 64, 32, 4, 1), (200, 131, 64, 50, 1, 2)]:
        im = torch.randn(2, 3, H, W, generator=g)
        r = Ref(im, ps, st, sf=sf, extra_bs=ebs)
        assert r.height_starts_list == oc.tile_starts(H, ps, st) == extract_starts(H, ps, st)
        assert r.width_starts_list == oc.tile_starts(W, ps, st) == extract_starts(W, ps, st)
        # feed a deterministic per-tile "result" through both accumulators
        res = torch.zeros(2, 3, H * sf, W * sf)
        cnt = torch.zeros_like(res)
        for pch, infos in r:
            out = torch.nn.functional.interpolate(pch, scale_factor=sf, mode="nearest") * 1.5 + 0.25
            r.update(out, infos)
            for t, (h0, h1, w0, w1) in enumerate(infos):
                res[:, :, h0:h1, w0:w1] += out[t * 2:(t + 1) * 2]
                cnt[:, :, h0:h1, w0:w1] += 1
        assert torch.equal(r.gather(), res / cnt)


def test_tiled_path_oracle_vs_reference_golden():
    """tests/golden/reference_tiled.npz was produced by the reference's own ImageSpliterTh + UNet / VQ-AE / diffusion loop
    (oracle/make_golden_tiled.py); the oracle's tiled restatement must reproduce it wherever the tests run."""
    import os

    from oracle import make_golden_tiled as mt

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_tiled.npz"))
    assert list(g["meta"]) == [mt.CHOP_SIZE, mt.CHOP_STRIDE, mt.CHOP_BS, mt.PAD_OFFSET, mt.SEED]
    up, ap, dp, _ = H.CASES["tiny"]
    usd, asd = H.weights(up, ap)
    y, calls = mt.tiled_inputs(dp["steps"])
    got = oc.sample_tiled(usd, up, asd, ap, dp, y, calls, chop_size=mt.CHOP_SIZE, chop_stride=mt.CHOP_STRIDE, chop_bs=mt.CHOP_BS,
                          padding_offset=mt.PAD_OFFSET)
    assert (got - torch.from_numpy(g["sample"])).abs().max().item() <= 2e-5


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")
def test_respaced_schedule_live_against_reference():
    """timestep_respacing < steps (respace.py:23-70): the oracle's and the product's tables and timestep map against the
    reference's own SpacedDiffusion object."""
    from resshift_amd import create_gaussian_diffusion

    _, _, create = ref_import.load()
    dp = dict(H.CASES["tiny"][2], steps=12, timestep_respacing=4)
    ref = create(**dp)
    s = oc.Schedule(dp)
    mine = create_gaussian_diffusion(**dp)
    assert list(ref.timestep_map) == s.timestep_map == mine.timestep_map == [0, 3, 6, 9]
    for name in ("sqrt_etas", "etas", "posterior_mean_coef1", "posterior_mean_coef2", "posterior_log_variance_clipped"):
        r = np.asarray(getattr(ref, name), dtype=np.float64)
        assert np.array_equal(r, np.asarray(getattr(s, name), dtype=np.float64)), name
        assert np.array_equal(r, np.asarray(getattr(mine, name), dtype=np.float64)), name


@pytest.mark.skipif(not ref_import.available(), reason="reference modules (tree or verified oracle/_ref copy) not present")
def test_reference_baseline_wrapper_runs_the_reference_loop_and_agrees_with_the_oracle(capsys):
    """bench.py's baseline legs (oracle/ref_baseline.py: `cpu_baseline.kind = "reference"`, `torch_rocm_autocast_baseline`) drive the UNMODIFIED
    modules' p_sample_loop_progressive + decode_first_stage with injected noise.  On the tiny case: same image / latent / VQ indices as the
    oracle's restatement (which is pinned to the reference elsewhere in this file), and nothing on stdout - bench.py's stdout is ONE JSON line
    and the reference prints notices while it is imported and constructed."""
    from oracle import ref_baseline

    up, ap, dp, _ = H.CASES["tiny"]
    usd, asd = H.weights(up, ap)
    y, noises, _ = H.synth.synthetic_inputs(H.SEED_X, 2, 16, 16, ap["embed_dim"], 16, 16, dp["steps"])
    capsys.readouterr()
    ref = ref_baseline.Reference(up, ap, dp, usd, asd)
    img, z, idx = ref.sample(y, noises)
    assert capsys.readouterr().out == ""
    o_img, aux = oc.sample_loop(usd, up, asd, ap, dp, y, noises, return_aux=True)
    assert torch.equal(idx, aux["indices"].reshape(2, -1))
    assert (img - o_img).abs().max().item() < 1e-4 and (z - aux["z_final"]).abs().max().item() < 1e-4


def test_winograd_study_transform_is_the_direct_convolution():
    """oracle/study_winograd.py decides a kernel design on the CPU; its F(2x2, 3x3) emulation must BE a 3x3 / pad-1 convolution:
    fp32 operands reproduce F.conv2d to fp32 rounding, pair operands to the pair's 2^-22."""
    from oracle import study_winograd as sw

    g = torch.Generator().manual_seed(3)
    x, w, b = torch.randn(2, 32, 8, 12, generator=g), torch.randn(48, 32, 3, 3, generator=g) / 17.0, torch.randn(48, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    sd = {"t.weight": w, "t.bias": b}
    for split, tol in ((False, 2e-6), (True, 4e-6)):
        sw._U.clear()
        got = sw.wino(sd, "t", x, split)
        assert (got.double() - ref).abs().max().item() / ref.abs().max().item() < tol
