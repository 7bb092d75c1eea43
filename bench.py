#!/usr/bin/env python
"""Benchmark of the ResShift sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--precision POLICY] [--config NAME]

Workload (BASELINE.json configs[1], the default): realsr_swinunet_realesrgan256, 64x64 -> 256x256, 15 diffusion steps,
batch 32 per GPU, random-init weights, synthetic inputs already resident in HBM.  One "step" of this benchmark = one
pass of the whole hot path over one batch: bicubic x4 -> VQ-f4 encode -> prior sample -> 15 x (Swin-UNet + posterior
update) -> VQ lookup -> VQ-f4 decode, i.e. one `rs_sample` call of the engine.  `--config` selects the other BASELINE
configurations (journal: 4 steps; faceir: 512x512, f8 autoencoder, batch 16; inpaint: 256x256 + mask, batch 16).

N > 1: one process per GPU (sampler.py:66-77).  Either the caller launches the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: WORLD_SIZE / RANK are in the environment) or `python bench.py --gpus N` launches
them itself: it re-executes this file under torch.distributed.run with a free rendezvous port on 127.0.0.1.  Backend RCCL
("nccl") when N GPUs are visible; fewer GPUs than ranks is refused unless RESSHIFT_DIST_BACKEND=gloo asks for the
ranks-share-a-GPU plumbing configuration.  Rank 0 packs the weights and the blob reaches the other ranks through ONE
broadcast; every rank then processes its own batch (weak scaling, no data-path collective).  The line carries `n_gpus` =
the world size of the process group, the per-rank images/sec, and the broadcast's bytes and time.

Rank 0 prints ONE JSON line with the contract fields.  EVERYTHING at the top level of the line - `value`, `ms_per_step`, `dtype`,
`roofline` (+ `per_kernel`, `groupnorm`, `traffic`), `cpu_baseline.gpu_vs_cpu_psnr_db` - describes ONE precision policy, the one
`--precision` names.  The default is `parity` (split-precision encoder + UNet, fp16 decoder): the fastest policy that meets
north_star's tolerance (image PSNR >= 60 dB against the reference CPU path).  Beside it:
  * `roofline`      MFMA implicit-GEMM kernel family of the headline pass (hipEvent timed on the launch stream in a dedicated pass):
                    `frac` = algorithmic flops / time / 2.5 PFLOP/s (`frac_whole_path` the same for the whole pass),
                    `mfma_issue_frac` = matrix-pipe utilisation (split storage issues three MFMAs per product); `roofline.per_kernel` per
                    kernel family, `roofline.groupnorm` the HBM-bound GroupNorm family, all live; `roofline.traffic` /
                    `roofline.offline_rocprofv3` / `roofline.mfma_busy` replay digest-stamped rocprofv3 results of this command from
                    profiles/ (a process cannot attach rocprofv3 to itself) and say so;
  * `ms_per_unet_step`  SURVEY.md 8(d)'s ms/step: one UNet forward + sampler update at the bench batch (hipEvents);
  * `cpu_baseline`  the reference's OWN modules (kind "reference": the verified copy under oracle/_ref, made by oracle/make_ref_copy.py and
                    shipped with the snapshot) timed on this host's cores (N = 1 only, bounded sample: the first 8 images); kind "port" =
                    the oracle's restatement when the copy is absent;
  * `parity_vs_cpu_oracle`  image PSNR, latent PSNR and VQ code agreement of the headline policy against that CPU run on up to
                    `--parity-images` (default: all 32) images of the batch, per-image minimum included;
  * `value_fp16_unqualified`  the all-fp16 policy (BASELINE.json's dtype) timed the same way, with ITS parity: fast, and below the tolerance;
  * `torch_rocm_autocast_baseline`  the reference's own modules moved to this GPU and run with stock PyTorch-ROCm ops under
                    torch.autocast(fp16) (what sampler.py:185 does) at the bench batch: images/sec and ITS parity against the fp32 CPU path
                    (`torch_rocm_autocast_restatement_baseline`: the oracle's restatement instead, when oracle/_ref is absent).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from resshift_amd import sharding  # noqa: E402
from resshift_amd.autoencoder import VQModelTorch  # noqa: E402
from resshift_amd.config import load_config, to_plain  # noqa: E402
from resshift_amd.gaussian_diffusion import create_gaussian_diffusion  # noqa: E402
from resshift_amd.spec import ae_param_spec, random_state_dict, unet_param_spec  # noqa: E402
from resshift_amd.unet import UNetModelSwin  # noqa: E402

# name -> (yaml, LR side, default batch, algorithmic GFLOP per image [SURVEY.md §6 / §8(d), 2*MAC], description)
CONFIGS = {
    "realsr": ("realsr_swinunet_realesrgan256", 64, 32, 2535.7, "1x3x64x64 LR -> 3x256x256, 15 steps"),
    "journal": ("realsr_swinunet_realesrgan256_journal", 64, 32, 1421.1, "1x3x64x64 LR -> 3x256x256, 4 steps"),
    "faceir": ("faceir_gfpgan512_lpips", 512, 16, 1891.6, "1x3x512x512 -> 3x512x512 face restoration, f8 autoencoder, 4 steps"),
    "inpaint": ("inpaint_lama256_imagenet", 256, 16, 1426.7, "1x3x256x256 + mask -> 3x256x256 inpainting, 4 steps"),
}
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md; "split" = three fp16 MFMAs per product
MFMA_PEAK_TFLOPS = {"fp16": 2500.0, "fp32": 157.3, "split": 2500.0 / 3.0}
HBM_PEAK_GBS = 8000.0

# precision policies: which parts run fp16-storage MFMA, the exact fp32 MFMA, or split storage ((hi, lo) fp16 pairs, three
# fp16 MFMAs per product: fp32-class GEMMs at 1/3 of the fp16 matrix rate)
POLICIES = {
    "fp16": dict(unet="fp16", encode="fp16", decode="fp16"),
    "fp32": dict(unet="fp32", encode="fp32", decode="fp32"),
    "split": dict(unet="split", encode="split", decode="split"),
    # the parity-qualified policy: everything in front of the VQ argmin (ldm/modules/vqvae/quantize.py:276-285) fp32-class,
    # the decoder behind it in fp16
    "parity": dict(unet="split", encode="split", decode="fp16"),
}
PARITY_POLICY = "parity"
# a cheaper mixture (profiles/r2_precision_sweep_lastk.txt): the first MIXED_FP16_STEPS sampling steps (t = T-1 ...) in fp16 - their
# error is damped by the posterior coefficients on the way to the final latent - the remaining steps and the encoder in split
# precision, fp16 decoder.  Reported beside the all-split policy, never instead of it and never as `value_at_parity`: it sits AT the
# criterion (60.6 - 63.9 dB, 99.88 - 99.94 % codes depending on the images: 2 - 5 flipped VQ codes per image instead of 0 - 1 per batch).
MIXED_FP16_STEPS = 3


def policy_args(name: str, steps: int):
    if name in POLICIES:
        p = POLICIES[name]
        return [p["unet"]] * steps, p["encode"], p["decode"]
    if name == "parity_mixed":
        return ["fp16" if t >= steps - MIXED_FP16_STEPS else "split" for t in range(steps)], "split", "fp16"
    if name.startswith("mixed"):
        # "mixed<k>": the last k timesteps (t = k-1 .. 0) in fp32, everything else fp16
        k = int(name[5:] or 1)
        return ["fp32" if t < k else "fp16" for t in range(steps)], "fp16", "fp16"
    raise SystemExit(f"unknown precision policy {name}")


_T0 = time.time()


def log(msg: str) -> None:
    """progress on stderr (stdout carries only the JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def kernel_source_digest() -> str:
    """sha256 over the HIP sources: PMC traffic files under profiles/ are stamped with it and ignored when it differs"""
    from resshift_amd import build as _b

    return _b._digest()[:16]


def per_kernel_rooflines(eng):
    """MFMA-path kernel families of the engine's last (profiled) call: achieved TFLOP/s against the dense MFMA peak of the
    arithmetic they run (split storage: three fp16 MFMAs per product -> 2500 / 3)"""
    fam_peak = [MFMA_PEAK_TFLOPS["fp16"], MFMA_PEAK_TFLOPS["split"], MFMA_PEAK_TFLOPS["fp16"], MFMA_PEAK_TFLOPS["split"],
                MFMA_PEAK_TFLOPS["fp32"], MFMA_PEAK_TFLOPS["fp16"], MFMA_PEAK_TFLOPS["fp16"], MFMA_PEAK_TFLOPS["split"], MFMA_PEAK_TFLOPS["split"],
                MFMA_PEAK_TFLOPS["fp16"], MFMA_PEAK_TFLOPS["split"], MFMA_PEAK_TFLOPS["split"]]
    # `frac` is SURVEY.md §8(d)'s: ALGORITHMIC flops / time / the dense fp16 MFMA peak (2.5 PFLOP/s) - a split-storage kernel issues three
    # MFMAs per algorithmic product and is not credited for the two extra ones; `mfma_issue_frac` is the matrix-pipe utilisation (the
    # same figure against the peak of the arithmetic the kernel actually issues: 2500 / 3 for split storage, 157.3 for fp32 MFMA)
    return [{"kernel": name, "bound": "mfma", "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": MFMA_PEAK_TFLOPS["fp16"], "unit": "TFLOP/s",
             "frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS["fp16"], 4), "mfma_issue_peak": round(pk, 1),
             "mfma_issue_frac": round(fl / (ms * 1e-3) / 1e12 / pk, 4), "ms_per_step": round(ms, 2), "launches_per_step": n}
            for (name, fl, ms, n), pk in zip(eng.profile_families(), fam_peak) if n and ms > 0]


def dominant_kernel(shapes, eng):
    """the kernel instantiation + grid (family, M, N; all its K) with the largest summed kernel time of the profiled pass, as one roofline row"""
    if not shapes:
        return None
    # one kernel INSTANTIATION on one grid = (family, M, N): its launches differ only in the K loop's length (the input channel count)
    groups = {}
    for r in shapes:
        g = groups.setdefault((r["part"], r["family"], r["M"], r["N"], r["z"]), {"ms": 0.0, "flops": 0.0, "launches": 0, "K": []})
        g["ms"] += r["ms"]; g["flops"] += r["flops"]; g["launches"] += r["launches"]; g["K"].append(r["K"])
    key = max(groups, key=lambda k: groups[k]["ms"])
    g = groups[key]
    s = {"part": key[0], "family": key[1], "M": key[2], "N": key[3], "z": key[4], "K": sorted(g["K"]), "ms": g["ms"], "flops": g["flops"], "launches": g["launches"]}
    tf = s["flops"] / (s["ms"] * 1e-3) / 1e12 if s["ms"] > 0 else 0.0
    fam = eng.FAMILIES[s["family"]] if s["family"] < len(eng.FAMILIES) else str(s["family"])
    split = "split" in fam
    return {"kernel": fam, "part": s["part"], "shape": {"M": s["M"], "N": s["N"], "K": s["K"], "z": s["z"]}, "launches_per_step": s["launches"],
            "ms_per_step": round(s["ms"], 2), "us_per_launch": round(1e3 * s["ms"] / max(1, s["launches"]), 1), "achieved": round(tf, 1),
            "peak": MFMA_PEAK_TFLOPS["fp16"], "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS["fp16"], 4),
            "mfma_issue_frac": round(tf / MFMA_PEAK_TFLOPS["split" if split else "fp16"], 4)}


def mfma_ms_by_level(shapes, B):
    """MFMA-family kernel ms per (part, plane side): M = B * side^2 for the conv / token launches (batched attention GEMMs keep their own M)"""
    out = {}
    for s in shapes:
        side = int(round((s["M"] / max(1, B)) ** 0.5)) if s["z"] == 1 else 0
        key = f"{s['part']}@{side}" if side and side * side * B == s["M"] else f"{s['part']}@other"
        out[key] = out.get(key, 0.0) + s["ms"]
    return {k: round(v, 2) for k, v in sorted(out.items(), key=lambda kv: -kv[1])}


def psnr_db(a, b, p2p):
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else float(10 * np.log10(p2p * p2p / mse))


def offline_profile(kind: str, precision: str, config: str, B: int):
    """digest-stamped rocprofv3 results of THIS command collected offline (scripts/collect_traffic.py / collect_gn_trace.py) - or
    (None, why) when the file is missing, was measured on other kernel sources, or for another workload"""
    path = os.path.join(ROOT, "profiles", f"r6_{kind}_{precision}.json")
    if not (config == "realsr" and B == 32):
        return None, "offline rocprofv3 files exist for the default workload only"
    if not os.path.exists(path):
        return None, f"{os.path.relpath(path, ROOT)} not collected"
    with open(path) as fh:
        j = json.load(fh)
    if j.get("kernel_source_digest") != kernel_source_digest():
        return None, f"{os.path.relpath(path, ROOT)} was collected on other kernel sources: not reported"
    j["file"] = os.path.relpath(path, ROOT)
    return j, None


def tiled_main(args):
    """`--tiled HxW`: throughput of the TILED large-image path (utils/util_image.py:889-979 ImageSpliterTh, sampler.py:186-208,
    inference_resshift.py:149-161) - one LR image of HxW pixels cut into overlapping `--chop-size` tiles, `chop_bs` tiles per sampler call,
    overlap-averaged on the GPU.  One JSON line: tiles/s with chop_bs = 1 and batched, output megapixels/s.  (Parity of this path: tests/ -
    tests/golden/reference_tiled.npz from the reference's own ImageSpliterTh, and the chop-512 tile under the parity policy.)"""
    from resshift_amd import ResShiftSampler
    from resshift_amd.config import ConfigNode
    from resshift_amd.tiling import TileSplitter

    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    cname = CONFIGS[args.config][0]
    cfg = to_plain(load_config(cname))
    up, aep, dp = cfg["model"]["params"], cfg["autoencoder"]["params"], cfg["diffusion"]["params"]
    H, W = (int(v) for v in args.tiled.lower().split("x"))
    chop = args.chop_size
    stride = chop - max(16, chop // 8)   # (the reference's default overlap: chop 512 -> stride 448, inference_resshift.py:54-58)
    uspec, _ = unet_param_spec(up)
    sds = {"model": random_state_dict(uspec, seed=1), "autoencoder": random_state_dict(ae_param_spec(aep), seed=2)}
    conf = ConfigNode(model=ConfigNode(target="models.unet.UNetModelSwin", ckpt_path=None, params=up),
                      diffusion=ConfigNode(target="models.script_util.create_gaussian_diffusion", params=dp),
                      autoencoder=ConfigNode(target="ldm.models.autoencoder.VQModelTorch", ckpt_path=None, params=aep))
    g = torch.Generator().manual_seed(5)
    y = (torch.rand(1, 3, H, W, generator=g) * 2 - 1).to(dev)
    sf = int(dp.get("sf", 4))
    ntiles = len(TileSplitter(y, chop, stride=stride, sf=sf, extra_bs=1))
    res = {}
    for bs in sorted({1, min(args.chop_bs, ntiles)}):
        smp = ResShiftSampler(conf, sf=sf, use_amp=True, chop_size=chop, chop_stride=stride, chop_bs=bs, padding_offset=64, seed=7, state_dicts=sds,
                              precision=args.precision)
        out = None
        ts = []
        for it in range(1 + args.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = smp.sample_tiled(y, noise_repeat=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        assert tuple(out.shape) == (1, 3, sf * H, sf * W) and torch.isfinite(out).all() and out.abs().max().item() <= 1.0
        sec = float(np.median(ts[1:]))   # (the first call allocates the arena)
        res[bs] = {"chop_bs": bs, "seconds_per_image": round(sec, 4), "tiles_per_sec": round(ntiles / sec, 3),
                   "output_megapixels_per_sec": round(sf * H * sf * W / sec / 1e6, 3), "first_call_s": round(ts[0], 3),
                   "arena_gib": round(smp.model.engine().arena_bytes() / 2 ** 30, 2)}
        del smp
        torch.cuda.empty_cache()
    best = max(res.values(), key=lambda r: r["tiles_per_sec"])
    print(json.dumps({"metric": "tiles/sec, tiled large-image path (ImageSpliterTh-equivalent on the GPU), 15 steps", "value": best["tiles_per_sec"],
                      "unit": "tiles/s", "n_gpus": 1, "steps": args.steps, "warmup": 1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": f"{args.precision} policy", "data": "synthetic",
                      "config": {"workload": f"{cname}: one {H}x{W} LR image -> {sf * H}x{sf * W}, chop_size {chop}, stride {stride}, {ntiles} tiles",
                                 "precision_policy": args.precision},
                      "by_chop_bs": list(res.values())}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="realsr", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default: the config's BASELINE batch)")
    ap.add_argument("--precision", default=os.environ.get("RESSHIFT_PRECISION", PARITY_POLICY),
                    help="the policy the whole line describes (default: parity = the fastest policy that meets the 60 dB tolerance)")
    ap.add_argument("--parity-images", type=int, default=32, help="images of the batch compared with the CPU oracle (chunks of 8)")
    ap.add_argument("--cpu-seconds", type=float, default=200.0, help="budget of CPU-oracle time: further chunks of 8 images are skipped once it is spent")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle (no cpu_baseline / parity legs)")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary policies (fp16 / parity beside the headline)")
    ap.add_argument("--mixed-leg", action="store_true", help="also time the parity_mixed mixture (reported, never credited)")
    ap.add_argument("--exact-leg", action="store_true", help="also run the fp32-policy parity/timing leg (one pass)")
    ap.add_argument("--no-exact-leg", action="store_true", help=argparse.SUPPRESS)   # (round-3 flag, now the default)
    ap.add_argument("--no-torch-baseline", action="store_true", help="skip the PyTorch-ROCm autocast leg")
    ap.add_argument("--no-unet-step", action="store_true", help="skip the ms_per_unet_step leg (profiling passes: the trace / PMC files must hold whole passes only)")
    ap.add_argument("--no-reference-modules", action="store_true", help="baseline legs on the oracle's restatement even when oracle/_ref is present")
    ap.add_argument("--tiled", default=None, metavar="HxW", help="time the tiled large-image path on one HxW LR image instead of the batch workload (one GPU)")
    ap.add_argument("--chop-size", type=int, default=128, help="--tiled: LR tile side (the reference's default for x4 models is 512)")
    ap.add_argument("--chop-bs", type=int, default=8, help="--tiled: tiles per sampler call of the batched leg (chop_bs = 1 is always timed too)")
    args = ap.parse_args()

    if args.tiled:
        if args.gpus != 1:
            raise SystemExit("[bench] --tiled times one image on one GPU")
        return tiled_main(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not started by a launcher: become one (one process per GPU, sampler.py:66-77).  The children see WORLD_SIZE and skip this.
        import subprocess

        try:
            backend = sharding.pick_backend(args.gpus)
        except RuntimeError as ex:
            raise SystemExit(f"[bench] refused: {ex}")
        cmd = sharding.launch_command(os.path.abspath(__file__), sys.argv[1:], args.gpus)
        print(f"[bench] launching {args.gpus} ranks ({backend}): {' '.join(cmd)}", file=sys.stderr, flush=True)
        env = dict(os.environ, RESSHIFT_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.setdefault("NCCL_DEBUG", "VERSION")   # one line with the RCCL version on stderr when the communicator comes up
        raise SystemExit(subprocess.call(cmd, env=env))

    world, rank = sharding.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.set_grad_enabled(False)

    cname, lr_side, def_batch, gflop_per_image, cdesc = CONFIGS[args.config]
    cfg = to_plain(load_config(cname))
    up, aep, dp = cfg["model"]["params"], cfg["autoencoder"]["params"], cfg["diffusion"]["params"]
    steps = int(dp["steps"])
    B = args.batch or def_batch
    with_mask = bool(up.get("cond_mask", False))

    # ---- models: rank 0 creates random-init weights of the architecture and packs them; one RCCL broadcast
    model = UNetModelSwin(**up).to(dev).eval()
    ae = VQModelTorch(**aep).to(dev).eval()
    uspec, ubuf = unet_param_spec(up)

    def load_fn():
        return random_state_dict(uspec, seed=1), random_state_dict(ae_param_spec(aep), seed=2)

    # the engine packs (and broadcasts) only the weight forms this run will ask for: the headline policy's, the secondary policy's, fp32
    # only with --exact-leg
    forms = set(policy_args(args.precision, steps)[0]) | set(policy_args(args.precision, steps)[1:])
    if not args.no_secondary:
        forms |= {"fp16", "split"}
    if args.mixed_leg:
        forms |= {"fp16", "split"}
    if args.exact_leg:
        forms.add("fp32")
    log(f"building models + packing weights ({sorted(forms)})")
    t0 = time.time()
    eng = sharding.build_engine_with_broadcast(model, ae, load_fn, rank, world, precisions=forms)
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    log(f"weights ready in {setup_s:.1f}s")
    diffusion = create_gaussian_diffusion(**dp)
    diffusion.adopt_engine(model, ae, eng)
    pu, pe, pd = policy_args(args.precision, steps)
    diffusion.set_precision(pu, pe, pd)
    tables = diffusion.step_tables()

    # ---- synthetic inputs resident in HBM (different per rank); SURVEY.md §8(d) input recipe
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    f = 2 ** (len(aep["ddconfig"]["ch_mult"]) - 1)
    hz = lr_side * diffusion.sf // f
    cz = int(aep["embed_dim"])
    y = (torch.rand(B, 3, lr_side, lr_side, generator=g) * 2 - 1).to(dev)
    noise = torch.randn(steps + 1, B, cz, hz, hz, generator=g).to(dev)
    mask = ((torch.rand(B, 1, lr_side, lr_side, generator=g) > 0.7).float() * 2 - 1).to(dev) if with_mask else None

    def run(pol, return_aux=False):
        return eng.sample(y, noise, tables, sf=diffusion.sf, scale_factor=diffusion.scale_factor, mask=mask, prec_unet=pol[0],
                          prec_encode=pol[1], prec_decode=pol[2], return_aux=return_aux)

    def timed(pol, n):
        run(pol)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            run(pol)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    def timed_all_ranks(pol, n, warm):
        """barrier-bracketed, max over ranks: the contract's timed region"""
        for i in range(warm):
            o = run(pol)
            torch.cuda.synchronize()
            log(f"warmup pass {i} done (arena {eng.arena_bytes() / 2**30:.2f} GiB)")
        torch.cuda.synchronize()
        sharding.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            o = run(pol)
        torch.cuda.synchronize()
        sharding.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t, o

    headline = (pu, pe, pd)
    elapsed, out = timed_all_ranks(headline, args.steps, args.warmup)
    per_rank = sharding.allgather_floats([elapsed, float(torch.cuda.current_device())], dev)
    elapsed = sharding.allreduce_max(elapsed, dev)
    assert torch.isfinite(out).all().item(), "non-finite output"
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    launches = eng.last_launch_count()
    log(f"timed region done ({args.precision}): {ms_per_step:.1f} ms/step, {value:.2f} img/s, {launches} kernel launches per step")

    # ---- N > 1: the OTHER reference policy (fp16 when the headline is parity, parity otherwise) timed across all ranks exactly like the
    # headline (barriers, max over ranks), so that the scaling curve exists for both.  Parity against the CPU oracle is an N = 1 matter.
    other_policy_all_ranks = None
    if world > 1 and not args.no_secondary:
        oname = "fp16" if args.precision == PARITY_POLICY else PARITY_POLICY
        el_p, _ = timed_all_ranks(policy_args(oname, steps), args.steps, max(1, args.warmup))
        el_p = sharding.allreduce_max(el_p, dev)
        other_policy_all_ranks = {"policy": oname, "value": round(world * B * args.steps / el_p, 3), "unit": "images/sec",
                                  "ms_per_step": round(el_p / args.steps * 1e3, 3), "n_gpus": world, "steps": args.steps,
                                  "note": "timed like `value` (barrier-bracketed, max over ranks)"}
        log(f"{oname} policy across {world} ranks: {other_policy_all_ranks['value']} img/s")

    def profile_pass(pol, pname):
        """one pass with hipEvent brackets around every MFMA-family and GroupNorm launch (on the launch stream): the `roofline` block"""
        eng.profile_enable(True)
        run(pol)
        torch.cuda.synchronize()
        st = eng.profile_get()
        per_kernel = per_kernel_rooflines(eng)
        shapes, parts = eng.profile_shapes()
        eng.profile_enable(False)
        fl = {"fp16": st["flops_f16"], "fp32": st["flops_f32"], "split": st["flops_split"]}
        tot = sum(fl.values())
        dom = max(fl, key=fl.get)
        # time-weighted peak when a policy mixes MFMA flavours: peak_eff = total flops / sum(flops_i / peak_i)
        peak_eff = tot / sum(v / MFMA_PEAK_TFLOPS[k] for k, v in fl.items()) if tot else MFMA_PEAK_TFLOPS["fp16"]
        achieved = tot / (st["igemm_ms"] * 1e-3) / 1e12 if st["igemm_ms"] > 0 else 0.0
        roof = {
            "bound": "mfma", "policy": pname, "roofline_schema": 3,   # (3: + dominant_kernel, ms_by_part, mfma_ms_by_level; 2 = round 5: frac against 2.5 PFLOP/s)
            "kernel": "igemm4_kernel<*> (halo 3x3 conv) / wino_kernel<*> (Winograd 3x3 conv) / igemm_split_kernel<*> / igemm2_kernel<*> / igemm3_kernel<*> / igemm_kernel<*> + swin_mlp*_kernel / "
                      "win_attn_qkv*_kernel / ae_flash_attn*_kernel (MFMA implicit-GEMM family incl. the fused Swin kernels and the streaming autoencoder attention)",
            "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS["fp16"], "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS["fp16"], 4),
            "frac_note": "SURVEY.md 8(d): algorithmic flops (2*M*N*K of the launches of the MFMA family) / their hipEvent time / 2.5 PFLOP/s dense fp16; "
                         "the emulation's extra MFMAs (split storage: three per product) are NOT credited - that figure is mfma_issue_frac",
            "mfma_issue_peak": round(peak_eff, 1), "mfma_issue_frac": round(achieved / peak_eff, 4) if peak_eff else None,
            "mfma_issue_note": "matrix-pipe utilisation: against the time-weighted dense peak of the arithmetic this policy issues (split storage 2500 / 3, fp16 2500, fp32 157.3 TFLOP/s)",
            "frac_whole_path": round(gflop_per_image * 1e9 * B / (ms_per_step * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS["fp16"], 4),
            "traffic": None, "traffic_unit": "MB of HBM traffic per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": None,
            "algorithmic_mb_per_launch": round(st["igemm_bytes"] / max(1, st["igemm_launches"]) / 1e6, 2),
            "algorithmic_gflop_per_launch": round(tot / max(1, st["igemm_launches"]) / 1e9, 2),
            "launches_per_step": st["igemm_launches"], "avg_launch_us": round(st["igemm_ms"] * 1e3 / max(1, st["igemm_launches"]), 2),
            "algorithmic_gflop_per_image_igemm": round(tot / B / 1e9, 1), "algorithmic_gflop_per_image_total": gflop_per_image,
            "igemm_ms_per_step": round(st["igemm_ms"], 2), "whole_path_tflops": round(gflop_per_image * 1e9 * B / (ms_per_step * 1e-3) / 1e12, 2),
            "dominant_precision": dom,
            # ONE kernel instantiation - the launch shape with the largest summed kernel time of the pass - with its own fraction:
            # algorithmic flops of its launches / their hipEvent time / 2.5 PFLOP/s
            "dominant_kernel": dominant_kernel(shapes, eng),
            # where the pass goes: wall ms of the reference modules' counterparts, and the MFMA family's kernel ms by plane size
            "ms_by_part": {k: round(v, 2) for k, v in parts.items()},
            "mfma_ms_by_level": mfma_ms_by_level(shapes, B),
            # per kernel family (north_star: "per-kernel achieved-fraction-of-roofline"): same hipEvent brackets, grouped
            "per_kernel": per_kernel,
        }
        # HBM traffic of the same kernel family from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes of this
        # very command, corrected as MI355X_MICROARCH.md prescribes) - collected offline by scripts/collect_traffic.py into profiles/,
        # because a process cannot attach rocprofv3 to itself.  Stamped with the digest of the kernel sources; a stale file is not reported.
        tj, why = offline_profile("pmc_traffic", pname, args.config, B)
        if tj:
            roof["traffic"] = round(tj["hbm_bytes_per_launch"] / 1e6, 2)
            roof["traffic_source"] = f"replayed from {tj['file']} (offline rocprofv3 --pmc of this command on these kernel sources, not observed by this run)"
        else:
            roof["traffic_source"] = why
        # matrix-pipe utilisation from the SQ counters (SQ_VALU_MFMA_BUSY_CYCLES / SQ_LDS_* per kernel instantiation; scripts/collect_mfma_busy.py),
        # replayed like the traffic figure: one rocprofv3 --pmc pass of this command on these kernel sources, collected offline
        mj, mwhy = offline_profile("pmc_mfma_busy", pname, args.config, B)
        roof["mfma_busy"] = ({"source": f"replayed from {mj['file']} (offline rocprofv3 --pmc of this command on these kernel sources, not observed by this run)",
                              "family_mfma_busy": mj.get("family_mfma_busy"), "per_kernel": mj.get("per_kernel", [])[:6]} if mj else mwhy)
        if st.get("gn_launches"):
            gbs = st["gn_bytes"] / (st["gn_ms"] * 1e-3) / 1e9 if st["gn_ms"] > 0 else 0.0
            gn = {
                "bound": "hbm", "kernel": "gn_stats_kernel / gn_apply_kernel / gn_fused_kernel (GroupNorm32 + SiLU / FiLM)",
                "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "algorithmic_mb_per_launch": round(st["gn_bytes"] / st["gn_launches"] / 1e6, 3),
                "note": "live hipEvent brackets (they over-credit 6 - 14 us kernels: see offline_rocprofv3); algorithmic bytes = every GroupNorm pass's "
                        "input read once + output written once, nothing where producer and consumer fold the normalisation",
                "launches_per_step": st["gn_launches"], "avg_launch_us": round(st["gn_ms"] * 1e3 / st["gn_launches"], 2), "ms_per_step": round(st["gn_ms"], 2),
                "traffic": round(tj["groupnorm"]["hbm_bytes_per_launch"] / 1e6, 3) if tj and tj.get("groupnorm", {}).get("launches_fetch_pass") else None,
                "traffic_unit": "MB of HBM traffic per kernel launch of the family (PMC; replayed like roofline.traffic)",
            }
            gj, gwhy = offline_profile("gn_trace", pname, args.config, B)
            if gj:   # the rocprofv3 kernel trace resolves the small kernels; reported BESIDE the live figure, labelled
                gn["offline_rocprofv3"] = {"source": f"replayed from {gj['file']} (rocprofv3 --kernel-trace of this command on these kernel sources)",
                                           "ms_per_step": round(gj["ms_per_pass"], 2), "kernel_launches_per_step": gj["launches_per_pass"],
                                           "achieved_gbs": round(st["gn_bytes"] / (gj["ms_per_pass"] * 1e-3) / 1e9, 1),
                                           "frac": round(st["gn_bytes"] / (gj["ms_per_pass"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            else:
                gn["offline_rocprofv3"] = gwhy
            roof["groupnorm"] = gn
        return roof

    roofline = None
    if rank == 0 and not args.no_profile_pass:
        roofline = profile_pass(headline, args.precision)
        log(f"roofline ({args.precision}): {roofline['achieved']} of {roofline['peak']} TFLOP/s (MFMA issue: of {roofline['mfma_issue_peak']})")

    # ---- ms/step as SURVEY.md 8(d) defines it: ONE UNet forward + sampler update (models/unet.py:865-895 + gaussian_diffusion.py:332-365) at
    # batch B under the headline policy, through the step-wise API (`p_sample`), hipEvent-bracketed on the launch stream
    unet_step = None
    if rank == 0 and not args.no_unet_step:
        ti = steps // 2
        xs = noise[1].contiguous()
        nz = noise[2].contiguous()
        tt = torch.full((B,), ti, dtype=torch.long)   # (a HOST tensor: p_sample reads int(t[0]) - on a device tensor that is a sync inside the timed loop)
        mk = {"lq": y}
        if mask is not None:
            mk["mask"] = mask
        for _ in range(2):
            diffusion.p_sample(model, xs, None, tt, clip_denoised=False, model_kwargs=mk, noise=nz)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        nrep = 10
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(nrep):
            diffusion.p_sample(model, xs, None, tt, clip_denoised=False, model_kwargs=mk, noise=nz)
        ev[1].record()
        torch.cuda.synchronize()
        unet_step = round(ev[0].elapsed_time(ev[1]) / nrep, 3)
        log(f"ms_per_unet_step (one UNet forward + posterior update at batch {B}, t = {ti}): {unet_step}")

    # ---- CPU baseline: the oracle (CPU restatement of the reference, fp32) on a bounded sample of the same workload; parity of the GPU
    # policies against it on (up to) all images of the batch
    cpu_baseline = parity = value_at_parity = value_fp16 = torch_baseline = None
    torch_key = "torch_rocm_autocast_baseline"
    extra = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_baseline, resshift_oracle as oc  # checker / baseline only; never on the measured GPU path

        # use the cores this process may actually run on (cgroup/affinity aware), never more than torch's own default
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:
            usable = os.cpu_count() or 1
        torch.set_num_threads(max(1, min(usable, torch.get_num_threads(), 64)))
        usd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        asd = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
        # the UNMODIFIED reference modules when they can be imported (oracle/_ref: the verified copy oracle/make_ref_copy.py ships to the GPU
        # box), the oracle's restatement otherwise
        reference = None
        if ref_baseline.available() and not args.no_reference_modules:
            try:
                reference = ref_baseline.Reference(up, aep, dp, usd, asd)
                log(f"cpu baseline runs the reference's own modules: {reference.source}")
            except Exception as ex:   # a broken copy must not cost the bench line: fall back to the port and say so
                log(f"reference modules unusable ({type(ex).__name__}: {ex}); cpu baseline falls back to the oracle port")
                reference = None
        CH = 8                                        # images per CPU call: the TIMED sample is the first chunk
        want = max(1, min(args.parity_images, B))
        refs, zrefs, irefs, cpu_first_s, cpu_total_s, nb = [], [], [], None, 0.0, 0
        log(f"cpu baseline: oracle on {torch.get_num_threads()} threads, chunks of {CH} images, up to {want} images / {args.cpu_seconds:.0f} s")
        while nb < want:
            n1 = min(CH, want - nb)
            if nb and cpu_total_s + cpu_total_s / nb * n1 > args.cpu_seconds:
                log(f"cpu budget spent after {nb} images ({cpu_total_s:.0f} s)")
                break
            sl = slice(nb, nb + n1)
            t0 = time.perf_counter()
            if reference is not None:
                r_img, r_z, r_idx = reference.sample(y[sl].cpu(), [noise[k, sl].cpu() for k in range(steps + 1)],
                                                     mask=mask[sl].cpu() if mask is not None else None)
                r_aux = {"z_final": r_z, "indices": r_idx}
            else:
                r_img, r_aux = oc.sample_loop(usd, up, asd, aep, dp, y[sl].cpu(), [noise[k, sl].cpu() for k in range(steps + 1)],
                                              mask=mask[sl].cpu() if mask is not None else None, return_aux=True)
            dt = time.perf_counter() - t0
            if cpu_first_s is None:
                cpu_first_s, first_n = dt, n1
            cpu_total_s += dt
            refs.append(r_img); zrefs.append(r_aux["z_final"]); irefs.append(r_aux["indices"].reshape(n1, -1))
            nb += n1
            log(f"  oracle images {sl.start}..{sl.stop - 1}: {dt:.1f} s")
        ref, zr, iref = torch.cat(refs), torch.cat(zrefs), torch.cat(irefs)
        hw = zr.shape[2] * zr.shape[3]
        zp2p = (zr.max() - zr.min()).item()

        def parity_of(name, img, z, idx):
            """image / latent PSNR and VQ code agreement against the CPU oracle (first nb images of the batch), whole sample and worst image"""
            img = img[:nb].float().cpu().clamp(-1, 1)
            z = z[:nb].float().cpu()
            same = (idx.reshape(-1)[: nb * hw].cpu().long().reshape(nb, hw) == iref)
            per_img = [psnr_db(img[i], ref[i].clamp(-1, 1), 2.0) for i in range(nb)]
            return {"policy": name, "images": nb, "checker": "reference modules (CPU fp32)" if reference is not None else "oracle port (CPU fp32)",
                    "image_psnr_db": round(psnr_db(img, ref.clamp(-1, 1), 2.0), 1), "image_psnr_db_worst_image": round(min(per_img), 1),
                    "latent_psnr_db": round(psnr_db(z, zr, zp2p), 1),
                    "vq_code_agreement": round(same.float().mean().item(), 5),
                    "vq_code_agreement_worst_image": round(same.float().mean(dim=1).min().item(), 5),
                    "vq_codes_flipped": int((~same).sum().item())}

        def engine_parity(name, pol):
            o, aux = run(pol, return_aux=True)
            torch.cuda.synchronize()
            return parity_of(name, o, aux["z_final"], aux["indices"])

        def meets(p):
            return bool(p["image_psnr_db"] >= 60.0 and p["vq_code_agreement"] >= 0.999)

        hp = engine_parity(args.precision, headline)
        hp["meets_criterion"] = meets(hp)
        parity = [hp]
        ref_mod = None
        rpath = os.path.join(ROOT, "profiles", "ref_cpu_timing.json")
        if os.path.exists(rpath):
            with open(rpath) as fh:
                rj = json.load(fh)
            ref_mod = {"what": "the UNMODIFIED reference modules' own p_sample_loop, measured by oracle/time_reference.py in the build container "
                               "(the reference tree does not exist on the GPU box) - read from profiles/ref_cpu_timing.json, not timed by this run",
                       "cores": rj.get("cores"), "cpu": rj.get("cpu"), "torch": rj.get("torch"),
                       "images_per_sec_by_batch": {str(r["batch"]): r["reference_images_per_sec"] for r in rj.get("rows", [])},
                       "oracle_over_reference_speed_same_host": {str(r["batch"]): r["oracle_over_reference"] for r in rj.get("rows", [])}}
        cpu_baseline = {"value": round(first_n / cpu_first_s, 4), "unit": "images/sec", "cores": torch.get_num_threads(),
                        "kind": "reference" if reference is not None else "port",
                        "what": (f"the reference's own modules, unmodified ({reference.source}): models.unet.UNetModelSwin + ldm.models.autoencoder.VQModelTorch driven by "
                                 "GaussianDiffusion.p_sample_loop_progressive + decode_first_stage, CPU fp32, timed by THIS run on this host") if reference is not None else
                                "oracle/ (functional restatement of the reference on torch CPU ops, fp32; oracle/_ref was not shipped, so the reference's own "
                                "modules could not be timed here: see reference_modules for their build-container number)",
                        "sample": f"{first_n} images (the first chunk; {nb - first_n} more ran untimed for the baseline, for the parity check), same "
                                  f"weights/inputs/noise as the first images of the GPU batch, full {steps}-step loop incl. VQ encode/decode",
                        "seconds": round(cpu_first_s, 2), "cpu_seconds_total": round(cpu_total_s, 1), "reference_modules": ref_mod,
                        "gpu_vs_cpu_psnr_db": hp["image_psnr_db"], "gpu_vs_cpu_psnr_db_worst_image": hp["image_psnr_db_worst_image"],
                        "gpu_vs_cpu_vq_code_agreement": hp["vq_code_agreement"], "gpu_vs_cpu_images": nb, "gpu_policy": args.precision,
                        "gpu_over_cpu": round(value / (first_n / cpu_first_s), 1)}
        log(f"cpu baseline {cpu_baseline['value']} img/s; headline parity {hp}")

        def secondary(pname, label, with_roofline=True):
            pol = policy_args(pname, steps)
            par = engine_parity(label, pol)
            par["meets_criterion"] = meets(par)
            for _ in range(max(0, args.warmup - 1)):   # timed like the headline: same warm-up and step counts
                run(pol)
            ms = timed(pol, args.steps)
            par.update({"ms_per_step": round(ms, 2), "images_per_sec": round(B / ms * 1e3, 2), "steps_timed": args.steps, "warmup": args.warmup})
            if with_roofline and not args.no_profile_pass:
                rf = profile_pass(pol, pname)
                par["roofline"] = {k: rf[k] for k in ("achieved", "peak", "unit", "frac", "traffic", "igemm_ms_per_step", "per_kernel")}
                if "groupnorm" in rf:
                    par["groupnorm_ms_per_step"] = rf["groupnorm"]["ms_per_step"]
            parity.append(par)
            log(f"{pname}: {({k: v for k, v in par.items() if k != 'roofline'})}")
            return par

        if not args.no_secondary:
            if args.precision != "fp16":
                pf = secondary("fp16", "fp16 (all-fp16 storage: BASELINE.json's dtype)")
                value_fp16 = {"value": pf["images_per_sec"], "unit": "images/sec", "ms_per_step": pf["ms_per_step"], "policy": "fp16", "dtype": "f16",
                              "image_psnr_db": pf["image_psnr_db"], "vq_code_agreement": pf["vq_code_agreement"], "images": nb,
                              "meets_criterion": pf["meets_criterion"],
                              "note": "NOT the headline: the reference's decoder re-quantises the latent with an 8192-way argmin (ldm/modules/vqvae/quantize.py:276-285) and "
                                      "fp16 rounding in front of it flips VQ codes; reported because BASELINE.json's config names fp16"}
            if args.precision != PARITY_POLICY:
                secondary(PARITY_POLICY, PARITY_POLICY + " (split-precision encoder + UNet, fp16 decoder)")
            if args.mixed_leg and steps > MIXED_FP16_STEPS + 1:
                pm = secondary("parity_mixed", f"parity_mixed (first {MIXED_FP16_STEPS} steps fp16, then split; split encoder, fp16 decoder)", with_roofline=False)
                extra["value_at_parity_mixed"] = {"value": pm["images_per_sec"], "unit": "images/sec", "policy": pm["policy"], "image_psnr_db": pm["image_psnr_db"],
                                                  "vq_code_agreement": pm["vq_code_agreement"], "images": nb, "meets_criterion": pm["meets_criterion"]}
        qualified = [p for p in parity if p.get("meets_criterion") and not p["policy"].startswith("parity_mixed")]
        if qualified:
            best = max(qualified, key=lambda p: p.get("images_per_sec", value))
            value_at_parity = {"value": best.get("images_per_sec", round(value, 3)), "unit": "images/sec", "policy": best["policy"],
                               "is_headline": best is hp,
                               "criterion": "image PSNR >= 60 dB and VQ code agreement >= 0.999 vs the CPU oracle",
                               "image_psnr_db": best["image_psnr_db"], "image_psnr_db_worst_image": best["image_psnr_db_worst_image"],
                               "vq_code_agreement": best["vq_code_agreement"], "images": nb}
        # the exact-kernel policy beside it (fp32 storage, v_mfma_f32_16x16x4_f32)
        if args.exact_leg and args.precision != "fp32":
            p32 = policy_args("fp32", steps)
            par32 = engine_parity("fp32 (exact fp32 MFMA everywhere)", p32)
            ms32 = timed(p32, 1)
            par32.update({"ms_per_step": round(ms32, 2), "images_per_sec": round(B / ms32 * 1e3, 2), "steps_timed": 1})
            parity.append(par32)
        # the reference run by stock PyTorch-ROCm ops (MIOpen / hipBLASLt) on this GPU under torch.autocast(float16), as
        # sampler.py:185 runs it, at the bench batch - its throughput (second call: MIOpen's find step is in the first) and ITS distance from
        # the fp32 CPU path.  The UNMODIFIED modules when oracle/_ref travelled (key `torch_rocm_autocast_baseline`), the oracle's
        # restatement of them otherwise (key `torch_rocm_autocast_restatement_baseline`).
        if not args.no_torch_baseline:
            ng = [noise[k] for k in range(steps + 1)]
            try:
                if reference is not None:
                    reference.to(dev)

                    def torch_run():
                        return reference.sample(y, ng, mask=mask, autocast_dtype=torch.float16)
                else:
                    usd_g = {k: v.to(dev) for k, v in usd.items()}
                    asd_g = {k: v.to(dev) for k, v in asd.items()}

                    def torch_run():
                        with torch.autocast("cuda", dtype=torch.float16):
                            o, a = oc.sample_loop(usd_g, up, asd_g, aep, dp, y, ng, mask=mask, return_aux=True)
                        return o, a["z_final"], a["indices"]

                t0 = time.perf_counter()
                torch_run()
                torch.cuda.synchronize()
                first_s = time.perf_counter() - t0
                t0 = time.perf_counter()
                o_t, z_t, i_t = torch_run()
                torch.cuda.synchronize()
                t_s = time.perf_counter() - t0
                tb = parity_of("torch autocast(fp16)", o_t, z_t, i_t)
                torch_baseline = {"value": round(B / t_s, 2), "unit": "images/sec", "batch": B, "seconds": round(t_s, 3),
                                  "first_call_seconds": round(first_s, 2), "kind": "reference" if reference is not None else "restatement",
                                  "what": (f"the reference's own modules, unmodified ({reference.source}), moved to this GPU and run under torch.autocast(float16) "
                                           "as sampler.py:185 does" if reference is not None else
                                           "oracle/ RESTATEMENT of the reference (not the unmodified modules: oracle/_ref was not shipped) on PyTorch-ROCm ops under "
                                           "torch.autocast(float16)") + ", eager, batch = the bench batch, second call (MIOpen find-db warm)",
                                  "parity_vs_cpu_fp32": tb, "engine_over_torch": round(value / (B / t_s), 1)}
                log(f"torch autocast baseline: {torch_baseline}")
                del o_t, z_t, i_t
                if reference is not None:
                    reference.to("cpu")
                torch.cuda.empty_cache()
            except Exception as ex:  # the baseline is informational: never fail the bench line over it
                torch_baseline = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            torch_key = "torch_rocm_autocast_baseline" if reference is not None else "torch_rocm_autocast_restatement_baseline"

    if rank == 0:
        dtypes = sorted(set(pu) | {pe, pd})
        names = {"fp16": "f16", "fp32": "f32", "split": "f16x2 (hi+lo fp16 pairs, 3 MFMAs per product: fp32-class)"}
        line = {
            "metric": f"images/sec ({cdesc.split(',')[0]}, {steps}-step ResShift sampling loop incl. VQ encode/decode)",
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "ms_per_diffusion_step": round(ms_per_step / steps, 3), "ms_per_unet_step": unet_step,
            "ms_per_unet_step_note": f"SURVEY.md 8(d)'s ms/step: ONE UNet forward + sampler update at batch {B} under the headline policy (step-wise API, "
                                     "hipEvents on the launch stream); ms_per_step is one whole pass of the path over a batch, ms_per_diffusion_step that / steps (autoencoder smeared in)",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": names.get(dtypes[0], dtypes[0]) if len(dtypes) == 1 else
                     f"f16x2 encoder + UNet (hi+lo fp16 pairs, 3 MFMAs per product: fp32-class) + f16 decoder ({args.precision} policy)" if args.precision == PARITY_POLICY
                     else "+".join(dtypes) + f" ({args.precision})",
            "data": "synthetic",
            "config": {"workload": f"{cname}: batch {B}/GPU x {world} GPU, {cdesc}, random-init weights",
                       "precision_policy": args.precision, "kernel_launches_per_step": launches, "weight_setup_s": round(setup_s, 2),
                       "weight_forms_packed": sorted(forms), "weight_blob_mb": round(eng.weight_blob().numel() / 1e6, 1),
                       "parallelism": f"dp{world} (batch sharded, one RCCL weight broadcast, no data-path collective)"},
            "ranks": {"world_size": dist.get_world_size() if dist.is_initialized() else 1,
                      "backend": dist.get_backend() if dist.is_initialized() else None,
                      "rccl_version": sharding.dist_info()["rccl_version"],
                      "gpus_visible": torch.cuda.device_count(),
                      "per_rank": [{"rank": r, "device": int(v[1]), "images_per_sec": round(B * args.steps / v[0], 3)} for r, v in enumerate(per_rank)],
                      "weight_broadcast_bytes": int(getattr(eng, "broadcast_bytes", 0)),
                      "weight_broadcast_ms": round(1e3 * float(getattr(eng, "broadcast_s", 0.0)), 3)},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity_vs_cpu_oracle": parity,
            "value_at_parity": value_at_parity, "value_fp16_unqualified": value_fp16, "other_policy_all_ranks": other_policy_all_ranks,
            torch_key: torch_baseline,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        sharding.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
