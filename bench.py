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

Rank 0 prints ONE JSON line with the contract fields plus
  * `roofline`      MFMA implicit-GEMM kernel family (hipEvent timed on the launch stream in a dedicated pass) and, under
                    `roofline.groupnorm`, the HBM-bound GroupNorm family measured the same way;
  * `cpu_baseline`  the CPU oracle timed on this host (N = 1 only, bounded sample);
  * `parity_vs_cpu_oracle` / `value_at_parity`  image PSNR, latent PSNR and VQ code agreement of the headline policy and of
                    the parity-qualified policy (split-precision encoder + UNet, fp16 decoder) against the CPU oracle on the
                    first images of the batch, and the throughput of that policy;
  * `torch_rocm_autocast_baseline`  the same restatement of the reference run with stock PyTorch-ROCm ops on this GPU under
                    torch.autocast(fp16) (what sampler.py:185 does): its images/sec and ITS parity against the fp32 CPU path.
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
                MFMA_PEAK_TFLOPS["fp16"]]
    return [{"kernel": name, "bound": "mfma", "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": round(pk, 1), "unit": "TFLOP/s",
             "frac": round(fl / (ms * 1e-3) / 1e12 / pk, 4), "ms_per_step": round(ms, 2), "launches_per_step": n}
            for (name, fl, ms, n), pk in zip(eng.profile_families(), fam_peak) if n and ms > 0]


def psnr_db(a, b, p2p):
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else float(10 * np.log10(p2p * p2p / mse))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="realsr", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default: the config's BASELINE batch)")
    ap.add_argument("--precision", default=os.environ.get("RESSHIFT_PRECISION", "fp16"))
    ap.add_argument("--parity-images", type=int, default=8, help="images of the batch compared with the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle (no cpu_baseline / parity legs)")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-exact-leg", action="store_true", help="skip the fp32-policy parity/timing leg")
    ap.add_argument("--no-torch-baseline", action="store_true", help="skip the PyTorch-ROCm autocast leg")
    args = ap.parse_args()

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

    log("building models + packing weights")
    t0 = time.time()
    eng = sharding.build_engine_with_broadcast(model, ae, load_fn, rank, world)
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

    headline = (pu, pe, pd)
    for i in range(args.warmup):
        out = run(headline)
        torch.cuda.synchronize()
        log(f"warmup pass {i} done (arena {eng.arena_bytes() / 2**30:.2f} GiB)")
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run(headline)
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank = sharding.allgather_floats([elapsed, float(torch.cuda.current_device())], dev)
    elapsed = sharding.allreduce_max(elapsed, dev)
    assert torch.isfinite(out).all().item(), "non-finite output"
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    log(f"timed region done: {ms_per_step:.1f} ms/step, {value:.2f} img/s")

    # ---- N > 1: the parity-qualified policy timed across all ranks exactly like the headline (barriers, max over ranks), so that the
    # scaling curve exists for the credited policy too.  Its parity against the CPU oracle is established by the N = 1 line
    # (`value_at_parity`) and the GPU tests; this leg only times it.
    value_parity_policy = None
    if world > 1 and args.precision != PARITY_POLICY:
        polp = policy_args(PARITY_POLICY, steps)
        for _ in range(max(1, args.warmup)):
            run(polp)
        torch.cuda.synchronize()
        sharding.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run(polp)
        torch.cuda.synchronize()
        sharding.barrier()
        torch.cuda.synchronize()
        el_p = sharding.allreduce_max(time.perf_counter() - t0, dev)
        value_parity_policy = {"value": round(world * B * args.steps / el_p, 3), "unit": "images/sec", "ms_per_step": round(el_p / args.steps * 1e3, 3),
                               "policy": PARITY_POLICY + " (split-precision encoder + UNet, fp16 decoder)", "n_gpus": world, "steps": args.steps,
                               "note": "timed like `value` (barrier-bracketed, max over ranks); the policy's parity against the CPU oracle is "
                                       "checked in the N = 1 line (`value_at_parity`) and in tests/test_engine_gpu.py"}
        log(f"parity policy across {world} ranks: {value_parity_policy['value']} img/s")

    # ---- roofline of the dominant kernel family (MFMA implicit GEMM) and of the GroupNorm family, measured in a dedicated
    # pass with hipEvents on the launch stream
    roofline = None
    launches = eng.last_launch_count()
    if rank == 0 and not args.no_profile_pass:
        eng.profile_enable(True)
        run(headline)
        torch.cuda.synchronize()
        st = eng.profile_get()
        eng.profile_enable(False)
        fl = {"fp16": st["flops_f16"], "fp32": st["flops_f32"], "split": st["flops_split"]}
        tot = sum(fl.values())
        dom = max(fl, key=fl.get)
        # time-weighted peak when a policy mixes MFMA flavours: peak_eff = total flops / sum(flops_i / peak_i)
        peak_eff = tot / sum(v / MFMA_PEAK_TFLOPS[k] for k, v in fl.items()) if tot else MFMA_PEAK_TFLOPS["fp16"]
        achieved = tot / (st["igemm_ms"] * 1e-3) / 1e12 if st["igemm_ms"] > 0 else 0.0
        # HBM traffic of the same kernel family from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc
        # passes of this very command, corrected as MI355X_MICROARCH.md prescribes) - collected offline by
        # scripts/collect_traffic.py into profiles/, because a process cannot attach rocprofv3 to itself.  The file is
        # stamped with the digest of the kernel sources it was measured on; a stale file is not reported.
        traffic = traffic_note = None
        tpath = os.path.join(ROOT, "profiles", f"r3_pmc_traffic_{args.precision}.json")
        gn_traffic = gn_trace_ms = None
        if args.config == "realsr" and B == 32 and os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("kernel_source_digest") == kernel_source_digest():
                traffic = round(tj["hbm_bytes_per_launch"] / 1e6, 2)  # MB per launch
                if tj.get("groupnorm", {}).get("launches_fetch_pass"):
                    gn_traffic = round(tj["groupnorm"]["hbm_bytes_per_launch"] / 1e6, 3)
            else:
                traffic_note = "profiles/ PMC file was collected on different kernel sources: not reported"
        roofline = {
            "bound": "mfma", "kernel": "igemm4_kernel<*> / igemm2_kernel<*> / igemm3_kernel<*> / igemm_split_kernel<*> / igemm_kernel<*> + swin_mlp*_kernel / win_attn_qkv*_kernel / ae_flash_attn_kernel (MFMA implicit-GEMM family incl. the fused Swin kernels and the streaming autoencoder attention)",
            "achieved": round(achieved, 2), "peak": round(peak_eff, 1), "unit": "TFLOP/s", "frac": round(achieved / peak_eff, 4) if peak_eff else None,
            "traffic": traffic, "traffic_unit": "MB of HBM traffic per launch (PMC)", "traffic_note": traffic_note,
            "algorithmic_mb_per_launch": round(st["igemm_bytes"] / max(1, st["igemm_launches"]) / 1e6, 2),
            "algorithmic_gflop_per_launch": round(tot / max(1, st["igemm_launches"]) / 1e9, 2),
            "launches_per_step": st["igemm_launches"], "avg_launch_us": round(st["igemm_ms"] * 1e3 / max(1, st["igemm_launches"]), 2),
            "algorithmic_gflop_per_image_igemm": round(tot / B / 1e9, 1), "algorithmic_gflop_per_image_total": gflop_per_image,
            "igemm_ms_per_step": round(st["igemm_ms"], 2), "whole_path_tflops": round(gflop_per_image * 1e9 * B / (ms_per_step * 1e-3) / 1e12, 2),
            "dominant_precision": dom,
        }
        # per kernel family (north_star: "per-kernel achieved-fraction-of-roofline"): same hipEvent brackets, grouped
        roofline["per_kernel"] = per_kernel_rooflines(eng)
        # GroupNorm family time from the committed rocprofv3 kernel trace of this command (scripts/collect_gn_trace.py; digest-stamped):
        # the hipEvent brackets cannot resolve 6 - 14 us kernels (their fixed cost is comparable to the kernels), the trace can
        gpath = os.path.join(ROOT, "profiles", f"r3_gn_trace_{args.precision}.json")
        if args.config == "realsr" and B == 32 and os.path.exists(gpath):
            with open(gpath) as fh:
                gj = json.load(fh)
            if gj.get("kernel_source_digest") == kernel_source_digest():
                gn_trace_ms = gj["ms_per_pass"]
        if st.get("gn_launches"):
            gn_ms = gn_trace_ms if gn_trace_ms else st["gn_ms"]
            gbs = st["gn_bytes"] / (gn_ms * 1e-3) / 1e9 if gn_ms > 0 else 0.0
            roofline["groupnorm"] = {
                "bound": "hbm", "kernel": "gn_stats_kernel / gn_apply_kernel / gn_fused_kernel (GroupNorm32 + SiLU / FiLM)",
                "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "algorithmic_mb_per_launch": round(st["gn_bytes"] / st["gn_launches"] / 1e6, 3),
                "note": "algorithmic bytes = every GroupNorm input read once + every output written once",
                "launches_per_step": st["gn_launches"], "avg_launch_us": round(gn_ms * 1e3 / st["gn_launches"], 2),
                "ms_per_step": round(gn_ms, 2), "ms_per_step_source": "rocprofv3 kernel trace (profiles/)" if gn_trace_ms else "hipEvent brackets",
                "ms_per_step_hipevents": round(st["gn_ms"], 2), "traffic": gn_traffic,
                "traffic_unit": "MB of HBM traffic per kernel launch of the family (PMC)",
            }

    # ---- CPU baseline: the oracle (CPU restatement of the reference, fp32) on a bounded sample of the same workload; parity
    # of the GPU policies against it on the same images
    cpu_baseline = parity = value_at_parity = torch_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import resshift_oracle as oc  # checker / baseline only; never on the measured GPU path

        # use the cores this process may actually run on (cgroup/affinity aware), never more than torch's own default
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:
            usable = os.cpu_count() or 1
        torch.set_num_threads(max(1, min(usable, torch.get_num_threads(), 64)))
        nb = max(1, min(args.parity_images, B))
        log(f"cpu baseline: oracle on {torch.get_num_threads()} threads, {nb} images")
        usd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        asd = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
        yc = y[:nb].cpu()
        nz = [noise[k, :nb].cpu() for k in range(steps + 1)]
        mc = mask[:nb].cpu() if mask is not None else None
        t0 = time.perf_counter()
        ref, ref_aux = oc.sample_loop(usd, up, asd, aep, dp, yc, nz, mask=mc, return_aux=True)
        cpu_s = time.perf_counter() - t0
        zr = ref_aux["z_final"]
        hw = zr.shape[2] * zr.shape[3]

        def parity_of(name, img, z, idx):
            """image / latent PSNR and VQ code agreement against the CPU oracle (first nb images of the batch)"""
            return {"policy": name, "images": nb,
                    "image_psnr_db": round(psnr_db(img[:nb].float().cpu().clamp(-1, 1), ref.clamp(-1, 1), 2.0), 1),
                    "latent_psnr_db": round(psnr_db(z[:nb].float().cpu(), zr, (zr.max() - zr.min()).item()), 1),
                    "vq_code_agreement": round((idx.reshape(-1)[: nb * hw].cpu().long() == ref_aux["indices"].reshape(-1)).float().mean().item(), 5)}

        def engine_parity(name, pol):
            o, aux = run(pol, return_aux=True)
            torch.cuda.synchronize()
            return parity_of(name, o, aux["z_final"], aux["indices"])

        parity = [engine_parity(args.precision, headline)]
        cpu_baseline = {"value": round(nb / cpu_s, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
                        "what": "oracle/ (functional restatement of the reference on torch CPU ops; 20-35 % FASTER than the reference "
                                "modules' own loop at B=1 (box-dependent, oracle/make_golden.py), so GPU/CPU ratios are understated)",
                        "sample": f"{nb} images, same weights/inputs/noise as the first {nb} images of the GPU batch, full {steps}-step loop, fp32",
                        "seconds": round(cpu_s, 2), "gpu_vs_cpu_psnr_db": parity[0]["image_psnr_db"]}
        log(f"cpu baseline {cpu_baseline['value']} img/s; headline parity {parity[0]}")
        # the parity-qualified policy: throughput + parity on the same inputs
        if args.precision != PARITY_POLICY:
            pol = policy_args(PARITY_POLICY, steps)
            par = engine_parity(PARITY_POLICY + " (split-precision encoder + UNet, fp16 decoder)", pol)
            for _ in range(max(0, args.warmup - 1)):   # timed like the headline: same warm-up and step counts
                run(pol)
            ms = timed(pol, args.steps)
            par.update({"ms_per_step": round(ms, 2), "images_per_sec": round(B / ms * 1e3, 2), "steps_timed": args.steps, "warmup": args.warmup})
            if not args.no_profile_pass:
                eng.profile_enable(True)
                run(pol)
                torch.cuda.synchronize()
                st2 = eng.profile_get()
                par["roofline_per_kernel"] = per_kernel_rooflines(eng)
                if st2.get("gn_launches") and st2["gn_ms"] > 0:
                    par["groupnorm_ms_per_step"] = round(st2["gn_ms"], 2)
                eng.profile_enable(False)
            parity.append(par)
            log(f"parity policy: {par}")
            # the cheaper mixture (first MIXED_FP16_STEPS steps fp16): its own entry, timed the same way
            if steps > MIXED_FP16_STEPS + 1:
                polm = policy_args("parity_mixed", steps)
                parm = engine_parity(f"parity_mixed (first {MIXED_FP16_STEPS} steps fp16, then split; split encoder, fp16 decoder)", polm)
                msm = timed(polm, args.steps)
                parm.update({"ms_per_step": round(msm, 2), "images_per_sec": round(B / msm * 1e3, 2), "steps_timed": args.steps})
                parity.append(parm)
                log(f"mixed policy: {parm}")
        qualified = [p for p in parity if p["image_psnr_db"] >= 60.0 and p["vq_code_agreement"] >= 0.999 and not p["policy"].startswith("parity_mixed")]
        if qualified:
            best = max(qualified, key=lambda p: p.get("images_per_sec", value))
            value_at_parity = {"value": best.get("images_per_sec", round(value, 3)), "unit": "images/sec", "policy": best["policy"],
                               "criterion": "image PSNR >= 60 dB and VQ code agreement >= 0.999 vs the CPU oracle",
                               "image_psnr_db": best["image_psnr_db"], "vq_code_agreement": best["vq_code_agreement"], "images": nb}
        # the exact-kernel policy beside it (fp32 storage, v_mfma_f32_16x16x4_f32)
        if args.precision != "fp32" and not args.no_exact_leg:
            p32 = policy_args("fp32", steps)
            par32 = engine_parity("fp32 (exact fp32 MFMA everywhere)", p32)
            ms32 = timed(p32, 1)
            par32.update({"ms_per_step": round(ms32, 2), "images_per_sec": round(B / ms32 * 1e3, 2), "steps_timed": 1})
            parity.append(par32)
        # the same restatement of the reference executed by stock PyTorch-ROCm ops (MIOpen / hipBLASLt) on this
        # GPU under torch.autocast, as sampler.py:185 runs the reference - its throughput and ITS distance from the fp32 CPU path
        if not args.no_torch_baseline:
            try:
                usd_g = {k: v.to(dev) for k, v in usd.items()}
                asd_g = {k: v.to(dev) for k, v in asd.items()}
                nt = min(B, 8)
                yg, ng = y[:nt], [noise[k, :nt] for k in range(steps + 1)]
                mg = mask[:nt] if mask is not None else None

                def torch_run():
                    with torch.autocast("cuda", dtype=torch.float16):
                        return oc.sample_loop(usd_g, up, asd_g, aep, dp, yg, ng, mask=mg, return_aux=True)

                t0 = time.perf_counter()
                torch_run()
                torch.cuda.synchronize()
                first_s = time.perf_counter() - t0
                t0 = time.perf_counter()
                o_t, aux_t = torch_run()
                torch.cuda.synchronize()
                t_s = time.perf_counter() - t0
                tb = parity_of("torch autocast(fp16)", o_t, aux_t["z_final"], aux_t["indices"]) if nt >= nb else {}
                torch_baseline = {"value": round(nt / t_s, 2), "unit": "images/sec", "batch": nt, "seconds": round(t_s, 3),
                                  "first_call_seconds": round(first_s, 2),
                                  "what": "oracle/ restatement on PyTorch-ROCm CUDA ops under torch.autocast(float16), eager",
                                  "parity_vs_cpu_fp32": tb}
                log(f"torch autocast baseline: {torch_baseline}")
            except Exception as ex:  # the baseline is informational: never fail the bench line over it
                torch_baseline = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    if rank == 0:
        dtypes = sorted(set(pu) | {pe, pd})
        line = {
            "metric": f"images/sec ({cdesc.split(',')[0]}, {steps}-step ResShift sampling loop incl. VQ encode/decode)",
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "ms_per_diffusion_step": round(ms_per_step / steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp16": "f16", "fp32": "f32", "split": "f16x2 (hi+lo pairs, 3 MFMAs per product)"}.get(dtypes[0], dtypes[0]) if len(dtypes) == 1
                     else "+".join(dtypes) + f" ({args.precision})",
            "data": "synthetic",
            "config": {"workload": f"{cname}: batch {B}/GPU x {world} GPU, {cdesc}, random-init weights",
                       "precision_policy": args.precision, "kernel_launches_per_step": launches, "weight_setup_s": round(setup_s, 2),
                       "parallelism": f"dp{world} (batch sharded, one RCCL weight broadcast, no data-path collective)"},
            "ranks": {"world_size": dist.get_world_size() if dist.is_initialized() else 1,
                      "backend": dist.get_backend() if dist.is_initialized() else None,
                      "gpus_visible": torch.cuda.device_count(),
                      "per_rank": [{"rank": r, "device": int(v[1]), "images_per_sec": round(B * args.steps / v[0], 3)} for r, v in enumerate(per_rank)],
                      "weight_broadcast_bytes": int(getattr(eng, "broadcast_bytes", 0)),
                      "weight_broadcast_ms": round(1e3 * float(getattr(eng, "broadcast_s", 0.0)), 3)},
            "value_at_parity": value_at_parity, "value_parity_policy": value_parity_policy,
            "value_at_parity_mixed": next(({"value": p["images_per_sec"], "unit": "images/sec", "policy": p["policy"], "image_psnr_db": p["image_psnr_db"],
                                            "vq_code_agreement": p["vq_code_agreement"], "images": p["images"],
                                            "meets_criterion": bool(p["image_psnr_db"] >= 60.0 and p["vq_code_agreement"] >= 0.999)}
                                           for p in (parity or []) if p["policy"].startswith("parity_mixed")), None),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity_vs_cpu_oracle": parity,
            "torch_rocm_autocast_baseline": torch_baseline,
        }
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
