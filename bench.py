#!/usr/bin/env python
"""Benchmark of the ResShift sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--precision POLICY]

Workload (BASELINE.json configs[1]): realsr_swinunet_realesrgan256, 64x64 -> 256x256, 15 diffusion steps, batch 32
per GPU, random-init weights, synthetic inputs already resident in HBM.  One "step" of this benchmark = one pass of
the whole hot path over one batch: bicubic x4 -> VQ-f4 encode -> prior sample -> 15 x (Swin-UNet + posterior update)
-> VQ lookup -> VQ-f4 decode, i.e. one `rs_sample` call of the engine.

For N > 1 the driver launches one process per GPU (torchrun); rank 0 packs the weights and the blob reaches the other
ranks through ONE RCCL broadcast; every rank then processes its own batch (weak scaling, no data-path collective).

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (MFMA implicit-GEMM kernel family, hipEvent
timed on the launch stream in a dedicated pass) and `cpu_baseline` (the CPU oracle timed on this host, N=1 only).
"""
from __future__ import annotations

import argparse
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

CONFIG = "realsr_swinunet_realesrgan256"
GFLOP_PER_IMAGE = 2535.7          # UNet 101.32 x 15 + encoder 345.24 + decoder 670.58 (2*MAC)
MFMA_PEAK_TFLOPS = {"fp16": 2500.0, "fp32": 157.3}  # dense, /opt/skills/guides/MI355X_MICROARCH.md

# precision policies: which kernels run fp16-storage MFMA and which run the exact fp32 MFMA
POLICIES = {
    "fp16": dict(unet="fp16", encode="fp16", decode="fp16"),
    "fp32": dict(unet="fp32", encode="fp32", decode="fp32"),
}


def policy_args(name: str, steps: int):
    if name in POLICIES:
        p = POLICIES[name]
        return [p["unet"]] * steps, p["encode"], p["decode"]
    if name.startswith("mixed"):
        # "mixed<k>": the last k timesteps (t = k-1 .. 0) and the encoder in fp32, everything else fp16
        k = int(name[5:] or 1)
        return ["fp32" if t < k else "fp16" for t in range(steps)], "fp16", "fp16"
    raise SystemExit(f"unknown precision policy {name}")


_T0 = time.time()


def log(msg: str) -> None:
    """progress on stderr (stdout carries only the JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--precision", default=os.environ.get("RESSHIFT_PRECISION", "fp16"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-exact-leg", action="store_true", help="skip the fp32-policy parity/timing leg")
    args = ap.parse_args()

    world, rank = sharding.init_distributed()
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.set_grad_enabled(False)

    cfg = to_plain(load_config(CONFIG))
    up, aep, dp = cfg["model"]["params"], cfg["autoencoder"]["params"], cfg["diffusion"]["params"]
    steps = int(dp["steps"])
    B = args.batch

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

    # ---- synthetic inputs resident in HBM (different per rank)
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    y = (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).to(dev)
    noise = torch.randn(steps + 1, B, 3, 64, 64, generator=g).to(dev)

    def one_pass():
        return eng.sample(y, noise, tables, sf=diffusion.sf, scale_factor=diffusion.scale_factor, prec_unet=pu, prec_encode=pe,
                          prec_decode=pd)

    for i in range(args.warmup):
        out = one_pass()
        torch.cuda.synchronize()
        log(f"warmup pass {i} done (arena {eng.arena_bytes() / 2**30:.2f} GiB)")
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.allreduce_max(elapsed, dev)
    assert torch.isfinite(out).all().item(), "non-finite output"
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    log(f"timed region done: {ms_per_step:.1f} ms/step, {value:.2f} img/s")

    # ---- roofline of the dominant kernel family (MFMA implicit GEMM), measured in a dedicated pass with hipEvents
    roofline = None
    launches = eng.last_launch_count()
    if rank == 0 and not args.no_profile_pass:
        eng.profile_enable(True)
        one_pass()
        torch.cuda.synchronize()
        st = eng.profile_get()
        eng.profile_enable(False)
        f16, f32 = st["flops_f16"], st["flops_f32"]
        dom = "fp16" if f16 >= f32 else "fp32"
        # time-weighted peak when a mixed policy runs both MFMA flavours: peak_eff = total flops / (f16/P16 + f32/P32)
        tot = f16 + f32
        peak_eff = tot / (f16 / MFMA_PEAK_TFLOPS["fp16"] + f32 / MFMA_PEAK_TFLOPS["fp32"]) if tot else MFMA_PEAK_TFLOPS["fp16"]
        achieved = tot / (st["igemm_ms"] * 1e-3) / 1e12 if st["igemm_ms"] > 0 else 0.0
        # HBM traffic of the same kernel family from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc
        # passes of this very command, corrected as MI355X_MICROARCH.md prescribes) - collected offline by
        # scripts/collect_traffic.py into profiles/, because a process cannot attach rocprofv3 to itself
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_pmc_traffic_igemm.json")
        if args.precision == "fp16" and B == 32 and os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = round(json.load(fh)["hbm_bytes_per_launch"] / 1e6, 2)  # MB per launch
        roofline = {
            "bound": "mfma", "kernel": "igemm2_kernel<*> / igemm3_kernel<*> / igemm_kernel<*> + swin_mlp_kernel / win_attn_qkv_kernel (MFMA implicit-GEMM family incl. the fused Swin kernels)", "achieved": round(achieved, 2),
            "peak": round(peak_eff, 1), "unit": "TFLOP/s", "frac": round(achieved / peak_eff, 4) if peak_eff else None,
            "traffic": traffic, "traffic_unit": "MB of HBM traffic per launch (PMC)",
            "algorithmic_mb_per_launch": round(st["igemm_bytes"] / max(1, st["igemm_launches"]) / 1e6, 2),
            "algorithmic_gflop_per_launch": round(tot / max(1, st["igemm_launches"]) / 1e9, 2),
            "launches_per_step": st["igemm_launches"], "avg_launch_us": round(st["igemm_ms"] * 1e3 / max(1, st["igemm_launches"]), 2),
            "algorithmic_gflop_per_image_igemm": round(tot / B / 1e9, 1), "algorithmic_gflop_per_image_total": GFLOP_PER_IMAGE,
            "igemm_ms_per_step": round(st["igemm_ms"], 2), "whole_path_tflops": round(GFLOP_PER_IMAGE * 1e9 * B / (ms_per_step * 1e-3) / 1e12, 2),
            "dominant_precision": dom,
        }

    # ---- CPU baseline: the oracle (CPU restatement of the reference, fp32) on a bounded sample of the same workload
    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import resshift_oracle as oc  # checker / baseline only; never on the measured GPU path

        # use the cores this process may actually run on (cgroup/affinity aware), never more than torch's own default
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:
            usable = os.cpu_count() or 1
        torch.set_num_threads(max(1, min(usable, torch.get_num_threads(), 64)))
        log(f"cpu baseline: oracle on {torch.get_num_threads()} threads")
        usd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        asd = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
        nb = 2
        yc = y[:nb].cpu()
        nz = [noise[k, :nb].cpu() for k in range(steps + 1)]
        t0 = time.perf_counter()
        ref, ref_aux = oc.sample_loop(usd, up, asd, aep, dp, yc, nz, return_aux=True)
        cpu_s = time.perf_counter() - t0

        def psnr_db(a, b, p2p):
            mse = torch.mean((a.double() - b.double()) ** 2).item()
            return float("inf") if mse == 0 else float(10 * np.log10(p2p * p2p / mse))

        def parity_of(policy_name, pu_, pe_, pd_):
            """image / latent PSNR and VQ code agreement of one precision policy against the CPU oracle (first nb images)"""
            o, aux = eng.sample(y, noise, tables, sf=diffusion.sf, scale_factor=diffusion.scale_factor, prec_unet=pu_, prec_encode=pe_,
                                prec_decode=pd_, return_aux=True)
            torch.cuda.synchronize()
            zr = ref_aux["z_final"]
            hw = zr.shape[2] * zr.shape[3]
            return {"policy": policy_name,
                    "image_psnr_db": round(psnr_db(o[:nb].cpu().clamp(-1, 1), ref.clamp(-1, 1), 2.0), 1),
                    "latent_psnr_db": round(psnr_db(aux["z_final"][:nb].cpu(), zr, (zr.max() - zr.min()).item()), 1),
                    "vq_code_agreement": round((aux["indices"][: nb * hw].cpu().long() == ref_aux["indices"].reshape(-1)).float().mean().item(), 4)}

        parity = [parity_of(args.precision, pu, pe, pd)]
        cpu_baseline = {"value": round(nb / cpu_s, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
                        "sample": f"{nb} images, same weights/inputs/noise as the first {nb} images of the GPU batch, full 15-step loop, fp32",
                        "seconds": round(cpu_s, 2), "gpu_vs_cpu_psnr_db": parity[0]["image_psnr_db"]}
        # the exact-kernel policy beside it (fp32 storage, v_mfma_f32_16x16x4_f32): the configuration the >= 60 dB parity tests
        # run; timed on the same inputs so that the price of bit-level VQ agreement is on the same line as the fp16 number
        if args.precision != "fp32" and not args.no_exact_leg:
            p32 = policy_args("fp32", steps)
            par32 = parity_of("fp32", *p32)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                eng.sample(y, noise, tables, sf=diffusion.sf, scale_factor=diffusion.scale_factor, prec_unet=p32[0], prec_encode=p32[1],
                           prec_decode=p32[2])
            torch.cuda.synchronize()
            ms32 = (time.perf_counter() - t0) / 2 * 1e3
            par32.update({"ms_per_step": round(ms32, 2), "images_per_sec": round(B / ms32 * 1e3, 2), "steps_timed": 2})
            parity.append(par32)
            # ... and a mixture in between (the last 8 UNet steps exact): how fast the VQ code agreement recovers with precision
            pm = policy_args("mixed8", steps)
            parm = parity_of("mixed8 (last 8 UNet steps fp32, rest fp16)", *pm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                eng.sample(y, noise, tables, sf=diffusion.sf, scale_factor=diffusion.scale_factor, prec_unet=pm[0], prec_encode=pm[1],
                           prec_decode=pm[2])
            torch.cuda.synchronize()
            msm = (time.perf_counter() - t0) / 2 * 1e3
            parm.update({"ms_per_step": round(msm, 2), "images_per_sec": round(B / msm * 1e3, 2), "steps_timed": 2})
            parity.append(parm)

    if rank == 0:
        line = {
            "metric": "images/sec (64->256 SR, 15-step ResShift sampling loop incl. VQ-f4 encode/decode)",
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "ms_per_diffusion_step": round(ms_per_step / steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp16": "f16", "fp32": "f32"}.get(args.precision, f"f16+f32 ({args.precision})"), "data": "synthetic",
            "config": {"workload": f"{CONFIG}: batch {B}/GPU x {world} GPU, 1x3x64x64 LR -> 3x256x256, 15 steps, random-init weights",
                       "precision_policy": args.precision, "kernel_launches_per_step": launches, "weight_setup_s": round(setup_s, 2),
                       "parallelism": f"dp{world} (batch sharded, one RCCL weight broadcast, no data-path collective)"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity_vs_cpu_oracle": parity,
        }
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
