"""Time the UNMODIFIED reference modules on this host's CPU cores and, beside them, the oracle
restatement on the same inputs, so that bench.py's `cpu_baseline` (kind "port": the oracle, timed on the GPU box's host) can
quote the reference's own number and the port / reference ratio next to it.

    python -m oracle.time_reference [--batches 1 4] [--runs 2]          (build container only: needs /root/reference)

Writes profiles/ref_cpu_timing.json.  Test / measurement infrastructure: nothing on the product path imports it, and nothing
on the GPU box can run it (the reference tree is not there) - bench.py only READS the committed JSON.
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import, resshift_oracle as oc, synth  # noqa: E402
from oracle.make_golden import ref_sample  # noqa: E402
from resshift_amd.config import load_config, to_plain  # noqa: E402
from resshift_amd.spec import ae_param_spec, unet_param_spec  # noqa: E402


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="realsr_swinunet_realesrgan256")
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--runs", type=int, default=2)
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    U, V, create = ref_import.load()
    cfg = to_plain(load_config(args.config))
    up, aep, dp = cfg["model"]["params"], cfg["autoencoder"]["params"], cfg["diffusion"]["params"]
    uspec, _ = unet_param_spec(up)
    usd = synth.synthetic_state_dict(uspec, 1, image_size=up["image_size"])
    asd = synth.synthetic_state_dict(ae_param_spec(aep), 1)
    um = U(**up).eval()
    um.load_state_dict(usd, strict=True)
    am = V(**aep).eval()
    am.load_state_dict(asd, strict=True)
    d = create(**dp)
    steps = int(dp["steps"])
    rows = []
    for B in args.batches:
        y, noises, _ = synth.synthetic_inputs(123, B, 64, 64, 3, 64, 64, steps)
        t_ref, t_or = [], []
        for r in range(args.runs + 1):          # run 0 = warm-up
            t0 = time.perf_counter()
            img, zf, idx = ref_sample(d, um, am, y, noises)
            t1 = time.perf_counter()
            o_img, aux = oc.sample_loop(usd, up, asd, aep, dp, y, noises, return_aux=True)
            t2 = time.perf_counter()
            if r:
                t_ref.append(t1 - t0)
                t_or.append(t2 - t1)
        agree = (aux["indices"] == idx).float().mean().item()
        # one UNet forward alone ("ms/step" of the metric)
        x = noises[1] * 1.3
        t = torch.full((B,), 7, dtype=torch.long)
        um(x, t, lq=y)
        t0 = time.perf_counter()
        um(x, t, lq=y)
        unet_ms = (time.perf_counter() - t0) * 1e3
        row = {"batch": B, "reference_seconds": [round(v, 3) for v in t_ref], "oracle_seconds": [round(v, 3) for v in t_or],
               "reference_images_per_sec": round(B / min(t_ref), 4), "oracle_images_per_sec": round(B / min(t_or), 4),
               "oracle_over_reference": round(min(t_ref) / min(t_or), 3), "reference_unet_forward_ms": round(unet_ms, 1),
               "vq_code_agreement_oracle_vs_reference": agree}
        print(row, flush=True)
        rows.append(row)
    out = {"what": "unmodified reference modules (models.unet.UNetModelSwin, ldm.models.autoencoder.VQModelTorch, "
                   "models.script_util.create_gaussian_diffusion; p_sample_loop_progressive + decode_first_stage) vs oracle/resshift_oracle.py, "
                   "CPU fp32, synthetic weights seed 1, inputs seed 123, best of the timed runs after one warm-up",
           "config": args.config, "steps": steps, "cores": cores, "torch_threads": torch.get_num_threads(), "cpu": cpu_model(),
           "torch": torch.__version__, "host": "build container (no GPU)", "rows": rows}
    path = os.path.join(ROOT, "profiles", "ref_cpu_timing.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
