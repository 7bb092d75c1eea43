"""Second half of the parity margin study (scripts/parity_margin_study.py): is a sample that misses 60 dB against the CPU reference a sample the
reference reproduces ITSELF?  For the worst samples of gpurun_out/parity_margin.json (and as many of the best, as controls) the fp32 CPU oracle runs twice
- with all host threads and with a different thread count (another reduction order inside the fp32 conv / matmul kernels, nothing else) - and the
two results are compared like engine and oracle are: image PSNR, flipped VQ codes, largest latent difference.  CPU only.

    python scripts/parity_margin_selfcheck.py gpurun_out/parity_margin.json [n=10] > profiles/r6_parity_margin_selfcheck.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402
from oracle import resshift_oracle as oc  # noqa: E402

torch.set_grad_enabled(False)
study = json.load(open(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = sorted(study["rows"], key=lambda r: r["psnr_db"])
pick = rows[:n] + rows[-n:]
up, ap, dp = H.realsr_params()
usd, asd = H.weights(up, ap)
d_ = np.load(os.path.join(ROOT, "tests", "golden", "val_sr_lq.npz"))
B, T = 32, dp["steps"]
y = (torch.from_numpy(d_["lq"][:B].astype(np.float32)).permute(0, 3, 1, 2).contiguous() / 255.0 - 0.5) / 0.5
nmax = torch.get_num_threads()
other = max(1, nmax // 2 - 1)
out = []
for r in pick:
    _, noises, _ = H.synth.synthetic_inputs(500 + r["seed"], B, 64, 64, 3, 64, 64, T)
    k = r["image"]
    res = []
    for nt in (nmax, other):
        torch.set_num_threads(nt)
        ref, aux = oc.sample_loop(usd, up, asd, ap, dp, y[k:k + 1], [z[k:k + 1] for z in noises], return_aux=True)
        res.append((ref.clamp(-1, 1).double(), aux["z_final"].double(), aux["indices"].view(-1)))
    (a, za, ia), (b, zb, ib) = res
    mse = ((a - b) ** 2).mean().item()
    o = {"seed": r["seed"], "image": k, "engine_vs_reference_psnr_db": r["psnr_db"], "engine_flipped_codes": r["flipped_codes"],
         "engine_latent_psnr_db": r["latent_psnr_db"],
         "reference_vs_itself_psnr_db": (None if mse == 0 else round(10 * np.log10(4.0 / mse), 2)), "reference_vs_itself_flipped_codes": int((ia != ib).sum()),
         "reference_vs_itself_latent_max_abs_diff": float((za - zb).abs().max())}
    out.append(o)
    print(f"[selfcheck] seed {o['seed']} image {k}: engine {o['engine_vs_reference_psnr_db']} dB / {o['engine_flipped_codes']} flips | reference({nmax} threads) vs "
          f"reference({other} threads): {o['reference_vs_itself_psnr_db']} dB / {o['reference_vs_itself_flipped_codes']} flips", file=sys.stderr, flush=True)
torch.set_num_threads(nmax)
bad = [o for o in out if o["engine_vs_reference_psnr_db"] < 60]
print(json.dumps({"what": "does the fp32 CPU reference reproduce itself (two host thread counts = two reduction orders) on the samples where the engine misses 60 dB?",
                  "threads": [nmax, other], "samples": out,
                  "summary": {"engine_samples_below_60_db": len(bad),
                              "of_those_the_reference_misses_60_db_against_itself": sum(1 for o in bad if (o["reference_vs_itself_psnr_db"] or 999) < 60),
                              "controls_where_the_reference_matches_itself_above_60_db": sum(1 for o in out if o["engine_vs_reference_psnr_db"] >= 60 and (o["reference_vs_itself_psnr_db"] or 999) >= 60)}}))
