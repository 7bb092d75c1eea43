"""GPU: image PSNR / VQ agreement of mixed-precision policies on the full-size realsr config vs the reference output
(tests/golden), plus batch-32 timing of each policy.  Not a test; results go to DESIGN.md."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from resshift_amd import UNetModelSwin, VQModelTorch, create_gaussian_diffusion
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
up, ap, dp = H.realsr_params()
g = H.golden()
usd, asd = H.weights(up, ap)
um = UNetModelSwin(**up).to(dev); um.load_state_dict(usd)
am = VQModelTorch(**ap).to(dev); am.load_state_dict(asd)
B = int(os.environ.get("NB", "4"))
# B images: image 0 is the golden one; others extra seeds compared against nothing (timing only)
y, noises, _ = H.synth.synthetic_inputs(H.SEED_X, 1, 64, 64, 3, 64, 64, dp["steps"])
ref_img = torch.from_numpy(g["realsr/sample"].astype(np.float32)).clamp(-1, 1)
ref_z = torch.from_numpy(g["realsr/sample_z"]); ref_idx = torch.from_numpy(g["realsr/sample_idx"].astype(np.int64))
d = create_gaussian_diffusion(**dp)
T = dp["steps"]
def pol(k_last=0, enc="fp16", dec="fp16", first=0):
    return ["fp32" if (t < k_last or t >= T - first) else "fp16" for t in range(T)], enc, dec
policies = {
    "fp16": pol(), "fp32": (["fp32"] * T, "fp32", "fp32"),
    "last1": pol(1), "last2": pol(2), "last3": pol(3), "last4": pol(4), "last6": pol(6), "last8": pol(8),
    "last1+enc32": pol(1, enc="fp32"), "last2+enc32": pol(2, enc="fp32"), "last4+enc32": pol(4, enc="fp32"),
    "unet32": (["fp32"] * T, "fp16", "fp16"), "unet32+enc32": (["fp32"] * T, "fp32", "fp16"),
    "dec32": pol(0, dec="fp32"),
    # split storage: (hi, lo) fp16 pairs, 3 fp16 MFMAs per product (fp32-class)
    "split": (["split"] * T, "split", "split"),
    "split+dec16": (["split"] * T, "split", "fp16"),
    "unet-split": (["split"] * T, "fp16", "fp16"),
    "split-last8+enc": (["split" if t < 8 else "fp16" for t in range(T)], "split", "fp16"),
}
# split-last<k>+enc: split-precision encoder, the first T-k sampling steps (t = T-1 .. k) in fp16, the last k (t = k-1 .. 0) in split
# precision, fp16 decoder.  An error made at step t reaches the final latent damped by the posterior coefficients
# (x_{t-1} = eta_{t-1}/eta_t x_t + alpha_t/eta_t x0_pred, models/gaussian_diffusion.py:218-221; eta_0 = 0), so the early steps
# tolerate fp16.
for k in (1, 2, 3, 4, 5, 6, 10, 12):
    policies[f"split-last{k}+enc"] = (["split" if t < k else "fp16" for t in range(T)], "split", "fp16")
    policies[f"split-last{k}+enc16"] = (["split" if t < k else "fp16" for t in range(T)], "fp16", "fp16")
if len(sys.argv) > 1:
    policies = {k: policies[k] for k in sys.argv[1:]}
yb = y.repeat(32, 1, 1, 1).to(dev); nb = torch.stack(noises, 0).repeat(1, 32, 1, 1, 1).to(dev)
for name, (pu, pe, pd) in policies.items():
    d.set_precision(pu, pe, pd)
    out, aux = d.p_sample_loop(y.to(dev), um, first_stage_model=am, noise=noises[0].to(dev), clip_denoised=False,
                               model_kwargs={"lq": y.to(dev)}, step_noises=[n.to(dev) for n in noises[1:]], return_aux=True)
    torch.cuda.synchronize()
    agree = (aux["indices"].cpu().long() == ref_idx).float().mean().item()
    p_img = H.psnr(out.cpu().clamp(-1, 1), ref_img)
    p_lat = H.psnr(aux["z_final"].cpu(), ref_z, peak_to_peak=(ref_z.max() - ref_z.min()).item())
    eng = d._fused_engine(um, am)
    tabs = d.step_tables()
    eng.sample(yb, nb, tabs, sf=4, scale_factor=1.0, prec_unet=pu, prec_encode=pe, prec_decode=pd); torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.sample(yb, nb, tabs, sf=4, scale_factor=1.0, prec_unet=pu, prec_encode=pe, prec_decode=pd); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"{name:14s} image PSNR {p_img:6.1f} dB  latent PSNR {p_lat:6.1f} dB  VQ agree {agree:.4f}  flips {int(round((1-agree)*4096)):4d}   B=32: {ms:8.1f} ms  {32e3/ms:6.1f} img/s", flush=True)
