"""CPU study (round 5): would Winograd F(2x2, 3x3) for the 3x3 / stride-1 convs of the encoder and the UNet - 2.25 x fewer
multiply-adds for ~ 80 % of the pass's flops - keep the parity criterion?  Decided here, with zero GPU minutes, BEFORE anybody writes the kernel
(the way oracle/study_fp8_cross.py decided the MX-fp8 cross terms).

The engine's parity policy stores activations and weights as (hi, lo) fp16 pairs (22 mantissa bits) and forms every product from three
fp16 MFMAs, W.X = Wh.Xh + 2^-11 (Wh.Xl + Wl.Xh), accumulated in fp32.  A Winograd halo kernel in that arithmetic would
    * take the normalised activation tile (the pair's joined value), form V = B^T d B with fp32 adds and RE-SPLIT it (round to the pair),
    * multiply with U = G g G^T - formed in double when the weights are packed, stored as a pair - as sixteen GEMMs over the input
      channels, three MFMAs per product, fp32 accumulation,
    * form Y = A^T M A in fp32 and add the bias.
F(2x2, 3x3)'s transforms are small-integer / half-integer combinations; what they cost is cancellation: the output transform subtracts
products that are larger than the result.  This script patches the ORACLE's conv helper with an emulation of exactly that arithmetic
(every other op stays what the mode says) and runs the whole sampling loop:
    exact       : fp32 (the reference CPU path: the thing compared against)
    split3      : the engine's parity-policy product on the DIRECT form              (calibrates the emulation: engine 137 dB latent)
    wino_f32    : Winograd with fp32 operands and fp32 accumulation                  (the transform's own error, no operand rounding)
    wino_split  : Winograd with pair operands and three-term products                (the proposal)
Attention matmuls, GroupNorm, softmax and the decoder stay fp32 in every mode (which FAVOURS the proposal, as in the fp8 study); strided and
1x1 convs and the upsampling convs keep the direct split3 form in the wino modes (a Winograd kernel would not serve them).

    python -m oracle.study_winograd [--config realsr] [--images 2] [--modes split3 wino_f32 wino_split] [--inputs synthetic|real]

Writes profiles/r5_winograd_study.json.  Test infrastructure only: nothing on the product path imports it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import resshift_oracle as oc  # noqa: E402
from oracle.study_fp8_cross import CONFIGS, LO, psnr, split16  # noqa: E402
from resshift_amd.config import load_config, to_plain  # noqa: E402
from resshift_amd.spec import ae_param_spec, random_state_dict, unet_param_spec  # noqa: E402

MODE = "exact"
# F(2x2, 3x3) (Lavin & Gray): Y = A^T [ (G g G^T) . (B^T d B) ] A
BT = torch.tensor([[1., 0., -1., 0.], [0., 1., 1., 0.], [0., -1., 1., 0.], [0., 1., 0., -1.]])
G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]])
AT = torch.tensor([[1., 1., 1., 0.], [0., 1., -1., -1.]])
_U = {}   # weight name -> transformed weights (made once, in double, like a packer would)


def pair(x):
    """round to the (hi, lo) fp16 pair's value"""
    h, l = split16(x)
    return h + l / LO


def direct(sd, name, x, stride, padding):
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    wh, wl = split16(w)
    xh, xl = split16(x)
    main = F.conv2d(xh, wh, None, stride=stride, padding=padding)
    cross = F.conv2d(xl, wh, None, stride=stride, padding=padding) + F.conv2d(xh, wl, None, stride=stride, padding=padding)
    return main + cross / LO + b.view(1, -1, 1, 1)


def wino(sd, name, x, split):
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    key = (name, split)
    if key not in _U:
        u = torch.einsum("ij,ocjk,lk->ocil", G.double(), w.double(), G.double()).float()   # [O, C, 4, 4]
        _U[key] = u
    u = _U[key]
    Bn, C, H, W = x.shape
    out = torch.empty(Bn, w.shape[0], H, W)
    for n in range(Bn):   # one image at a time: the transformed tensor is 4 x the activation
        xp = F.pad(x[n:n + 1], (1, 1, 1, 1))
        d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                   # [1, C, H/2, W/2, 4, 4]
        v = torch.einsum("ij,nchwjk,lk->nchwil", BT, d, BT)                      # fp32 adds (exact small-integer weights)
        if split:
            vh, vl = split16(v)
            uh, ul = split16(u)
            m = torch.einsum("ocil,nchwil->nohwil", uh, vh) + (torch.einsum("ocil,nchwil->nohwil", uh, vl) + torch.einsum("ocil,nchwil->nohwil", ul, vh)) / LO
        else:
            m = torch.einsum("ocil,nchwil->nohwil", u, v)
        yt = torch.einsum("ij,nohwjk,lk->nohwil", AT, m, AT)                    # [1, O, H/2, W/2, 2, 2]
        out[n] = yt.permute(0, 1, 2, 4, 3, 5).reshape(1, w.shape[0], H, W)[0]
    return out + b.view(1, -1, 1, 1)


def emu_conv(sd, name, x, stride=1, padding=0):
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    if MODE == "exact":
        return F.conv2d(x, w, b, stride=stride, padding=padding)
    is_wino = (MODE.startswith("wino") and tuple(w.shape[2:]) == (3, 3) and stride == 1 and padding == 1 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
               and w.shape[1] >= 32 and w.shape[0] >= 32)   # (the tiny-channel input / output convs are direct kernels on the engine)
    if is_wino:
        return wino(sd, name, x, MODE == "wino_split")
    return direct(sd, name, x, stride, padding)


def emu_linear(sd, name, x):
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    if MODE == "exact" or x.dim() < 2 or x.shape[-1] < 32 or ".emb_layers." in name or name.startswith("time_embed"):
        return F.linear(x, w, b)
    wh, wl = split16(w)
    xh, xl = split16(x)
    return F.linear(xh, wh) + (F.linear(xl, wh) + F.linear(xh, wl)) / LO + b


def main():
    global MODE
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", nargs="+", default=["realsr"])
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--modes", nargs="+", default=["split3", "wino_f32", "wino_split"])
    ap.add_argument("--inputs", default="synthetic", choices=["synthetic", "real"])
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r5_winograd_study.json"))
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    # self-check of the transform: fp32 Winograd == direct conv to fp32 rounding
    g = torch.Generator().manual_seed(0)
    xs, ws, bs = torch.randn(1, 32, 8, 8, generator=g), torch.randn(32, 32, 3, 3, generator=g) / 17.0, torch.randn(32, generator=g)
    ref = F.conv2d(xs.double(), ws.double(), bs.double(), padding=1)
    got = wino({"t.weight": ws, "t.bias": bs}, "t", xs, False)
    err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-5, err
    _U.clear()
    oc._conv, oc._linear = emu_conv, emu_linear
    orig_decode = oc.vq_decode

    def decode_exact(*a, **k):
        global MODE
        m, MODE = MODE, "exact"
        try:
            return orig_decode(*a, **k)
        finally:
            MODE = m

    oc.vq_decode = decode_exact
    results = {"what": __doc__.split("\n\n")[0], "criterion": "image PSNR >= 60 dB AND VQ code agreement >= 0.999 on every image",
               "transform_self_check_rel_err": err, "runs": []}
    if os.path.exists(args.out):
        with open(args.out) as fh:
            results = json.load(fh)
    for cname in args.config:
        yaml, lr = CONFIGS[cname]
        cfg = to_plain(load_config(yaml))
        up, aep, dp = cfg["model"]["params"], cfg["autoencoder"]["params"], cfg["diffusion"]["params"]
        steps = int(dp["steps"])
        uspec, _ = unet_param_spec(up)
        usd, asd = random_state_dict(uspec, seed=1), random_state_dict(ae_param_spec(aep), seed=2)   # bench.py's weights
        gg = torch.Generator().manual_seed(1000)                                                         # bench.py's rank-0 inputs
        f = 2 ** (len(aep["ddconfig"]["ch_mult"]) - 1)
        sf = int(dp.get("sf", 4))
        hz, cz, B = lr * sf // f, int(aep["embed_dim"]), args.images
        y = torch.rand(32 if cname != "faceir" else 16, 3, lr, lr, generator=gg)[:B] * 2 - 1
        if args.inputs == "real":
            import numpy as np

            assert lr == 64, "the bundled validation images are 64 x 64"
            lq = np.load(os.path.join(ROOT, "tests", "golden", "val_sr_lq.npz"))["lq"][:B].astype(np.float32)
            y = (torch.from_numpy(lq).permute(0, 3, 1, 2).contiguous() / 255.0 - 0.5) / 0.5
        noise = torch.randn(steps + 1, B, cz, hz, hz, generator=gg)
        mask = (torch.rand(B, 1, lr, lr, generator=gg) > 0.7).float() * 2 - 1 if up.get("cond_mask", False) else None
        nz = [noise[k] for k in range(steps + 1)]
        MODE = "exact"
        t0 = time.time()
        ref, raux = oc.sample_loop(usd, up, asd, aep, dp, y, nz, mask=mask, return_aux=True)
        print(f"[{cname}] exact: {time.time() - t0:.0f} s", flush=True)
        zr, ir = raux["z_final"], raux["indices"].reshape(B, -1)
        for mode in args.modes:
            MODE = mode
            _U.clear()
            t0 = time.time()
            img, aux = oc.sample_loop(usd, up, asd, aep, dp, y, nz, mask=mask, return_aux=True)
            same = aux["indices"].reshape(B, -1) == ir
            per = [psnr(img[i].clamp(-1, 1), ref[i].clamp(-1, 1), 2.0) for i in range(B)]
            row = {"config": cname, "mode": mode, "images": B, "inputs": args.inputs, "latent_psnr_db": round(psnr(aux["z_final"], zr, (zr.max() - zr.min()).item()), 1),
                   "vq_code_agreement": round(same.float().mean().item(), 5), "vq_code_agreement_worst_image": round(same.float().mean(1).min().item(), 5),
                   "image_psnr_db": round(psnr(img.clamp(-1, 1), ref.clamp(-1, 1), 2.0), 1), "image_psnr_db_worst_image": round(min(per), 1),
                   "seconds": round(time.time() - t0)}
            row["meets_criterion"] = bool(row["image_psnr_db_worst_image"] >= 60.0 and row["vq_code_agreement_worst_image"] >= 0.999)
            print(row, flush=True)
            results["runs"] = [r for r in results["runs"] if not (r["config"] == cname and r["mode"] == mode and r["images"] == B and r.get("inputs", "synthetic") == args.inputs)] + [row]
            with open(args.out, "w") as fh:
                json.dump(results, fh, indent=1)
    MODE = "exact"


if __name__ == "__main__":
    main()
