"""Debugging aid for wino.hip: small cases against a float64 conv, with the error broken down by output-pixel parity, tile and channel."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resshift_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def run(B, H, W, Cin, Cout, coef, res, seed=0, kind="rand"):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    if kind == "center":
        w = torch.zeros(Cout, Cin, 3, 3)
        for n in range(Cout):
            w[n, n % Cin, 1, 1] = 1.0
    bias = torch.randn(Cout, generator=g)
    xr = x.double()
    cd = None
    if coef:
        a = torch.randn(B, Cin, generator=g) * 0.3 + 1.0
        d = torch.randn(B, Cin, generator=g) * 0.5
        cd = torch.stack([a, d], 1).contiguous().to(dev)
        xr = F.silu(xr * a.double()[:, :, None, None] + d.double()[:, :, None, None])
    ref = F.conv2d(xr, w.double(), bias.double(), padding=1)
    rd = None
    if res:
        r = torch.randn(ref.shape, generator=g)
        rd = ops.convert(r.permute(0, 2, 3, 1).contiguous().to(dev), ops.SPLIT)
        ref = ref + r.double()
    xs = ops.convert(x.permute(0, 2, 3, 1).contiguous().to(dev), ops.SPLIT)
    y, st = ops.conv3x3_wino(xs, w, bias, coef=cd, act_in=2 if coef else 0, res=rd, want_stats=True)
    torch.cuda.synchronize()
    got = ops.convert(y, ops.F32).cpu().permute(0, 3, 1, 2).double()
    err = (got - ref).abs()
    sc = ref.abs().max().item()
    print(f"case B={B} {H}x{W} {Cin}->{Cout} coef={coef} res={res} kind={kind}: max err {err.max().item():.3e} / scale {sc:.3e} = {err.max().item()/sc:.2e}")
    if err.max().item() > 3e-6 * sc:
        for yy in range(2):
            for xx in range(2):
                print(f"   parity ({yy},{xx}): {err[:, :, yy::2, xx::2].max().item():.3e}")
        ec = err.amax(dim=(0, 2, 3))
        print("   per channel block of 16:", [f"{ec[i:i+16].max().item():.2e}" for i in range(0, Cout, 16)])
        et = err.amax(dim=(0, 1))
        print("   per 8x16 tile:", [[f"{et[i:i+8, j:j+16].max().item():.1e}" for j in range(0, W, 16)] for i in range(0, H, 8)])
        e2 = err[0].amax(0)[:16, :16]
        print("   first tile rows (max over channels):")
        for i in range(16):
            print("    ", " ".join(f"{e2[i, j].item():7.1e}" for j in range(16)))
    sref = torch.stack([ref.reshape(B, Cout, -1).sum(-1), (ref * ref).reshape(B, Cout, -1).sum(-1)], -1)
    se = ((st.cpu().double().sum(1) - sref).abs() / (sref.abs() + 1.0)).max().item()
    print(f"   stats rel err {se:.2e}")


run(1, 16, 16, 32, 64, False, False, kind="center")
run(1, 16, 16, 32, 64, False, False)
run(1, 64, 64, 32, 64, True, True)
run(1, 48, 48, 64, 128, False, False)
run(4, 32, 32, 160, 160, True, True)
run(1, 64, 64, 160, 160, False, False)
run(2, 32, 32, 320, 320, True, True)
run(2, 64, 64, 128, 256, True, True)
