"""GPU micro-benchmark: Winograd F(2x2,3x3) kernel (wino.hip) against the split-storage halo kernel (igemm4) on the 3x3 conv shapes of one
parity pass of realsr_swinunet_realesrgan256 at batch 32 (GroupNorm affine + SiLU folded in, residual where the layer has one).

    python scripts/wino_bench.py [reps]

Prints us per launch for both kernels, algorithmic TFLOP/s (2 M N 9 Cin) and the weighted ms per pass (launch counts from profiles/r5_shapes_parity.txt).
"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resshift_amd import _lib, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
# name, B, H, Cin, Cout, residual, launches per pass
SHAPES = [
    ("unet 160->160 @64", 32, 64, 160, 160, 1, 90),
    ("unet 320->160 @64", 32, 64, 320, 160, 0, 30),
    ("unet 480->160 @64", 32, 64, 480, 160, 0, 15),
    ("unet 320->320 @32", 32, 32, 320, 320, 1, 75),
    ("unet 640->320 @32", 32, 32, 640, 320, 0, 30),
    ("unet 480->320 @32", 32, 32, 480, 320, 0, 15),
    ("unet 160->320 @32", 32, 32, 160, 320, 0, 15),
    ("ae 512->512 @64", 32, 64, 512, 512, 1, 7),
    ("ae 256->256 @128", 32, 128, 256, 256, 1, 3),
    ("ae 128->128 @256", 8, 256, 128, 128, 1, 4 * 4),
]
only = [t for t in os.environ.get("RS_BENCH_ONLY", "").split(",") if t]
if only:
    SHAPES = [s for s in SHAPES if any(t in s[0] for t in only)]
lib = _lib.load()
g = torch.Generator().manual_seed(0)
tot_w = tot_h = 0.0
print(f"{'shape':22s} {'M':>8s} {'wino us':>9s} {'TF/s':>7s} {'halo us':>9s} {'TF/s':>7s} {'ratio':>6s} {'n':>4s}")
for name, B, H, Cin, Cout, res, cnt in SHAPES:
    x = ops.convert((torch.randn(B, H, H, Cin, generator=g) * 1.5 + 0.2).to(dev), ops.SPLIT)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    coef = torch.stack([torch.randn(B, Cin, generator=g) * 0.3 + 1.0, torch.randn(B, Cin, generator=g) * 0.5], 1).contiguous().to(dev)
    r = ops.convert(torch.randn(B, H, H, Cout, generator=g).to(dev), ops.SPLIT) if res else None
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    (y, st), ms_w = ops.conv3x3_wino(x, w, bias, coef=coef, act_in=2, res=r, want_stats=True, reps=reps)
    # the halo kernel through its own op entry, timed with torch events around `reps` calls of the packed-weight bench entry
    wp = torch.empty(Cout, 2, 9 * Cin, dtype=torch.float16)
    wk = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
    wp[:, 0] = wk.half()
    wp[:, 1] = ((wk - wk.half().float()) * 2048.0).half()
    wpd = wp.to(dev).contiguous()
    y2 = torch.empty_like(y)
    ms = ctypes.c_float(0.0)
    rc = lib.rs_op_conv2d_bench(x.data_ptr(), wpd.data_ptr(), bias.to(dev).data_ptr(), r.data_ptr() if r is not None else None, y2.data_ptr(),
                                B, H, H, Cin, Cout, 3, 3, 1, 1, H, H, 1, 0, 2, 2, reps, ctypes.byref(ms), _lib.current_stream_ptr())
    _lib.check(rc, "bench")
    ms_h = ms.value
    tot_w += ms_w * cnt
    tot_h += ms_h * cnt
    print(f"{name:22s} {B*H*H:8d} {ms_w*1e3:9.1f} {fl/ms_w/1e9:7.1f} {ms_h*1e3:9.1f} {fl/ms_h/1e9:7.1f} {ms_h/ms_w:6.2f} {cnt:4d}", flush=True)
print(f"weighted ms per pass: wino {tot_w:.1f}   halo (plain conv, no GroupNorm fold) {tot_h:.1f}")
