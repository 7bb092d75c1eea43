"""GPU micro-benchmark of the implicit-GEMM kernel on the real layer shapes of realsr_swinunet_realesrgan256 at B=32.

    python scripts/igemm_bench.py [fp16|fp32|split] [reps]

Prints ms / TFLOP/s per shape and the weighted total for one full sampling pass (counts = launches per pass).
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resshift_amd import _lib  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
P = {"fp16": 0, "fp32": 1, "split": 2}[prec]
dt = torch.float16 if P == 0 else torch.float32


def to_store(t):
    """fp32 device tensor [..., C] -> storage of the benchmarked precision (split: [..., C hi | C lo] fp16 pairs, common.h)"""
    if P != 2:
        return t.to(dt)
    hi = t.half()
    lo = ((t - hi.float()) * 2048.0).half()
    return torch.cat([hi, lo], dim=-1).contiguous()


B = 32
# name, H(out), Cin, Cout, k, stride, up, act, res, launches per pass (15 UNet forwards + AE)
U = 15
SHAPES = [
    ("unet c3 160->160 @64", 64, 160, 160, 3, 1, 1, 0, 1, 7 * U),
    ("unet c3 320->160 @64", 64, 320, 160, 3, 1, 1, 0, 0, 2 * U),
    ("unet c3 480->160 @64", 64, 480, 160, 3, 1, 1, 0, 0, 1 * U),
    ("unet up 320->320 @64", 64, 320, 320, 3, 1, 2, 0, 0, 1 * U),
    ("unet c3 320->320 @32", 32, 320, 320, 3, 1, 1, 0, 1, 7 * U),
    ("unet c3 640->320 @32", 32, 640, 320, 3, 1, 1, 0, 0, 2 * U),
    ("unet c3 320->320 @16", 16, 320, 320, 3, 1, 1, 0, 1, 7 * U),
    ("unet c3 640->320 @16", 16, 640, 320, 3, 1, 1, 0, 0, 2 * U),
    ("unet c3 640->640 @8", 8, 640, 640, 3, 1, 1, 0, 1, 10 * U),
    ("unet c3 1280->640 @8", 8, 1280, 640, 3, 1, 1, 0, 0, 2 * U),
    ("swin fc1 192->768 @64", 64, 192, 768, 1, 1, 1, 1, 0, 4 * U),
    ("swin fc1-noact 192->768 @64", 64, 192, 768, 1, 1, 1, 0, 0, 0),   # ablation shape: fc1 without the GELU epilogue (count 0)
    ("swin fc2 768->192 @64", 64, 768, 192, 1, 1, 1, 0, 1, 4 * U),
    ("swin qkv 192->576 @64", 64, 192, 576, 1, 1, 1, 0, 0, 4 * U),
    ("swin proj 192->192 @64", 64, 192, 192, 1, 1, 1, 0, 1, 4 * U),
    ("swin fc1 192->768 @32", 32, 192, 768, 1, 1, 1, 1, 0, 4 * U),
    ("swin emb 160->192 @64", 64, 160, 192, 1, 1, 1, 0, 0, 2 * U),
    ("unet skip 320->160 @64", 64, 320, 160, 1, 1, 1, 0, 0, 2 * U),
    ("swin qkv 192->576 @16", 16, 192, 576, 1, 1, 1, 0, 0, 4 * U),
    ("swin proj 192->192 @16", 16, 192, 192, 1, 1, 1, 0, 1, 4 * U),
    ("swin fc1 192->768 @16", 16, 192, 768, 1, 1, 1, 1, 0, 4 * U),
    ("swin fc2 768->192 @16", 16, 768, 192, 1, 1, 1, 0, 1, 4 * U),
    ("swin qkv 192->576 @8", 8, 192, 576, 1, 1, 1, 0, 0, 6 * U),
    ("swin proj 192->192 @8", 8, 192, 192, 1, 1, 1, 0, 1, 6 * U),
    ("swin fc1 192->768 @8", 8, 192, 768, 1, 1, 1, 1, 0, 6 * U),
    ("swin fc2 768->192 @8", 8, 768, 192, 1, 1, 1, 0, 1, 6 * U),
    ("swin emb 640->192 @8", 8, 640, 192, 1, 1, 1, 0, 0, 3 * U),
    ("swin unemb 192->640 @8", 8, 192, 640, 1, 1, 1, 0, 0, 3 * U),
    ("ae c3 512->512 @64", 64, 512, 512, 3, 1, 1, 0, 1, 17),
    ("ae c3 128->128 @256", 256, 128, 128, 3, 1, 1, 0, 1, 9),
    ("ae c3 256->256 @128", 128, 256, 256, 3, 1, 1, 0, 1, 8),
    ("ae up 512->512 @128", 128, 512, 512, 3, 1, 2, 0, 0, 1),
    ("ae up 256->256 @256", 256, 256, 256, 3, 1, 2, 0, 0, 1),
    ("ae c3 512->256 @128", 128, 512, 256, 3, 1, 1, 0, 0, 1),
    ("ae c3 256->128 @256", 256, 256, 128, 3, 1, 1, 0, 0, 1),
    ("ae q 512->512 @64 (1x1)", 64, 512, 512, 1, 1, 1, 0, 0, 8),
]
only = [t for t in os.environ.get("RS_BENCH_ONLY", "").split(",") if t]   # substring filter on the shape names
if only:
    SHAPES = [sh for sh in SHAPES if any(t in sh[0] for t in only)]
lib = _lib.load()
dev = torch.device("cuda:0")
st = _lib.current_stream_ptr()
tot_ms = tot_fl = 0.0
print(f"{'shape':28s} {'M':>8s} {'N':>5s} {'K':>6s} {'ms':>8s} {'TF/s':>8s} {'n/pass':>6s} {'ms/pass':>8s}")
for name, Ho, Cin, Cout, k, stride, up, act, res, cnt in SHAPES:
    Hs = Ho * stride // up
    x = to_store(torch.randn(B, Hs, Hs, Cin, device=dev))
    w = to_store(torch.randn(Cout, k * k * Cin, device=dev) / (k * k * Cin) ** 0.5)
    bias = torch.randn(Cout, device=dev)
    y = torch.empty(B, Ho, Ho, Cout, device=dev, dtype=torch.float32 if P == 2 else dt)   # (split: 4 bytes per element)
    r = to_store(torch.randn(B, Ho, Ho, Cout, device=dev)) if res else None
    ms = C.c_float(0)
    rc = lib.rs_op_conv2d_bench(x.data_ptr(), w.data_ptr(), bias.data_ptr(), r.data_ptr() if r is not None else None, y.data_ptr(), B, Hs, Hs,
                                Cin, Cout, k, k, stride, k // 2, Ho, Ho, up, act, P, P, reps, C.byref(ms), st)
    assert rc == 0, _lib.last_error()
    M, K = B * Ho * Ho, k * k * Cin
    fl = 2.0 * M * Cout * K
    print(f"{name:28s} {M:8d} {Cout:5d} {K:6d} {ms.value:8.3f} {fl / ms.value / 1e9:8.1f} {cnt:6d} {ms.value * cnt:8.2f}", flush=True)
    if hasattr(lib, "rs_igemm4_phase_cycles") and os.environ.get("RS_IG4_PHASES"):   # ablate builds: in-kernel phase timing of the halo kernel
        out3 = (C.c_double * 3)()
        lib.rs_igemm4_phase_cycles.argtypes = [C.c_int, C.POINTER(C.c_double)]
        if lib.rs_igemm4_phase_cycles(256, out3) == 0:
            print(f"    igemm4 phases (mean s_memtime ticks over 256 workgroups): setup {out3[0]:.0f}  K loop {out3[1]:.0f}  epilogue {out3[2]:.0f}", flush=True)
    if hasattr(lib, "rs_igemm2_phase_cycles") and os.environ.get("RS_IG2_PHASES"):   # ablate builds: phases of igemm2
        out3 = (C.c_double * 3)()
        lib.rs_igemm2_phase_cycles.argtypes = [C.c_int, C.POINTER(C.c_double)]
        if lib.rs_igemm2_phase_cycles(64, out3) == 0:
            print(f"    igemm2 phases (mean s_memtime ticks over 64 workgroups): setup {out3[0]:.0f}  K loop {out3[1]:.0f}  epilogue {out3[2]:.0f}", flush=True)
    if hasattr(lib, "rs_igemm_split_phase_cycles") and os.environ.get("RS_IGS_PHASES"):   # ablate builds: phases of the split implicit GEMM
        out3 = (C.c_double * 3)()
        lib.rs_igemm_split_phase_cycles.argtypes = [C.c_int, C.POINTER(C.c_double)]
        if lib.rs_igemm_split_phase_cycles(64, out3) == 0:
            print(f"    igemm_split phases (mean s_memtime ticks over 64 workgroups): setup {out3[0]:.0f}  K loop {out3[1]:.0f}  epilogue {out3[2]:.0f}", flush=True)
    tot_ms += ms.value * cnt
    tot_fl += fl * cnt
    del x, w, y, r
print(f"weighted total: {tot_ms:.1f} ms/pass for {tot_fl / 1e12:.1f} TFLOP -> {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
