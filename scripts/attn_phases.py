"""Phase stamps of the fused Swin window-attention kernels (ablate build: RS_BUILD_ABLATE=1, RESSHIFT_HIP_LIB=<that library>):
wave 0's s_memtime at the phase boundaries, mean over the first workgroups of the last launch, next to the launch time.
    RESSHIFT_HIP_LIB=ab/lib_ablate.so python scripts/attn_phases.py [fp16|split|both]"""
import ctypes, math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from resshift_amd import ops

gpu = torch.device("cuda:0")
lib = ops._lib.load()
which = sys.argv[1] if len(sys.argv) > 1 else "both"
heads, E = 6, 192
NAMES = ["tokens->LDS", "GN fold", "q pass", "k pass", "v pass + V^T", "barrier", "attn w0 / scores+softmax", "attn w1 / P V", "-> proj barrier", "proj + store"]


def phases(fn, nwg):
    f = getattr(lib, fn, None)
    if f is None:
        return None
    f.restype = ctypes.c_int
    out = (ctypes.c_double * 16)()
    rc = f(ctypes.c_int(nwg), ctypes.c_int(11), out)
    return [out[i] for i in range(11)] if rc == 0 else None


def timeit(run, n=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, W) in [(32, 64, 64), (32, 32, 32), (32, 16, 16)]:
    for shift in (0, 4):
        g = torch.Generator().manual_seed(H + shift)
        x = torch.randn(B, H, W, E, generator=g)
        wqkv = torch.randn(3 * E, E, generator=g) / math.sqrt(E)
        bqkv = torch.randn(3 * E, generator=g) * 0.2
        table = torch.randn(225, heads, generator=g) * 0.5
        wproj = torch.randn(E, E, generator=g) / math.sqrt(E)
        bproj = torch.randn(E, generator=g) * 0.2
        res = torch.randn(B, H, W, E, generator=g)
        coef = torch.stack([1.0 + 0.3 * torch.randn(B, E, generator=g), 0.2 * torch.randn(B, E, generator=g)], 1)
        nwin = B * (H // 8) * (W // 8)
        flop = 2.0 * B * H * W * E * (3 * E + E) + 4.0 * nwin * heads * 64 * 64 * 32
        th, tp = ops._hostf(table)
        sp = ops._lib.current_stream_ptr
        bd, bpd = bqkv.to(gpu), bproj.to(gpu)
        if which in ("fp16", "both"):
            xh, rh = x.to(gpu, torch.float16), res.to(gpu, torch.float16)
            wd, wpd = wqkv.to(gpu, torch.float16), wproj.to(gpu, torch.float16)
            out = torch.empty_like(xh)

            def run16():
                rc = lib.rs_op_window_attention_qkv(xh.data_ptr(), wd.data_ptr(), bd.data_ptr(), wpd.data_ptr(), bpd.data_ptr(), rh.data_ptr(), out.data_ptr(),
                                                    tp, B, H, W, heads, shift, sp())
                assert rc == 0
            us = timeit(run16)
            ph = phases("rs_attn_phase_cycles", min(4096, nwin // 2))
            print(f"fp16  B={B} {H}x{W} shift {shift}: {us:7.1f} us/launch (events around the op entry: includes its per-call bias-table upload + sync)  {flop / us / 1e6:6.1f} TFLOP/s")
            if ph:
                print("      " + "  ".join(f"{n}: {v:.0f}" for n, v in zip(NAMES, ph[:10])) + f"   total {sum(ph[:10]):.0f}   launch span {ph[10]:.0f} ticks")
        if which in ("split", "both"):
            xs, rs = ops.convert(x.to(gpu), ops.SPLIT), ops.convert(res.to(gpu), ops.SPLIT)
            wd2, wpd2 = ops.split_pack_rows(wqkv).to(gpu), ops.split_pack_rows(wproj).to(gpu)
            xc = coef.to(gpu).contiguous()
            out2 = torch.empty_like(xs)

            def runs():
                rc = lib.rs_op_window_attention_qkv_split(xs.data_ptr(), wd2.data_ptr(), bd.data_ptr(), wpd2.data_ptr(), bpd.data_ptr(), rs.data_ptr(), out2.data_ptr(),
                                                          tp, xc.data_ptr(), B, H, W, heads, shift, sp())
                assert rc == 0
            us = timeit(runs)
            ph = phases("rs_attn_split_phase_cycles", min(4096, nwin))
            print(f"split B={B} {H}x{W} shift {shift}: {us:7.1f} us/launch (events around the op entry: includes its per-call bias-table upload + sync)  {flop / us / 1e6:6.1f} TFLOP/s (x3 MFMA work)")
            if ph:
                print("      " + "  ".join(f"{n}: {v:.0f}" for n, v in zip(NAMES, ph[:10])) + f"   total {sum(ph[:10]):.0f}   launch span {ph[10]:.0f} ticks")
