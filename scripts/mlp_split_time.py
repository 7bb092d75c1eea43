import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, math, time
from resshift_amd import ops
gpu = torch.device("cuda:0")
E, HD, M = 192, 768, 131072
g = torch.Generator().manual_seed(1)
x = torch.randn(M, E, generator=g)
w1 = torch.randn(HD, E, generator=g) / math.sqrt(E); w2 = torch.randn(E, HD, generator=g) / math.sqrt(HD)
b1, b2 = torch.randn(HD, generator=g) * .3, torch.randn(E, generator=g) * .3
xs = ops.convert(x.to(gpu), ops.SPLIT)
for _ in range(3): y = ops.swin_mlp(xs, w1, b1, w2, b2, xs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
lib = ops._lib.load()
w1d, w2d = ops.split_pack_rows(w1).to(gpu), ops.split_pack_rows(w2).to(gpu)
b1d, b2d = b1.to(gpu), b2.to(gpu)
y = torch.empty_like(xs)
e0.record()
for _ in range(20):
    lib.rs_op_swin_mlp_split(xs.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), xs.data_ptr(), y.data_ptr(), M, E, HD, ops._lib.current_stream_ptr())
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"split fused mlp M={M}: {ms*1e3:.1f} us  {4.0*M*E*HD/ms/1e9:.1f} TFLOP/s (x3 MFMA: {12.0*M*E*HD/ms/1e9:.1f})")


def phases(tag):
    """ablate builds: wave 0's phase stamps of the last launch (mean over the first 1024 workgroups)"""
    import ctypes
    f = getattr(lib, "rs_mlp_phase_cycles", None)
    if f is None:
        return
    out = (ctypes.c_double * 6)()
    if f(ctypes.c_int(min(1024, M // 128)), out) == 0:
        names = ["prologue (weight DMA issue, token fragments, GN fold)", "first step", "steps 1 .. n-1", "barrier", "b2 + residual (+ statistics)", "split / staging / stores"]
        print(f"  {tag} phases: " + "  ".join(f"{n}: {out[i]:.0f}" for i, n in enumerate(names)) + f"   total {sum(out):.0f}")


phases("split")

# fp16 fused kernel on the same shape for comparison
xh = x.to(gpu, torch.float16)
w1h, w2h = w1.to(gpu, torch.float16), w2.to(gpu, torch.float16)
yh = torch.empty_like(xh)
for _ in range(3):
    lib.rs_op_swin_mlp(xh.data_ptr(), w1h.data_ptr(), b1d.data_ptr(), w2h.data_ptr(), b2d.data_ptr(), xh.data_ptr(), yh.data_ptr(), M, E, HD, ops._lib.current_stream_ptr())
e0.record()
for _ in range(20):
    lib.rs_op_swin_mlp(xh.data_ptr(), w1h.data_ptr(), b1d.data_ptr(), w2h.data_ptr(), b2d.data_ptr(), xh.data_ptr(), yh.data_ptr(), M, E, HD, ops._lib.current_stream_ptr())
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"fp16 fused mlp M={M}: {ms*1e3:.1f} us  {4.0*M*E*HD/ms/1e9:.1f} TFLOP/s")
phases("fp16")
