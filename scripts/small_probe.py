"""GPU probe: small-M (8x8 / 16x16 UNet level) conv shapes under different split-K settings (env RS_SPLITK_*)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resshift_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = _lib.current_stream_ptr()
def run(Ho, Cin, N, k, res=1):
    B = 32
    x = torch.randn(B, Ho, Ho, Cin, device=dev).half(); w = torch.randn(N, k*k*Cin, device=dev).half() * (k*k*Cin) ** -0.5
    b = torch.randn(N, device=dev); y = torch.empty(B, Ho, Ho, N, device=dev, dtype=torch.half)
    r = torch.randn(B, Ho, Ho, N, device=dev).half() if res else None
    ms = C.c_float(0)
    rc = lib.rs_op_conv2d_bench(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), B, Ho, Ho, Cin, N, k, k, 1, k // 2, Ho, Ho, 1, 0, 0, 0, 30, C.byref(ms), st)
    assert rc == 0
    fl = 2.0 * B * Ho * Ho * N * k * k * Cin
    print(f"  {Ho:3d}^2 {Cin:5d}->{N:4d} k{k}: {ms.value*1e3:7.1f} us {fl/ms.value/1e9:7.1f} TF/s", flush=True)
print("target", os.environ.get("RS_SPLITK_TARGET"), "minstages", os.environ.get("RS_SPLITK_MINSTAGES"))
for shp in ((8, 640, 640, 3), (8, 1280, 640, 3), (8, 960, 640, 3), (16, 320, 320, 3), (16, 640, 320, 3), (16, 960, 320, 3), (8, 192, 768, 1), (8, 768, 192, 1), (8, 640, 192, 1), (16, 192, 768, 1), (16, 768, 192, 1), (16, 192, 576, 1)):
    run(*shp)
