"""GPU probe: fixed per-workgroup overhead vs per-K-stage cost of the implicit-GEMM kernel (1x1 conv shapes)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resshift_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = _lib.current_stream_ptr()
def run(M, N, K, act=0, res=0, k=1, H=None):
    B = 32; Ho = int((M // B) ** 0.5)
    Cin = K // (k * k)
    x = torch.randn(B, Ho, Ho, Cin, device=dev).half(); w = torch.randn(N, K, device=dev).half() * K ** -0.5
    b = torch.randn(N, device=dev); y = torch.empty(B, Ho, Ho, N, device=dev, dtype=torch.half)
    r = torch.randn(B, Ho, Ho, N, device=dev).half() if res else None
    ms = C.c_float(0)
    rc = lib.rs_op_conv2d_bench(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), B, Ho, Ho, Cin, N, k, k, 1, k // 2, Ho, Ho, 1, act, 0, 0, 20, C.byref(ms), st)
    assert rc == 0
    fl = 2.0 * M * N * K
    print(f"M={M:7d} N={N:4d} K={K:5d} k={k} act={act} res={res}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:7.1f} TF/s  bytes/us={(M*Cin*2+M*N*2*(1+res))/ms.value/1e3:8.1f} MB/ms", flush=True)
M = 131072
for K in (64, 128, 192, 384, 768, 1536, 3072):
    run(M, 768, K)
for N in (128, 256, 384, 768):
    run(M, N, 192)
run(M, 768, 192, act=1)
run(M, 192, 192, res=1)
run(M, 128, 64)
run(32768, 768, 192)
run(M, 128, 1152, k=3)
run(M, 128, 2304, k=3)
run(M, 128, 4608, k=3)
