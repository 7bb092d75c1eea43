"""Run ONE conv shape a few times (for rocprofv3 --pmc): python scripts/one_conv.py Ho Cin Cout k [reps]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resshift_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = _lib.current_stream_ptr()
Ho, Cin, N, k = [int(a) for a in sys.argv[1:5]]; reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
B = 32
x = torch.randn(B, Ho, Ho, Cin, device=dev).half(); w = torch.randn(N, k*k*Cin, device=dev).half() * (k*k*Cin) ** -0.5
b = torch.randn(N, device=dev); y = torch.empty(B, Ho, Ho, N, device=dev, dtype=torch.half)
ms = C.c_float(0)
rc = lib.rs_op_conv2d_bench(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), B, Ho, Ho, Cin, N, k, k, 1, k // 2, Ho, Ho, 1, 0, 0, 0, reps, C.byref(ms), st)
print("ms", ms.value, "TF", 2.0 * B * Ho * Ho * N * k * k * Cin / ms.value / 1e9)
