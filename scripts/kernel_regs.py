"""print VGPR / spill / scratch / LDS of the kernels whose mangled name contains every given substring (reads the built library's
code objects with llvm-readelf; no GPU needed)
    python scripts/kernel_regs.py win_attn igemm4"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_host_cpu import _kernel_resource_table
from resshift_amd import _lib
t = _kernel_resource_table(_lib.LIB_PATH)
for k, v in sorted(t.items()):
    if any(a in k for a in sys.argv[1:]) or len(sys.argv) == 1:
        print(f"{k[:110]:110s} {v}")
