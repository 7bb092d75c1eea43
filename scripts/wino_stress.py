"""Race hunt for wino.hip: the same conv many times, count launches / elements whose result differs from the float64 reference by more than rounding."""
import math, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resshift_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B, H, Cin, Cout = 8, 64, 160, 160
x = torch.randn(B, Cin, H, H, generator=g)
w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
ref = F.conv2d(x.double(), w.double(), None, padding=1)
xs = ops.convert(x.permute(0, 2, 3, 1).contiguous().to(dev), ops.SPLIT)
bad_l = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    y = ops.conv3x3_wino(xs, w, None)
    torch.cuda.synchronize()
    got = ops.convert(y, ops.F32).cpu().permute(0, 3, 1, 2).double()
    err = (got - ref).abs()
    nb = int((err > 2e-5).sum())
    tiles = int((err.amax(1).reshape(B, 8, 8, 4, 16).amax(dim=(2, 4)) > 2e-5).sum())
    bad_l += nb > 0
    print(f"iter {it}: max err {err.max().item():.2e}, bad elements {nb}, bad tiles {tiles}", flush=True)
print("launches with errors:", bad_l)
