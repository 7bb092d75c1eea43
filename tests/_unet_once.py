"""Helper of test_fused_swin_paths_match_unfused: one full-size fp16 UNet forward (batch 4) with whatever RS_* knobs the
parent put in the environment (they are read once per process); saves the output for a bitwise / tolerance comparison."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H  # noqa: E402
from resshift_amd.engine import F16, Engine  # noqa: E402

torch.set_grad_enabled(False)
up, ap, dp = H.realsr_params()
usd, _ = H.weights(up, ap)
dev = torch.device("cuda:0")
eng = Engine(unet_params=up, ae_params=None, enable_f32=False, device=dev)
eng.load_state_dicts(unet_sd=usd)
eng.mark_weights_ready()
g = torch.Generator().manual_seed(21)
x = torch.randn(4, 3, 64, 64, generator=g).to(dev)
lq = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).to(dev)
out = eng.unet_forward(x, [7, 7, 7, 7], lq=lq, prec=F16)
torch.cuda.synchronize()
torch.save(out.cpu(), sys.argv[1])
