"""Subprocess half (the RS_* knobs are read once per process, so an A/B needs a fresh one) of test_fused_swin_paths_match_unfused / test_groupnorm_tails_change_no_bit: one full-size UNet forward (fp16, batch 4 - or
RS_TEST_PREC / RS_TEST_B) with whatever RS_* knobs the parent put in the environment (they are read once per process); saves the output
(RS_TEST_META=1: a dict with the engine's kernel-launch count as well) for a bitwise / tolerance comparison."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H  # noqa: E402
from resshift_amd.engine import Engine, parse_precision  # noqa: E402

torch.set_grad_enabled(False)
up, ap, dp = H.realsr_params()
usd, _ = H.weights(up, ap)
dev = torch.device("cuda:0")
prec = parse_precision(os.environ.get("RS_TEST_PREC", "fp16"))
B = int(os.environ.get("RS_TEST_B", "4"))
eng = Engine(unet_params=up, ae_params=None, enable_f32=False, device=dev)
eng.load_state_dicts(unet_sd=usd)
eng.mark_weights_ready()
g = torch.Generator().manual_seed(21)
x = torch.randn(B, 3, 64, 64, generator=g).to(dev)
lq = (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).to(dev)
out = eng.unet_forward(x, [7] * B, lq=lq, prec=prec)
torch.cuda.synchronize()
if os.environ.get("RS_TEST_META"):
    out2 = eng.unet_forward(x, [7] * B, lq=lq, prec=prec)   # a second call: the ticket pool is re-zeroed per call, the plan re-made
    torch.cuda.synchronize()
    torch.save({"out": out.cpu(), "out2": out2.cpu(), "launches": eng.last_launch_count()}, sys.argv[1])
else:
    torch.save(out.cpu(), sys.argv[1])
