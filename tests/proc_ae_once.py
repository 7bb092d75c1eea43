"""Worker of the AE attention row-block tests (tests/test_engine_gpu.py): encode (+ decode) one seeded image with the engine and
save the result.  RS_ATTN_S_FLOATS (read once per process) bounds the materialised score block.
    python tests/proc_ae_once.py <out.pt> <tiny|realsr> <side> <prec>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402
from resshift_amd import VQModelTorch  # noqa: E402

torch.set_grad_enabled(False)
out, which, side, prec = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
dev = torch.device("cuda:0")
if which == "tiny":
    up, ap, _, _ = H.CASES["tiny"]
else:
    up, ap, _ = H.realsr_params()
_, asd = H.weights(up, ap)
am = VQModelTorch(**ap).to(dev).eval()
am.load_state_dict(asd, strict=True)
g = torch.Generator().manual_seed(side)
img = (torch.rand(1, 3, side, side, generator=g) * 2 - 1).to(dev)
z = am.encode(img, prec=prec)
res = {"z": z.cpu()}
if which == "tiny" or os.environ.get("RS_TEST_DECODE"):
    res["img"] = am.decode(z, prec=prec).cpu()
torch.cuda.synchronize()
torch.save(res, out)
