"""One tile of the reference's DEFAULT tile size (inference_resshift.py:54-58,149-161: --chop_size 512 for the x4 models): a
512 x 512 LR input is a 512 x 512 latent for the UNet (64 x the constructed 64 x 64) and a 2048 x 2048 autoencoder image whose
mid-block attention runs over T = 262 144 tokens (2 x ~141 TFLOP).  Records wall time per pass, the scratch arena, the number of
query-row blocks the attention is processed in, and the per-family kernel times of one profiled pass.

    python scripts/tiled_chop512.py [side=512] [policy=fp16|parity] > gpurun_out/tiled_chop512.txt
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402
from resshift_amd import ResShiftSampler  # noqa: E402
from resshift_amd.config import ConfigNode  # noqa: E402

torch.set_grad_enabled(False)
side = int(sys.argv[1]) if len(sys.argv) > 1 else 512
policy = sys.argv[2] if len(sys.argv) > 2 else "fp16"
dev = torch.device("cuda:0")
up, ap, dp = H.realsr_params()
usd, asd = H.weights(up, ap)
cfg = ConfigNode(model=ConfigNode(target="models.unet.UNetModelSwin", ckpt_path=None, params=up),
                 diffusion=ConfigNode(target="models.script_util.create_gaussian_diffusion", params=dp),
                 autoencoder=ConfigNode(target="ldm.models.autoencoder.VQModelTorch", ckpt_path=None, params=ap))
s = ResShiftSampler(cfg, sf=4, use_amp=True, chop_size=side, chop_stride=side - 64, chop_bs=1, padding_offset=64, seed=7,
                    state_dicts={"model": usd, "autoencoder": asd})
T = dp["steps"]
if policy == "parity":
    s.set_precision(["split"] * T, "split", "fp16")
g = torch.Generator().manual_seed(3)
y = (torch.rand(1, 3, side, side, generator=g) * 2 - 1).to(dev)
eng = s.model.engine()
times = []
for it in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = s.sample_tiled(y, noise_repeat=True)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
assert tuple(out.shape) == (1, 3, 4 * side, 4 * side) and torch.isfinite(out).all() and out.abs().max().item() <= 1.0
Ttok = side * side
budget = int(os.environ.get("RS_ATTN_S_FLOATS", str(1 << 30)))
rows = Ttok if Ttok * Ttok <= budget else max(128, min(Ttok, (budget // Ttok) // 128 * 128))
print(f"tile {side} x {side} LR (latent {side} x {side}, autoencoder image {4 * side} x {4 * side}), policy {policy}, B = 1, {T} steps")
print(f"wall time per tile: first call {times[0]:.3f} s (includes arena allocation), second call {times[1]:.3f} s")
print(f"scratch arena: {eng.arena_bytes() / 2**30:.2f} GiB; kernel launches per tile: {eng.last_launch_count()}")
print(f"AE mid-block attention: T = {Ttok} tokens, {rows} query rows per block -> {(Ttok + rows - 1) // rows} blocks per attention "
      f"(QK^T + PV = {4.0 * Ttok * Ttok * 512 / 1e12:.1f} TFLOP per attention, two attentions per tile)")
eng.profile_enable(True)
os.environ["RS_PROF_SHAPES"] = "1"
out = s.sample_tiled(y, noise_repeat=True)
torch.cuda.synchronize()
st = eng.profile_get()
print(f"whole tile (one rs_sample call) MFMA-family kernel time {st['igemm_ms']:.1f} ms")
for name, fl, ms, n in eng.profile_families():
    if n:
        print(f"  {name:75s} {ms:9.2f} ms  {n:5d} launches  {fl / max(ms, 1e-9) / 1e9:8.1f} TFLOP/s")
eng.profile_enable(False)
