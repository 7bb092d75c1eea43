"""Does sub-batch concurrency pay?  Runs the full sampling pass for 32 images as (a) one engine x 32 on one stream and
(b) S engines x 32/S on S streams (engines share nothing but the packed weights), all fed by one host thread.

    python scripts/two_stream_probe.py [steps]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resshift_amd.autoencoder import VQModelTorch  # noqa: E402
from resshift_amd.config import load_config, to_plain  # noqa: E402
from resshift_amd.engine import Engine  # noqa: E402
from resshift_amd.gaussian_diffusion import create_gaussian_diffusion  # noqa: E402
from resshift_amd.spec import ae_param_spec, random_state_dict, unet_param_spec  # noqa: E402
from resshift_amd.unet import UNetModelSwin  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = to_plain(load_config("realsr_swinunet_realesrgan256"))
up, aep, dp = cfg["model"]["params"], cfg["autoencoder"]["params"], cfg["diffusion"]["params"]
steps = int(dp["steps"])
usd = random_state_dict(unet_param_spec(up)[0], seed=1)
asd = random_state_dict(ae_param_spec(aep), seed=2)
diffusion = create_gaussian_diffusion(**dp)
tables = diffusion.step_tables()
B = 32
g = torch.Generator().manual_seed(1000)
y = (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).to(dev)
noise = torch.randn(steps + 1, B, 3, 64, 64, generator=g).to(dev)

eng0 = Engine(unet_params=up, ae_params=aep, device=dev)
eng0.load_state_dicts(unet_sd=usd, ae_sd=asd)
eng0.mark_weights_ready()


def clone_engine():
    e = Engine(unet_params=up, ae_params=aep, device=dev)
    e.weight_blob().copy_(eng0.weight_blob())
    torch.cuda.synchronize()
    e.mark_weights_ready()
    return e


def run(engines, streams, ys, nzs, reps):
    outs = [None] * len(engines)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    def worker(i):
        # one host thread per stream: ctypes drops the GIL inside rs_sample, so the enqueue loops run in parallel
        with torch.cuda.stream(streams[i]):
            for _ in range(reps):
                outs[i] = engines[i].sample(ys[i], nzs[i], tables, sf=diffusion.sf, scale_factor=diffusion.scale_factor)

    import threading
    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(engines))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, host / reps * 1e3, outs


ref = None
for S in (1, 2, 4):
    engines = [eng0] + [clone_engine() for _ in range(S - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    nb = B // S
    ys = [y[i * nb:(i + 1) * nb].contiguous() for i in range(S)]
    nzs = [noise[:, i * nb:(i + 1) * nb].contiguous() for i in range(S)]
    torch.cuda.synchronize()
    run(engines, streams, ys, nzs, 1)  # warm-up (arena sizing, FiLM tables)
    ms, host_ms, outs = run(engines, streams, ys, nzs, K)
    out = torch.cat(outs, 0)
    if ref is None:
        ref = out
    d = (out - ref).abs().max().item()
    print(f"S={S}: {ms:8.2f} ms per 32 images ({B / ms * 1e3:6.1f} img/s), host enqueue {host_ms:7.2f} ms, max|diff vs S=1| {d:.3e}", flush=True)
    del engines[1:]
