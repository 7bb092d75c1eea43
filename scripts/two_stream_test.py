"""GPU experiment: one batch of 32 as N concurrent sub-batches on N HIP streams (N engines sharing nothing but the device).
The small-level kernels of a sub-batch are latency-bound and leave most CUs idle; another sub-batch's kernels can fill them.
    python scripts/two_stream_test.py [fp16|parity] [N ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from resshift_amd import UNetModelSwin, VQModelTorch, create_gaussian_diffusion
from resshift_amd.engine import Engine
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
pol = sys.argv[1] if len(sys.argv) > 1 else "fp16"
ways = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
up, ap, dp = H.realsr_params()
usd, asd = H.weights(up, ap)
um = UNetModelSwin(**up).to(dev); um.load_state_dict(usd)
am = VQModelTorch(**ap).to(dev); am.load_state_dict(asd)
d = create_gaussian_diffusion(**dp)
T = dp["steps"]
pu, pe, pd = (["split"] * T, "split", "fp16") if pol == "parity" else (["fp16"] * T, "fp16", "fp16")
tabs = d.step_tables()
B = 32
y, noises, _ = H.synth.synthetic_inputs(H.SEED_X, B, 64, 64, 3, 64, 64, T)
y = y.to(dev); nb = torch.stack(noises, 0).to(dev)
engines = [d._fused_engine(um, am)]
ref = None
for n in ways:
    while len(engines) < n:
        e = Engine(unet_params=um.params, ae_params=am.params, device=dev)
        e.load_state_dicts(unet_sd=um.state_dict(), ae_sd=am.state_dict())
        engines.append(e)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
    per = B // n
    def run():
        outs = []
        cur = torch.cuda.current_stream(dev)
        for k in range(n):
            streams[k].wait_stream(cur)
            with torch.cuda.stream(streams[k]):
                outs.append(engines[k].sample(y[k * per:(k + 1) * per], nb[:, k * per:(k + 1) * per].contiguous(), tabs, sf=4, scale_factor=1.0,
                                              prec_unet=pu, prec_encode=pe, prec_decode=pd))
        for k in range(n):
            cur.wait_stream(streams[k])
        return torch.cat(outs, 0)
    for _ in range(2):
        out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        out = run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    if ref is None:
        ref = out
    same = torch.equal(out, ref)
    print(f"{pol}: {n} concurrent sub-batches of {per}: {ms:8.1f} ms / batch of 32  {32e3 / ms:6.1f} img/s   bit-identical to the single-stream output: {same}", flush=True)
