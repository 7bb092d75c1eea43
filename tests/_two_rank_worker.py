"""Worker of test_two_ranks_share_one_gpu (tests/test_engine_gpu.py): one of WORLD_SIZE processes on ONE GPU, gloo
rendezvous (RESSHIFT_DIST_BACKEND=gloo).  Exercises the whole multi-rank product path: rank 0 packs the weights, the blob
travels by broadcast, every rank takes its slice of the global batch and of the globally drawn noise (sampler.py:273-277)
and runs the fused loop; the per-rank outputs are gathered on every rank and rank 0 saves them."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402
from resshift_amd import UNetModelSwin, VQModelTorch, create_gaussian_diffusion, sharding  # noqa: E402


def main():
    out_path, policy = sys.argv[1], sys.argv[2]
    torch.set_grad_enabled(False)
    world, rank = sharding.init_distributed()
    dev = torch.device("cuda", torch.cuda.current_device())
    up, ap, dp, _ = H.CASES["tiny"]
    um, am = UNetModelSwin(**up).to(dev).eval(), VQModelTorch(**ap).to(dev).eval()
    eng = sharding.build_engine_with_broadcast(um, am, lambda: H.weights(up, ap), rank, world)
    d = create_gaussian_diffusion(**dp)
    d.adopt_engine(um, am, eng)
    T = dp["steps"]
    d.set_precision(*{"fp32": ("fp32", "fp32", "fp32"), "parity": (["split"] * T, "split", "fp16")}[policy])
    B = 5                                    # odd on purpose: ceil(5/2) = 3 images on rank 0, 2 on rank 1
    y, noises, _ = H.synth.synthetic_inputs(31, B, 16, 16, 3, 16, 16, T)
    noise_all = torch.stack(noises, 0)       # [T+1, B, ...] drawn for the GLOBAL batch
    y_loc = sharding.shard_batch(y, rank, world).to(dev)
    n_loc = sharding.shard_noise(noise_all, rank, world).to(dev)
    out = d.p_sample_loop(y_loc, um, first_stage_model=am, noise=n_loc[0], clip_denoised=False, model_kwargs={"lq": y_loc},
                          step_noises=list(n_loc[1:]))
    # the step-wise API on a rank whose module shells never saw the checkpoint: must use the broadcast weights
    z = am.encode(torch.nn.functional.interpolate(y_loc, scale_factor=4, mode="nearest"), prec="fp32")
    torch.cuda.synchronize()
    full = sharding.gather_images(out.cpu() if torch.distributed.get_backend() == "gloo" else out, B, rank, world)
    zsum = torch.tensor([float(z.abs().sum())], dtype=torch.float64)
    torch.distributed.all_reduce(zsum)
    if rank == 0:
        torch.save({"out": full.cpu(), "zsum": float(zsum)}, out_path)
    sharding.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
