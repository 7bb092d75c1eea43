"""Drop-in `ResShiftSampler` (reference: sampler.py:26-308) on top of the HIP engine.

Same constructor and `sample_func` / `inference` signatures as the reference.  Differences that do not
change results: models come from the engine-backed classes (the YAML `target:` strings of the
reference are mapped onto them), every rank can receive the packed weights through ONE RCCL
broadcast instead of re-reading the checkpoint (sharding.py), and image file I/O uses PIL (the
reference uses cv2, which is a host-side detail outside the hot path).
"""
from __future__ import annotations

import math
import os
import random
from contextlib import nullcontext
from pathlib import Path
from typing import Mapping, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import sharding
from .autoencoder import VQModelTorch
from .config import ConfigNode, load_config
from .gaussian_diffusion import create_gaussian_diffusion
from .unet import UNetModelSwin

# the reference's only plug-in mechanism is the `target:` string (utils/util_common.py:19-29)
TARGETS = {
    "models.unet.UNetModelSwin": UNetModelSwin,
    "models.script_util.create_gaussian_diffusion": create_gaussian_diffusion,
    "ldm.models.autoencoder.VQModelTorch": VQModelTorch,
    "resshift_amd.unet.UNetModelSwin": UNetModelSwin,
    "resshift_amd.gaussian_diffusion.create_gaussian_diffusion": create_gaussian_diffusion,
    "resshift_amd.autoencoder.VQModelTorch": VQModelTorch,
}


def instantiate_from_config(config: Mapping):
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    try:
        cls = TARGETS[config["target"]]
    except KeyError as e:
        raise NotImplementedError(f"target {config['target']} is outside the accelerated hot path") from e
    return cls(**dict(config.get("params", {})))


def reload_model(model: torch.nn.Module, ckpt: Mapping[str, torch.Tensor]) -> None:
    """utils/util_net.py:86-98 semantics: tolerate `module.` / `_orig_mod.` prefixes, require every key."""
    first = next(iter(ckpt.keys()))
    module_flag = first.startswith("module.")
    compile_flag = "_orig_mod" in first
    for key, value in model.state_dict().items():
        tkey = key
        if compile_flag:
            tkey = "_orig_mod." + tkey
        if module_flag:
            tkey = "module." + tkey
        assert tkey in ckpt, f"checkpoint is missing {tkey}"
        value.copy_(ckpt[tkey])


class BaseSampler:
    # precision policies by name: (UNet, encoder, decoder) storage types
    POLICIES = {"parity": ("split", "split", "fp16"), "fp16": ("fp16", "fp16", "fp16"), "fp32": ("fp32", "fp32", "fp32"),
                "split": ("split", "split", "split")}

    def __init__(self, configs, sf=4, use_amp=True, chop_size=128, chop_stride=128, chop_bs=1, padding_offset=16, seed=10000,
                 state_dicts: Optional[Mapping[str, Mapping[str, torch.Tensor]]] = None, blob_cache=None, precision=None, pack="policy"):
        """`state_dicts` ({"model": sd, "autoencoder": sd}) replaces checkpoint files, e.g. for synthetic-weight runs.
        `blob_cache`: file that keeps the packed device weights between runs (sharding.build_engine_with_broadcast).
        `precision`: "parity" | "fp16" | "fp32" | "split" (POLICIES).  Default: "parity" when `use_amp` (the reduced-precision path of the
        reference, sampler.py:185 - here the fastest policy that still reproduces the reference's CPU output to >= 60 dB: split-precision
        encoder + UNet, fp16 decoder; the reference's own autocast path, and "fp16" here, flip 2 - 7 % of the VQ codes), "fp32" otherwise.
        `pack`: "policy" - the engine packs, broadcasts and caches only the weight forms that policy needs (a later set_precision to a
        form that was not packed fails loudly); "all" - every form (a process that switches policies)."""
        self.configs = configs if isinstance(configs, Mapping) else load_config(configs)
        self.sf = sf
        self.chop_size, self.chop_stride, self.chop_bs = chop_size, chop_stride, chop_bs
        self.seed = seed
        self.use_amp = use_amp
        self.precision = precision if precision is not None else ("parity" if use_amp else "fp32")
        if self.precision not in self.POLICIES:
            raise ValueError(f"unknown precision policy {self.precision!r} (one of {sorted(self.POLICIES)})")
        if pack not in ("policy", "all"):
            raise ValueError("pack must be 'policy' or 'all'")
        self.pack = pack
        self.padding_offset = padding_offset
        self._state_dicts = state_dicts
        self._blob_cache = blob_cache
        self.setup_dist()
        self.setup_seed()
        self.build_model()

    def setup_seed(self, seed=None):
        seed = self.seed if seed is None else seed
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)

    def setup_dist(self, gpu_id=None):
        """One process per GPU (torchrun); RCCL through torch.distributed backend 'nccl' (sampler.py:66-77)."""
        self.num_gpus, self.rank = sharding.init_distributed()
        self.device = torch.device("cuda", torch.cuda.current_device())

    def write_log(self, log_str):
        if self.rank == 0:
            print(log_str, flush=True)

    def _load_sd(self, which: str, ckpt_path):
        if self._state_dicts is not None and which in self._state_dicts:
            return self._state_dicts[which]
        assert ckpt_path is not None, f"configs.{which}.ckpt_path is required"
        state = torch.load(ckpt_path, map_location="cpu")
        return state["state_dict"] if "state_dict" in state else state

    def build_model(self):
        c = self.configs
        self.write_log(f"Building the diffusion model with length: {c['diffusion']['params']['steps']}...")
        self.base_diffusion = instantiate_from_config(c["diffusion"])
        model = instantiate_from_config(c["model"]).to(self.device)
        autoencoder = instantiate_from_config(c["autoencoder"]).to(self.device) if c.get("autoencoder") is not None else None
        if autoencoder is None:
            raise NotImplementedError("the accelerated path samples in the VQ latent space (all shipped configs)")
        # rank 0 reads + packs the weights once; the packed blob reaches the other ranks by one RCCL broadcast
        eng = sharding.build_engine_with_broadcast(
            model, autoencoder,
            load_fn=lambda: (self._load_sd("model", c["model"].get("ckpt_path")), self._load_sd("autoencoder", c["autoencoder"].get("ckpt_path"))),
            rank=self.rank, world=self.num_gpus, blob_cache=self._blob_cache,
            cache_fingerprint=sharding.checkpoint_fingerprint([c["model"].get("ckpt_path"), c["autoencoder"].get("ckpt_path")]),
            precisions=None if self.pack == "all" else set(self.POLICIES[self.precision]))
        self.base_diffusion.adopt_engine(model, autoencoder, eng)
        self.model = model.eval()
        self.autoencoder = autoencoder.eval()
        self.engine = eng
        self.base_diffusion.set_precision(*self.POLICIES[self.precision])

    def set_precision(self, unet=None, encode=None, decode=None):
        self.base_diffusion.set_precision(unet, encode, decode)


class ResShiftSampler(BaseSampler):
    def sample_func(self, y0, noise_repeat=False, mask=False, noise=None, step_noises=None):
        """y0: [n,c,h,w] in [-1,1]; returns [n,c,h*sf,w*sf] in [-1,1] (sampler.py:119-165)."""
        if noise_repeat:
            self.setup_seed()
        if mask is False:
            mask = None
        offset = self.padding_offset
        ori_h, ori_w = y0.shape[2:]
        flag_pad = not (ori_h % offset == 0 and ori_w % offset == 0)
        if flag_pad:
            pad_h = (math.ceil(ori_h / offset)) * offset - ori_h
            pad_w = (math.ceil(ori_w / offset)) * offset - ori_w
            y0 = sharding.reflect_pad(y0, pad_h, pad_w)
            if mask is not None:
                # (deviation: the reference pads y0 only, sampler.py:130-138, and then fails in the UNet's channel concat when an
                # inpainting input needs padding; padding the mask alike keeps such inputs usable and changes nothing otherwise)
                mask = sharding.reflect_pad(mask, pad_h, pad_w)
        cond_lq = self.configs["model"]["params"].get("cond_lq", True)
        if cond_lq and mask is not None:
            model_kwargs = {"lq": y0, "mask": mask}
        elif cond_lq:
            model_kwargs = {"lq": y0}
        else:
            model_kwargs = None
        results = self.base_diffusion.p_sample_loop(y=y0, model=self.model, first_stage_model=self.autoencoder, noise=noise,
                                                    noise_repeat=noise_repeat, clip_denoised=(self.autoencoder is None),
                                                    denoised_fn=None, model_kwargs=model_kwargs, progress=False,
                                                    step_noises=step_noises)
        if flag_pad:
            results = results[:, :, : ori_h * self.sf, : ori_w * self.sf]
        return results.clamp_(-1.0, 1.0)

    def sample_tiled(self, im_lq, mask=None, noise_repeat=False, tile_noises=None):
        """sampler.py:176-216 (`_process_per_image`): inputs larger than `chop_size` are cut into overlapping
        `chop_size` tiles (stride `chop_stride`, `chop_bs` tiles per sampler call), sampled independently and
        overlap-averaged on the GPU; smaller inputs go straight to `sample_func`.  Returns [-1,1] like sample_func.
        `tile_noises[k] = (noise, step_noises)` injects the draws of the k-th sampler call (parity runs)."""
        from .tiling import TileSplitter

        if not (im_lq.shape[2] > self.chop_size or im_lq.shape[3] > self.chop_size):
            nz = tile_noises[0] if tile_noises else (None, None)
            return self.sample_func(im_lq, noise_repeat=noise_repeat, mask=mask, noise=nz[0], step_noises=nz[1])
        x = torch.cat([im_lq, mask], dim=1) if mask is not None else im_lq
        splitter = TileSplitter(x, self.chop_size, stride=self.chop_stride, sf=self.sf, extra_bs=self.chop_bs)
        for k, (pch, index_infos) in enumerate(splitter):
            if mask is not None:
                pch, mask_pch = pch[:, :-1].contiguous(), pch[:, -1:].contiguous()
            else:
                mask_pch = None
            nz = tile_noises[k] if tile_noises else (None, None)
            out = self.sample_func(pch, noise_repeat=noise_repeat, mask=mask_pch, noise=nz[0], step_noises=nz[1])
            splitter.update(out, index_infos)
        return splitter.gather()

    # ------------------------------------------------------------------ file-level demo driver
    @staticmethod
    def _read_image_u8(path, gray=False) -> torch.Tensor:
        """uint8 HWC tensor (RGB, or one channel for masks); decoding is host work, everything after it runs on the GPU."""
        from PIL import Image

        im = np.asarray(Image.open(path).convert("L" if gray else "RGB"), dtype=np.uint8)
        return torch.from_numpy(im.reshape(im.shape[0], im.shape[1], -1).copy())

    def inference(self, in_path, out_path, mask_path=None, mask_back=True, bs=1, noise_repeat=False):
        """sampler.py:167-308: batches of `bs` images are sharded over the ranks exactly like sampler.py:273-277; every
        rank writes its own PNGs.  uint8 -> [-1,1] (datapipe/datasets.py:59-63), the inpainting blend (sampler.py:218-222)
        and the final clamp / round to uint8 (utils/util_image.py:245-269) run on the device (rs_u8_to_input /
        rs_output_to_u8): only uint8 pixels cross PCIe.  Inputs larger than `chop_size` take the tiled path."""
        in_path, out_path = Path(in_path), Path(out_path)
        if self.rank == 0:
            out_path.mkdir(parents=True, exist_ok=True)
        sharding.barrier()
        single = not in_path.is_dir()
        if single:
            files = [in_path]
        else:
            # utils/util_common.py:68-87 with recursive=True (sampler.py:246,258): one recursive glob per extension, in the
            # reference's extension order, each sorted; the inpainting loader adds 'PNG' (sampler.py:259)
            exts = ["png", "jpg", "jpeg", "JPEG", "bmp"] + (["PNG"] if mask_path is not None else [])
            files = [p for e in exts for p in sorted(in_path.glob(f"**/*.{e}"))]
        from PIL import Image

        micro = math.ceil(bs / self.num_gpus)   # sampler.py:274-277: the slice width comes from bs, also on the last, partial batch
        for b0 in range(0, len(files), bs):
            batch = files[b0:b0 + bs]
            mine = batch[self.rank * micro:(self.rank + 1) * micro]
            if mine:
                lq = self.engine.u8_to_input(torch.stack([self._read_image_u8(p) for p in mine]).to(self.device))
                mask = None
                if mask_path is not None:
                    # a directory input looks the mask up by file name (datapipe/datasets.py:470); a single input file takes
                    # mask_path as the mask file itself (sampler.py:296-297)
                    mpaths = [Path(mask_path)] if single else [Path(mask_path) / p.name for p in mine]
                    mask = self.engine.u8_to_input(torch.stack([self._read_image_u8(m, gray=True) for m in mpaths]).to(self.device))
                sr = self.sample_tiled(lq, mask=mask, noise_repeat=noise_repeat)
                blend = mask is not None and mask_back
                out_u8 = self.engine.output_to_u8(sr, lq=lq if blend else None, mask=mask if blend else None).cpu().numpy()
                for p, im in zip(mine, out_u8):
                    Image.fromarray(im if im.shape[2] != 1 else im[:, :, 0]).save(out_path / f"{p.stem}.png")
        sharding.barrier()
        self.write_log(f"Processing done, enjoy the results in {out_path}")
