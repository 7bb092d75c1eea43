"""Baseline legs of bench.py on the UNMODIFIED reference modules - measurement infrastructure only.

`available()` is true where the reference tree - or the verified byte-identical copy oracle/make_ref_copy.py leaves in the git-ignored
oracle/_ref/ (that is what travels to the GPU box) - can be imported.  `Reference(up, aep, dp, usd, asd)` instantiates
`models.unet.UNetModelSwin`, `ldm.models.autoencoder.VQModelTorch` and `models.script_util.create_gaussian_diffusion` with the YAML's
parameters, loads bench.py's weights with `load_state_dict(strict=True)` and runs the reference's own `p_sample_loop_progressive` +
`decode_first_stage` (models/gaussian_diffusion.py:367-498) with the bench's noise injected where the loop calls `th.randn_like`
(:446, :358) - on the CPU in fp32 (`cpu_baseline.kind = "reference"`), or moved to the GPU under
`torch.autocast(float16)` exactly as sampler.py:185 runs it (`torch_rocm_autocast_baseline`).  The product path never imports this.
"""
from __future__ import annotations

import contextlib
import sys

import torch

from . import ref_import


def available() -> bool:
    return ref_import.available()


class Reference:
    def __init__(self, up, aep, dp, usd, asd):
        from .make_golden import ref_sample   # drives the reference loop with injected noise

        with contextlib.redirect_stdout(sys.stderr):   # the reference prints notices on import / construction; bench.py's stdout is ONE JSON line
            U, V, create = ref_import.load()
            self._ref_sample = ref_sample
            self.unet = U(**up).eval()
            self.unet.load_state_dict(usd, strict=True)
            self.ae = V(**aep).eval()
            self.ae.load_state_dict(asd, strict=True)
            self.diffusion = create(**dp)
        self.source = ("verified copy of the reference's modules (oracle/_ref/reference_modules.zip, sha256 manifest checked)" if ref_import.is_copy()
                       else f"reference tree at {ref_import.REF}")

    def to(self, dev):
        self.unet.to(dev)
        self.ae.to(dev)
        return self

    @torch.no_grad()
    def sample(self, y, noises, mask=None, autocast_dtype=None):
        """-> (image, final latent, VQ indices [B, h*w]) like oracle.sample_loop(..., return_aux=True)"""
        if autocast_dtype is not None:
            with torch.autocast(y.device.type, dtype=autocast_dtype):
                img, z, idx = self._ref_sample(self.diffusion, self.unet, self.ae, y, list(noises), mask=mask)
        else:
            img, z, idx = self._ref_sample(self.diffusion, self.unet, self.ae, y, list(noises), mask=mask)
        return img.float(), z.float(), idx.reshape(y.shape[0], -1)
