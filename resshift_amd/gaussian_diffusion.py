"""Residual-shifting diffusion sampler, host side (reference: models/gaussian_diffusion.py:32-66,107-609,
models/respace.py:23-70, models/script_util.py:7-56).

`create_gaussian_diffusion(**yaml.diffusion.params)` returns an object with the reference's sampling
API (`p_sample_loop`, `p_sample_loop_progressive`, `p_sample`, `p_mean_variance`,
`encode_first_stage`, `decode_first_stage`, `prior_sample`, `_scale_input`, `num_timesteps`, ...).
Schedule constants are computed in float64 numpy exactly like the reference; all tensor work is done by
the HIP engine.  When `model` / `first_stage_model` are the engine-backed `UNetModelSwin` /
`VQModelTorch`, `p_sample_loop` runs the whole loop in ONE native call (`rs_sample`).
Training-side methods (q_sample, training_losses) are out of scope.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .autoencoder import VQModelTorch
from .engine import F16, F32, Engine, parse_precision
from .unet import UNetModelSwin, params_version


def get_named_eta_schedule(schedule_name, num_diffusion_timesteps, min_noise_level, etas_end=0.99, kappa=1.0, kwargs=None):
    """sqrt(eta_t), t = 0..T-1 (gaussian_diffusion.py:32-66); only the 'exponential' family is shipped."""
    if schedule_name != "exponential":
        raise ValueError(f"Unknow schedule_name {schedule_name}")
    power = kwargs.get("power", None)
    T = num_diffusion_timesteps
    etas_start = min(min_noise_level / kappa, min_noise_level)
    growth = math.exp(math.log(etas_end / etas_start) / (T - 1))
    exponent = np.linspace(0, 1, T, endpoint=True) ** power * (T - 1)
    return np.power(np.full([T], growth), exponent) * etas_start


def space_timesteps(num_timesteps, sample_timesteps):
    """respace.py:6-18"""
    return set(int((num_timesteps / sample_timesteps) * x) for x in range(sample_timesteps))


class ResShiftDiffusion:
    """Sampling half of GaussianDiffusion + SpacedDiffusion (predict_type 'xstart')."""

    def __init__(self, *, sqrt_etas, kappa, use_timesteps, sf=4, scale_factor=None, normalize_input=True, latent_flag=True,
                 predict_type="xstart"):
        if predict_type != "xstart":
            raise NotImplementedError("engine implements predict_type='xstart' (all shipped configs)")
        self.kappa = kappa
        self.sf = sf
        self.scale_factor = scale_factor
        self.normalize_input = normalize_input
        self.latent_flag = latent_flag
        use = set(use_timesteps)
        self.original_num_steps = len(sqrt_etas)
        self.timestep_map = [i for i in range(len(sqrt_etas)) if i in use]  # respace.py:31-46
        self.sqrt_etas = np.array([sqrt_etas[i] for i in self.timestep_map], dtype=np.float64)
        self.etas = self.sqrt_etas ** 2
        assert (self.etas > 0).all() and (self.etas <= 1).all()
        self.num_timesteps = int(self.etas.shape[0])
        # posterior q(x_{t-1} | x_t, x_0) (gaussian_diffusion.py:149-161)
        self.etas_prev = np.append(0.0, self.etas[:-1])
        self.alpha = self.etas - self.etas_prev
        self.posterior_variance = kappa ** 2 * self.etas_prev / self.etas * self.alpha
        self.posterior_variance_clipped = np.append(self.posterior_variance[1], self.posterior_variance[1:])
        self.posterior_log_variance_clipped = np.log(self.posterior_variance_clipped)
        self.posterior_mean_coef1 = self.etas_prev / self.etas
        self.posterior_mean_coef2 = self.alpha / self.etas
        # precision policy of the engine-backed loop: None -> fp16 under autocast, else fp32
        self.precision_unet = None
        self.precision_encode = None
        self.precision_decode = None
        self._fused: Dict[tuple, tuple] = {}

    # ---- scalar tables handed to the native loop
    def step_tables(self) -> Dict[str, np.ndarray]:
        f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)  # _extract_into_tensor(...).float()
        if self.normalize_input and self.latent_flag:
            inv_std = 1.0 / np.sqrt(f32(self.etas).astype(np.float64) * self.kappa ** 2 + 1)
        elif self.normalize_input:
            inv_std = 1.0 / (f32(self.sqrt_etas).astype(np.float64) * self.kappa * 3 + 1)
        else:
            inv_std = np.ones_like(self.etas)
        return {
            "inv_std": f32(inv_std),
            "coef1": f32(self.posterior_mean_coef1),
            "coef2": f32(self.posterior_mean_coef2),
            "sigma": f32(np.exp(0.5 * f32(self.posterior_log_variance_clipped).astype(np.float64))),
            "tmap": np.asarray(self.timestep_map, dtype=np.int32),
            "prior_scale": np.float32(self.kappa * self.sqrt_etas[-1]),
        }

    def _prec(self, which: Optional[int]) -> int:
        if which is not None:
            return parse_precision(which)
        return F16 if torch.is_autocast_enabled() else F32

    def set_precision(self, unet=None, encode=None, decode=None):
        """unet: 'fp16' | 'fp32' | 'split' (alias 'fp16x3') | list with one entry per timestep t (index = t); encode / decode: one name."""
        self.precision_unet, self.precision_encode, self.precision_decode = unet, encode, decode

    def _unet_precisions(self):
        pu = self.precision_unet
        if isinstance(pu, (list, tuple)):
            assert len(pu) == self.num_timesteps
            return [parse_precision(p) for p in pu]
        return [self._prec(pu)] * self.num_timesteps

    # ---- fused engine (UNet + AE in one native object)
    def _fused_engine(self, model: UNetModelSwin, ae: VQModelTorch) -> Engine:
        key = (id(model), id(ae))
        ver = (params_version(model), params_version(ae))
        ent = self._fused.get(key)
        dev = next(model.parameters()).device
        if ent is None or ent[0].device != dev:
            ent = (Engine(unet_params=model.params, ae_params=ae.params, device=dev), None)
        if ent[1] != ver:
            ent[0].load_state_dicts(unet_sd=model.state_dict(), ae_sd=ae.state_dict())
            ent = (ent[0], ver)
        self._fused[key] = ent
        return ent[0]

    def adopt_engine(self, model: UNetModelSwin, ae: VQModelTorch, engine: Engine):
        """Use an already-loaded engine (e.g. one whose weights arrived by RCCL broadcast) for this model pair.  The module
        shells adopt it too: on ranks > 0 (and after a blob-cache hit) their own parameters were never filled, so
        `model.engine()` / `autoencoder.engine()` - the step-wise API, `model(x, t)`, `encode` / `decode` - must not rebuild an
        engine from those zeros."""
        self._fused[(id(model), id(ae))] = (engine, (params_version(model), params_version(ae)))
        for m in (model, ae):
            m._engine, m._engine_version = engine, params_version(m)

    # ---- reference API
    @staticmethod
    def _axpbypcz(x, z, n, a, b, c, engine=None):
        """a*x + b*z + c*n on the engine's elementwise kernel.  There is deliberately no torch/CPU arithmetic path: the
        step-wise API needs the engine-backed UNetModelSwin (it owns the device kernels)."""
        if engine is None or not x.is_cuda:
            raise RuntimeError("step-wise sampling needs the HIP engine (engine-backed UNetModelSwin on a GPU); no CPU fallback")
        return engine.axpbypcz(x, z, n, a, b, c).to(x.dtype)

    def _scale_input(self, inputs, t, engine=None):
        tab = self.step_tables()["inv_std"]
        ti = int(t[0]) if torch.is_tensor(t) else int(t)
        return self._axpbypcz(inputs, None, None, float(tab[ti]), 0.0, 0.0, engine)

    def prior_sample(self, y, noise=None, engine=None):
        if noise is None:
            noise = torch.randn_like(y)
        return self._axpbypcz(y, None, noise, 1.0, 0.0, float(np.float32(self.kappa * self.sqrt_etas[-1])), engine)

    def q_posterior_mean_variance(self, x_start, x_t, t, engine=None):
        ti = int(t[0]) if torch.is_tensor(t) else int(t)
        tab = self.step_tables()
        mean = self._axpbypcz(x_t, x_start, None, float(tab["coef1"][ti]), float(tab["coef2"][ti]), 0.0, engine)
        var = torch.full_like(x_t, float(np.float32(self.posterior_variance[ti])))
        logvar = torch.full_like(x_t, float(np.float32(self.posterior_log_variance_clipped[ti])))
        return mean, var, logvar

    def encode_first_stage(self, y, first_stage_model, up_sample=False):
        """gaussian_diffusion.py:500-515"""
        if first_stage_model is None:
            if up_sample and self.sf != 1:
                raise RuntimeError("bicubic upsampling needs an engine; pass the autoencoder")
            return y
        eng = first_stage_model.engine()
        if up_sample and self.sf != 1:
            y = eng.bicubic(y, self.sf)
        z = first_stage_model.encode(y, prec=self._prec(self.precision_encode))
        return _lib.window_copy(z, scale=self.scale_factor) if self.scale_factor != 1.0 else z   # (one fp32 multiply on the device)

    def decode_first_stage(self, z_sample, first_stage_model=None, consistencydecoder=None):
        """gaussian_diffusion.py:474-498"""
        if consistencydecoder is not None:
            raise NotImplementedError("consistency decoder is out of scope")
        if first_stage_model is None:
            return z_sample
        if self.scale_factor != 1.0:
            z_sample = _lib.window_copy(z_sample, scale=1.0 / self.scale_factor)
        return first_stage_model.decode(z_sample, prec=self._prec(self.precision_decode))

    def p_mean_variance(self, model, x_t, y, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """gaussian_diffusion.py:234-307 (START_X).  `t`: [B] tensor of equal indices."""
        model_kwargs = model_kwargs or {}
        ti = int(t[0])
        prec = self._unet_precisions()[ti]
        ts = [self.timestep_map[ti]] * x_t.shape[0]  # _WrappedModel (respace.py:67-70)
        if not isinstance(model, UNetModelSwin):
            raise NotImplementedError("p_mean_variance drives the engine-backed UNetModelSwin only")
        eng = model.engine()
        pred = model(self._scale_input(x_t, ti, eng), ts, prec=prec, **model_kwargs)
        if denoised_fn is not None:
            pred = denoised_fn(pred)
        if clip_denoised:
            pred = pred.clamp(-1, 1)
        mean, var, logvar = self.q_posterior_mean_variance(pred, x_t, ti, eng)
        return {"mean": mean, "variance": var, "log_variance": logvar, "pred_xstart": pred}

    def p_sample(self, model, x, y, t, clip_denoised=True, denoised_fn=None, model_kwargs=None, noise_repeat=False, noise=None):
        """gaussian_diffusion.py:332-365; `noise` may be injected for parity runs (draws randn_like otherwise)."""
        out = self.p_mean_variance(model, x, y, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        if noise is None:
            noise = torch.randn_like(x)
        if noise_repeat:
            noise = noise[0,].repeat(x.shape[0], 1, 1, 1)
        ti = int(t[0])
        sigma = float(self.step_tables()["sigma"][ti]) if ti != 0 else 0.0
        eng = model.engine() if isinstance(model, UNetModelSwin) else None
        sample = self._axpbypcz(out["mean"], None, noise, 1.0, 0.0, sigma, eng)
        return {"sample": sample, "pred_xstart": out["pred_xstart"], "mean": out["mean"]}

    def p_sample_loop_progressive(self, y, model, first_stage_model=None, noise=None, noise_repeat=False, clip_denoised=True,
                                  denoised_fn=None, model_kwargs=None, device=None, progress=False, step_noises=None):
        """gaussian_diffusion.py:421-472: generator of per-step dicts {"sample","pred_xstart","mean"}."""
        z_y = self.encode_first_stage(y, first_stage_model, up_sample=True)
        if noise is None:
            noise = torch.randn_like(z_y)
        if noise_repeat:
            noise = noise[0,].repeat(z_y.shape[0], 1, 1, 1)
        z_sample = self.prior_sample(z_y, noise, model.engine() if isinstance(model, UNetModelSwin) else None)
        for k, i in enumerate(list(range(self.num_timesteps))[::-1]):
            t = torch.tensor([i] * y.shape[0])   # (kept on the host: the shells read the step index from it - a device tensor would cost a sync per step)
            out = self.p_sample(model, z_sample, z_y, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                model_kwargs=model_kwargs, noise_repeat=noise_repeat,
                                noise=None if step_noises is None else step_noises[k])
            yield out
            z_sample = out["sample"]

    def p_sample_loop(self, y, model, first_stage_model=None, consistencydecoder=None, noise=None, noise_repeat=False,
                      clip_denoised=True, denoised_fn=None, model_kwargs=None, device=None, progress=False, step_noises=None,
                      return_aux=False):
        """gaussian_diffusion.py:367-419.  Fast path: one native `rs_sample` call for the whole loop."""
        fused_ok = (isinstance(model, UNetModelSwin) and isinstance(first_stage_model, VQModelTorch) and consistencydecoder is None
                    and denoised_fn is None and not clip_denoised)
        if not fused_ok:
            final = None
            for sample in self.p_sample_loop_progressive(y, model, first_stage_model=first_stage_model, noise=noise,
                                                         noise_repeat=noise_repeat, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                                         model_kwargs=model_kwargs, device=device, progress=progress,
                                                         step_noises=step_noises):
                final = sample["sample"]
            return self.decode_first_stage(final, first_stage_model=first_stage_model, consistencydecoder=consistencydecoder)
        eng = self._fused_engine(model, first_stage_model)
        B, _, h, w = y.shape
        f = 2 ** (int(eng.cfg.ae.n_levels) - 1)
        zshape = (B, int(eng.cfg.ae.embed_dim), h * self.sf // f, w * self.sf // f)
        T = self.num_timesteps
        # RNG draws in the reference's order: prior noise (:446), then one randn_like per step (:358)
        draws = []
        for k in range(T + 1):
            if k == 0 and noise is not None:
                n = noise
            elif k > 0 and step_noises is not None:
                n = step_noises[k - 1]
            else:
                n = torch.randn(zshape, device=y.device, dtype=torch.float32)
            n = n.to(y.device, torch.float32)
            if noise_repeat:
                n = n[0,].repeat(B, 1, 1, 1)
            draws.append(n)
        noise_all = torch.stack(draws, 0)
        mask = (model_kwargs or {}).get("mask", None)
        lq = (model_kwargs or {}).get("lq", None)
        if lq is not None and lq.data_ptr() != y.data_ptr() and not torch.equal(lq, y):
            raise NotImplementedError("fused loop conditions the UNet on y itself (sampler.py:140-148)")
        res = eng.sample(y, noise_all, self.step_tables(), sf=self.sf, scale_factor=self.scale_factor, mask=mask,
                         prec_unet=self._unet_precisions(), prec_encode=self._prec(self.precision_encode),
                         prec_decode=self._prec(self.precision_decode), return_aux=return_aux)
        return res


def create_gaussian_diffusion(*, normalize_input, schedule_name, sf=4, min_noise_level=0.01, steps=1000, kappa=1, etas_end=0.99,
                              schedule_kwargs=None, weighted_mse=False, predict_type="xstart", timestep_respacing=None,
                              scale_factor=None, latent_flag=True):
    """models/script_util.py:7-56 — same keyword signature; returns the engine-backed sampler."""
    sqrt_etas = get_named_eta_schedule(schedule_name, num_diffusion_timesteps=steps, min_noise_level=min_noise_level,
                                       etas_end=etas_end, kappa=kappa, kwargs=schedule_kwargs)
    if timestep_respacing is None:
        timestep_respacing = steps
    else:
        assert isinstance(timestep_respacing, int)
    return ResShiftDiffusion(sqrt_etas=sqrt_etas, kappa=kappa, use_timesteps=space_timesteps(steps, timestep_respacing), sf=sf,
                             scale_factor=1.0 if scale_factor is None else scale_factor, normalize_input=normalize_input,
                             latent_flag=latent_flag, predict_type=predict_type)
