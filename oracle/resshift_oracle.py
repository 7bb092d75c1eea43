"""ORACLE — test infrastructure only.  Never imported by the product (resshift_amd/).

A CPU, fp32, functional restatement of the reference's sampling hot path, driven directly by a
reference-format ``state_dict`` (plain dict name -> tensor) and the YAML ``params`` blocks.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.

Arithmetic: torch CPU fp32 operators — the very ATen kernels the reference's own CPU path runs
(every conv / GroupNorm / softmax / GELU / bicubic / argmin on this path is a stock
torch op; the reference ships no native code).  Each function cites the reference lines it restates.

Pinning: ``oracle/make_golden.py`` runs the *unmodified reference modules* (imported from
/root/reference in the build container) on seeded synthetic weights and inputs, checks this file
against them bit-for-bit / to fp32 round-off, and stores the outputs under ``tests/golden/``;
``tests/test_oracle.py`` re-checks the oracle against those fixtures wherever it runs.  The
reference itself has no tests or golden vectors for this path.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- small helpers
def _listify(v, n):
    return [int(v)] * n if isinstance(v, int) else [int(x) for x in v]


def _gn(sd: SD, name: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    # models/basic_ops.py:15-17,96 (GroupNorm32, 32 groups, fp32) / model.py:46-47 (eps 1e-6)
    return F.group_norm(x.float(), 32, sd[name + ".weight"], sd[name + ".bias"], eps)


def _conv(sd: SD, name: str, x: torch.Tensor, stride: int = 1, padding: int = 0) -> torch.Tensor:
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def _linear(sd: SD, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    # models/basic_ops.py:99-117 — cos first, then sin
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# ----------------------------------------------------------------------------- Swin pieces
def _rel_index(ws: int) -> torch.Tensor:
    # models/swin_transformer.py:93-102
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _shift_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    # models/swin_transformer.py:214-236 (called with the runtime size on every forward, quirk Q1)
    # NB the reference builds img_mask as (1,1,H,W) but indexes it `img_mask[:, h, w, :]`, i.e. its "h" slices hit
    # the singleton dim (only the last one is non-empty) and its "w" slices hit the ROW axis.  The effective region
    # id therefore depends on the row band only: rows [0,H-ws) -> 6, [H-ws,H-shift) -> 7, [H-shift,H) -> 8.
    img = torch.zeros(1, 1, H, W)
    for cnt, rows in zip((6, 7, 8), (slice(0, -ws), slice(-ws, -shift), slice(-shift, None))):
        img[:, :, rows, :] = cnt
    mw = _partition(img, ws).permute(0, 2, 3, 1).reshape(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def _partition(x: torch.Tensor, ws: int) -> torch.Tensor:
    # models/swin_transformer.py:35-47: [B,C,H,W] -> [B*nW, ws, ws, C]
    B, C, H, W = x.shape
    return x.view(B, C, H // ws, ws, W // ws, ws).permute(0, 2, 4, 3, 5, 1).reshape(-1, ws, ws, C)


def _reverse(w: torch.Tensor, ws: int, H: int, W: int) -> torch.Tensor:
    # models/swin_transformer.py:49-63
    B = w.shape[0] // ((H // ws) * (W // ws))
    return w.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 5, 1, 3, 2, 4).reshape(B, -1, H, W)


def swin_block(sd: SD, name: str, x: torch.Tensor, heads: int, ws: int, shift: int) -> torch.Tensor:
    # models/swin_transformer.py:238-281 + WindowAttention.forward :114-145
    B, C, H, W = x.shape
    shortcut = x
    h = _gn(sd, name + ".norm1", x, 1e-5)
    if shift > 0:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(2, 3))
    win = _partition(h, ws).reshape(-1, ws * ws, C)
    nB = win.shape[0]
    qkv = _linear(sd, name + ".attn.qkv", win).reshape(nB, ws * ws, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    table = sd[name + ".attn.relative_position_bias_table"]
    bias = table[_rel_index(ws).view(-1).to(table.device)].view(ws * ws, ws * ws, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift > 0:  # the mask of a non-shifted block is identically zero
        mask = _shift_mask(H, W, ws, shift).to(attn.device)
        nW = mask.shape[0]
        attn = (attn.view(nB // nW, nW, heads, ws * ws, ws * ws) + mask[None, :, None]).view(-1, heads, ws * ws, ws * ws)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(nB, ws * ws, C)
    o = _linear(sd, name + ".attn.proj", o)
    o = _reverse(o.view(-1, ws, ws, C), ws, H, W)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(2, 3))
    x = shortcut + o
    m = _conv(sd, name + ".mlp.fc1", _gn(sd, name + ".norm2", x, 1e-5))
    m = _conv(sd, name + ".mlp.fc2", F.gelu(m))  # exact-erf GELU (swin_transformer.py:18)
    return x + m


def basic_layer(sd: SD, name: str, x: torch.Tensor, p: dict, ds: int) -> torch.Tensor:
    # models/swin_transformer.py:427-442; shift_size fixed at construction (:189-194,416)
    ws = int(p.get("window_size", 8))
    nhc = int(p.get("num_head_channels", -1))
    heads = int(p.get("num_heads", 1)) if nhc == -1 else int(p["swin_embed_dim"]) // nhc
    h = _conv(sd, name + ".patch_embed.proj", x)
    for d in range(int(p.get("swin_depth", 2))):
        shift = ws // 2 if (d % 2 == 1 and ds > ws) else 0
        h = swin_block(sd, f"{name}.blocks.{d}", h, heads, ws, shift)
    return _conv(sd, name + ".patch_unembed.proj", h)


# ----------------------------------------------------------------------------- UNetModelSwin
def res_block(sd: SD, name: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    # models/unet.py:186-206 with use_scale_shift_norm=True, dropout 0
    h = _conv(sd, name + ".in_layers.2", F.silu(_gn(sd, name + ".in_layers.0", x, 1e-5)), padding=1)
    e = _linear(sd, name + ".emb_layers.1", F.silu(emb))[..., None, None]
    scale, shift = torch.chunk(e, 2, dim=1)
    h = _gn(sd, name + ".out_layers.0", h, 1e-5) * (1 + scale) + shift
    h = _conv(sd, name + ".out_layers.3", F.silu(h), padding=1)
    skip = _conv(sd, name + ".skip_connection", x) if (name + ".skip_connection.weight") in sd else x
    return skip + h


def unet_forward(sd: SD, p: dict, x: torch.Tensor, t: torch.Tensor, lq: Optional[torch.Tensor] = None,
                 mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """models/unet.py:865-895 UNetModelSwin.forward; block plan re-derived as in :704-863."""
    mc = int(p["model_channels"])
    mult = [int(m) for m in p.get("channel_mult", (1, 2, 4, 8))]
    nrb = _listify(p["num_res_blocks"], len(mult))
    attn_res = [int(a) for a in p["attention_resolutions"]]
    emb = _linear(sd, "time_embed.2", F.silu(_linear(sd, "time_embed.0", timestep_embedding(t, mc))))
    if lq is not None:
        if mask is not None:
            lq = torch.cat([lq, mask], dim=1)
        ii = 0
        while f"feature_extractor.{3 * ii}.weight" in sd:  # unet.py:693-702 (Identity when absent)
            lq = F.silu(_conv(sd, f"feature_extractor.{3 * ii}", lq, padding=1))
            lq = _conv(sd, f"feature_extractor.{3 * ii + 2}.op", lq, stride=2, padding=1)
            ii += 1
        x = torch.cat([x, lq], dim=1)
    hs: List[torch.Tensor] = []
    h = _conv(sd, "input_blocks.0.0", x, padding=1)
    hs.append(h)
    n = 1
    ds = int(p["image_size"])
    for level in range(len(mult)):
        for jj in range(nrb[level]):
            h = res_block(sd, f"input_blocks.{n}.0", h, emb)
            if ds in attn_res and jj == 0:
                h = basic_layer(sd, f"input_blocks.{n}.1", h, p, ds)
            hs.append(h)
            n += 1
        if level != len(mult) - 1:
            h = _conv(sd, f"input_blocks.{n}.0.op", h, stride=2, padding=1)  # unet.py:97-101
            hs.append(h)
            n += 1
            ds //= 2
    h = res_block(sd, "middle_block.0", h, emb)
    h = basic_layer(sd, "middle_block.1", h, p, ds)
    h = res_block(sd, "middle_block.2", h, emb)
    n = 0
    for level in reversed(range(len(mult))):
        for i in range(nrb[level] + 1):
            h = torch.cat([h, hs.pop()], dim=1)  # decoder features first (unet.py:891)
            sub = 0
            h = res_block(sd, f"output_blocks.{n}.{sub}", h, emb)
            sub += 1
            if ds in attn_res and i == 0:
                h = basic_layer(sd, f"output_blocks.{n}.{sub}", h, p, ds)
                sub += 1
            if level and i == nrb[level]:
                h = F.interpolate(h, scale_factor=2, mode="nearest")  # unet.py:78
                h = _conv(sd, f"output_blocks.{n}.{sub}.conv", h, padding=1)
                ds *= 2
            n += 1
    return _conv(sd, "out.2", F.silu(_gn(sd, "out.0", h, 1e-5)), padding=1)


# ----------------------------------------------------------------------------- VQModelTorch
def _resnet(sd: SD, name: str, x: torch.Tensor) -> torch.Tensor:
    # ldm/modules/diffusionmodules/model.py:129-149 (temb None)
    h = _conv(sd, name + ".conv1", F.silu(_gn(sd, name + ".norm1", x, 1e-6)), padding=1)
    h = _conv(sd, name + ".conv2", F.silu(_gn(sd, name + ".norm2", h, 1e-6)), padding=1)
    if (name + ".nin_shortcut.weight") in sd:
        x = _conv(sd, name + ".nin_shortcut", x)
    return x + h


def _attn_block(sd: SD, name: str, x: torch.Tensor) -> torch.Tensor:
    # ldm/modules/diffusionmodules/model.py:179-203
    B, C, H, W = x.shape
    n = _gn(sd, name + ".norm", x, 1e-6)
    q = _conv(sd, name + ".q", n).reshape(B, C, H * W).permute(0, 2, 1)
    k = _conv(sd, name + ".k", n).reshape(B, C, H * W)
    v = _conv(sd, name + ".v", n).reshape(B, C, H * W)
    w = torch.bmm(q, k) * (int(C) ** -0.5)
    w = F.softmax(w, dim=2)
    o = torch.bmm(v, w.permute(0, 2, 1)).reshape(B, C, H, W)
    return x + _conv(sd, name + ".proj_out", o)


def vq_encode(sd: SD, p: dict, x: torch.Tensor) -> torch.Tensor:
    """ldm/models/autoencoder.py:28-31 + Encoder.forward model.py:522-547."""
    dd = p["ddconfig"]
    mult = [int(m) for m in dd["ch_mult"]]
    nrb = _listify(dd["num_res_blocks"], len(mult))
    h = _conv(sd, "encoder.conv_in", x, padding=1)
    for l in range(len(mult)):
        for i in range(nrb[l]):
            h = _resnet(sd, f"encoder.down.{l}.block.{i}", h)
        if l != len(mult) - 1:
            h = _conv(sd, f"encoder.down.{l}.downsample.conv", F.pad(h, (0, 1, 0, 1)), stride=2)  # model.py:80-84
    h = _resnet(sd, "encoder.mid.block_1", h)
    h = _attn_block(sd, "encoder.mid.attn_1", h)
    h = _resnet(sd, "encoder.mid.block_2", h)
    h = _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.norm_out", h, 1e-6)), padding=1)
    return _conv(sd, "quant_conv", h)


def vq_quantize(sd: SD, z: torch.Tensor):
    """ldm/modules/vqvae/quantize.py:271-312 — returns (z_q [B,C,H,W], indices [B*H*W])."""
    e = sd["quantize.embedding.weight"]
    zp = z.permute(0, 2, 3, 1).contiguous()
    zf = zp.view(-1, e.shape[1])
    d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(e ** 2, dim=1) - 2 * torch.einsum("bd,dn->bn", zf, e.t())
    idx = torch.argmin(d, dim=1)
    zq = e[idx].view(zp.shape)
    zq = zp + (zq - zp)  # straight-through expression kept for bit parity (quantize.py:298)
    return zq.permute(0, 3, 1, 2).contiguous(), idx


def vq_decode(sd: SD, p: dict, h: torch.Tensor, force_not_quantize: bool = False, return_indices: bool = False):
    """ldm/models/autoencoder.py:33-40 + Decoder.forward model.py:627-660."""
    dd = p["ddconfig"]
    mult = [int(m) for m in dd["ch_mult"]]
    nrb = _listify(dd["num_res_blocks"], len(mult))
    idx = None
    if not force_not_quantize:
        h, idx = vq_quantize(sd, h)
    h = _conv(sd, "post_quant_conv", h)
    h = _conv(sd, "decoder.conv_in", h, padding=1)
    h = _resnet(sd, "decoder.mid.block_1", h)
    h = _attn_block(sd, "decoder.mid.attn_1", h)
    h = _resnet(sd, "decoder.mid.block_2", h)
    for l in reversed(range(len(mult))):
        for i in range(nrb[l] + 1):
            h = _resnet(sd, f"decoder.up.{l}.block.{i}", h)
        if l != 0:
            h = _conv(sd, f"decoder.up.{l}.upsample.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"), padding=1)
    out = _conv(sd, "decoder.conv_out", F.silu(_gn(sd, "decoder.norm_out", h, 1e-6)), padding=1)
    return (out, idx) if return_indices else out


# ----------------------------------------------------------------------------- diffusion
def eta_schedule(steps: int, min_noise_level: float, etas_end: float, kappa: float, power: float) -> np.ndarray:
    # models/gaussian_diffusion.py:45-58 ('exponential' schedule), float64
    etas_start = min(min_noise_level / kappa, min_noise_level)
    increaser = math.exp(1 / (steps - 1) * math.log(etas_end / etas_start))
    base = np.ones([steps]) * increaser
    power_timestep = np.linspace(0, 1, steps, endpoint=True) ** power
    power_timestep *= steps - 1
    return np.power(base, power_timestep) * etas_start


class Schedule:
    """Scalar tables of GaussianDiffusion.__init__ (gaussian_diffusion.py:123-174) + SpacedDiffusion (respace.py:31-47)."""

    def __init__(self, dp: dict):
        steps = int(dp["steps"])
        self.kappa = float(dp["kappa"])
        self.sf = int(dp.get("sf", 4))
        self.scale_factor = float(dp["scale_factor"]) if dp.get("scale_factor") is not None else 1.0
        sqrt_etas = eta_schedule(steps, float(dp["min_noise_level"]), float(dp.get("etas_end", 0.99)), self.kappa,
                                 float(dp["schedule_kwargs"]["power"]))
        respacing = dp.get("timestep_respacing") or steps
        use = set(int((steps / respacing) * x) for x in range(respacing))
        self.timestep_map = [i for i in range(steps) if i in use]
        self.sqrt_etas = np.array([sqrt_etas[i] for i in self.timestep_map])
        self.etas = self.sqrt_etas ** 2
        self.num_timesteps = len(self.etas)
        self.etas_prev = np.append(0.0, self.etas[:-1])
        self.alpha = self.etas - self.etas_prev
        self.posterior_variance = self.kappa ** 2 * self.etas_prev / self.etas * self.alpha
        self.posterior_variance_clipped = np.append(self.posterior_variance[1], self.posterior_variance[1:])
        self.posterior_log_variance_clipped = np.log(self.posterior_variance_clipped)
        self.posterior_mean_coef1 = self.etas_prev / self.etas
        self.posterior_mean_coef2 = self.alpha / self.etas
        self.normalize_input = bool(dp.get("normalize_input", True))
        self.latent_flag = bool(dp.get("latent_flag", True))


def _f32(a: np.ndarray, i: int) -> torch.Tensor:
    # _extract_into_tensor (gaussian_diffusion.py:92-105): float64 table -> float32 scalar
    return torch.from_numpy(a)[i].float()


def sample_loop(unet_sd: SD, unet_p: dict, ae_sd: SD, ae_p: dict, dp: dict, y: torch.Tensor, noises: Sequence[torch.Tensor],
                mask: Optional[torch.Tensor] = None, return_aux: bool = False):
    """gaussian_diffusion.py:367-472 (p_sample_loop) with the RNG draws replaced by explicit tensors.

    noises[0] is the prior noise (:446), noises[k] (k>=1) the draw of the k-th loop iteration (:358),
    i.e. timestep t = T-k.  Returns the decoded, un-clamped image (clamp is sampler.py:165).
    """
    s = Schedule(dp)
    with torch.no_grad():
        up = F.interpolate(y, scale_factor=s.sf, mode="bicubic") if s.sf != 1 else y  # :503-504
        z_y = vq_encode(ae_sd, ae_p, up) * s.scale_factor
        T = s.num_timesteps
        x = z_y + _f32(s.kappa * s.sqrt_etas, T - 1) * noises[0]  # prior_sample :517-529
        kwargs = {"lq": y}
        if mask is not None:
            kwargs["mask"] = mask
        for k, i in enumerate(reversed(range(T)), start=1):
            if s.normalize_input and s.latent_flag:
                xin = x / torch.sqrt(_f32(s.etas, i) * s.kappa ** 2 + 1)  # _scale_input :598-609
            elif s.normalize_input:
                xin = x / (_f32(s.sqrt_etas, i) * s.kappa * 3 + 1)
            else:
                xin = x
            t = torch.tensor([s.timestep_map[i]] * y.shape[0], device=y.device)
            pred = unet_forward(unet_sd, unet_p, xin, t, **kwargs)  # START_X, no clipping (:278, sampler.py:156)
            mean = _f32(s.posterior_mean_coef1, i) * x + _f32(s.posterior_mean_coef2, i) * pred  # :218-221
            nonzero = 1.0 if i != 0 else 0.0
            x = mean + nonzero * torch.exp(0.5 * _f32(s.posterior_log_variance_clipped, i)) * noises[k]  # :361-364
        z_final = x
        out, idx = vq_decode(ae_sd, ae_p, (1 / s.scale_factor) * z_final, return_indices=True)  # :490-492
    if return_aux:
        return out, {"z_y": z_y, "z_final": z_final, "indices": idx}
    return out


def sample_func(unet_sd, unet_p, ae_sd, ae_p, dp, y0, noises, mask=None, padding_offset=64):
    """sampler.py:119-165: reflect-pad to a multiple of padding_offset, run the loop, crop, clamp."""
    sf = int(dp.get("sf", 4))
    H, W = y0.shape[2:]
    ph = (math.ceil(H / padding_offset)) * padding_offset - H
    pw = (math.ceil(W / padding_offset)) * padding_offset - W
    if ph or pw:
        y0 = F.pad(y0, (0, pw, 0, ph), mode="reflect")
    out = sample_loop(unet_sd, unet_p, ae_sd, ae_p, dp, y0, noises, mask=mask)
    if ph or pw:
        out = out[:, :, : H * sf, : W * sf]
    return out.clamp_(-1.0, 1.0)


# ----------------------------------------------------------------------------- tiled large-image path
def tile_starts(length: int, pch_size: int, stride: int) -> List[int]:
    """utils/util_image.py:922-931 (ImageSpliterTh.extract_starts)."""
    if length <= pch_size:
        return [0]
    starts = list(range(0, length, stride))
    for ii in range(len(starts)):
        if starts[ii] + pch_size > length:
            starts[ii] = length - pch_size
    return sorted(set(starts), key=starts.index)


def sample_tiled(unet_sd, unet_p, ae_sd, ae_p, dp, im_lq, tile_noises, mask=None, chop_size=64, chop_stride=48, chop_bs=1,
                 padding_offset=64):
    """sampler.py:176-216 + utils/util_image.py:889-979: overlapping tiles, `chop_bs` tiles per sample_func call,
    sum / count averaging.  tile_noises[k] = list of noise tensors for the k-th sample_func call."""
    sf = int(dp.get("sf", 4))
    B, _, H, W = im_lq.shape
    if not (H > chop_size or W > chop_size):
        return sample_func(unet_sd, unet_p, ae_sd, ae_p, dp, im_lq, tile_noises[0], mask=mask, padding_offset=padding_offset)
    x = torch.cat([im_lq, mask], dim=1) if mask is not None else im_lq
    starts = [(i, j) for i in tile_starts(H, chop_size, chop_stride) for j in tile_starts(W, chop_size, chop_stride)]
    res = count = None
    for k, c0 in enumerate(range(0, len(starts), chop_bs)):
        cur = starts[c0:c0 + chop_bs]
        pch = torch.cat([x[:, :, h0:h0 + chop_size, w0:w0 + chop_size] for h0, w0 in cur], dim=0)
        m = None
        if mask is not None:
            pch, m = pch[:, :-1], pch[:, -1:]
        out = sample_func(unet_sd, unet_p, ae_sd, ae_p, dp, pch, tile_noises[k], mask=m, padding_offset=padding_offset)
        if res is None:
            res = torch.zeros(B, out.shape[1], H * sf, W * sf)
            count = torch.zeros_like(res)
        for t, (h0, w0) in enumerate(cur):
            res[:, :, h0 * sf:(h0 + chop_size) * sf, w0 * sf:(w0 + chop_size) * sf] += out[t * B:(t + 1) * B]
            count[:, :, h0 * sf:(h0 + chop_size) * sf, w0 * sf:(w0 + chop_size) * sf] += 1
    assert torch.all(count != 0)
    return res / count
