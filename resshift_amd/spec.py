"""state_dict layouts (key -> shape) of the two networks on the hot path.

The engine is drop-in at the checkpoint level: `utils/util_net.py:86-98` (reload_model) walks
`model.state_dict()` and `copy_`s the checkpoint tensor of the same name into each entry, so the
host-side shells must expose exactly the reference's key names and shapes.  The functions below
derive them from the YAML `params` blocks, following the construction order of
`models/unet.py:658-863` (UNetModelSwin), `models/swin_transformer.py:368-425,79-112,197-212`
(BasicLayer / WindowAttention / SwinTransformerBlock) and
`ldm/models/autoencoder.py:20-26` + `ldm/modules/diffusionmodules/model.py:452-521,550-626`.

Entries whose dtype is not float32 (the two Swin index/mask buffers) are listed in BUFFERS.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Tuple

Spec = "OrderedDict[str, Tuple[int, ...]]"


def _listify(v, n):
    return [int(v)] * n if isinstance(v, int) else [int(x) for x in v]


def unet_heads(p: dict) -> int:
    nhc = int(p.get("num_head_channels", -1))
    return int(p.get("num_heads", 1)) if nhc == -1 else int(p["swin_embed_dim"]) // nhc


def _conv(spec, name, cin, cout, k):
    spec[name + ".weight"] = (cout, cin, k, k)
    spec[name + ".bias"] = (cout,)


def _linear(spec, name, cin, cout):
    spec[name + ".weight"] = (cout, cin)
    spec[name + ".bias"] = (cout,)


def _norm(spec, name, c):
    spec[name + ".weight"] = (c,)
    spec[name + ".bias"] = (c,)


def _resblock(spec, name, cin, cout, emb):
    _norm(spec, name + ".in_layers.0", cin)
    _conv(spec, name + ".in_layers.2", cin, cout, 3)
    _linear(spec, name + ".emb_layers.1", emb, 2 * cout)
    _norm(spec, name + ".out_layers.0", cout)
    _conv(spec, name + ".out_layers.3", cout, cout, 3)
    if cin != cout:
        _conv(spec, name + ".skip_connection", cin, cout, 1)


def swin_shift(block_index: int, ds: int, window: int) -> int:
    """shift_size of block `block_index` built for a ds x ds map (swin_transformer.py:189-194,416)."""
    return window // 2 if (block_index % 2 == 1 and ds > window) else 0


def _basic_layer(spec, buffers, name, c, p, ds):
    e = int(p["swin_embed_dim"])
    heads = unet_heads(p)
    ws = int(p.get("window_size", 8))
    hidden = int(e * float(p.get("mlp_ratio", 2.0)))
    _conv(spec, name + ".patch_embed.proj", c, e, 1)
    _conv(spec, name + ".patch_unembed.proj", e, c, 1)  # registered before the blocks (swin_transformer.py:392-407)
    for d in range(int(p.get("swin_depth", 2))):
        b = f"{name}.blocks.{d}"
        if swin_shift(d, ds, ws) > 0:
            spec[b + ".attn_mask"] = ((ds // ws) ** 2, ws * ws, ws * ws)
        _norm(spec, b + ".norm1", e)
        spec[b + ".attn.relative_position_bias_table"] = ((2 * ws - 1) ** 2, heads)
        spec[b + ".attn.relative_position_index"] = (ws * ws, ws * ws)
        buffers.add(b + ".attn.relative_position_index")
        _linear(spec, b + ".attn.qkv", e, 3 * e)
        _linear(spec, b + ".attn.proj", e, e)
        _norm(spec, b + ".norm2", e)
        _conv(spec, b + ".mlp.fc1", e, hidden, 1)
        _conv(spec, b + ".mlp.fc2", hidden, e, 1)


def unet_param_spec(p: dict):
    """(spec, buffers) for `models.unet.UNetModelSwin(**p)`; buffers = names that are not float parameters."""
    spec: Dict[str, Tuple[int, ...]] = OrderedDict()
    buffers = set()
    mc = int(p["model_channels"])
    mult = [int(m) for m in p.get("channel_mult", (1, 2, 4, 8))]
    nrb = _listify(p["num_res_blocks"], len(mult))
    attn_res = [int(a) for a in p["attention_resolutions"]]
    emb = 4 * mc
    image_size, lq_size = int(p["image_size"]), int(p.get("lq_size", 256))
    cond_lq, cond_mask = bool(p.get("cond_lq", True)), bool(p.get("cond_mask", False))
    _linear(spec, "time_embed.0", mc, emb)
    _linear(spec, "time_embed.2", emb, emb)
    if cond_lq and lq_size == image_size:
        base = 4 if cond_mask else 3
    else:
        feat = 4 if cond_mask else 3
        base = 16
        for ii in range(int(math.log(lq_size / image_size) / math.log(2))):
            _conv(spec, f"feature_extractor.{3 * ii}", feat, base, 3)
            _conv(spec, f"feature_extractor.{3 * ii + 2}.op", base, base * 2, 3)
            base *= 2
            feat = base
    ch = input_ch = mult[0] * mc
    _conv(spec, "input_blocks.0.0", int(p["in_channels"]) + base, ch, 3)
    chans = [ch]
    n = 1
    ds = image_size
    for level, m in enumerate(mult):
        for jj in range(nrb[level]):
            _resblock(spec, f"input_blocks.{n}.0", ch, m * mc, emb)
            ch = m * mc
            if ds in attn_res and jj == 0:
                _basic_layer(spec, buffers, f"input_blocks.{n}.1", ch, p, ds)
            chans.append(ch)
            n += 1
        if level != len(mult) - 1:
            _conv(spec, f"input_blocks.{n}.0.op", ch, ch, 3)
            chans.append(ch)
            n += 1
            ds //= 2
    _resblock(spec, "middle_block.0", ch, ch, emb)
    _basic_layer(spec, buffers, "middle_block.1", ch, p, ds)
    _resblock(spec, "middle_block.2", ch, ch, emb)
    n = 0
    for level in reversed(range(len(mult))):
        m = mult[level]
        for i in range(nrb[level] + 1):
            ich = chans.pop()
            sub = 0
            _resblock(spec, f"output_blocks.{n}.{sub}", ch + ich, mc * m, emb)
            sub += 1
            ch = mc * m
            if ds in attn_res and i == 0:
                _basic_layer(spec, buffers, f"output_blocks.{n}.{sub}", ch, p, ds)
                sub += 1
            if level and i == nrb[level]:
                _conv(spec, f"output_blocks.{n}.{sub}.conv", ch, ch, 3)
                ds *= 2
            n += 1
    _norm(spec, "out.0", ch)
    _conv(spec, "out.2", input_ch, int(p["out_channels"]), 3)
    for k in spec:
        if k.endswith(".attn_mask"):
            buffers.add(k)
    return spec, buffers


def _resnet(spec, name, cin, cout):
    _norm(spec, name + ".norm1", cin)
    _conv(spec, name + ".conv1", cin, cout, 3)
    _norm(spec, name + ".norm2", cout)
    _conv(spec, name + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(spec, name + ".nin_shortcut", cin, cout, 1)


def _attn(spec, name, c):
    _norm(spec, name + ".norm", c)
    for s in ("q", "k", "v", "proj_out"):
        _conv(spec, f"{name}.{s}", c, c, 1)


def ae_param_spec(p: dict):
    """spec for `ldm.models.autoencoder.VQModelTorch(**p)` (p = autoencoder.params)."""
    dd = p["ddconfig"]
    if list(dd.get("attn_resolutions", [])):
        raise NotImplementedError("autoencoder attn_resolutions must be empty (true for every shipped config)")
    spec: Dict[str, Tuple[int, ...]] = OrderedDict()
    ch, mult = int(dd["ch"]), [int(m) for m in dd["ch_mult"]]
    nrb = _listify(dd["num_res_blocks"], len(mult))
    zc, embed, n_embed = int(dd["z_channels"]), int(p["embed_dim"]), int(p["n_embed"])
    nl = len(mult)
    # encoder
    _conv(spec, "encoder.conv_in", int(dd["in_channels"]), ch, 3)
    block_in = ch
    for l in range(nl):
        block_in = ch * ((1,) + tuple(mult))[l]
        block_out = ch * mult[l]
        for i in range(nrb[l]):
            _resnet(spec, f"encoder.down.{l}.block.{i}", block_in, block_out)
            block_in = block_out
        if l != nl - 1:
            _conv(spec, f"encoder.down.{l}.downsample.conv", block_in, block_in, 3)
    _resnet(spec, "encoder.mid.block_1", block_in, block_in)
    _attn(spec, "encoder.mid.attn_1", block_in)
    _resnet(spec, "encoder.mid.block_2", block_in, block_in)
    _norm(spec, "encoder.norm_out", block_in)
    _conv(spec, "encoder.conv_out", block_in, 2 * zc if dd.get("double_z", True) else zc, 3)
    # decoder
    block_in = ch * mult[-1]
    _conv(spec, "decoder.conv_in", zc, block_in, 3)
    _resnet(spec, "decoder.mid.block_1", block_in, block_in)
    _attn(spec, "decoder.mid.attn_1", block_in)
    _resnet(spec, "decoder.mid.block_2", block_in, block_in)
    ups: List["OrderedDict"] = [OrderedDict() for _ in range(nl)]
    for l in reversed(range(nl)):
        block_out = ch * mult[l]
        for i in range(nrb[l] + 1):
            _resnet(ups[l], f"decoder.up.{l}.block.{i}", block_in, block_out)
            block_in = block_out
        if l != 0:
            _conv(ups[l], f"decoder.up.{l}.upsample.conv", block_in, block_in, 3)
    for l in range(nl):  # `self.up.insert(0, up)` -> state_dict lists level 0 first
        spec.update(ups[l])
    _norm(spec, "decoder.norm_out", block_in)
    _conv(spec, "decoder.conv_out", block_in, int(dd["out_ch"]), 3)
    spec["quantize.embedding.weight"] = (n_embed, embed)
    _conv(spec, "quant_conv", zc, embed, 1)
    _conv(spec, "post_quant_conv", embed, zc, 1)
    return spec


def relative_position_index(ws: int = 8):
    """int64 [ws*ws, ws*ws] table of swin_transformer.py:93-102, recomputed (it is a deterministic buffer)."""
    import torch

    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_attn_mask(H: int, W: int, ws: int, shift: int):
    """[nW, ws*ws, ws*ws] 0/-100 mask of swin_transformer.py:214-236."""
    import torch

    # The reference indexes its (1,1,H,W) mask image as `[:, h, w, :]`, so only row bands are ever labelled
    # (rows [0,H-ws), [H-ws,H-shift), [H-shift,H)); columns do not matter.  Reproduced as is.
    img = torch.zeros(1, 1, H, W)
    for cnt, rows in zip((6, 7, 8), (slice(0, -ws), slice(-ws, -shift), slice(-shift, None))):
        img[:, :, rows, :] = cnt
    # ... and the partitioned windows [nW,ws,ws,1] are permuted (0,2,3,1) once more before flattening (:230),
    # which transposes the token order inside each window.  Also reproduced as is.
    mw = img.view(1, 1, H // ws, ws, W // ws, ws).permute(0, 2, 4, 3, 5, 1).reshape(-1, ws, ws, 1)
    mw = mw.permute(0, 2, 3, 1).reshape(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def random_state_dict(spec, seed: int = 0, buffers=(), window: int = 8):
    """Random-init weights of the right shapes (fan-in scaled normals; every tensor non-zero, unlike the reference's
    zero-initialised ResBlock tails) for benchmarking and smoke runs when no checkpoint is available."""
    import torch

    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in spec.items():
        if name.endswith("relative_position_index"):
            sd[name] = relative_position_index(window)
        elif name.endswith("attn_mask"):
            side = int(round(shape[0] ** 0.5)) * window
            sd[name] = shift_attn_mask(side, side, window, window // 2)
        elif name.endswith("relative_position_bias_table"):
            sd[name] = torch.randn(shape, generator=g) * 0.5
        elif name == "quantize.embedding.weight":
            sd[name] = torch.randn(shape, generator=g) * 0.6
        elif len(shape) == 1:
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g) if name.endswith(".weight") else 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            scale = fan_in ** -0.5
            if ".emb_layers." in name:
                scale *= 0.5
            if name == "decoder.conv_out.weight":
                scale *= 0.3
            sd[name] = torch.randn(shape, generator=g) * scale
    return sd
