"""GPU-side bisect helper: engine debug trace vs oracle intermediates for a tiny UNet (a checker tool, not a collected test;
it lives under tests/ because only test infrastructure may import oracle/).    python tests/debug_unet.py [case] [fp32|fp16]"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H
from oracle import resshift_oracle as oc
from resshift_amd import UNetModelSwin

torch.set_grad_enabled(False)
tag = sys.argv[1] if len(sys.argv) > 1 else "tiny"
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
up, ap, dp, with_mask = H.CASES[tag]
usd, asd = H.weights(up, ap)
dev = torch.device("cuda:0")
um = UNetModelSwin(**up).to(dev); um.load_state_dict(usd, strict=True)
y, noises, mask = H.case_inputs(up, ap, dp, with_mask)
x, t = noises[1] * 1.3, torch.tensor([2, 2])
kw = {"lq": y}
if with_mask: kw["mask"] = mask
eng = um.engine(); eng.debug_enable(True)
got = um(x.to(dev), t, prec=prec, **{k: v.to(dev) for k, v in kw.items()})
tr = eng.debug_trace()
# oracle intermediates
sd, p = usd, up
ref = {}
mc = p["model_channels"]; mult = p["channel_mult"]; nrb = p["num_res_blocks"]; ar = p["attention_resolutions"]
emb = oc._linear(sd, "time_embed.2", F.silu(oc._linear(sd, "time_embed.0", oc.timestep_embedding(t, mc))))
lq = y
if mask is not None: lq = torch.cat([lq, mask], 1)
ii = 0
while f"feature_extractor.{3*ii}.weight" in sd:
    lq = F.silu(oc._conv(sd, f"feature_extractor.{3*ii}", lq, padding=1)); lq = oc._conv(sd, f"feature_extractor.{3*ii+2}.op", lq, stride=2, padding=1); ii += 1
h = oc._conv(sd, "input_blocks.0.0", torch.cat([x, lq], 1), padding=1); ref["in.0"] = h
hs = [h]; n = 1; ds = p["image_size"]
def res_parts(name, xx, pre):
    g1 = F.silu(oc._gn(sd, name + ".in_layers.0", xx, 1e-5)); ref[pre + "gn1"] = g1
    c1 = oc._conv(sd, name + ".in_layers.2", g1, padding=1); ref[pre + "conv1"] = c1
    e = oc._linear(sd, name + ".emb_layers.1", F.silu(emb))[..., None, None]; sc, sh = torch.chunk(e, 2, 1)
    ref[pre + "gn2film"] = F.silu(oc._gn(sd, name + ".out_layers.0", c1, 1e-5) * (1 + sc) + sh)
for level in range(len(mult)):
    for jj in range(nrb[level]):
        res_parts(f"input_blocks.{n}.0", h, f"in.{n}.res.")
        h = oc.res_block(sd, f"input_blocks.{n}.0", h, emb)
        if ds in ar and jj == 0:
            ref[f"in.{n}.res"] = h
            ref[f"in.{n}.swin.embed"] = oc._conv(sd, f"input_blocks.{n}.1.patch_embed.proj", h)
            e0 = ref[f"in.{n}.swin.embed"]
            ref[f"in.{n}.swin.blk0.out"] = oc.swin_block(sd, f"input_blocks.{n}.1.blocks.0", e0, p["swin_embed_dim"] // 32, 8, 0)
            h = oc.basic_layer(sd, f"input_blocks.{n}.1", h, p, ds)
        ref[f"in.{n}"] = h; hs.append(h); n += 1
    if level != len(mult) - 1:
        h = oc._conv(sd, f"input_blocks.{n}.0.op", h, stride=2, padding=1); ref[f"in.{n}"] = h; hs.append(h); n += 1; ds //= 2
h = oc.res_block(sd, "middle_block.0", h, emb); ref["mid.res1"] = h
h = oc.basic_layer(sd, "middle_block.1", h, p, ds); ref["mid.swin"] = h
h = oc.res_block(sd, "middle_block.2", h, emb); ref["mid.res2"] = h
n = 0
for level in reversed(range(len(mult))):
    for i in range(nrb[level] + 1):
        h = torch.cat([h, hs.pop()], 1); sub = 0
        h = oc.res_block(sd, f"output_blocks.{n}.{sub}", h, emb); sub += 1
        if ds in ar and i == 0: h = oc.basic_layer(sd, f"output_blocks.{n}.{sub}", h, p, ds); sub += 1
        if level and i == nrb[level]:
            h = oc._conv(sd, f"output_blocks.{n}.{sub}.conv", F.interpolate(h, scale_factor=2, mode="nearest"), padding=1); ds *= 2
        ref[f"out.{n}"] = h; n += 1
final = oc._conv(sd, "out.2", F.silu(oc._gn(sd, "out.0", h, 1e-5)), padding=1)
for name, tns in tr.items():
    if name in ref:
        r = ref[name]
        if tuple(r.shape) != tuple(tns.shape):
            print(f"{name:28s} SHAPE {tuple(tns.shape)} vs {tuple(r.shape)}"); continue
        print(f"{name:28s} rel err {H.rel_err(tns, r):.3e}")
    else:
        print(f"{name:28s} (no oracle counterpart) absmax {tns.abs().max().item():.3e}")
print("final rel err", H.rel_err(got, final))
