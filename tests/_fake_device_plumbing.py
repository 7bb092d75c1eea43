"""Helper of test_dry_and_real_pass_agree_without_a_gpu (tests/test_host_cpu.py): one rs_sample call with RS_FAKE_DEVICE=1 - the engine's real
pass walks its whole control flow on a host-memory arena while every launch simply fails (there is no GPU), and prints what the dry
sizing pass and the real pass each counted: coefficient-pool bytes, GroupNorm-tail tickets, producer / GroupNorm sequence numbers,
kernel launches.  Usage: _fake_device_plumbing.py <config yaml name> <batch> <precision 0 fp16 | 1 fp32 | 2 split>"""
import os, sys, ctypes as C
os.environ["RS_FAKE_DEVICE"]="1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from resshift_amd import _lib
from resshift_amd.engine import _fill_unet, _fill_ae
from resshift_amd.config import load_config, to_plain
from resshift_amd.gaussian_diffusion import create_gaussian_diffusion
lib=_lib.load()
cname=sys.argv[1] if len(sys.argv)>1 else "realsr_swinunet_realesrgan256"
B=int(sys.argv[2]) if len(sys.argv)>2 else 32
prec=int(sys.argv[3]) if len(sys.argv)>3 else 2
cfgy=to_plain(load_config(cname))
up, aep, dp = cfgy["model"]["params"], cfgy["autoencoder"]["params"], cfgy["diffusion"]["params"]
cfg=_lib.Config(); _fill_unet(cfg.unet, up); cfg.has_unet=1; _fill_ae(cfg.ae, aep); cfg.has_ae=1
cfg.enable_f16=cfg.enable_f32=cfg.enable_split=1
h=lib.rs_create(C.byref(cfg)); assert h
n=lib.rs_weight_bytes(h)
lib.rs_bind_weight_blob(h, 256*1024, n)   # fake, aligned address: never dereferenced on the host
assert lib.rs_weights_ready(h)==0
d=create_gaussian_diffusion(**dp); tables=d.step_tables(); steps=len(tables["coef1"])
a=_lib.SampleArgs()
a.y=a.noise=a.out=4096; a.mask=4096 if up.get("cond_mask") else None
lr={"realsr_swinunet_realesrgan256":64,"faceir_gfpgan512_lpips":512,"inpaint_lama256_imagenet":256}.get(cname,64)
a.B,a.h,a.w,a.sf,a.steps=B,lr,lr,int(d.sf),steps
for t in range(steps):
    a.inv_std[t]=float(tables["inv_std"][t]); a.coef1[t]=float(tables["coef1"][t]); a.coef2[t]=float(tables["coef2"][t]); a.sigma[t]=float(tables["sigma"][t]); a.tmap[t]=int(tables["tmap"][t]); a.prec_unet[t]=prec
a.prior_scale=float(tables["prior_scale"]); a.scale_factor=float(d.scale_factor); a.prec_encode=prec; a.prec_decode=0
rc=lib.rs_sample(h, C.byref(a))
print("rc",rc,_lib.last_error(), "launches", lib.rs_last_launch_count(h))
