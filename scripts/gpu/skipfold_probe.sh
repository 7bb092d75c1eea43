#!/bin/bash
# (Record of the round-4 probe; the patch is in the tree since 60692e8.)  Probe of the shortcut fold (a ResBlock's 1x1 shortcut as extra K stages of its second 3x3 conv): the patched
# library is built OUTSIDE the product path (scripts/probe/libresshift_skipfold.so, RESSHIFT_HIP_LIB) - the in-tree sources and their
# digest-stamped profiles stay as committed.  (1) one UNet forward, fold on vs off; (2) A/B of the parity pass; (3) parity vs the CPU oracle.
R=$(pwd); O=$R/gpurun_out/r4sf; mkdir -p $O; export TMPDIR=/tmp
export RESSHIFT_HIP_LIB=$R/scripts/probe/libresshift_skipfold.so
for f in 1 0; do RS_TEST_PREC=split RS_TEST_META=1 RS_SKIP_FOLD=$f timeout 300 python tests/proc_unet_once.py $O/unet_$f.pt > $O/unet_$f.log 2>&1; echo "unet fold=$f rc=$?"; done
python - <<PY
import torch
a=torch.load("$O/unet_1.pt"); b=torch.load("$O/unet_0.pt")
d=(a["out"].double()-b["out"].double()).abs().max().item(); s=b["out"].double().abs().max().item()
print("unet split B=4: fold on vs off max|diff| %.3e (max|out| %.3e, rel %.3e); second call identical: %s; launches %d vs %d" % (d, s, d/s, bool((a["out"]==a["out2"]).all()), a["launches"], b["launches"]))
PY
B="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline"
ab() { RS_SKIP_FOLD=$1 timeout 200 python bench.py $B > $O/ab_$1_$2.json 2> $O/ab_$1_$2.err; echo "ab fold=$1 run $2 rc=$?"; python - <<PY
import json
d=json.load(open("$O/ab_$1_$2.json")); r=d["roofline"]
print("fold=$1", d["ms_per_step"], d["config"].get("kernel_launches_per_step"), [(k["kernel"][:22], k["ms_per_step"], k["launches_per_step"]) for k in r["per_kernel"]], r["groupnorm"].get("ms_per_step"))
PY
}
ab 0 1; ab 1 1
timeout 400 python bench.py --steps 5 --warmup 2 --parity-images 8 --cpu-seconds 60 --no-secondary --no-torch-baseline > $O/bench_fold_parity.json 2> $O/bench_fold_parity.err; echo "parity bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_fold_parity.json')); print(d['value'], d['ms_per_step'], d.get('parity_vs_cpu_oracle'))"
ab 0 2; ab 1 2
