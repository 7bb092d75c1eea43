#!/bin/bash
# Round 5: the sub-pixel form of the upsampling convs (engine.hip add_upfold) - correctness tests, then an A/B on one box: RS_UPFOLD=0 (the folded-address
# 3x3 conv) against the default (low-resolution M >= 16384: the 32 -> 64 UNet step and both decoder steps).  (The record in profiles/r5_upfold_ab.txt also
# has a leg with every UNet step in the new form - an A/B knob that was removed with the result: slower.)
#   gpurun --timeout 900 -- bash scripts/gpu/r5_upfold.sh
R=$(pwd); O=$R/gpurun_out/r5u2; mkdir -p $O; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -s -k "subpixel or unet_forward_vs_oracle or groupnorm_tails" > $O/pytest_eng.log 2>&1; echo "engine tests rc=$?"; grep -E "sub-pixel|passed|failed|Error" $O/pytest_eng.log | tail -6
B="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline"
for v in on off on; do
  e="RS_UPFOLD=1"; [ $v = off ] && e="RS_UPFOLD=0"
  env $e timeout 200 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench upfold=$v rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json")); r=d["roofline"]
print("$v", d["ms_per_step"], d["ms_per_unet_step"], d["config"].get("kernel_launches_per_step"), [(k["kernel"][:22], k["ms_per_step"], k["launches_per_step"]) for k in r["per_kernel"]], r["groupnorm"].get("ms_per_step"))
PY
done
timeout 400 python bench.py --steps 5 --warmup 2 --parity-images 8 --cpu-seconds 80 --no-torch-baseline --no-secondary > $O/bench_parity8.json 2> $O/bench_parity8.err; echo "parity bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_parity8.json')); print(d['value'], d['ms_per_step'], d['parity_vs_cpu_oracle'][0], d['cpu_baseline']['kind'], d['cpu_baseline']['value'])"
