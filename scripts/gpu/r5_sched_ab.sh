#!/bin/bash
# Round 5: A/B of the halo kernel's K-loop refill (igemm4_kernel.h; libs built with RS_BUILD_DEFS="RS_IG4_FAST=f RS_IG4_SCHED=s" RS_BUILD_OUT=ablib/f<f>s<s>.so):
#   FAST  1 = precomputed per-lane offsets + scalar chunk / tap offset, 0 = round 4's address arithmetic in every stage
#   SCHED 0 = refill right behind the barrier (round 4), 1 = waves 0-3 in front of their MFMAs / waves 4-7 behind the first channel fragments,
#         2 = every wave behind its first channel fragment.     main = FAST 1, SCHED 2 (split) / 1 (fp16): the shipped defaults.
#   gpurun --timeout 900 -- bash scripts/gpu/r5_sched_ab.sh
R=$(pwd); O=$R/gpurun_out/r5s2; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "halo or conv_igemm or conv_concat" > $O/pytest_ops.log 2>&1; echo "op tests rc=$?"; tail -2 $O/pytest_ops.log
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -s -k "patch_unembed_fold or shortcut_fold or groupnorm_tails or unet_forward_vs_oracle or autoencoder_vs_oracle" > $O/pytest_eng.log 2>&1; echo "engine tests rc=$?"; grep -E "fold|passed|failed|Error" $O/pytest_eng.log | tail -8
for v in main f1s1 f1s0 f0s2 f0s1; do
  L=$R/ablib/$v.so; [ $v = main ] && L=$R/resshift_amd/libresshift_hip.so
  for prec in split fp16; do
    RS_BENCH_ONLY="c3" RESSHIFT_HIP_LIB=$L timeout 200 python scripts/igemm_bench.py $prec 5 > $O/ib_${v}_$prec.txt 2>&1; echo "== $v $prec rc=$? $(tail -1 $O/ib_${v}_$prec.txt)"
  done
done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline"
for v in main f1s1 f1s0 main; do
  L=$R/ablib/$v.so; [ $v = main ] && L=$R/resshift_amd/libresshift_hip.so
  RESSHIFT_HIP_LIB=$L timeout 200 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json")); r=d["roofline"]
print("$v", d["ms_per_step"], d["ms_per_unet_step"], d["config"].get("kernel_launches_per_step"), [(k["kernel"][:22], k["ms_per_step"], k["launches_per_step"]) for k in r["per_kernel"]])
PY
done
