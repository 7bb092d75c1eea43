#!/bin/bash
# Round 5: A/B of the halo kernel's refill placement (igemm4_kernel.h SCHED; libs built with RS_BUILD_DEFS="RS_IG4_SCHED=n" RS_BUILD_OUT=ablib/schedn.so):
#   main = SCHED 1 (waves 0-3 refill in front of their MFMAs, waves 4-7 behind the first channel fragments), sched0 = round 4's order,
#   sched2 = every wave refills behind its first channel fragment.   gpurun --timeout 900 -- bash scripts/gpu/r5_sched_ab.sh
R=$(pwd); O=$R/gpurun_out/r5s; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "halo or conv_igemm or conv_concat" > $O/pytest_ops.log 2>&1; echo "op tests rc=$?"; tail -2 $O/pytest_ops.log
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -s -k "patch_unembed_fold or shortcut_fold or groupnorm_tails or unet_forward_vs_oracle" > $O/pytest_eng.log 2>&1; echo "engine tests rc=$?"; grep -E "fold|passed|failed|Error" $O/pytest_eng.log | tail -8
for v in main sched0 sched2; do
  L=$R/ablib/$v.so; [ $v = main ] && L=$R/resshift_amd/libresshift_hip.so
  for prec in split fp16; do
    RS_BENCH_ONLY="c3" RESSHIFT_HIP_LIB=$L timeout 200 python scripts/igemm_bench.py $prec 5 > $O/ib_${v}_$prec.txt 2>&1; echo "== $v $prec rc=$?"; cat $O/ib_${v}_$prec.txt | grep -v "^shape"
  done
done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline"
for v in main sched0 sched2 main; do
  L=$R/ablib/$v.so; [ $v = main ] && L=$R/resshift_amd/libresshift_hip.so
  RESSHIFT_HIP_LIB=$L timeout 200 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json")); r=d["roofline"]
print("$v", d["ms_per_step"], d["ms_per_unet_step"], d["config"].get("kernel_launches_per_step"), [(k["kernel"][:22], k["ms_per_step"], k["launches_per_step"]) for k in r["per_kernel"]])
PY
done
