#!/bin/bash
# Round 5: A/B of the split halo kernel's fragment read-ahead (XPF) and of the static wave priority (PRIO), plus the hardened parity tests.
#   libs: main (XPF 1, PRIO 1), ablib/xpf0.so (RS_IG4_XPF=0), ablib/prio0.so (RS_IG4_PRIO=0).   gpurun --timeout 1200 -- bash scripts/gpu/r5_xpf_ab.sh
R=$(pwd); O=$R/gpurun_out/r5x; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "halo or conv_igemm or conv_concat" > $O/pytest_ops.log 2>&1; echo "op tests rc=$?"; tail -2 $O/pytest_ops.log
timeout 400 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -s -k "shortcut_fold or groupnorm_tails or unet_forward_vs_oracle or autoencoder_vs_oracle" > $O/pytest_eng.log 2>&1; echo "engine tests rc=$?"; grep -E "fold|passed|failed|Error" $O/pytest_eng.log | tail -6
for v in main xpf0 prio0; do
  L=$R/ablib/$v.so; [ $v = main ] && L=$R/resshift_amd/libresshift_hip.so
  for prec in split fp16; do
    RS_BENCH_ONLY="c3" RESSHIFT_HIP_LIB=$L timeout 200 python scripts/igemm_bench.py $prec 5 > $O/ib_${v}_$prec.txt 2>&1; echo "== $v $prec rc=$? $(tail -1 $O/ib_${v}_$prec.txt)"
  done
done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline"
for v in main xpf0 prio0 main; do
  L=$R/ablib/$v.so; [ $v = main ] && L=$R/resshift_amd/libresshift_hip.so
  RESSHIFT_HIP_LIB=$L timeout 200 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json")); r=d["roofline"]
print("$v", d["ms_per_step"], d["ms_per_unet_step"], d["config"].get("kernel_launches_per_step"), [(k["kernel"][:22], k["ms_per_step"], k["launches_per_step"]) for k in r["per_kernel"]])
PY
done
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -s -k "other_baseline_configs_at_the_bench_batch or reference_faceir_and_inpainting or batch32_parity" > $O/pytest_parity.log 2>&1; echo "parity tests rc=$?"; grep -E "parity policy|B=32|passed|failed|Error|assert" $O/pytest_parity.log | cut -c1-400 | tail -14
