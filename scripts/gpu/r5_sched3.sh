#!/bin/bash
# A/B: SCHED 3 (one LDS-DMA piece per channel fragment, wave halves half a fragment apart; ablib/s3.so) against the shipped SCHED 2, split storage
R=$(pwd); O=$R/gpurun_out/r5s3; mkdir -p $O; export TMPDIR=/tmp
RESSHIFT_HIP_LIB=$R/ablib/s3.so timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "halo" > $O/pytest_ops.log 2>&1; echo "op tests (s3) rc=$?"; tail -1 $O/pytest_ops.log
for v in main s3; do
  L=$R/ablib/$v.so; [ $v = main ] && L=$R/resshift_amd/libresshift_hip.so
  RS_BENCH_ONLY="c3" RESSHIFT_HIP_LIB=$L timeout 200 python scripts/igemm_bench.py split 5 > $O/ib_${v}_split.txt 2>&1; echo "== $v split $(tail -1 $O/ib_${v}_split.txt)"; grep "^unet c3 160->160 @64\|^unet c3 320->160 @64\|^ae c3 512->512\|^ae c3 128->128" $O/ib_${v}_split.txt
done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline --no-unet-step"
for v in main s3 main s3; do
  L=$R/ablib/$v.so; [ $v = main ] && L=$R/resshift_amd/libresshift_hip.so
  RESSHIFT_HIP_LIB=$L timeout 200 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json")); r=d["roofline"]
print("$v", d["ms_per_step"], [(k["kernel"][:22], k["ms_per_step"]) for k in r["per_kernel"][:2]])
PY
done
