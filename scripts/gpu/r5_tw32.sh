#!/bin/bash
# A/B: 8 x 32 halo tiles for the 128-channel-tile shapes (RS_IGEMM_V4_TW32 bit 0 split, bit 1 fp16) against 4 x 64
R=$(pwd); O=$R/gpurun_out/r5w; mkdir -p $O; export TMPDIR=/tmp
for v in 0 3; do
  for prec in split fp16; do
    RS_IGEMM_V4_TW32=$v RS_BENCH_ONLY="ae c3" timeout 200 python scripts/igemm_bench.py $prec 5 > $O/ib_${v}_$prec.txt 2>&1; echo "== tw32=$v $prec"; grep "^ae\|weighted" $O/ib_${v}_$prec.txt
  done
done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline"
for v in 0 1 3 0; do
  RS_IGEMM_V4_TW32=$v timeout 200 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench tw32=$v rc=$?"
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json")); r=d["roofline"]
print("$v", d["ms_per_step"], [(k["kernel"][:22], k["ms_per_step"]) for k in r["per_kernel"][:2]])
PY
done
