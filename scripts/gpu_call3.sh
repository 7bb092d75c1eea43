#!/bin/bash
# round 3, call 3: split-K target sweep of the small-plane halo kernel (microbench) + parity bench with the hoisted attention weights
O=gpurun_out/r3c3; mkdir -p $O; export TMPDIR=/tmp
export RS_BENCH_ONLY="@16,@8"
for prec in split fp16; do
  echo "== $prec, generic kernels (RS_IGEMM_V4_SEG=0)"; RS_IGEMM_V4_SEG=0 python scripts/igemm_bench.py $prec 20 2>&1 | grep -E "c3|total"
  for tgt in 128 192 256 384 512; do
    echo "== $prec, halo small planes, RS_IGEMM_V4_SKTARGET=$tgt"; RS_IGEMM_V4_SEG=3 RS_IGEMM_V4_SKTARGET=$tgt python scripts/igemm_bench.py $prec 20 2>&1 | grep -E "c3|total"
  done
done > $O/sk_sweep.txt 2>&1
cat $O/sk_sweep.txt
unset RS_BENCH_ONLY
RS_PROF_SHAPES=1 timeout 300 python bench.py --precision parity --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_parity.json 2> $O/bench_parity.err; echo "bench rc=$?"; cut -c1-330 $O/bench_parity.json; grep shapes $O/bench_parity.err | grep -E " f7 | f8 " 
