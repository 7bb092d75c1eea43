#!/bin/bash
# round 3, call 6: chunk-aligned split-K slices of the small-plane halo kernel: correctness, microbench A/B, bench
O=gpurun_out/r3c6; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "halo or small_plane" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -4 $O/pytest_ops.log
export RS_BENCH_ONLY="@16,@8"
for prec in split fp16; do
  echo "== $prec, generic kernels (RS_IGEMM_V4_SEG=0)"; RS_IGEMM_V4_SEG=0 python scripts/igemm_bench.py $prec 20 2>&1 | grep -E "c3|total"
  for tgt in 128 256; do
    echo "== $prec, halo small planes (RS_IGEMM_V4_SEG=7), RS_IGEMM_V4_SKTARGET=$tgt"; RS_IGEMM_V4_SEG=7 RS_IGEMM_V4_SKTARGET=$tgt python scripts/igemm_bench.py $prec 20 2>&1 | grep -E "c3|total"
  done
done > $O/sk_sweep.txt 2>&1
cat $O/sk_sweep.txt
unset RS_BENCH_ONLY
RS_PROF_SHAPES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-torch-baseline --no-exact-leg --parity-images 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
