#!/bin/bash
# round 3, call 9: halo kernel as two 4-wave workgroups per CU (RS_IGEMM_V4_W4): correctness, microbench, bench A/B
O=gpurun_out/r3c9; mkdir -p $O; export TMPDIR=/tmp
RS_IGEMM_V4_W4=3 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "halo and not small" > $O/pytest_ops_w4.log 2>&1; echo "ops(w4) rc=$?"; tail -3 $O/pytest_ops_w4.log
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "halo or small_plane" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/pytest_ops.log
RS_IGEMM_V4_W4=3 timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "batch32 or realsr_full" > $O/pytest_eng_w4.log 2>&1; echo "eng(w4) rc=$?"; tail -3 $O/pytest_eng_w4.log
for prec in fp16 split; do
  for w4 in 0 3; do
    echo "== $prec RS_IGEMM_V4_W4=$w4"; RS_IGEMM_V4_W4=$w4 RS_BENCH_ONLY="c3" python scripts/igemm_bench.py $prec 20 2>&1 | grep -E "c3|total"
  done
done > $O/w4_microbench.txt 2>&1; cat $O/w4_microbench.txt
for w4 in 0 3 0 3; do
  for pol in fp16 parity; do
    RS_IGEMM_V4_W4=$w4 timeout 300 python bench.py --precision $pol --steps 8 --warmup 2 --no-cpu-baseline --no-profile-pass > $O/b.json 2> $O/b.err; echo "$pol w4=$w4 rc=$? $(python -c "import json;d=json.load(open('$O/b.json'));print(d['ms_per_step'])")"
  done
done
