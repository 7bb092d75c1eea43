#!/bin/bash
# round 4, call 4: wave-0 tails + DPP statistics sums: bit identity, the new evidence tests, same-box A/B against the round-3 library
R=$(pwd); O=$R/gpurun_out/r4c4; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "groupnorm_tails or fused_swin_paths" -s > $O/pytest_tail.log 2>&1; echo "tail tests rc=$?"; tail -3 $O/pytest_tail.log; grep "kernel launches" $O/pytest_tail.log
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py -m gpu -q -k "at_the_bench_batch or outside_baseline or large_weights or vq_nearest or tiles_larger_than_image_size or unet_forward_vs_oracle or batch32_parity" -s > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -4 $O/pytest_new.log; grep "PSNR\|rel err\|ulp\|weight blob\|rows differ" $O/pytest_new.log | cut -c1-200
short() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', d['ms_per_step'], d['config']['kernel_launches_per_step'], [(k['kernel'][:16], k['ms_per_step']) for k in d['roofline']['per_kernel']], d['roofline'].get('groupnorm',{}).get('ms_per_step'), d['roofline'].get('groupnorm',{}).get('launches_per_step'))"; }
for rep in 1 2; do
  RESSHIFT_HIP_LIB=$R/ab/lib_r3.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_r3_$rep.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_r3_$rep.json "r3lib"
  for knob in "RS_GN_TAIL=1" "RS_GN_TAIL=0" "RS_GN_GEN_STATS=0"; do
    env $knob timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_${knob}_$rep.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_${knob}_$rep.json "$knob"
  done
done
RESSHIFT_HIP_LIB=$R/ab/lib_r3.so timeout 300 python bench.py --precision fp16 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/abf_r3.json 2> $O/ab.err; echo "rc=$?"; short $O/abf_r3.json "fp16 r3lib"
timeout 300 python bench.py --precision fp16 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/abf_new.json 2> $O/ab.err; echo "rc=$?"; short $O/abf_new.json "fp16 new"
