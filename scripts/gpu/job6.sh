#!/bin/bash
# round 4, call 6: tails v3 (wave-0 arrive, batched all-thread finish), split streaming AE attention: tests, then A/B
R=$(pwd); O=$R/gpurun_out/r4c6; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "groupnorm_tails" -s > $O/pytest_tail.log 2>&1; echo "tail tests rc=$?"; tail -3 $O/pytest_tail.log; grep "kernel launches" $O/pytest_tail.log
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "ae_flash or vq_nearest or gemm_nt_batched" -s > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/pytest_ops.log; grep "rows differ\|ulp" $O/pytest_ops.log | cut -c1-200
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "realsr_full_size or batch32_parity or large_weights or streaming_ae_attention" -s > $O/pytest_net.log 2>&1; echo "net rc=$?"; tail -3 $O/pytest_net.log; grep "PSNR\|rel err" $O/pytest_net.log | cut -c1-220
short() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', d['ms_per_step'], d['config']['kernel_launches_per_step'], [(k['kernel'][:16], k['ms_per_step']) for k in d['roofline']['per_kernel']], d['roofline'].get('groupnorm',{}).get('ms_per_step'), d['roofline'].get('groupnorm',{}).get('launches_per_step'))"; }
for rep in 1 2; do
  for knob in "RS_GN_TAIL=1" "RS_GN_TAIL=0" "RS_AE_FLASH=0"; do
    env $knob timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_${knob}_$rep.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_${knob}_$rep.json "new $knob"
  done
done
RESSHIFT_HIP_LIB=$R/ab/lib_r3.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_r3.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_r3.json "r3lib"
