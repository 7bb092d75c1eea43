#!/bin/bash
# round 4, call 13: split-K slices of the small-plane halo kernels mapped onto the XCDs (weights of a slice stay in one L2): tests, A/B against the previous build
R=$(pwd); O=$R/gpurun_out/r4c13; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -q -x -k "halo or unet_forward_vs_oracle or groupnorm_tails or sample_loop_vs_oracle" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log
short() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', d['ms_per_step'], d['config']['kernel_launches_per_step'], [(k['kernel'][:16], k['ms_per_step']) for k in d['roofline']['per_kernel']], d['roofline'].get('groupnorm',{}).get('ms_per_step'))"; }
for rep in 1 2 3; do
  RESSHIFT_HIP_LIB=$R/ab/lib_prev.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_prev_$rep.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_prev_$rep.json "prev"
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_new_$rep.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_new_$rep.json "new"
done
RS_PROF_SHAPES=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/shapes.json 2> $O/shapes.err; grep "^\[shapes\] f1 M=2048" $O/shapes.err
