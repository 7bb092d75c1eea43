#!/bin/bash
# round 4, call 10: fused-attention tail with one arrival per window: bit identity, A/B
R=$(pwd); O=$R/gpurun_out/r4c10; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "groupnorm_tails and split" -s > $O/pytest_tail.log 2>&1; echo "tail tests rc=$?"; tail -3 $O/pytest_tail.log; grep "kernel launches" $O/pytest_tail.log
short() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', d['ms_per_step'], d['config']['kernel_launches_per_step'], [(k['kernel'][:16], k['ms_per_step']) for k in d['roofline']['per_kernel']], d['roofline'].get('groupnorm',{}).get('ms_per_step'), d['roofline'].get('groupnorm',{}).get('launches_per_step'))"; }
for rep in 1 2 3; do
  for knob in "RS_GN_SWIN_STATS_SPLIT=1" "RS_GN_SWIN_STATS_SPLIT=0"; do
    env $knob timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_${knob}_$rep.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_${knob}_$rep.json "$knob"
  done
done
