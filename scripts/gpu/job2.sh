#!/bin/bash
# round 4, call 2: GroupNorm tails (stage 1 + 2): bit identity against RS_GN_TAIL=0, the GroupNorm / halo-conv op tests, A/B on the parity and fp16 passes
R=$(pwd); O=$R/gpurun_out/r4c2; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "groupnorm_tails or fused_swin_paths or smoke" -s > $O/pytest_tail.log 2>&1; echo "tail tests rc=$?"; tail -5 $O/pytest_tail.log; grep "kernel launches" $O/pytest_tail.log
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "groupnorm or halo" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/pytest_ops.log
short() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', d['ms_per_step'], d['config']['kernel_launches_per_step'], [(k['kernel'][:16], k['ms_per_step']) for k in d['roofline']['per_kernel']], d['roofline'].get('groupnorm',{}).get('ms_per_step'), d['roofline'].get('groupnorm',{}).get('launches_per_step'))"; }
for rep in 1 2; do for knob in "RS_GN_TAIL=1" "RS_GN_TAIL=0"; do
  env $knob timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_${knob}_$rep.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_${knob}_$rep.json "$knob"
done; done
for knob in "RS_GN_TAIL=1" "RS_GN_TAIL=0"; do
  env $knob timeout 300 python bench.py --precision fp16 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/abf_${knob}.json 2> $O/ab.err; echo "rc=$?"; short $O/abf_${knob}.json "fp16 $knob"
done
