#!/bin/bash
# round 4, call 5 (after a lost box): the wave-0 tails alone - bit identity, then one short bench - before anything bigger is tried again
R=$(pwd); O=$R/gpurun_out/r4c5; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "groupnorm_tails" -s > $O/pytest_tail.log 2>&1; echo "tail tests rc=$?"; tail -3 $O/pytest_tail.log; grep "kernel launches" $O/pytest_tail.log
short() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', d['ms_per_step'], d['config']['kernel_launches_per_step'], [(k['kernel'][:16], k['ms_per_step']) for k in d['roofline']['per_kernel']], d['roofline'].get('groupnorm',{}).get('ms_per_step'), d['roofline'].get('groupnorm',{}).get('launches_per_step'))"; }
for rep in 1 2; do
  RESSHIFT_HIP_LIB=$R/ab/lib_r3.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_r3_$rep.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_r3_$rep.json "r3lib"
  RESSHIFT_HIP_LIB=$R/ab/lib_resfrags.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_resfrags_$rep.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_resfrags_$rep.json "new, residual as fragments"
  for knob in "RS_GN_TAIL=1" "RS_GN_TAIL=0" "RS_GN_GEN_STATS=0"; do
    env $knob timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_${knob}_$rep.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_${knob}_$rep.json "new $knob"
  done
done
