#!/bin/bash
# round 4, call 8: fused output heads (GroupNorm + SiLU + conv3x3 -> <= 4 channels): network tests, then A/B
R=$(pwd); O=$R/gpurun_out/r4c8; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "unet_forward_vs_oracle or autoencoder_vs_oracle or sample_loop_vs_oracle or realsr_full_size or offsize or tiles_larger or groupnorm_tails or sampler_drop_in" > $O/pytest_net.log 2>&1; echo "net rc=$?"; tail -4 $O/pytest_net.log
short() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', d['ms_per_step'], d['config']['kernel_launches_per_step'], [(k['kernel'][:16], k['ms_per_step']) for k in d['roofline']['per_kernel']], d['roofline'].get('groupnorm',{}).get('ms_per_step'), d['roofline'].get('groupnorm',{}).get('launches_per_step'))"; }
for rep in 1 2 3; do
  for knob in "RS_HEAD_FUSED=1" "RS_HEAD_FUSED=0"; do
    env $knob timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_${knob}_$rep.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_${knob}_$rep.json "$knob"
  done
done
for knob in "RS_HEAD_FUSED=1" "RS_HEAD_FUSED=0"; do
  env $knob timeout 300 python bench.py --precision fp16 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/abf_${knob}.json 2> $O/ab.err; echo "rc=$?"; short $O/abf_${knob}.json "fp16 $knob"
done
