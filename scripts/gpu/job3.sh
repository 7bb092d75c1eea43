#!/bin/bash
# round 4, call 3: GroupNorm launch diet (tails + producer statistics): bit identity, network parity tests, A/B on the parity and fp16 passes
R=$(pwd); O=$R/gpurun_out/r4c3; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "groupnorm_tails or fused_swin_paths or smoke" -s > $O/pytest_tail.log 2>&1; echo "tail tests rc=$?"; tail -5 $O/pytest_tail.log; grep "kernel launches" $O/pytest_tail.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "unet_forward_vs_oracle or autoencoder_vs_oracle or sample_loop_vs_oracle or batch32_parity" > $O/pytest_net.log 2>&1; echo "net tests rc=$?"; tail -5 $O/pytest_net.log
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "groupnorm or halo or conv" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/pytest_ops.log
short() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', d['ms_per_step'], d['config']['kernel_launches_per_step'], [(k['kernel'][:16], k['ms_per_step']) for k in d['roofline']['per_kernel']], d['roofline'].get('groupnorm',{}).get('ms_per_step'), d['roofline'].get('groupnorm',{}).get('launches_per_step'))"; }
for rep in 1 2; do for knob in "RS_GN_TAIL=1" "RS_GN_TAIL=0" "RS_GN_GEN_STATS=0"; do
  env $knob timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_${knob}_$rep.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_${knob}_$rep.json "$knob"
done; done
for knob in "RS_GN_TAIL=1" "RS_GN_TAIL=0"; do
  env $knob timeout 300 python bench.py --precision fp16 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/abf_${knob}.json 2> $O/ab.err; echo "rc=$?"; short $O/abf_${knob}.json "fp16 $knob"
done
timeout 600 python bench.py --steps 4 --warmup 1 --parity-images 8 --no-torch-baseline --no-secondary > $O/bench_parity8.json 2> $O/bench_parity8.err; echo "rc=$?"; python -c "
import json; d=json.load(open('$O/bench_parity8.json')); print(d['value'], d['parity_vs_cpu_oracle'][0])"
