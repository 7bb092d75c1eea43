#!/bin/bash
# A/B of a change to the fused split window-attention kernel on ONE box: op tests on the new library, phase stamps (ablate builds) and the parity pass
# with both libraries (resshift_amd/lib_ab_base.so = the previous build, libresshift_hip.so = the tree's).  Outputs: gpurun_out/r6a
R=$(pwd); O=$R/gpurun_out/r6a; mkdir -p $O
RESSHIFT_HIP_LIB=${NEW_LIB:+$R/$NEW_LIB} timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "window_attention or win_attn or swin" > $O/pytest_attn.txt 2>&1; echo "attn tests rc=$?"; tail -2 $O/pytest_attn.txt
for v in base new; do
  RESSHIFT_HIP_LIB=$R/resshift_amd/lib_ab_abl_$v.so timeout 300 python scripts/attn_phases.py split > $O/phases_$v.txt 2>&1; echo "phases $v rc=$?"; grep -A1 "64x64 shift 0" $O/phases_$v.txt | cut -c1-420
done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline --no-unet-step"
for v in base new base new; do
  if [ $v = base ]; then export RESSHIFT_HIP_LIB=$R/resshift_amd/lib_ab_base.so; elif [ -n "$NEW_LIB" ]; then export RESSHIFT_HIP_LIB=$R/$NEW_LIB; else unset RESSHIFT_HIP_LIB; fi
  timeout 600 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); r=d['roofline']; a=[k for k in r['per_kernel'] if 'win_attn_qkv_split' in k['kernel']]; print('$v', d['value'], d['ms_per_step'], a[0]['ms_per_step'] if a else None)"
done
