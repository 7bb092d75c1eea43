#!/bin/bash
# same-box A/B of two builds of the library on the parity pass: resshift_amd/lib_ab_base.so (the previous build) against the tree's libresshift_hip.so, alternating;
# first the op tests named in $TESTS (pytest -k expression) and the engine tests on the new library.  Outputs: gpurun_out/r6b
R=$(pwd); O=$R/gpurun_out/r6b; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "${TESTS:-halo or window_attention or igemm}" > $O/pytest_ops.txt 2>&1; echo "op tests rc=$?"; tail -2 $O/pytest_ops.txt
if [ -n "$ENGINE_TESTS" ]; then timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "$ENGINE_TESTS" > $O/pytest_engine.txt 2>&1; echo "engine tests rc=$?"; tail -2 $O/pytest_engine.txt; fi
B="--steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-unet-step"
for v in base new base new; do
  if [ $v = base ]; then export RESSHIFT_HIP_LIB=$R/resshift_amd/lib_ab_base.so; else unset RESSHIFT_HIP_LIB; fi
  timeout 600 python bench.py $B > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], (d.get('value_fp16_unqualified') or {}).get('value'), r['ms_by_part'])"
done
