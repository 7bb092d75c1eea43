#!/bin/bash
# round 3, call 14: fragment-major weights + relative position bias from a 225-entry LDS table in the fused window-attention kernels: tests, bench A/B (old / new / new with late proj weights)
R=$(pwd); O=$R/gpurun_out/r3c14; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "window_attention" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -5 $O/pytest_ops.log
RESSHIFT_HIP_LIB=$R/ab/lib_latew.so timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "window_attention" > $O/pytest_ops_latew.log 2>&1; echo "ops latew rc=$?"; tail -3 $O/pytest_ops_latew.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "unet_forward or sample_loop_vs_oracle or offsize" > $O/pytest_eng.log 2>&1; echo "eng rc=$?"; tail -5 $O/pytest_eng.log
for rep in 1 2; do
  for v in old new latew; do
    lib=$R/ab/lib_$v.so; [ $v = new ] && lib=$R/resshift_amd/libresshift_hip.so
    for pol in fp16 parity; do
      RESSHIFT_HIP_LIB=$lib timeout 300 python bench.py --precision $pol --steps 8 --warmup 2 --no-cpu-baseline > $O/b.json 2> $O/b.err; echo "$v $pol rc=$? $(python -c "import json;d=json.load(open('$O/b.json'));print(d['ms_per_step'], [ (k['kernel'][:24],k['ms_per_step'],k['frac']) for k in d['roofline']['per_kernel'] if 'win_attn' in k['kernel']])")"
    done
  done
done
