#!/bin/bash
# round 3, call 17: attention softmax in base 2 with MFMA row sums + mask-free path, shortcut DMA behind the V^T barrier, fp16 MLP shortcut prefetched: tests, bench A/B
R=$(pwd); O=$R/gpurun_out/r3c17; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "window_attention or swin_mlp or mlp" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -5 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "unet_forward or sample_loop_vs_oracle or offsize or batch32" > $O/pytest_eng.log 2>&1; echo "eng rc=$?"; tail -5 $O/pytest_eng.log
for rep in 1 2; do
  for v in prev new; do
    lib=$R/ab/lib_$v.so; [ $v = new ] && lib=$R/resshift_amd/libresshift_hip.so
    for pol in fp16 parity; do
      RESSHIFT_HIP_LIB=$lib timeout 300 python bench.py --precision $pol --steps 8 --warmup 2 --no-cpu-baseline > $O/b.json 2> $O/b.err; echo "$v $pol rc=$? $(python -c "import json;d=json.load(open('$O/b.json'));print(d['ms_per_step'], [ (k['kernel'][:24],k['ms_per_step'],k['frac']) for k in d['roofline']['per_kernel'] if 'win_attn' in k['kernel'] or 'swin_mlp' in k['kernel']])")"
    done
  done
done
