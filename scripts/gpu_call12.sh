#!/bin/bash
# round 3, call 12: twelve-wave fused window attention (fp16): op tests, engine parity subset, bench A/B against the six-wave kernel
R=$(pwd); O=$R/gpurun_out/r3c12; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "window_attention" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -5 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "unet_forward or sample_loop_vs_oracle or offsize" > $O/pytest_eng.log 2>&1; echo "eng rc=$?"; tail -5 $O/pytest_eng.log
for w12 in 1 0 1 0; do
  RS_ATTN_W12=$w12 timeout 300 python bench.py --precision fp16 --steps 8 --warmup 2 --no-cpu-baseline > $O/b.json 2> $O/b.err; echo "fp16 w12=$w12 rc=$? $(python -c "import json;d=json.load(open('$O/b.json'));print(d['ms_per_step'], [ (k['kernel'][:20],k['ms_per_step'],k['frac']) for k in d['roofline']['per_kernel'] if 'attn' in k['kernel']])")"
done
