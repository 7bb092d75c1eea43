#!/bin/bash
# round 3, call 7: GroupNorm statistics from the fused attention / MLP epilogues: parity tests + A/B bench
O=gpurun_out/r3c7; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "batch32 or realsr_full or fused_swin or unet_forward_vs or 128_tile" > $O/pytest_eng.log 2>&1; echo "eng rc=$?"; tail -5 $O/pytest_eng.log
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "small_plane or swin_mlp or window_attention" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/pytest_ops.log
for st in 0 1; do
  for pol in fp16 parity; do
    RS_GN_SWIN_STATS=$st timeout 300 python bench.py --precision $pol --steps 8 --warmup 2 --no-cpu-baseline --no-profile-pass > $O/bench_${pol}_stats$st.json 2> $O/bench_${pol}_stats$st.err; echo "$pol swin_stats=$st rc=$?"; cut -c1-260 $O/bench_${pol}_stats$st.json | cut -c100-260
  done
done
