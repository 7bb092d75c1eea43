#!/bin/bash
# round 3, call 2: fused split attention + small-plane halo kernel: op tests, engine parity, A/B bench
O=gpurun_out/r3c2; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "fused_qkv_split or small_planes or halo or conv_igemm or window_attention" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -30 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "batch32 or realsr_full or other_baseline or fused_swin or 128_tile or sample_loop_vs or unet_forward_vs" > $O/pytest_eng.log 2>&1; echo "eng rc=$?"; tail -15 $O/pytest_eng.log
RS_PROF_SHAPES=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-torch-baseline --no-exact-leg --parity-images 2 > $O/bench_new.json 2> $O/bench_new.err; echo "bench rc=$?"; cut -c1-400 $O/bench_new.json
for pol in fp16 parity; do
  RS_IGEMM_V4_SEG=0 RS_ATTN_FUSED_SPLIT=0 timeout 300 python bench.py --precision $pol --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass > $O/bench_old_$pol.json 2> $O/bench_old_$pol.err; echo "old $pol rc=$?"; cut -c1-330 $O/bench_old_$pol.json
done
RS_ATTN_FUSED_SPLIT=0 timeout 300 python bench.py --precision parity --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass > $O/bench_noattn_parity.json 2> $O/bench_noattn.err; echo "seg only parity rc=$?"; cut -c1-330 $O/bench_noattn_parity.json
