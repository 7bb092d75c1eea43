#!/bin/bash
# round 3, call 4: MFMA-shape probe, early (overlapped) GroupNorm pass in the halo kernel: correctness + A/B microbench + bench
O=gpurun_out/r3c4; mkdir -p $O; export TMPDIR=/tmp
{ echo "== wave8_probe (16x16x32)"; ./scripts/probe/wave8_probe; echo "== wave8_probe32 (32x32x16)"; ./scripts/probe/wave8_probe32; } > $O/probe_mfma_shape.txt 2>&1; cat $O/probe_mfma_shape.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "halo or conv_igemm or small_plane" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -8 $O/pytest_ops.log
for prec in fp16 split; do
  for early in 0 1; do
    echo "== $prec RS_IG4_EARLY_GN=$early"; RS_IG4_EARLY_GN=$early RS_BENCH_ONLY="c3" python scripts/igemm_bench.py $prec 20 2>&1 | grep -E "c3|total"
  done
done > $O/early_gn_microbench.txt 2>&1; cat $O/early_gn_microbench.txt
for early in 0 1; do
  RS_IG4_EARLY_GN=$early timeout 300 python bench.py --precision fp16 --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass > $O/bench_fp16_early$early.json 2> $O/bench_fp16_early$early.err; echo "fp16 early=$early rc=$?"; cut -c1-330 $O/bench_fp16_early$early.json
  RS_IG4_EARLY_GN=$early timeout 300 python bench.py --precision parity --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass > $O/bench_parity_early$early.json 2> $O/bench_parity_early$early.err; echo "parity early=$early rc=$?"; cut -c1-330 $O/bench_parity_early$early.json
done
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "batch32 or realsr_full or fused_swin" > $O/pytest_eng.log 2>&1; echo "eng rc=$?"; tail -5 $O/pytest_eng.log
