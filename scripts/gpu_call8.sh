#!/bin/bash
# round 3, call 8: wave-local staging sync (A/B against a full-barrier build of the same sources), coalesced coefficient kernel + Swin statistics A/B
O=gpurun_out/r3c8; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "batch32 or realsr_full or fused_swin or unet_forward_vs or sample_loop_vs" > $O/pytest_eng.log 2>&1; echo "eng rc=$?"; tail -3 $O/pytest_eng.log
run() { # name, env...
  local name=$1; shift
  for pol in fp16 parity; do
    env "$@" timeout 300 python bench.py --precision $pol --steps 8 --warmup 2 --no-cpu-baseline --no-profile-pass > $O/bench_${pol}_$name.json 2> $O/bench_${pol}_$name.err; echo "$pol $name rc=$? $(python -c "import json;d=json.load(open('$O/bench_${pol}_$name.json'));print(d['ms_per_step'], d['config']['kernel_launches_per_step'])")"
  done
}
run new X=1
run fullbarrier RESSHIFT_HIP_LIB=$PWD/ab/lib_fullbarrier.so
run nostats RS_GN_SWIN_STATS=0
run new2 X=1
