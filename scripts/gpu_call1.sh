#!/bin/bash
# round 3, call 1: new parity tests, bench self-launch, chop_size 512 tile, baseline shapes profile
O=gpurun_out/r3c1; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "offsize or 128_tile or tiles_larger or other_baseline or bench_launches or natural_images or two_ranks or blob_cache or rejects" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_new.log
timeout 600 python scripts/tiled_chop512.py 512 fp16 > $O/tiled_chop512.txt 2> $O/tiled_chop512.err; echo "chop512 rc=$?"; cat $O/tiled_chop512.txt; tail -5 $O/tiled_chop512.err
RS_PROF_SHAPES=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-torch-baseline --no-exact-leg --parity-images 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-900 $O/bench.json
