#!/bin/bash
# round 4, call 7: the new evidence tests, one group per pytest process (a lost box in call 4 took their output with it)
R=$(pwd); O=$R/gpurun_out/r4c7; mkdir -p $O; export TMPDIR=/tmp
run() { timeout $1 python -m pytest $2 -m gpu -q -x -k "$3" -s > $O/$4.log 2>&1; echo "$4 rc=$?"; tail -2 $O/$4.log; grep "PSNR\|rel err\|weight blob\|LR tile" $O/$4.log | cut -c1-230; free -g | sed -n 2p; }
run 600 tests/test_engine_gpu.py "large_weights" large_weights
run 600 tests/test_engine_gpu.py "tiles_larger_than_image_size" tiles_larger
run 900 tests/test_engine_gpu.py "outside_baseline" outside_baseline
run 900 tests/test_engine_gpu.py "at_the_bench_batch and journal" bench_batch_journal
run 900 tests/test_engine_gpu.py "at_the_bench_batch and inpaint" bench_batch_inpaint
run 900 tests/test_engine_gpu.py "at_the_bench_batch and faceir" bench_batch_faceir
run 900 tests/test_engine_gpu.py "chop_size_512" chop512
