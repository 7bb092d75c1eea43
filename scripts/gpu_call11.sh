#!/bin/bash
# round 3, call 11: phase stamps of the fused window-attention kernels (ablate build), the self-launch test with the parity-policy leg
R=$(pwd); O=$R/gpurun_out/r3c11; mkdir -p $O; export TMPDIR=/tmp
export RESSHIFT_HIP_LIB=$R/ab/lib_ablate.so
timeout 300 python scripts/attn_phases.py both > $O/attn_phases.txt 2>&1; echo "phases rc=$?"
RS_ATTN_ABL=1 timeout 300 python scripts/attn_phases.py fp16 > $O/attn_phases_abl1.txt 2>&1; echo "phases abl rc=$?"
RS_ATTN_NW=1 timeout 300 python scripts/attn_phases.py fp16 > $O/attn_phases_nw1.txt 2>&1; echo "phases nw1 rc=$?"
unset RESSHIFT_HIP_LIB
cat $O/attn_phases.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "bench_launches" > $O/pytest_bench.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_bench.log
