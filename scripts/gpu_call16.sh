#!/bin/bash
# round 3, call 16: phase stamps of the fused MLP kernels and of the reworked attention kernels (ablate build)
R=$(pwd); O=$R/gpurun_out/r3c16; mkdir -p $O; export TMPDIR=/tmp
export RESSHIFT_HIP_LIB=$R/ab/lib_ablate.so
timeout 300 python scripts/mlp_split_time.py > $O/mlp_phases.txt 2>&1; echo "mlp rc=$?"; cat $O/mlp_phases.txt
timeout 300 python scripts/attn_phases.py both > $O/attn_phases.txt 2>&1; echo "attn rc=$?"; head -8 $O/attn_phases.txt
