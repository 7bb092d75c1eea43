#!/bin/bash
# rocprofv3 kernel trace of the parity pass with two builds of the library on one box (lib_ab_base.so vs the tree's): per-family summary side by side.  Outputs: gpurun_out/r6c
R=$(pwd); O=$R/gpurun_out/r6c; mkdir -p $O; export TMPDIR=/tmp
B="--no-cpu-baseline --no-profile-pass --no-secondary --no-unet-step"
for v in base new; do
  if [ $v = base ]; then export RESSHIFT_HIP_LIB=$R/resshift_amd/lib_ab_base.so; else unset RESSHIFT_HIP_LIB; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$v -o t -- python $R/bench.py --steps 2 --warmup 1 $B > $O/trace_$v.log 2>&1)
  db=$(ls $O/trace_$v/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then python scripts/rocpd_summary.py $db --top 30 > $O/kernel_trace_$v.txt; rm -rf $O/trace_$v; fi
  echo "== $v"; sed -n 4,22p $O/kernel_trace_$v.txt | cut -c1-110
done
