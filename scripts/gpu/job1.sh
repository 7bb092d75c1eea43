#!/bin/bash
# round 4, call 1: the parity-headline bench line end to end, knob A/Bs on the parity pass, per-shape table, kernel trace (baseline of the round)
R=$(pwd); O=$R/gpurun_out/r4c1; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --steps 8 --warmup 2 > $O/bench_parity.json 2> $O/bench_parity.err; echo "bench rc=$?"; tail -c 600 $O/bench_parity.json | head -c 600; echo
short() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', d['ms_per_step'], d['config']['kernel_launches_per_step'], [(k['kernel'][:16], k['ms_per_step']) for k in d['roofline']['per_kernel']], d['roofline'].get('groupnorm',{}).get('ms_per_step'))"; }
for knob in "X=0" "RS_MLP_FUSED_MINM=2048" "RS_MLP_FUSED_MINM=8192" "RS_GN_SWIN_STATS=1" "RS_IGEMM_V4_SEG=6" "X=1"; do
  env $knob timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/ab_$knob.json 2> $O/ab.err; echo "rc=$?"; short $O/ab_$knob.json "$knob"
done
RS_PROF_SHAPES=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/shapes.json 2> $O/shapes.err; grep "^\[shapes\]" $O/shapes.err > $O/shapes_parity.txt; wc -l $O/shapes_parity.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_parity -o parity -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --no-secondary > $O/trace_parity.log 2>&1)
db=$(ls $O/trace_parity/*.db 2>/dev/null | head -1); echo "db=$db"
[ -n "$db" ] && python scripts/rocpd_summary.py $db --top 30 > $O/kernel_trace_parity.txt && rm -rf $O/trace_parity
head -30 $O/kernel_trace_parity.txt
