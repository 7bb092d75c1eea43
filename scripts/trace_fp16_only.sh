R=$(pwd); O=$R/gpurun_out/trace_r2d; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fp16 -o fp16 -- python $R/bench.py --precision fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass > $O/trace_fp16.log 2>&1)
python scripts/rocpd_summary.py $(ls $O/trace_fp16/*.db | head -1) --top 24 > $O/kernel_trace_fp16.txt; head -8 $O/kernel_trace_fp16.txt
rm -rf $O/trace_fp16
python bench.py --no-cpu-baseline > $O/bench_nocpu.json 2>/dev/null; tail -1 $O/bench_nocpu.json | cut -c1-400
