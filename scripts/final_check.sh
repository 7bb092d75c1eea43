#!/bin/bash
# Round-end verification on the GPU box: tests, smoke, default bench, kernel traces of the fp16 and the parity policy (+ the GroupNorm family
# time per pass from them), the two PMC traffic passes of the fp16 bench (MFMA family + GroupNorm family), the other BASELINE configurations.
# usage: scripts/final_check.sh <tag> [notests]   (outputs under gpurun_out/final_<tag>/)
tag=${1:-r3}
R=$(pwd); O=$R/gpurun_out/final_$tag; mkdir -p $O; export TMPDIR=/tmp
if [ "$2" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
  timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json
for pol in fp16 parity; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$pol -o $pol -- python $R/bench.py --precision $pol --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass > $O/trace_$pol.log 2>&1)
  db=$(ls $O/trace_$pol/*.db | head -1)
  python scripts/rocpd_summary.py $db --top 24 > $O/kernel_trace_$pol.txt; head -14 $O/kernel_trace_$pol.txt
  python scripts/collect_gn_trace.py $db 3 $O/gn_trace_$pol.json
  rm -rf $O/trace_$pol
done
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass > $O/pmc_$c.log 2>&1)
done
python scripts/collect_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic_fp16.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE   # raw CSVs are large; the JSON carries the per-launch figures
for cfg in journal faceir inpaint; do
  timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --parity-images 2 --no-torch-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc=$?"; cut -c1-300 $O/bench_$cfg.json
done
