#!/bin/bash
# slim re-collection after a kernel-source change late in a round (gpurun --timeout 460 -- bash scripts/gpu/adopt_measure.sh; outputs under
# gpurun_out/r4c13): the digest-stamped PMC traffic and GroupNorm trace of the parity pass first (copied into profiles/ at once, so that the
# bench line of the same call replays them), then the kernel trace, the default bench line with a bounded CPU-oracle budget, the per-shape table
# and two quick tests.  The full version (other configs, fp16 trace, multi-rank plumbing) is scripts/gpu/final_measure.sh.
R=$(pwd); O=$R/gpurun_out/r4c13; mkdir -p $O; export TMPDIR=/tmp
B="--no-cpu-baseline --no-profile-pass --no-secondary"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 $B > $O/pmc_$c.log 2>&1)
done
python scripts/collect_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic_parity.json && cp $O/pmc_traffic_parity.json profiles/r4_pmc_traffic_parity.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_parity -o parity -- python $R/bench.py --steps 2 --warmup 1 $B > $O/trace_parity.log 2>&1)
db=$(ls $O/trace_parity/*.db 2>/dev/null | head -1); echo "db=$db"
if [ -n "$db" ]; then python scripts/rocpd_summary.py $db --top 30 > $O/kernel_trace_parity.txt; python scripts/collect_gn_trace.py $db 3 $O/gn_trace_parity.json && cp $O/gn_trace_parity.json profiles/r4_gn_trace_parity.json; rm -rf $O/trace_parity; fi
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds ${CPU_SECONDS:-150} > $O/bench_parity_final.json 2> $O/bench_parity_final.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_parity_final.json')); print(d['value'], d['ms_per_step'], d['config']['kernel_launches_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['groupnorm'].get('source'), d['cpu_baseline']['gpu_vs_cpu_psnr_db'], d['parity_vs_cpu_oracle'][0]['images'], d['value_fp16_unqualified']['value'])"
RS_PROF_SHAPES=1 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/shapes.json 2> $O/shapes.err; grep "^\[shapes\]" $O/shapes.err > $O/shapes_parity.txt; wc -l $O/shapes_parity.txt
timeout 300 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "groupnorm_tails or large_weights" > $O/pytest_quick.log 2>&1; echo "quick tests rc=$?"; tail -2 $O/pytest_quick.log
