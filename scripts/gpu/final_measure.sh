#!/bin/bash
# final measurement call of a round (gpurun --timeout 2400 -- bash scripts/gpu/final_measure.sh; outputs under gpurun_out/r4c12): rocprofv3 kernel trace + GroupNorm trace + PMC traffic of the parity pass on the FINAL sources (digest-stamped,
# copied into profiles/ so that the bench line of the same call replays them), the per-shape table, then the full default bench line and the
# other BASELINE configurations
R=$(pwd); O=$R/gpurun_out/r4c12; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "groupnorm_tails or smoke or unet_forward_vs_oracle" > $O/pytest_quick.log 2>&1; echo "quick tests rc=$?"; tail -2 $O/pytest_quick.log
B="--no-cpu-baseline --no-profile-pass --no-secondary"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_parity -o parity -- python $R/bench.py --steps 2 --warmup 1 $B > $O/trace_parity.log 2>&1)
db=$(ls $O/trace_parity/*.db 2>/dev/null | head -1); echo "db=$db"
if [ -n "$db" ]; then python scripts/rocpd_summary.py $db --top 30 > $O/kernel_trace_parity.txt; python scripts/collect_gn_trace.py $db 3 $O/gn_trace_parity.json; rm -rf $O/trace_parity; fi
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 $B > $O/pmc_$c.log 2>&1)
done
python scripts/collect_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic_parity.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cp $O/gn_trace_parity.json profiles/r4_gn_trace_parity.json; cp $O/pmc_traffic_parity.json profiles/r4_pmc_traffic_parity.json
RS_PROF_SHAPES=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/shapes.json 2> $O/shapes.err; grep "^\[shapes\]" $O/shapes.err > $O/shapes_parity.txt; wc -l $O/shapes_parity.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_parity_final.json 2> $O/bench_parity_final.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_parity_final.json')); print(d['value'], d['ms_per_step'], d['config']['kernel_launches_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['gpu_vs_cpu_psnr_db'], d['value_fp16_unqualified']['value'])"
for c in journal faceir inpaint; do
  timeout 600 python bench.py --config $c --steps 8 --warmup 2 --parity-images 8 --no-torch-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"
  python -c "
import json; d=json.load(open('$O/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['parity_vs_cpu_oracle'][0]['image_psnr_db'], d['parity_vs_cpu_oracle'][0]['vq_code_agreement'], d['value_fp16_unqualified']['value'])"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fp16 -o fp16 -- python $R/bench.py --precision fp16 --steps 2 --warmup 1 $B > $O/trace_fp16.log 2>&1)
db=$(ls $O/trace_fp16/*.db 2>/dev/null | head -1)
if [ -n "$db" ]; then python scripts/rocpd_summary.py $db --top 24 > $O/kernel_trace_fp16.txt; rm -rf $O/trace_fp16; fi
# the exact multi-GPU commands of BASELINE.json configs[4] / [2], as plumbing runs on this one-GPU box (ranks share the device: gloo)
RESSHIFT_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 4 --config inpaint --steps 2 --warmup 1 --no-profile-pass > $O/bench_gpus4_inpaint.json 2> $O/bench_gpus4_inpaint.err; echo "gpus4 inpaint rc=$?"
RESSHIFT_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --config journal --steps 2 --warmup 1 --no-profile-pass > $O/bench_gpus8_journal.json 2> $O/bench_gpus8_journal.err; echo "gpus8 journal rc=$?"
for f in bench_gpus4_inpaint bench_gpus8_journal; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['n_gpus'], d['value'], d['ranks']['backend'], len(d['ranks']['per_rank']), d['ranks']['weight_broadcast_bytes'], d['other_policy_all_ranks'])"; done
