#!/bin/bash
# final measurement call of a round (gpurun --timeout 3000 -- bash scripts/gpu/final_measure.sh; outputs under gpurun_out/r6f): rocprofv3 kernel trace + GroupNorm
# trace + PMC traffic + matrix-pipe PMC of the parity pass on the FINAL sources (digest-stamped, copied into profiles/ so that the bench line of the same call
# replays them), then the full default bench line (the driver's command), the other BASELINE configurations, the fp16 trace and the multi-GPU plumbing lines
R=$(pwd); O=$R/gpurun_out/r6f; mkdir -p $O; export TMPDIR=/tmp
B="--no-cpu-baseline --no-profile-pass --no-secondary --no-unet-step"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_parity -o parity -- python $R/bench.py --steps 2 --warmup 1 $B > $O/trace_parity.log 2>&1)
db=$(ls $O/trace_parity/*.db 2>/dev/null | head -1); echo "db=$db"
if [ -n "$db" ]; then python scripts/rocpd_summary.py $db --top 34 > $O/kernel_trace_parity.txt; python scripts/collect_gn_trace.py $db 3 $O/gn_trace_parity.json; rm -rf $O/trace_parity; fi
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 $B > $O/pmc_$c.log 2>&1)
done
python scripts/collect_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic_parity.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_busy -- python $R/bench.py --steps 1 --warmup 1 $B > $O/pmc_busy.log 2>&1)
python scripts/collect_mfma_busy.py $(find $O/pmc_busy -name "*counter_collection.csv" | head -1) $O/pmc_mfma_busy_parity.json 14 parity > $O/pmc_mfma_busy_parity.txt 2>&1
rm -rf $O/pmc_busy
cp $O/gn_trace_parity.json profiles/r6_gn_trace_parity.json; cp $O/pmc_traffic_parity.json profiles/r6_pmc_traffic_parity.json; cp $O/pmc_mfma_busy_parity.json profiles/r6_pmc_mfma_busy_parity.json
RS_PROF_SHAPES=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-unet-step > $O/shapes.json 2> $O/shapes.err; grep "^\[shapes\]" $O/shapes.err > $O/shapes_parity.txt; wc -l $O/shapes_parity.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_parity_final.json 2> $O/bench_parity_final.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_parity_final.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], d['config']['kernel_launches_per_step'], r['frac'], r['traffic'], d['cpu_baseline']['gpu_vs_cpu_psnr_db'], d['value_fp16_unqualified']['value']); print(r['dominant_kernel']); print(r['ms_by_part'], r['mfma_ms_by_level'])"
for c in journal faceir inpaint; do
  timeout 600 python bench.py --config $c --steps 8 --warmup 2 --parity-images 8 --no-torch-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"
  python -c "
import json; d=json.load(open('$O/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['parity_vs_cpu_oracle'][0]['image_psnr_db'], d['parity_vs_cpu_oracle'][0]['vq_code_agreement'], d['value_fp16_unqualified']['value'])"
done
RS_WINO=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline --no-unet-step > $O/bench_parity_wino_off.json 2> $O/bench_parity_wino_off.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline --no-unet-step > $O/bench_parity_wino_on.json 2> $O/bench_parity_wino_on.err
python -c "
import json
for n in ('off','on'):
    d=json.load(open('$O/bench_parity_wino_'+n+'.json')); print('RS_WINO', n, d['value'], d['ms_per_step'])"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fp16 -o fp16 -- python $R/bench.py --precision fp16 --steps 2 --warmup 1 $B > $O/trace_fp16.log 2>&1)
db=$(ls $O/trace_fp16/*.db 2>/dev/null | head -1)
if [ -n "$db" ]; then python scripts/rocpd_summary.py $db --top 24 > $O/kernel_trace_fp16.txt; rm -rf $O/trace_fp16; fi
# the exact multi-GPU commands of BASELINE.json configs[4] / [2], as plumbing runs on this one-GPU box (ranks share the device: gloo)
RESSHIFT_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 4 --config inpaint --steps 2 --warmup 1 --no-profile-pass > $O/bench_gpus4_inpaint.json 2> $O/bench_gpus4_inpaint.err; echo "gpus4 inpaint rc=$?"
RESSHIFT_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --config journal --steps 2 --warmup 1 --no-profile-pass > $O/bench_gpus8_journal.json 2> $O/bench_gpus8_journal.err; echo "gpus8 journal rc=$?"
for f in bench_gpus4_inpaint bench_gpus8_journal; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['n_gpus'], d['value'], d['ranks']['backend'], len(d['ranks']['per_rank']), d['ranks']['weight_broadcast_bytes'], d['other_policy_all_ranks'])"; done
