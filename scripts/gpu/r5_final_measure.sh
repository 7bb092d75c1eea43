#!/bin/bash
# Final measurement call of round 5 (gpurun --timeout 2400 -- bash scripts/gpu/r5_final_measure.sh; outputs under gpurun_out/r5f, the digest-stamped
# files are copied into profiles/ at once so that the bench line of the SAME call replays them): rocprofv3 kernel trace + GroupNorm trace, PMC HBM
# traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and PMC matrix-pipe / LDS utilisation of the parity pass on the FINAL kernel sources, the
# per-shape table, the full default bench line (reference modules on the CPU and under autocast on the GPU: oracle/_ref travelled with the
# snapshot), the other BASELINE configurations, and the 4-rank plumbing run.
R=$(pwd); O=$R/gpurun_out/r5f; mkdir -p $O; export TMPDIR=/tmp
B="--no-cpu-baseline --no-profile-pass --no-secondary --no-unet-step"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_parity -o parity -- python $R/bench.py --steps 2 --warmup 1 $B > $O/trace_parity.log 2>&1)
db=$(ls $O/trace_parity/*.db 2>/dev/null | head -1); echo "db=$db"
if [ -n "$db" ]; then python scripts/rocpd_summary.py $db --top 36 > $O/kernel_trace_parity.txt; python scripts/collect_gn_trace.py $db 3 $O/gn_trace_parity.json; rm -rf $O/trace_parity; fi
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 $B > $O/pmc_$c.log 2>&1)
done
python scripts/collect_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic_parity.json > /dev/null
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python $R/bench.py --steps 1 --warmup 1 $B > $O/pmc_mfma.log 2>&1)
python scripts/collect_mfma_busy.py $(find $O/pmc_mfma -name "*counter_collection.csv" | head -1) $O/pmc_mfma_busy_parity.json 14 > $O/pmc_mfma_busy_parity.txt 2>&1; tail -16 $O/pmc_mfma_busy_parity.txt | cut -c1-220
rm -rf $O/pmc_mfma
for f in gn_trace_parity pmc_traffic_parity pmc_mfma_busy_parity; do [ -s $O/$f.json ] && cp $O/$f.json profiles/r5_$f.json; done
RS_PROF_SHAPES=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-unet-step > $O/shapes.json 2> $O/shapes.err; grep "^\[shapes\]" $O/shapes.err > $O/shapes_parity.txt; wc -l $O/shapes_parity.txt
(time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5) 2> $O/bench_time.txt > $O/bench_parity_final.json; echo "bench rc=$? $(grep real $O/bench_time.txt)"
python -c "
import json; d=json.load(open('$O/bench_parity_final.json')); r=d['roofline']; c=d['cpu_baseline']; t=d.get('torch_rocm_autocast_baseline') or d.get('torch_rocm_autocast_restatement_baseline')
print(d['value'], d['ms_per_step'], d['ms_per_unet_step'], d['config']['kernel_launches_per_step'], 'frac', r['frac'], 'issue', r['mfma_issue_frac'], 'path', r['frac_whole_path'], 'traffic', r['traffic'], 'busy', (r['mfma_busy'] or {}).get('family_mfma_busy') if isinstance(r['mfma_busy'], dict) else r['mfma_busy'])
print('cpu', c['kind'], c['value'], c['cores'], c['gpu_vs_cpu_psnr_db'], c['gpu_vs_cpu_psnr_db_worst_image'], c['gpu_vs_cpu_images'], 'fp16', d['value_fp16_unqualified']['value'], 'torch', t and (t.get('kind'), t.get('value'), (t.get('parity_vs_cpu_fp32') or {}).get('image_psnr_db')))"
for c in journal faceir inpaint; do
  timeout 600 python bench.py --config $c --steps 8 --warmup 2 --parity-images 8 --no-torch-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"
  python -c "
import json; d=json.load(open('$O/bench_$c.json')); p=d['parity_vs_cpu_oracle'][0]; print('$c', d['value'], d['ms_per_step'], p['image_psnr_db'], p['image_psnr_db_worst_image'], p['vq_code_agreement'], p['checker'], d['value_fp16_unqualified']['value'])"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fp16 -o fp16 -- python $R/bench.py --precision fp16 --steps 2 --warmup 1 $B > $O/trace_fp16.log 2>&1)
db=$(ls $O/trace_fp16/*.db 2>/dev/null | head -1)
if [ -n "$db" ]; then python scripts/rocpd_summary.py $db --top 24 > $O/kernel_trace_fp16.txt; rm -rf $O/trace_fp16; fi
RESSHIFT_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 4 --config inpaint --steps 2 --warmup 1 --no-profile-pass > $O/bench_gpus4_inpaint.json 2> $O/bench_gpus4_inpaint.err; echo "gpus4 inpaint rc=$?"
python -c "
import json; d=json.load(open('$O/bench_gpus4_inpaint.json')); print('gpus4', d['n_gpus'], d['value'], d['ranks']['backend'], len(d['ranks']['per_rank']), d['ranks']['weight_broadcast_bytes'])"
