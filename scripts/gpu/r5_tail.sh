#!/bin/bash
# last call of round 5: the two multi-GPU commands of BASELINE configs[4] / [2] as gloo plumbing runs (ranks share the one GPU) and one more default bench line on another box
R=$(pwd); O=$R/gpurun_out/r5z; mkdir -p $O; export TMPDIR=/tmp
RESSHIFT_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 4 --config inpaint --steps 2 --warmup 1 --no-profile-pass > $O/bench_gpus4_inpaint.json 2> $O/bench_gpus4_inpaint.err; echo "gpus4 inpaint rc=$?"
RESSHIFT_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --config journal --steps 2 --warmup 1 --no-profile-pass > $O/bench_gpus8_journal.json 2> $O/bench_gpus8_journal.err; echo "gpus8 journal rc=$?"
for f in bench_gpus4_inpaint bench_gpus8_journal; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['n_gpus'], d['value'], d['ranks']['backend'], len(d['ranks']['per_rank']), d['ranks']['weight_broadcast_bytes'], d['other_policy_all_ranks'])"; done
(time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5) 2> $O/bench_time.txt > $O/bench_parity_second_box.json; echo "bench rc=$? $(grep real $O/bench_time.txt)"
python -c "
import json; d=json.load(open('$O/bench_parity_second_box.json')); r=d['roofline']; c=d['cpu_baseline']; print(d['value'], d['ms_per_step'], d['ms_per_unet_step'], r['frac'], r['mfma_issue_frac'], r['traffic'], c['kind'], c['value'], c['gpu_vs_cpu_psnr_db'], c['gpu_vs_cpu_images'])"
