R=$(pwd); O=$R/gpurun_out/r6g; mkdir -p $O
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_parity_final.json 2> $O/bench_parity_final.err; echo "bench rc=$?"
RS_WINO=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline --no-unet-step > $O/bench_parity_wino_off.json 2> $O/wo.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline --no-unet-step > $O/bench_parity_wino_on.json 2> $O/won.err
python -c "
import json
d=json.load(open('$O/bench_parity_final.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], d['cpu_baseline']['gpu_vs_cpu_psnr_db'], d['cpu_baseline']['gpu_vs_cpu_images'], d['value_fp16_unqualified']['value'])
for n in ('off','on'):
    d=json.load(open('$O/bench_parity_wino_'+n+'.json')); print('RS_WINO', n, d['value'], d['ms_per_step'])"
