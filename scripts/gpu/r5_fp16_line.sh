#!/bin/bash
# the all-fp16 policy (BASELINE.json's dtype; NOT the credited number: 46 dB) as a full bench line + its PMC matrix-pipe utilisation, final round-5 sources
R=$(pwd); O=$R/gpurun_out/r5h; mkdir -p $O; export TMPDIR=/tmp
B="--no-cpu-baseline --no-profile-pass --no-secondary --no-unet-step"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python $R/bench.py --precision fp16 --steps 1 --warmup 1 $B > $O/pmc_mfma.log 2>&1)
python scripts/collect_mfma_busy.py $(find $O/pmc_mfma -name "*counter_collection.csv" | head -1) $O/pmc_mfma_busy_fp16.json 12 fp16 > $O/pmc_mfma_busy_fp16.txt 2>&1; tail -13 $O/pmc_mfma_busy_fp16.txt | cut -c1-200
rm -rf $O/pmc_mfma; cp $O/pmc_mfma_busy_fp16.json profiles/r5_pmc_mfma_busy_fp16.json
timeout 600 python bench.py --precision fp16 --steps 20 --warmup 5 --parity-images 8 --cpu-seconds 70 --no-torch-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_fp16.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], d['ms_per_unet_step'], d['config']['kernel_launches_per_step'], r['frac'], r['mfma_issue_frac'], d['parity_vs_cpu_oracle'][0]['image_psnr_db'], [(k['kernel'][:18], k['ms_per_step'], k['frac']) for k in r['per_kernel']])"
