#!/bin/bash
# split-K plan of the 8 x 8 level's halo convs (igemm4.hip rs_igemm4_plan: RS_IGEMM_V4_SKTARGET = workgroups a launch should reach, RS_IGEMM_V4_SKMINSTAGES = shortest
# slice): parity pass per setting on one box.  Outputs: gpurun_out/r6s
R=$(pwd); O=$R/gpurun_out/r6s; mkdir -p $O
B="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline --no-unet-step"
for cfg in "256 6" "128 6" "512 6" "256 12" "512 3" "384 6" "256 6"; do
  set -- $cfg
  RS_IGEMM_V4_SKTARGET=$1 RS_IGEMM_V4_SKMINSTAGES=$2 timeout 600 python bench.py $B > $O/b_$1_$2.json 2> $O/b_$1_$2.err
  python -c "
import json; d=json.load(open('$O/b_$1_$2.json')); r=d['roofline']; print('target $1 minstages $2:', d['value'], d['ms_per_step'], r['mfma_ms_by_level'].get('unet@8'), d['config']['kernel_launches_per_step'])"
done
