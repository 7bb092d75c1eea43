#!/bin/bash
# round 3, call 5 (ablate build): in-kernel phase stamps of the halo kernel, split vs fp16 storage
O=gpurun_out/r3c5; mkdir -p $O; export TMPDIR=/tmp
for prec in split fp16; do
  echo "== $prec"; RS_IG4_PHASES=1 RS_BENCH_ONLY="c3 160->160 @64,c3 320->160 @64,c3 320->320 @32,c3 640->320 @32,ae c3 128->128,ae c3 512->512 @64,ae c3 256->256" python scripts/igemm_bench.py $prec 5 2>&1 | grep -E "c3|phases"
done > $O/phases.txt 2>&1; cat $O/phases.txt
