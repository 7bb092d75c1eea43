#!/bin/bash
# round 3, call 18: halo kernel (fp16) fetches its residual as rows through the staging tile: tests, A/B, and - only if it pays - the digest-stamped artifacts again
R=$(pwd); O=$R/gpurun_out/r3c18; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "halo or conv_igemm" > $O/pytest_ops.log 2>&1; rc1=$?; echo "ops rc=$rc1"; tail -3 $O/pytest_ops.log
timeout 300 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "unet_forward or autoencoder_vs" > $O/pytest_eng.log 2>&1; rc2=$?; echo "eng rc=$rc2"; tail -3 $O/pytest_eng.log
[ $rc1 -ne 0 -o $rc2 -ne 0 ] && exit 1
for rep in 1 2; do
  for v in prev new; do
    lib=$R/ab/lib_$v.so; [ $v = new ] && lib=$R/resshift_amd/libresshift_hip.so
    RESSHIFT_HIP_LIB=$lib timeout 200 python bench.py --precision fp16 --steps 8 --warmup 2 --no-cpu-baseline > $O/b_$v$rep.json 2> $O/b.err; echo "$v fp16 rc=$? $(python -c "import json;d=json.load(open('$O/b_$v$rep.json'));print(d['ms_per_step'], [ (k['kernel'][:14],k['ms_per_step']) for k in d['roofline']['per_kernel'] if 'igemm4' in k['kernel']])")"
  done
done
gain=$(python -c "
import json
m=lambda v:sum(json.load(open('$O/b_%s%d.json'%(v,r)))['ms_per_step'] for r in (1,2))/2
print(1 if m('prev')-m('new')>0.25 else 0)")
echo "adopt=$gain"
[ "$gain" != "1" ] && exit 0
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_fp16 -o fp16 -- python $R/bench.py --precision fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass > $O/trace_fp16.log 2>&1)
db=$(ls $O/trace_fp16/*.db | head -1)
python scripts/rocpd_summary.py $db --top 24 > $O/kernel_trace_fp16.txt; python scripts/collect_gn_trace.py $db 3 $O/gn_trace_fp16.json; rm -rf $O/trace_fp16
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass > $O/pmc_$c.log 2>&1)
done
python scripts/collect_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic_fp16.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
