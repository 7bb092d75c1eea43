R=$(pwd); O=$R/gpurun_out/pmc_r2d; mkdir -p $O; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass > $O/pmc_$c.log 2>&1)
done
python scripts/collect_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic_igemm_fp16.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/pmc_traffic_igemm_fp16.json
