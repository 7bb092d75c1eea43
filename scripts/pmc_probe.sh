#!/bin/bash
# usage: [RS_PROBE_CMD="python scripts/x.py" RS_PROBE_KERNEL=substr] scripts/pmc_probe.sh <tag> <shape filter> ; writes gpurun_out/pmc_<tag>/ and a summary gpurun_out/pmc_<tag>.txt
tag=$1; only=$2
export TMPDIR=/tmp RS_BENCH_ONLY="$only"
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 120 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- ${RS_PROBE_CMD:-python $root/scripts/igemm_bench.py ${RS_PROBE_PREC:-fp16} 3} > $out/p$i.log 2>&1)
done
python scripts/pmc_probe.py $out ${RS_PROBE_KERNEL:-igemm} > gpurun_out/pmc_$tag.txt 2>&1
cat gpurun_out/pmc_$tag.txt
