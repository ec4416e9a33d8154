#!/bin/bash
# PMC passes over the bf16 attention forward (one counter group per pass).
set -u
R=$(pwd); OUT=$R/gpurun_out/pmc_attn; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "2 1025" "2 6145"; do
  tag=$(echo $cfg | tr ' ' '_')
  for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES"; do
    d=$OUT/${tag}_$(echo $ctr | tr ' ' '+' | cut -c1-40)
    timeout -k 20 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python $R/tools/pmc_attn.py $cfg > /dev/null 2>&1
    echo "== $cfg :: $ctr"; python $R/tools/pmc_summary.py $d sat_attn_fwd
  done
done > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
