#!/bin/bash
# SQ counters of the strided convs' weight-gradient kernel (sat_wgrad_small_bf16x3_kernel<2>) at two Oobleck levels.
set -u
R=$(pwd); OUT=$R/gpurun_out/pmc_wgs; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kind in wgrads2 wgrads8; do
  for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
    d=$OUT/${kind}_$(echo $ctr | tr ' ' '+' | cut -c1-30)
    timeout -k 20 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python $R/tools/pmc_conv.py $kind > /dev/null 2>&1
    echo "== $kind :: $ctr"; python $R/tools/pmc_summary.py $d sat_wgrad_small
  done
done > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
