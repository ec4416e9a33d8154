#!/bin/bash
# SQ counters of the k = 7 conv kernels at C = 128, T = 2 097 152 (the level that dominates the train step): the direct kernel
# (conv1d_bf16x3_k7.h: round 2's path at this level), the shipped one (conv1d_bf16x3_k7q.h), and the discriminator's 3 x 9 kernel
# (disc_conv.hip).  Separate passes per counter group, --kernel-trace only (no other trace domains).
set -u
R=$(pwd); OUT=$R/gpurun_out/pmc_k7; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kind in conv7 conv7q disc9; do
  for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAVES"; do
    d=$OUT/${kind}_$(echo $ctr | tr ' ' '+' | cut -c1-30)
    timeout -k 20 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python $R/tools/pmc_conv.py $kind 128 2097152 > /dev/null 2>&1
    echo "== $kind (C = 128, T = 2097152; disc9: 64 -> 64, 3 x 9, 8189 frames x 513 bins) :: $ctr"; python $R/tools/pmc_summary.py $d sat_conv1d sat_disc_conv
  done
done > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
