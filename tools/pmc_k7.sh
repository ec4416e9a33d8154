#!/bin/bash
# SQ counters of the k = 7 conv kernels (direct and planes) at one level shape: C T
set -u
R=$(pwd); OUT=$R/gpurun_out/pmc_k7; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
 for cfg in "128 2097152" "1024 8192"; do
  tag=$(echo $cfg | tr ' ' '_')_planes$mode
  for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAVES"; do
    d=$OUT/${tag}_$(echo $ctr | tr ' ' '+' | cut -c1-30)
    SAT_K7_PLANES=$mode SAT_K7_PLANES_MIN=1 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python $R/tools/pmc_conv.py conv7 $cfg > /dev/null 2>&1
    echo "== C T = $cfg planes=$mode :: $ctr"; python $R/tools/pmc_summary.py $d sat_conv1d
  done
 done
done > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
