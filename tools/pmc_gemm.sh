#!/bin/bash
# PMC passes over the projection GEMM (one counter group per pass; FETCH_SIZE and WRITE_SIZE do not share a pass on gfx950).
set -u
R=$(pwd); OUT=$R/gpurun_out/pmc_gemm; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "ff1 2050 0" "ff1 2050 1" "qkv 2050 0"; do
  tag=$(echo $cfg | tr ' ' '_')
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
    d=$OUT/${tag}_$(echo $ctr | tr ' ' '+' | cut -c1-40)
    timeout -k 20 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python $R/tools/pmc_gemm.py $cfg > /dev/null 2>&1
    echo "== $cfg :: $ctr"; python $R/tools/pmc_summary.py $d sat_gemm
  done
done > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
