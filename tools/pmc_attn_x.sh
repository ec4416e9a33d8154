#!/bin/bash
# PMC passes over experimental attention-forward variants (one counter group per pass; --pmc with --kernel-trace only).
#   bash tools/pmc_attn_x.sh "n 15" "n 62"      (args: "<lib tag> <variant>")
set -u
R=$(pwd); OUT=$R/gpurun_out/pmc_attn_x; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
 for cfg in "2 6145" "2 1025"; do
  tag=$(echo "$v $cfg" | tr ' ' '_')
  for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES"; do
    d=$OUT/${tag}_$(echo $ctr | tr ' ' '+' | cut -c1-30)
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python $R/tools/pmc_attn_x.py $v $cfg > /dev/null 2>&1
    echo "== $v :: $cfg :: $ctr"; python $R/tools/pmc_summary.py $d attn_fwd
  done
 done
done > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
