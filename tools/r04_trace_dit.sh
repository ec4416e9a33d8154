#!/bin/bash
# per-kernel in-model times of the bf16 sampler and the N = 6145 fp8 sampler (rocprofv3 --kernel-trace --stats): gpurun_out/r04_trace_dit/
set -u
R=$(pwd)
OUT=$R/gpurun_out/r04_trace_dit
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/lc -- python $R/bench.py --workload long_context --steps 4 --warmup 1 --no-cpu-baseline > $OUT/lc.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/ds -- python $R/bench.py --workload dit_sample --steps 10 --warmup 2 --no-cpu-baseline > $OUT/ds.log 2>&1
cd $R
for w in lc ds; do python tools/rocpd_stats.py $(ls $OUT/$w/*/*.db | head -1) $OUT/${w}_stats.csv; python tools/rocpd_launches.py $(ls $OUT/$w/*/*.db | head -1) sat_gemm > $OUT/${w}_gemm_launches.txt; done
find $OUT -name "*.db" -delete; rm -rf $OUT/lc $OUT/ds
head -14 $OUT/lc_stats.csv | cut -c1-160
cat $OUT/lc_gemm_launches.txt | head -30
cat $OUT/ds_gemm_launches.txt | head -30
