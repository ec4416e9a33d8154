#!/bin/bash
# per-launch durations (grouped by grid shape) of the conv-stack kernels of the generator step: gpurun_out/r04_trace/
set -u
R=$(pwd)
OUT=$R/gpurun_out/r04_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GEN="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-graph"
rocprofv3 --kernel-trace --stats -d $OUT/vae -- python $R/bench.py --steps 3 --warmup 1 $GEN > $OUT/vae_prof.log 2>&1
cd $R
DB=$(ls $OUT/vae/*/*.db | head -1)
python tools/rocpd_stats.py $DB $OUT/vae_stats.csv
for k in wgrad_small wgrad7 "conv1d_bf16x3_kernel" k7q stft rowsum reduce_splits; do
  echo "=== $k" >> $OUT/launches.txt
  python tools/rocpd_launches.py $DB $k >> $OUT/launches.txt
done
find $OUT -name "*.db" -delete; rm -rf $OUT/vae
head -70 $OUT/launches.txt
