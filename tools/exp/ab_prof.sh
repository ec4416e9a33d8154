set -u
R=$(pwd); OUT=$R/gpurun_out/c13; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GEN="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep"
rocprofv3 --kernel-trace --stats -d $OUT/prio -- python $R/bench.py --steps 3 --warmup 1 $GEN > $OUT/prio.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/noprio -- python $R/tools/bench_with_lib.py $R/tools/exp/libsat_amd_noprio.so --steps 3 --warmup 1 $GEN > $OUT/noprio.log 2>&1
cd $R
for w in prio noprio; do python tools/rocpd_stats.py $(ls $OUT/$w/*/*.db | head -1) $OUT/${w}_stats.csv; done
find $OUT -name "*.db" -delete; rm -rf $OUT/prio $OUT/noprio
