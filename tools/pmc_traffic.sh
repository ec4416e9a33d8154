#!/bin/bash
# HBM traffic of the dominant conv kernels from the L2 memory-side counters, one rocprofv3 --pmc pass per counter
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), plus a calibration pass on launches of known traffic
# (MI355X_MICROARCH.md, HBM section).  Run on the GPU box from the repo root; writes gpurun_out/pmc_traffic/summary.txt.
set -u
R=$(pwd)
OUT=$R/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kind in calib conv7 dgrad7 conv1 wgrad7; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 20 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${kind}_$ctr -- python $R/tools/pmc_conv.py $kind > /dev/null 2>&1
  done
done
cd $R
for kind in calib conv7 dgrad7 conv1 wgrad7; do
  echo "== $kind (C=128, T=2097152; 4 launches)"
  for ctr in FETCH_SIZE WRITE_SIZE; do python tools/pmc_summary.py $OUT/${kind}_$ctr sat_ ; python tools/pmc_summary.py $OUT/${kind}_$ctr elementwise ; done
done > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
