#!/bin/bash
# round 5, call 12: kernel stats of the step with the 16-wave sat_wn_grad_splits, and three more A / B pairs
set -u
R=$(pwd); OUT=$R/gpurun_out/r05_call12; rm -rf $OUT; mkdir -p $OUT
GEN="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-dit-train"
cd /tmp && export TMPDIR=/tmp
timeout -k 20 200 rocprofv3 --kernel-trace --stats -d $OUT/vae -- python $R/bench.py --steps 3 --warmup 1 $GEN > $OUT/vae_prof.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $OUT/vae/*/*.db | head -1) $OUT/vae_stats.csv; rm -rf $OUT/vae
grep -E "wn_|reduce_splits|rowsum|multi_copy|copyBuffer|k7q" $OUT/vae_stats.csv | cut -c1-150
for i in 1 2 3; do
  timeout 200 python bench.py --steps 8 --warmup 2 $GEN >> $OUT/vae_wn_in_unit.json 2>> $OUT/ab.err
  timeout 200 python bench.py --steps 8 --warmup 2 $GEN --ops-set wn_fused=0 >> $OUT/vae_wn_separate.json 2>> $OUT/ab.err
done
python - <<PY
import json
for f in ("vae_wn_in_unit","vae_wn_separate"):
    for l in open("$OUT/%s.json"%f):
        r=json.loads(l); print(f, round(r["ms_per_step"],2), r["config"]["launch"]["ms_per_step"], "k7q frac", round(r["roofline"]["frac"],3))
PY
