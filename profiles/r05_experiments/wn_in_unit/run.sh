#!/bin/bash
# round 5, call 10: weight norm inside the conv autograd units (sat_wn_grad_splits) + gradient gather table in kernel arguments — A / B, then the
# round's final bench line, kernel stats and GPU test log on the same box
set -u
R=$(pwd); OUT=$R/gpurun_out/r05_call10; rm -rf $OUT; mkdir -p $OUT
if ! timeout 300 python __graft_entry__.py smoke > $OUT/smoke_first.log 2>&1; then tail -5 $OUT/smoke_first.log; echo "smoke failed on this box: stopping"; exit 1; fi
timeout 600 python -m pytest tests/test_conv_kernels.py tests/test_train_step.py -m gpu -q -x > $OUT/tests_first.log 2>&1; echo "tests exit $?" >> $OUT/tests_first.log
tail -n 3 $OUT/tests_first.log
GEN="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-dit-train"
for i in 1 2; do
  timeout 300 python bench.py --steps 5 --warmup 2 $GEN >> $OUT/vae_wn_in_unit.json 2>> $OUT/ab.err
  timeout 300 python bench.py --steps 5 --warmup 2 $GEN --ops-set wn_fused=0 >> $OUT/vae_wn_separate.json 2>> $OUT/ab.err
done
python - <<PY
import json
for f in ("vae_wn_in_unit","vae_wn_separate"):
    for l in open("$OUT/%s.json"%f):
        r=json.loads(l); print(f, round(r["ms_per_step"],2), r["config"]["launch"]["ms_per_step"], "k7q frac", round(r["roofline"]["frac"],3))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 20 400 rocprofv3 --kernel-trace --stats -d $OUT/vae -- python $R/bench.py --steps 3 --warmup 1 $GEN > $OUT/vae_prof.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $OUT/vae/*/*.db | head -1) $OUT/vae_stats.csv; rm -rf $OUT/vae
head -12 $OUT/vae_stats.csv | cut -c1-150
timeout 900 python bench.py > $OUT/bench_vae_train.json 2> $OUT/bench_vae_train.err
tail -c 600 $OUT/bench_vae_train.json
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/gpu_tests.log 2>&1
tail -3 $OUT/gpu_tests.log
