#!/bin/bash
# round 5, call 7: the swizzled transposed images of sat_ru_k1_bwd (kernel tests + duration in the model) and the batched SnakeBeta constants (A / B)
set -u
R=$(pwd); OUT=$R/gpurun_out/r05_call7; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv_kernels.py tests/test_vae_parity.py tests/test_train_step.py -m gpu -q -x > $OUT/tests.log 2>&1; echo "tests exit $?" >> $OUT/tests.log
tail -n 4 $OUT/tests.log
GEN="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-graph --no-dit-train"
for i in 1 2; do
  timeout 300 python bench.py --steps 5 --warmup 2 $GEN >> $OUT/vae_batched.json 2>> $OUT/ab.err
  timeout 300 python bench.py --steps 5 --warmup 2 $GEN --ops-set snake_batch=0 >> $OUT/vae_per_pair.json 2>> $OUT/ab.err
done
python - <<PY
import json
for f in ("vae_batched","vae_per_pair"):
    for l in open("$OUT/%s.json"%f):
        r=json.loads(l); print(f, round(r["ms_per_step"],2), "k7q frac", round(r["roofline"]["frac"],3))
        for k in r["roofline"]["all_conv_kernels"][:8]: print("     ", k["kernel"], k.get("launches"), k.get("total_ms"), k.get("frac"))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/vae -- python $R/bench.py --steps 3 --warmup 1 $GEN > $OUT/vae_prof.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $OUT/vae/*/*.db | head -1) $OUT/vae_stats.csv; rm -rf $OUT/vae
head -30 $OUT/vae_stats.csv | cut -c1-150
