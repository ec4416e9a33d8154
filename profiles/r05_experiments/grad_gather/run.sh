#!/bin/bash
# round 5, call 6: full GPU suite on the current tree, then A / B of the gradient gather (adopt + one sat_multi_copy launch vs per-parameter adds)
set -u
R=$(pwd); OUT=$R/gpurun_out/r05_call6; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; echo "tests exit $?" >> $OUT/gpu_tests.log
tail -n 4 $OUT/gpu_tests.log
GEN="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-graph --no-dit-train"
for i in 1 2; do
  timeout 300 python bench.py --steps 5 --warmup 2 $GEN >> $OUT/vae_steal.json 2>> $OUT/ab.err
  timeout 300 python bench.py --steps 5 --warmup 2 $GEN --no-grad-steal >> $OUT/vae_adds.json 2>> $OUT/ab.err
done
timeout 300 python bench.py --workload dit_train --batch 4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dit_train_steal.json 2>> $OUT/ab.err
python - <<PY
import json
for f in ("vae_steal","vae_adds","dit_train_steal"):
    for l in open("$OUT/%s.json"%f):
        r=json.loads(l); print(f, round(r["ms_per_step"],2), r["value"])
PY
tail -n 5 $OUT/ab.err
