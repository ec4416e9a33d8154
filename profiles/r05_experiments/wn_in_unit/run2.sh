#!/bin/bash
# round 5, call 11 (the last GPU minutes): the 16-wave sat_wn_grad_splits (z parts, 16-byte loads) — kernel tests, then A / B / A / B against separate weight-norm nodes
set -u
R=$(pwd); OUT=$R/gpurun_out/r05_call11; rm -rf $OUT; mkdir -p $OUT
timeout 240 python -m pytest tests/test_conv_kernels.py tests/test_train_step.py -m gpu -q -x -k "wn_grad_from_slabs or multi_copy or ru_k1 or alternating or graphed" > $OUT/tests.log 2>&1; echo "tests exit $?" >> $OUT/tests.log
tail -n 3 $OUT/tests.log
GEN="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-dit-train"
for i in 1 2; do
  timeout 200 python bench.py --steps 5 --warmup 2 $GEN >> $OUT/vae_wn_in_unit.json 2>> $OUT/ab.err
  timeout 200 python bench.py --steps 5 --warmup 2 $GEN --ops-set wn_fused=0 >> $OUT/vae_wn_separate.json 2>> $OUT/ab.err
done
python - <<PY
import json
for f in ("vae_wn_in_unit","vae_wn_separate"):
    for l in open("$OUT/%s.json"%f):
        r=json.loads(l); print(f, round(r["ms_per_step"],2), r["config"]["launch"]["ms_per_step"], "k7q frac", round(r["roofline"]["frac"],3))
PY
