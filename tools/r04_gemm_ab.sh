#!/bin/bash
# Round-4 projection-GEMM experiment (one gpurun call): correctness of the new tiles on the hardware, the micro-benchmark over every
# shipped tile, and the in-model A/B of the tile policy (SAT_GEMM_POLICY=r3 = round 3's rule) on the sampler, the fp8 long-context
# sampler and the DiT train step.  Output: gpurun_out/r04_gemm/
set -u
R=$(pwd)
OUT=$R/gpurun_out/r04_gemm
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gemm_kernels.py tests/test_long_context.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests exit $?" >> $OUT/tests.log
SAT_TILES=0,4,7,8 SAT_SPLITS=2,3,4 timeout 600 python tools/gemm_bench.py 2050 4100 12290 > $OUT/gemm_bench.jsonl 2> $OUT/gemm_bench.err
for pol in r3 model r3 model; do
  SAT_GEMM_POLICY=$pol timeout 300 python bench.py --workload dit_sample --no-cpu-baseline >> $OUT/dit_sample_$pol.json 2>> $OUT/dit_sample.err
done
for pol in r3 model; do
  SAT_GEMM_POLICY=$pol timeout 300 python bench.py --workload long_context --no-cpu-baseline >> $OUT/long_context_$pol.json 2>> $OUT/long_context.err
  SAT_GEMM_POLICY=$pol timeout 300 python bench.py --workload dit_train --no-cpu-baseline >> $OUT/dit_train_$pol.json 2>> $OUT/dit_train.err
done
tail -3 $OUT/tests.log
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*_r3.json") + glob.glob("$OUT/*_model.json")):
    for l in open(f):
        try:
            r = json.loads(l)
        except Exception:
            continue
        print(f.split('/')[-1], round(r["value"], 2), r["unit"], "proj", r.get("roofline", {}).get("projections", {}).get("frac"))
PY
