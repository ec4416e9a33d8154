#!/bin/bash
# Round-4 experiment (one gpurun call): fp8 activations quantised per row in one pass (sat_quant_fp8_rows + row_alpha) against the
# per-tensor pair of passes (SAT_FP8_ROW_SCALES=0), on the N = 6145 sampler.  Output: gpurun_out/r04_fp8rows/
set -u
R=$(pwd)
OUT=$R/gpurun_out/r04_fp8rows
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gemm_kernels.py tests/test_long_context.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests exit $?" >> $OUT/tests.log
for v in 0 1 0 1; do
  SAT_FP8_ROW_SCALES=$v timeout 300 python bench.py --workload long_context --no-cpu-baseline >> $OUT/long_context_rows$v.json 2>> $OUT/long_context.err
done
timeout 400 python bench.py --workload long_context > $OUT/long_context_full.json 2> $OUT/long_context_full.err
tail -3 $OUT/tests.log
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/long_context_*.json")):
    for l in open(f):
        try:
            r = json.loads(l)
        except Exception:
            continue
        lc = r["long_context"]
        fp = lc["fp8_projections"]
        print(f.split('/')[-1], round(r["value"], 2), "fp8 frac", fp["frac"], "gemm ms", fp["total_ms"], "quant", fp.get("quant_launches"), fp.get("quant_total_ms"), "parity", json.dumps(lc.get("parity"))[:300])
PY
