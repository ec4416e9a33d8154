#!/bin/bash
# Round-4 experiment: LayerNorm straight to fp8 rows (sat_layernorm_fwd_fp8) against LayerNorm -> bf16 -> sat_quant_fp8_rows
# (SAT_FP8_LN_FUSED=0) on the N = 6145 sampler.  Output: gpurun_out/r04_lnfp8/
set -u
R=$(pwd)
OUT=$R/gpurun_out/r04_lnfp8
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_dit_kernels.py tests/test_long_context.py tests/test_gemm_kernels.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests exit $?" >> $OUT/tests.log
for v in 0 1 0 1; do
  SAT_FP8_LN_FUSED=$v timeout 300 python bench.py --workload long_context --no-cpu-baseline >> $OUT/long_context_fused$v.json 2>> $OUT/long_context.err
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
        print(f.split('/')[-1], round(r["value"], 2), "parity", json.dumps(lc.get("parity"))[:260])
PY
