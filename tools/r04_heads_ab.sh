#!/bin/bash
# Round-4 experiment: the head-split / rotary / plane-layout epilogue without per-lane divisions (one per window) and with 16-byte rotary
# table loads, against the previous build of gemm.hip (tools/exp/libsat_amd_gemmbase.so).  Output: gpurun_out/r04_heads/
set -u
R=$(pwd)
OUT=$R/gpurun_out/r04_heads
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gemm_kernels.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests exit $?" >> $OUT/tests.log
timeout 300 python tools/qkv_bench.py > $OUT/qkv_new.jsonl 2> $OUT/qkv.err
SAT_EXP_LIB=tools/exp/libsat_amd_gemmbase.so timeout 300 python tools/qkv_bench.py > $OUT/qkv_base.jsonl 2>> $OUT/qkv.err
for i in 1 2; do
  timeout 300 python tools/bench_with_lib.py tools/exp/libsat_amd_gemmbase.so --workload dit_sample --no-cpu-baseline >> $OUT/dit_sample_base.json 2>> $OUT/ds.err
  timeout 300 python bench.py --workload dit_sample --no-cpu-baseline >> $OUT/dit_sample_new.json 2>> $OUT/ds.err
done
timeout 300 python tools/bench_with_lib.py tools/exp/libsat_amd_gemmbase.so --workload long_context --no-cpu-baseline >> $OUT/long_context_base.json 2>> $OUT/lc.err
timeout 300 python bench.py --workload long_context --no-cpu-baseline >> $OUT/long_context_new.json 2>> $OUT/lc.err
tail -3 $OUT/tests.log
cat $OUT/qkv_base.jsonl; echo; cat $OUT/qkv_new.jsonl
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/dit_sample_*.json") + glob.glob("$OUT/long_context_*.json")):
    for l in open(f):
        try:
            r = json.loads(l)
        except Exception:
            continue
        print(f.split('/')[-1], round(r["value"], 2), r["unit"])
PY
