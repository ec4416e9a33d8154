#!/bin/bash
# (historical: the script of round 5, call 1 — the SAT_*_LEAN switches it sets were removed from the tree in the commit that followed this call)
# Round-5 opener: every arm written blind at the end of round 4 (lean attention fwd/bwd, lean GEMM K loops bf16 + fp8, two-row LayerNorm,
# the native RCCL exchange behind the C-ABI) executed on gfx950 and timed against the product kernels, in ONE call.
# Output: gpurun_out/r05_lean/.  A trimmed version of tools/r05_attn_lean_ab.sh (GPU-minute budget).
set -u
export SAT_TEST_LEAN_ARMS=1
R=$(pwd)
OUT=$R/gpurun_out/r05_lean
rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline"
timeout 900 python -m pytest tests/test_dit_kernels.py tests/test_gemm_kernels.py -m gpu -q > $OUT/tests.log 2>&1; echo "tests exit $?" >> $OUT/tests.log
timeout 600 python -m pytest tests/test_train_step.py -m gpu -q -k "native_exchange_gpu or single_rank_rccl or lean or graph_replay" > $OUT/native_tests.log 2>&1; echo "tests exit $?" >> $OUT/native_tests.log
timeout 120 python tools/fuzz_lean.py 1 30 gpu > $OUT/fuzz.log 2>&1; echo "fuzz exit $?" >> $OUT/fuzz.log
for i in 1 2; do
  SAT_ATTN_LEAN=0 timeout 200 python tools/attn_bench.py >> $OUT/attn_product.jsonl 2>> $OUT/attn.err
  SAT_ATTN_LEAN=1 timeout 200 python tools/attn_bench.py >> $OUT/attn_lean.jsonl 2>> $OUT/attn.err
done
for i in 1 2; do
  SAT_TILES=4,7,8 SAT_GEMM_LEAN=0 timeout 300 python tools/gemm_bench.py 2050 4100 >> $OUT/gemm_product.jsonl 2>> $OUT/gemm.err
  SAT_TILES=4,7,8 SAT_GEMM_LEAN=1 timeout 300 python tools/gemm_bench.py 2050 4100 >> $OUT/gemm_lean.jsonl 2>> $OUT/gemm.err
done
for i in 1 2; do
  timeout 300 python bench.py --workload dit_sample $B >> $OUT/dit_sample_product.json 2>> $OUT/ds.err
  SAT_GEMM_LEAN=1 SAT_ATTN_LEAN=1 SAT_LN_LEAN=1 timeout 300 python bench.py --workload dit_sample $B >> $OUT/dit_sample_all_lean.json 2>> $OUT/ds.err
done
SAT_ATTN_LEAN=1 timeout 300 python bench.py --workload dit_sample $B >> $OUT/dit_sample_attn_lean.json 2>> $OUT/ds.err
SAT_GEMM_LEAN=1 timeout 300 python bench.py --workload dit_sample $B >> $OUT/dit_sample_gemm_lean.json 2>> $OUT/ds.err
SAT_LN_LEAN=1 timeout 300 python bench.py --workload dit_sample $B >> $OUT/dit_sample_ln_lean.json 2>> $OUT/ds.err
timeout 300 python bench.py --workload dit_train $B >> $OUT/dit_train_product.json 2>> $OUT/dt.err
SAT_ATTN_LEAN=1 SAT_ATTN_BWD_LEAN=1 timeout 300 python bench.py --workload dit_train $B >> $OUT/dit_train_attn_lean.json 2>> $OUT/dt.err
SAT_GEMM_LEAN=1 SAT_ATTN_LEAN=1 SAT_ATTN_BWD_LEAN=1 SAT_LN_LEAN=1 timeout 300 python bench.py --workload dit_train $B >> $OUT/dit_train_all_lean.json 2>> $OUT/dt.err
timeout 300 python bench.py --workload long_context $B >> $OUT/long_context_product.json 2>> $OUT/lc.err
SAT_GEMM_LEAN=1 SAT_ATTN_LEAN=1 SAT_LN_LEAN=1 timeout 300 python bench.py --workload long_context $B >> $OUT/long_context_all_lean.json 2>> $OUT/lc.err
D="--ddp-single-rank --steps 3 --warmup 1 --no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-graph"
SAT_DDP_NATIVE=0 timeout 600 python bench.py $D >> $OUT/ddp_torch.json 2>> $OUT/ddp.err
SAT_DDP_NATIVE=1 timeout 600 python bench.py $D >> $OUT/ddp_native.json 2>> $OUT/ddp.err
tail -4 $OUT/tests.log $OUT/native_tests.log $OUT/fuzz.log
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    for l in open(f):
        try:
            r = json.loads(l)
        except Exception:
            continue
        ro = r.get("roofline") or {}
        lc = r.get("long_context") or {}
        print(f.split('/')[-1], round(r["value"], 3), r["unit"], "ms", r.get("ms_per_step"), "attn", ro.get("frac"), "proj", (ro.get("projections") or {}).get("frac"),
              "bwd", ro.get("backward"), "ddp", (r.get("config") or {}).get("ddp"))
PY
