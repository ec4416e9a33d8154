#!/bin/bash
# Round-4 experiment (one gpurun call): the vector-staged strided weight-gradient kernel (sat_wgrad_small_bf16x3_kernel<2>): round-3 build
# of the file (tools/exp/libsat_amd_wgbase.so) / quads loaded in the stage (…wgnocarry.so) / quads carried across the MFMA stage (product),
# on the generator step; per-launch durations by grid shape; the training-shape GEMM micro-benchmark.  Output: gpurun_out/r04_wgs/
set -u
R=$(pwd)
OUT=$R/gpurun_out/r04_wgs
rm -rf $OUT; mkdir -p $OUT
GEN="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-graph"
timeout 900 python -m pytest tests/test_conv_kernels.py tests/test_vae_parity.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests exit $?" >> $OUT/tests.log
timeout 300 python bench.py --steps 4 --warmup 1 $GEN > $OUT/carry_1.json 2> $OUT/carry_1.err
timeout 300 python tools/bench_with_lib.py tools/exp/libsat_amd_wgbase.so --steps 4 --warmup 1 $GEN > $OUT/base_1.json 2> $OUT/base_1.err
timeout 300 python tools/bench_with_lib.py tools/exp/libsat_amd_wgnocarry.so --steps 4 --warmup 1 $GEN > $OUT/nocarry_1.json 2> $OUT/nocarry_1.err
timeout 300 python bench.py --steps 4 --warmup 1 $GEN > $OUT/carry_2.json 2> $OUT/carry_2.err
SAT_BENCH_TRAIN=1 SAT_TILES=0,4,7,8 timeout 300 python tools/gemm_bench.py > $OUT/gemm_train.jsonl 2> $OUT/gemm_train.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/vae -- python $R/bench.py --steps 3 --warmup 1 $GEN > $OUT/vae_prof.log 2>&1
cd $R
DB=$(ls $OUT/vae/*/*.db | head -1)
python tools/rocpd_stats.py $DB $OUT/vae_stats.csv
python tools/rocpd_launches.py $DB wgrad_small > $OUT/launches.txt
find $OUT -name "*.db" -delete; rm -rf $OUT/vae
tail -3 $OUT/tests.log
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    for l in open(f):
        try:
            r = json.loads(l)
        except Exception:
            continue
        ks = r.get("roofline", {}).get("all_conv_kernels") or []
        short = {d["kernel"].replace("sat_", "").replace("_kernel", "")[:30]: round(d["total_ms"] / r["steps"], 2) for d in ks}
        print(f.split('/')[-1], round(r["value"], 3), r["unit"], round(r["ms_per_step"], 2), "ms", short)
PY
cat $OUT/launches.txt
cat $OUT/gemm_train.jsonl
