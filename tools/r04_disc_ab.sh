#!/bin/bash
# Round-4 experiment (one gpurun call): the discriminator's 3 x 3 layers on a three-stage LDS-DMA ring (sat_disc_conv_kernel<3>) against
# the round-3 build of disc_conv.hip (tools/exp/libsat_amd_discbase.so): correctness on the hardware, the per-layer micro-benchmark, the
# real (alternating discriminator / generator) step; the DiT train step's in-model time per projection shape.  Output: gpurun_out/r04_disc/
set -u
R=$(pwd)
OUT=$R/gpurun_out/r04_disc
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_disc_conv.py tests/test_discriminator.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests exit $?" >> $OUT/tests.log
SAT_DISC_OLD=0 timeout 300 python tools/disc_bench.py 1024 256 > $OUT/disc_bench_new.jsonl 2> $OUT/disc_bench.err
SAT_DISC_OLD=0 SAT_EXP_LIB=tools/exp/libsat_amd_discbase.so timeout 300 python tools/disc_bench.py 1024 256 > $OUT/disc_bench_base.jsonl 2>> $OUT/disc_bench.err
REAL="--no-cpu-baseline --no-secondary --no-parity --no-long-context --no-batch-sweep --no-graph"
timeout 400 python bench.py --steps 2 --warmup 1 $REAL > $OUT/real_new.json 2> $OUT/real_new.err
timeout 400 python tools/bench_with_lib.py tools/exp/libsat_amd_discbase.so --steps 2 --warmup 1 $REAL > $OUT/real_base.json 2> $OUT/real_base.err
for pol in r3 model; do
  SAT_BENCH_GEMM_SHAPES=1 SAT_GEMM_POLICY=$pol timeout 300 python bench.py --workload dit_train --no-cpu-baseline > $OUT/dit_train_$pol.json 2>> $OUT/dit_train.err
done
tail -3 $OUT/tests.log
python - <<PY
import json, glob
for f in ("disc_bench_base", "disc_bench_new"):
    for l in open("$OUT/" + f + ".jsonl"):
        r = json.loads(l)
        print(f, r["n_fft"], r["kernel"], "conv", r["conv_emit_ms"], r["conv_emit_tf"], "dgrad", r["dgrad_ms"], r["dgrad_tf"], "wgrad", r["wgrad_ms"])
for f in ("real_base", "real_new"):
    for l in open("$OUT/" + f + ".json"):
        r = json.loads(l)
        print(f, round(r["ms_per_step"], 2), "real", r.get("real_step_ms"), r.get("real_step_samples_per_s"), json.dumps(r["config"].get("real_step", {}).get("discriminator_kernels"))[:600])
sh = {}
for pol in ("r3", "model"):
    for l in open("$OUT/dit_train_%s.json" % pol):
        r = json.loads(l)
        print("dit_train", pol, round(r["value"], 2), r["roofline"]["projections"]["frac"])
        for row in r["roofline"]["projections"].get("shapes", []):
            k = tuple(row["key"][:7])
            sh.setdefault(k, {})[pol] = (row["key"][7], row["launches"], row["avg_us"], row["total_ms"])
for k, v in sorted(sh.items(), key=lambda kv: -max(x[3] for x in kv[1].values())):
    print(k, v)
PY
