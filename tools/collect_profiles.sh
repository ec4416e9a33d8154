#!/bin/bash
# Collects the round's committed evidence on the GPU box (run from the repo root; TAG names the round, e.g. r02):
#   gpurun_out/final/vae_stats.csv        rocprofv3 --kernel-trace --stats of `bench.py` (VAE generator train step only)
#   gpurun_out/final/real_stats.csv       the same with the alternating discriminator / generator step
#   gpurun_out/final/dit_{sample,train}_stats.csv
#   gpurun_out/final/pmc_{FETCH,WRITE}_SIZE.txt + pmc_traffic.json   per-kernel HBM counters of the same bench command (separate passes)
#   gpurun_out/final/bench_*.json         the bench lines themselves (with cpu_baseline)
#   gpurun_out/final/{gemm,k7,ru,disc,qkv}_bench.jsonl, pmc_k7_summary.txt   the kernel micro-benchmarks and SQ counters
set -u
R=$(pwd)
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GEN="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep"
rocprofv3 --kernel-trace --stats -d $OUT/vae -- python $R/bench.py --steps 3 --warmup 1 $GEN > $OUT/vae_prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/real -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-parity --no-long-context --no-batch-sweep > $OUT/real_prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/dit_sample -- python $R/bench.py --workload dit_sample --steps 10 --warmup 2 --no-cpu-baseline > $OUT/dit_sample_prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/dit_train -- python $R/bench.py --workload dit_train --steps 3 --warmup 1 --no-cpu-baseline > $OUT/dit_train_prof.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$ctr -- python $R/bench.py --steps 1 --warmup 1 $GEN > /dev/null 2>&1
done
cd $R
for w in vae real dit_sample dit_train; do python tools/rocpd_stats.py $(ls $OUT/$w/*/*.db | head -1) $OUT/${w}_stats.csv; done
for ctr in FETCH_SIZE WRITE_SIZE; do python tools/pmc_summary.py $OUT/pmc_$ctr sat_ > $OUT/pmc_$ctr.txt 2>&1; done
python tools/pmc_traffic_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 1 $GEN, MI355X (tools/collect_profiles.sh)" > $OUT/pmc_traffic.log 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +3M -delete
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/vae $OUT/real $OUT/dit_sample $OUT/dit_train
cp $OUT/pmc_traffic.json $R/profiles/r04_pmc_traffic.json      # bench.py reads it for roofline.traffic / roofline.hbm
python bench.py > $OUT/bench_vae_train.json 2> $OUT/bench_vae_train.err
python bench.py --workload dit_train --no-cpu-baseline > $OUT/bench_dit_train.json 2> /dev/null
SAT_TILES=0,4 SAT_SPLITS=2,3 python tools/gemm_bench.py 2050 4100 12290 > $OUT/gemm_bench.jsonl 2> /dev/null
python tools/k7_bench.py > $OUT/k7_bench.jsonl 2> /dev/null
python tools/ru_bench.py > $OUT/ru_bench.jsonl 2> /dev/null
python tools/disc_bench.py > $OUT/disc_bench.jsonl 2> /dev/null
python tools/qkv_bench.py > $OUT/qkv_bench.jsonl 2> /dev/null
python tools/attn_bench.py > $OUT/attn_bench.jsonl 2> /dev/null
python bench.py --ddp-single-rank --ddp-mode reduce_scatter --no-secondary --no-real-step --no-batch-sweep --no-parity --no-cpu-baseline > $OUT/bench_ddp_single_rank.json 2> /dev/null
bash tools/pmc_k7.sh > /dev/null 2>&1; cp gpurun_out/pmc_k7/summary.txt $OUT/pmc_k7_summary.txt
tail -c 3000 $OUT/bench_vae_train.json
