#!/bin/bash
# Collects the round's committed evidence on the GPU box (run from the repo root) into gpurun_out/final/:
#   gpu_tests.log                         tail of `pytest tests -m gpu` + smoke()
#   {vae,real,dit_sample,long_context}_stats.csv   rocprofv3 --kernel-trace --stats of the bench.py command named below
#   vae_launches_by_grid.txt              the conv-stack kernels' launches grouped by grid shape
#   pmc_{FETCH,WRITE}_SIZE.txt + pmc_traffic.json   per-kernel HBM counters of the same bench command (separate passes)
#   pmc_sq_{vae,dit_train,dit_sample,long_context}.json   MFMA-busy / VALU:MFMA / wait fractions per kernel (tools/pmc_mfma_busy.py)
#   dit_train_b{4,16}_stats.csv           rocprofv3 kernel summary of bench.py --workload dit_train
#   bench_*.json                          the bench lines themselves (the default line with cpu_baseline, parity, real step, secondary ...)
# One call of ~20 GPU-minutes; A/B experiments live with their run scripts under profiles/r0N_experiments/.
set -u
R=$(pwd)
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
# a box whose GPU faults on everything (round 5, call 7) must not eat the budget: one smoke() first, stop if it fails
if ! timeout 300 python __graft_entry__.py smoke > $OUT/smoke_first.log 2>&1; then tail -5 $OUT/smoke_first.log; echo "smoke failed on this box: stopping"; exit 1; fi
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25; timeout 200 python __graft_entry__.py smoke 2>&1 | tail -3 ) > $OUT/gpu_tests.log 2>&1
cd /tmp && export TMPDIR=/tmp
GEN="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep"
timeout -k 20 400 rocprofv3 --kernel-trace --stats -d $OUT/vae -- python $R/bench.py --steps 3 --warmup 1 $GEN > $OUT/vae_prof.log 2>&1
timeout -k 20 400 rocprofv3 --kernel-trace --stats -d $OUT/real -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-parity --no-long-context --no-batch-sweep --no-graph > $OUT/real_prof.log 2>&1
timeout -k 20 400 rocprofv3 --kernel-trace --stats -d $OUT/dit_sample -- python $R/bench.py --workload dit_sample --steps 10 --warmup 2 --no-cpu-baseline > $OUT/dit_sample_prof.log 2>&1
timeout -k 20 400 rocprofv3 --kernel-trace --stats -d $OUT/long_context -- python $R/bench.py --workload long_context --steps 4 --warmup 1 --no-cpu-baseline > $OUT/long_context_prof.log 2>&1
for b in 4 16; do
  timeout -k 20 400 rocprofv3 --kernel-trace --stats -d $OUT/dit_train_b$b -- python $R/bench.py --workload dit_train --batch $b --steps 3 --warmup 1 --no-cpu-baseline > $OUT/dit_train_b${b}_prof.log 2>&1
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 20 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$ctr -- python $R/bench.py --steps 1 --warmup 1 $GEN --no-graph > /dev/null 2>&1
done
# MFMA utilisation of the kernels that ship (north_star: "rocprof-reported HBM GB/s and MFMA utilisation"): SQ counters over one generator
# step and one DiT train step (B = 4) / one sampler step, own passes, --kernel-trace only
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
timeout -k 20 400 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_vae -- python $R/bench.py --steps 1 --warmup 1 $GEN --no-graph > /dev/null 2>&1
timeout -k 20 400 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_dit_train -- python $R/bench.py --workload dit_train --batch 4 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout -k 20 400 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_dit_sample -- python $R/bench.py --workload dit_sample --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout -k 20 400 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_long_context -- python $R/bench.py --workload long_context --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
for w in pmc_sq_vae pmc_sq_dit_train pmc_sq_dit_sample pmc_sq_long_context; do python tools/pmc_mfma_busy.py $OUT/$w "$w: rocprofv3 --pmc $SQ --kernel-trace (own pass), MI355X (tools/collect_profiles.sh)" > $OUT/${w}.json 2> $OUT/${w}.err; rm -rf $OUT/$w; done
for w in vae real dit_sample long_context dit_train_b4 dit_train_b16; do python tools/rocpd_stats.py $(ls $OUT/$w/*/*.db | head -1) $OUT/${w}_stats.csv; done
for k in wgrad_small wgrad7 conv1d_bf16x3_kernel k7q ru_k1_bwd stft; do echo "=== $k"; python tools/rocpd_launches.py $(ls $OUT/vae/*/*.db | head -1) $k; done > $OUT/vae_launches_by_grid.txt 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do python tools/pmc_summary.py $OUT/pmc_$ctr sat_ > $OUT/pmc_$ctr.txt 2>&1; done
python tools/pmc_traffic_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 1 $GEN --no-graph, MI355X (tools/collect_profiles.sh)" > $OUT/pmc_traffic.log 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +3M -delete
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/vae $OUT/real $OUT/dit_sample $OUT/long_context $OUT/dit_train_b4 $OUT/dit_train_b16
# (these copies live on the GPU box only — bench.py reads them there for roofline.traffic / mfma_utilisation; gpurun brings back gpurun_out/ alone, so
#  after the call copy gpurun_out/final/{pmc_traffic,pmc_sq_*}.json over profiles/r06_pmc_{traffic,mfma_*}.json by hand, with the other files)
cp $OUT/pmc_traffic.json $R/profiles/r06_pmc_traffic.json; cp $OUT/pmc_sq_vae.json $R/profiles/r06_pmc_mfma_vae.json; cp $OUT/pmc_sq_dit_train.json $R/profiles/r06_pmc_mfma_dit_train.json; cp $OUT/pmc_sq_dit_sample.json $R/profiles/r06_pmc_mfma_dit_sample.json; cp $OUT/pmc_sq_long_context.json $R/profiles/r06_pmc_mfma_long_context.json      # bench.py reads it for roofline.traffic / roofline.hbm
timeout 900 python bench.py > $OUT/bench_vae_train.json 2> $OUT/bench_vae_train.err
timeout 300 python bench.py --workload dit_train --no-cpu-baseline > $OUT/bench_dit_train.json 2> /dev/null
tail -8 $OUT/gpu_tests.log
tail -c 2500 $OUT/bench_vae_train.json
