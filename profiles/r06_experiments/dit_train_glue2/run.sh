mkdir -p gpurun_out/dit_glue2
timeout 900 python -m pytest tests/test_dit_parity.py tests/test_dit_train_step.py tests/test_gemm_kernels.py tests/test_dit_kernels.py tests/test_full_width.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/dit_glue2/tests.log
for arm in 1 0 1 0; do for b in 4 16; do
  timeout 300 python bench.py --workload dit_train --batch $b --steps 5 --warmup 2 --no-cpu-baseline --ops-set cast_pair=$arm 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'cast_pair': $arm, 'batch': $b, 'samples_per_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> gpurun_out/dit_glue2/ab.jsonl
done; done
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout -k 20 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dit_glue2/prof -- python $R/bench.py --workload dit_train --batch 4 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/dit_glue2/prof.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls gpurun_out/dit_glue2/prof/*/*.db | head -1) gpurun_out/dit_glue2/dit_train_b4_stats.csv
rm -rf gpurun_out/dit_glue2/prof
cat gpurun_out/dit_glue2/tests.log gpurun_out/dit_glue2/ab.jsonl; grep -i "layernorm\|cast" gpurun_out/dit_glue2/dit_train_b4_stats.csv
