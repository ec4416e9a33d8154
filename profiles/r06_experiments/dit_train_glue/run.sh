mkdir -p gpurun_out/dit_glue
timeout 900 python -m pytest tests/test_dit_parity.py tests/test_dit_train_step.py tests/test_gemm_kernels.py tests/test_dit_kernels.py tests/test_full_width.py tests/test_long_context.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/dit_glue/tests.log
for arm in 1 0 1 0; do for b in 4 16; do
  timeout 300 python bench.py --workload dit_train --batch $b --steps 5 --warmup 2 --no-cpu-baseline --ops-set train_fused_nodes=$arm 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'arm': $arm, 'batch': $b, 'samples_per_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> gpurun_out/dit_glue/ab.jsonl
done; done
cat gpurun_out/dit_glue/tests.log gpurun_out/dit_glue/ab.jsonl
