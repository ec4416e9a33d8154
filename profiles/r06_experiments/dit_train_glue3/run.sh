mkdir -p gpurun_out/dit_glue3
timeout 900 python -m pytest tests/test_dit_parity.py tests/test_dit_train_step.py tests/test_dit_kernels.py tests/test_full_width.py tests/test_boundary.py tests/test_stft_parity.py tests/test_train_step.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/dit_glue3/tests.log
for arm in 1 0 1 0; do for b in 4 16; do
  timeout 300 python bench.py --workload dit_train --batch $b --steps 5 --warmup 2 --no-cpu-baseline --ops-set ln_residual=$arm 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'ln_residual': $arm, 'batch': $b, 'samples_per_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> gpurun_out/dit_glue3/ab.jsonl
done; done
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-dit-train > gpurun_out/dit_glue3/bench_vae.json 2> gpurun_out/dit_glue3/bench_vae.err
cat gpurun_out/dit_glue3/tests.log gpurun_out/dit_glue3/ab.jsonl; head -c 400 gpurun_out/dit_glue3/bench_vae.json
