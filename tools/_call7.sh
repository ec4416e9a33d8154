set -u
O=gpurun_out/call7; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_kernels.py -x -q -m gpu -k "edge" 2>&1 | tail -40 > $O/tests_edge.log
timeout 900 python -m pytest tests/test_conv_kernels.py tests/test_vae_parity.py -q -m gpu -k "not edge_conv_gpu" 2>&1 | tail -8 > $O/tests.log
G="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-dit-train --no-graph --steps 6 --warmup 2"
for i in 1 2; do
timeout 300 python bench.py $G > $O/vae_persist_$i.json 2> $O/vae_persist_$i.err
timeout 300 python bench.py $G --ops-set k7q_persist=0 > $O/vae_classic_$i.json 2> $O/vae_classic_$i.err
done
cat $O/tests_edge.log | tail -30; cat $O/tests.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/call7/vae_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],2), d['roofline']['frac'], d['roofline']['avg_launch_ms'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-600:])
PY
