set -u
O=gpurun_out/call4; mkdir -p $O
timeout 600 python profiles/r06_experiments/prep_ablation/run.py > $O/prep_ablation.json 2> $O/prep_ablation.err
for i in 1 2; do
timeout 300 python bench.py --workload dit_train --no-cpu-baseline --batch 4 > $O/dit_train_new_$i.json 2> $O/dit_train_new_$i.err
timeout 300 python bench.py --workload dit_train --no-cpu-baseline --batch 4 --ops-set cross_kernels=0 > $O/dit_train_nocross_$i.json 2> $O/dit_train_nocross_$i.err
done
cat $O/prep_ablation.json; tail -3 $O/prep_ablation.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/call4/dit_train_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],2))
    except Exception as e: print(f,'ERR',e)
PY
