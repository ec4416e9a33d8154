set -u
O=gpurun_out/call6; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_kernels.py tests/test_vae_parity.py -x -q -m gpu -k "edge or vae or golden" 2>&1 | tail -6 > $O/tests.log
G="--no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-dit-train --no-graph --steps 6 --warmup 2"
for i in 1 2; do
timeout 300 python bench.py $G > $O/vae_edge_$i.json 2> $O/vae_edge_$i.err
timeout 300 python bench.py $G --ops-set edge_convs=0 > $O/vae_noedge_$i.json 2> $O/vae_noedge_$i.err
done
cd /tmp && export TMPDIR=/tmp
timeout -k 20 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/vae -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-step --no-secondary --no-parity --no-long-context --no-batch-sweep --no-dit-train > $GRAFT_REPO_ROOT/$O/vae_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(ls $O/vae/*/*.db | head -1) $O/vae_stats.csv
rm -rf $O/vae
cat $O/tests.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/call6/vae_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],2), d['roofline']['frac'])
    except Exception as e: print(f,'ERR',e)
PY
grep -i "edge\|k7_kernel\|k7_planes\|wgrad7_bf16x3_kernel" $O/vae_stats.csv
