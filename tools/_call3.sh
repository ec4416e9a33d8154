set -u
O=gpurun_out/call3; mkdir -p $O
timeout 600 python -m pytest tests/test_dit_kernels.py -x -q -m gpu -k attention 2>&1 | tail -5 > $O/tests_attn.log
timeout 300 python tools/cross_attn_bench.py > $O/cross_attn_bench.jsonl 2> $O/cross_attn_bench.err
timeout 300 python tools/attn_bench.py > $O/attn_bench.jsonl 2> $O/attn_bench.err
timeout 300 python bench.py --workload long_context --no-cpu-baseline > $O/bench_lc_auto.json 2> $O/bench_lc_auto.err
timeout 300 python bench.py --workload long_context --no-cpu-baseline --ops-set attn_q64=0 > $O/bench_lc_q32.json 2> $O/bench_lc_q32.err
timeout 400 python bench.py --workload dit_train --no-cpu-baseline > $O/bench_dit_train.json 2> $O/bench_dit_train.err
timeout 400 python bench.py --workload dit_train --no-cpu-baseline --ops-set attn_q64=0 --ops-set cross_kernels=0 > $O/bench_dit_train_old.json 2> $O/bench_dit_train_old.err
cat $O/tests_attn.log; cat $O/cross_attn_bench.jsonl $O/attn_bench.jsonl; tail -2 $O/*.err
