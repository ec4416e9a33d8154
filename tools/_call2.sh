set -u
O=gpurun_out/call2; mkdir -p $O
timeout 900 python -m pytest tests/test_dit_kernels.py tests/test_gemm_kernels.py tests/test_long_context.py tests/test_dit_parity.py -x -q -m gpu 2>&1 | tail -25 > $O/tests_a.log
timeout 300 python tools/cross_attn_bench.py > $O/cross_attn_bench.jsonl 2> $O/cross_attn_bench.err
timeout 600 python bench.py --workload long_context > $O/bench_long_context.json 2> $O/bench_long_context.err
timeout 400 python bench.py --workload dit_train --no-cpu-baseline > $O/bench_dit_train.json 2> $O/bench_dit_train.err
timeout 400 python bench.py --workload dit_train --no-cpu-baseline --ops-set cross_kernels=0 > $O/bench_dit_train_nocross.json 2> $O/bench_dit_train_nocross.err
timeout 300 python bench.py --workload dit_sample --no-cpu-baseline > $O/bench_dit_sample.json 2> $O/bench_dit_sample.err
timeout 900 python -m pytest tests/test_headline_parity.py -x -q -m gpu -s 2>&1 | tail -25 > $O/tests_headline.log
cat $O/tests_a.log | tail -5; cat $O/cross_attn_bench.jsonl; tail -3 $O/tests_headline.log
