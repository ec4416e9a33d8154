O=gpurun_out/call9; mkdir -p $O
timeout 600 python profiles/r06_experiments/fp8_policy/run.py > $O/fp8_policy.jsonl 2> $O/fp8_policy.err
cat $O/fp8_policy.jsonl; tail -3 $O/fp8_policy.err
timeout 600 python -m pytest tests/test_long_context.py -x -q -m gpu -s -k "fp8" 2>&1 | grep -i "N=6145\|policy\|passed\|failed\|Error" | tail -12
