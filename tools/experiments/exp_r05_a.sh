#!/bin/bash
# round 5, call a: the tests touched by the first batch (split-K by name, fresh state under the auto-captured graph, sampler seeding,
# bench.py --gpus self-launch, gated competitive heads) + the full default bench line
out=gpurun_out/r05_a; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_policy.py tests/test_gpu_sampling.py tests/test_gpu_bench_dist.py tests/test_gpu_idm.py "tests/test_gpu_configs.py::test_input_driven_actions_on_competitive_heads" \
  "tests/test_gpu_configs.py::test_idm_4x_forward" tests/test_gpu_dropin.py -q -m gpu --durations=8 -s -x > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|Error" $out/gpu_tests.log | cut -c1-600 | tail -30
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -c 12000 $out/bench.json; tail -5 $out/bench.err
