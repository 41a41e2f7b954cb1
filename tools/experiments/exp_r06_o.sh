#!/bin/bash
# Round 6, call O: end-of-round evidence on the final tree -- all GPU tests, smoke, the driver's bench command, rocprofv3 kernel statistics of the forward
# (single CNN stream, both formats), forward + BC, the single-stream BC step, the acting-step latency.
out=gpurun_out/r06o; mkdir -p $out
timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $out/gpu_tests.log 2>&1; tail -2 $out/gpu_tests.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err; tail -c 600 $out/bench_line.json
VPT_PROF_SKIP_PMC=1 bash tools/profile_round.sh r06g > $out/profile_round.log 2>&1; tail -3 $out/profile_round.log
