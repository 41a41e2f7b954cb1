#!/bin/bash
# One GPU round: the -m gpu tests (all of them, failures collected), smoke(), bench.py.  tools/gpu_round.sh [tag] [pytest args...]
tag=${1:-r04_full}; shift
out=gpurun_out/$tag; mkdir -p $out
timeout 2400 python -m pytest tests/ -q -m gpu --durations=12 -s "$@" > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/gpu_tests.log | cut -c1-400 | tail -40
timeout 600 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -8 $out/smoke.log | cut -c1-400
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -c 9000 $out/bench.json; tail -5 $out/bench.err
