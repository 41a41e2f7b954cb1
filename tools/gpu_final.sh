#!/bin/bash
# End-of-round evidence in one call: all GPU tests, smoke, bench, profiles.
bash tools/gpu_round.sh r03_final
bash tools/profile_round.sh r03 > gpurun_out/r03_final/profile_round.log 2>&1; tail -3 gpurun_out/r03_final/profile_round.log
