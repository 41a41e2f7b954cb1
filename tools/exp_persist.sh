#!/bin/bash
out=gpurun_out/exp_zc; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py tests/test_gpu_training.py -q -k "conv3x3 or dgrad or conv_layer or bc_gradients" > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log; tail -4 $out/test.log
bash tools/exp_ab2.sh zc ref new
