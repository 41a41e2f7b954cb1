#!/bin/bash
B=$PWD/video-pre-training_amd/build
timeout 300 python -m pytest tests/test_gpu_training.py -q -k "conv_first" 2>&1 | grep "passed\|failed\|PARITY"
VPT_HIP_LIB=$B/libvpt_ref.so python tools/conv_first_bwd_bench.py 2>&1 | grep conv_first
python tools/conv_first_bwd_bench.py 2>&1 | grep conv_first
