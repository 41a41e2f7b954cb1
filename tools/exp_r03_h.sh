#!/bin/bash
B=$PWD/video-pre-training_amd/build
for r in 1 2; do for v in 0 1 16 17 8 4 31; do VPT_HIP_LIB=$B/libvpt_cfb$v.so timeout 120 python tools/conv_first_bwd_bench.py 1024 5 2>&1 | grep median; done; done
timeout 120 python tools/conv_first_bench.py 2>&1 | grep -i "ms" | head -5
