#!/bin/bash
out=gpurun_out/exp_trace2; mkdir -p $out
A=$PWD/video-pre-training_amd/build/libvpt_abl1.so
VPT_CONV_ANTIPHASE_US=20 python tools/conv_trace.py 64 128 256 0 512 2>&1 | grep -v amdgpu.ids > $out/m0_ap.log; echo "== mode0 antiphase"; sed -n 1,9p $out/m0_ap.log; grep -A12 "CU key 0x0" $out/m0_ap.log
VPT_HIP_LIB=$A VPT_CONV_ANTIPHASE_US=20 python tools/conv_trace.py 64 128 256 0 512 2>&1 | grep -v amdgpu.ids > $out/m0_ap_nostore.log; echo "== mode0 antiphase, no stores"; sed -n 1,9p $out/m0_ap_nostore.log; grep -A12 "CU key 0x0" $out/m0_ap_nostore.log
