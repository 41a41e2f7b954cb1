#!/bin/bash
# round-2 experiment: asymmetric wave priority between the two co-resident conv workgroups (VPT_CONV_PRIO)
out=gpurun_out/exp_prio; mkdir -p $out
VPT_CONV_PRIO=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "conv3x3" -x > $out/test_prio1.log 2>&1; echo "test rc=$?" >> $out/test_prio1.log
run() { tag=$1; shift; env "$@" timeout 300 python tools/conv_bench.py 512 5 > $out/$tag.log 2>&1; echo "== $tag"; cat $out/$tag.log | grep -v Warn; }
run base
run prio1 VPT_CONV_PRIO=1
run prio2 VPT_CONV_PRIO=2
run prio3 VPT_CONV_PRIO=3
run prio1_counted VPT_CONV_PRIO=1 VPT_CONV_COUNTED=1
run base2
run one_wg VPT_CONV_EXTRA_LDS=8192
run one_wg_noepi VPT_CONV_EXTRA_LDS=8192 VPT_CONV_ABLATE=1
run two_wg_noepi VPT_CONV_ABLATE=1
run prio1_b VPT_CONV_PRIO=1
