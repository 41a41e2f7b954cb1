"""Bit-stability of the kernels while OTHER processes use the same GPU (round 6).

Round 5's open defect -- the 2-rank BC gradients deviating by 1e-3 .. 1e-2 on most tensors in about one call in ten -- was a single kernel:
vpt_ln_bwd_kernel returned one row of dx shifted by a constant when another process's waves shared its SIMDs (the compiler's SLP-packed update of
the two row sums; video-pre-training_amd/build.py EXTRA_FLAGS has the analysis).  No single-process test can see that class of fault: a kernel launched
alone has its SIMDs to itself at these sizes.  tools/kernel_stress.py launches every candidate -- the LayerNorm backward shapes, the kernels around it,
the whole inference forward, the acting step, the IDM and one complete BC gradient computation -- thousands of times in three concurrent processes and
compares every output with the process's first one bit for bit."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernels_are_bit_stable_beside_other_processes():
    env = dict(os.environ, STRESS_PATHS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_stress.py"), "3", "1500"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = re.findall(r"rank (\d) (.+?): (\d+) mismatching outputs in (\d+) launches", p.stdout)
    assert len(rows) >= 3 * 12, p.stdout[-2000:]            # every case ran in every process (none skipped)
    assert "skipped" not in p.stdout, p.stdout[-2000:]
    bad = [(r, name, int(n), int(of)) for r, name, n, of in rows if int(n)]
    launches = sum(int(of) for _, _, _, of in rows)
    print(f"PARITY concurrency: {len(rows)} (process, kernel) pairs, {launches} launches compared bit for bit beside two other processes, {len(bad)} pairs with mismatches")
    assert not bad, bad
