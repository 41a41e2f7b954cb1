"""The residual epilogues of vpt_conv3x3_kernel take their ReLU from the CLAMP modifier of an inline-assembly v_pk_fma_f32 (vpt_common.h:pk_fma_clamp01;
round 5: -17 % epilogue instructions).  Inline assembly is invisible to the compiler's hazard recogniser, so the claim "bit-identical to the plain
fp32 v_max ReLU" is held two ways: tests/test_kernel_resources_cpu.py checks in the ISA that every such instruction sits >= 32 instructions behind the
last MFMA, and THIS test runs tools/conv_hash.py's 28 hashed cases (modes 0 / 1 / 5, the policy's layer shapes and ragged ones, zeros, negatives
everywhere, 16-bit extremes, the frame statistics) through the shipped library and through build/libvpt_noclamp.so (the same sources with
-DVPT_EPI_NO_CLAMP_RELU=1, built by build.py) and compares the hashes (a tool promoted to a test)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOCLAMP = os.path.join(ROOT, "video-pre-training_amd", "build", "libvpt_noclamp.so")


def _hashes(env_extra):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "conv_hash.py")], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    return [ln for ln in p.stdout.splitlines() if ln.strip()]


def test_clamp_relu_epilogue_is_bit_identical_to_the_plain_relu_build():
    import __graft_entry__ as ge
    ge.build()
    assert os.path.exists(NOCLAMP), "build.py did not produce the A/B library (no hipcc on this box and none shipped)"
    a, b = _hashes({}), _hashes({"VPT_HIP_LIB": NOCLAMP})
    import re
    n_hash = sum(len(re.findall(r"\b[0-9a-f]{16}\b", ln)) for ln in a)
    assert len(a) >= 20 and len(a) == len(b) and n_hash >= 56, (len(a), len(b), n_hash)      # (tensor + statistics hashes of >= 28 launches)
    diff = [(x, y) for x, y in zip(a, b) if x != y]
    print(f"PARITY conv epilogue clamp-ReLU vs plain ReLU build: {len(a)} lines / {n_hash} hashes (outputs and frame statistics), {len(diff)} lines differ")
    assert not diff, diff[:4]
