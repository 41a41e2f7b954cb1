"""bench.py's N > 1 branch, executed: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` exactly as the
driver launches it for the scaling runs, except that the 1-GPU test box cannot host two RCCL ranks (RCCL refuses two ranks on one
device): VPT_DIST_BACKEND=gloo keeps every line of the distributed branch -- rendezvous from the torchrun environment, per-rank
build barrier, per-rank batches, barrier + max-over-ranks timing, the BC step's bucketed gradient all-reduce, the rank-0 JSON
line -- and only swaps the transport.  No scaling number is claimed from this."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_through_torchrun():
    env = dict(os.environ, VPT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--bc-steps", "1", "--bc-warmup", "1", "--batch", "2", "--seq", "16", "--model", "1x"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 prints ONE line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak" and rec["dtype"] == "bf16"
    assert rec["config"]["global_batch"] == 4 and "world_size=2" in rec["config"]["parallelism"] and "backend=gloo" in rec["config"]["parallelism"]
    assert rec["value"] > 0 and abs(rec["value"] - 2 * 2 * 16 * 2 / (rec["ms_per_step"] * 2e-3)) / rec["value"] < 1e-2   # whole-job frames / max-over-ranks time
    bc = rec["bc_step"]
    assert "error" not in bc, bc
    assert bc["global_batch"] == 4 and "all-reduce" in bc["allreduce"] and bc["ms_per_step"] > 0
    assert math.isfinite(bc["loss_first"]) and math.isfinite(bc["loss_last"]) and 5.0 < bc["loss_first"] < 25.0
    print("bench.py --gpus 2 (gloo, both ranks on cuda:0):", {k: rec[k] for k in ("value", "ms_per_step", "n_gpus")}, bc["ms_per_step"], bc["loss_first"], bc["loss_last"])
