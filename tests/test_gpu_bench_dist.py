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
    assert p.stdout.strip() == lines[0], p.stdout[-600:]          # ... and stdout holds nothing else (RCCL's version banner, written through C stdio, used to follow it)
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak" and rec["dtype"] == "bf16"
    assert rec["config"]["global_batch"] == 4 and "world_size=2" in rec["config"]["parallelism"] and "backend=gloo" in rec["config"]["parallelism"]
    assert "collective_ranks=2" in rec["config"]["parallelism"]
    assert rec["value"] > 0 and abs(rec["value"] - 2 * 2 * 16 * 2 / (rec["ms_per_step"] * 2e-3)) / rec["value"] < 1e-2   # whole-job frames / max-over-ranks time
    bc = rec["bc_step"]
    assert "error" not in bc, bc
    assert bc["global_batch"] == 4 and "all-reduce" in bc["allreduce"] and bc["ms_per_step"] > 0
    assert math.isfinite(bc["loss_first"]) and math.isfinite(bc["loss_last"]) and 5.0 < bc["loss_first"] < 25.0
    print("bench.py --gpus 2 (gloo, both ranks on cuda:0):", {k: rec[k] for k in ("value", "ms_per_step", "n_gpus")}, bc["ms_per_step"], bc["loss_first"], bc["loss_last"])


def _check_two_rank_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    assert stdout.strip() == lines[0], stdout[-600:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4
    assert "world_size=2" in rec["config"]["parallelism"] and "collective_ranks=2" in rec["config"]["parallelism"]
    return rec


def test_bench_gpus_2_launches_its_own_ranks():
    """plain `python bench.py --gpus 2` (no launcher) must run TWO ranks and say n_gpus = 2 -- it used to parse --gpus and
    never read it, so the same command line would have printed a 1-GPU number.  (gloo: both ranks share the test box's one GPU.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(VPT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--bc-steps", "1", "--bc-warmup", "1",
           "--batch", "2", "--seq", "16", "--model", "1x"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    rec = _check_two_rank_line(p.stdout)
    bc = rec["bc_step"]
    assert "error" not in bc, bc
    ad = bc["allreduce_detail"]
    assert ad["ranks"] == 2 and ad["backend"] == "gloo" and ad["ms_standalone"] > 0 and ad["bytes"] > 4 * 60e6     # the 1x model's 71 M parameters
    assert 0.0 <= ad["overlap_frac"] <= 1.0 and ad["step_ms_without_exchange"] > 0 and ad["exposed_ms"] >= 0       # (how much of the exchange is hidden)
    assert list(rec)[-2:] == ["value_blocks", "roofline"] or list(rec)[-1] in ("roofline", "cpu_baseline")        # the tail of the line holds the graded objects
    print("bench.py --gpus 2 (self-launched, gloo):", rec["value"], bc["ms_per_step"], ad)


def test_bench_refuses_more_ranks_than_gpus_and_a_mismatched_launcher():
    """... and must fail LOUDLY instead of printing an N = 1 line: --gpus beyond the visible devices without the gloo override, and a
    launcher whose WORLD_SIZE differs from --gpus."""
    n_dev = __import__("torch").cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "VPT_DIST_BACKEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_dev + 1), "--steps", "1", "--warmup", "0", "--bc-steps", "0", "--batch", "1", "--seq", "4",
           "--model", "1x", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], p.stdout[-2000:]
    assert "GPU(s) visible" in p.stderr
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", VPT_DIST_BACKEND="gloo")
    cmd[cmd.index("--gpus") + 1] = "2"
    p = subprocess.run(cmd, cwd=ROOT, env=env2, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], p.stdout[-2000:]
    assert "WORLD_SIZE" in p.stderr


def test_bench_stdout_is_exactly_one_json_line_with_the_rccl_probe():
    """The driver's single-GPU command runs the one-rank RCCL probe (bc_step.dp_path_one_rank); RCCL writes a version banner to descriptor 1 through C stdio, which
    a pipe delivers at exit -- behind the line.  bench.py points descriptor 1 at stderr and writes the line to the saved descriptor."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--bc-steps", "1", "--bc-warmup", "1", "--batch", "2", "--seq", "16",
           "--model", "1x", "--no-cpu-baseline", "--no-ingest", "--value-blocks", "0"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = p.stdout.strip()
    assert out.startswith("{") and "\n" not in out, p.stdout[-800:]
    rec = json.loads(out)
    probe = rec["bc_step"].get("dp_path_one_rank")
    assert probe and "error" not in probe and probe["ranks"] == 1, probe          # the probe ran (so RCCL was initialised in this process)
