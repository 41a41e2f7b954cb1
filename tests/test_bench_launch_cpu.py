"""bench.py's `--gpus N` contract as a pure function, checked without a GPU: a plain `--gpus N` starts N ranks, a launcher
whose WORLD_SIZE differs from N is refused, N beyond the visible devices is refused unless the gloo test transport is named -- the script can
never print an N-GPU line from fewer ranks, nor a silent 1-GPU line for N > 1.  (The spawned run itself: tests/test_gpu_bench_dist.py.)"""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("_bench_for_test", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(bench)


def test_single_process_default():
    assert bench.launch_plan(1, {}, 1, "nccl") == ("run", 1)
    assert bench.launch_plan(1, {}, 8, "nccl") == ("run", 1)


def test_plain_gpus_n_spawns_its_own_ranks():
    assert bench.launch_plan(8, {}, 8, "nccl") == ("spawn", 8)
    assert bench.launch_plan(2, {}, 1, "gloo") == ("spawn", 2)          # the tests' transport: both ranks on one device


def test_under_a_launcher_the_world_must_match():
    assert bench.launch_plan(4, {"WORLD_SIZE": "4", "RANK": "2"}, 8, "nccl") == ("run", 4)
    what, msg = bench.launch_plan(8, {"WORLD_SIZE": "1"}, 8, "nccl")
    assert what == "refuse" and "WORLD_SIZE=1" in msg
    what, msg = bench.launch_plan(1, {"WORLD_SIZE": "2"}, 8, "nccl")
    assert what == "refuse" and "--gpus 1" in msg


def test_more_ranks_than_devices_is_refused_for_rccl():
    what, msg = bench.launch_plan(8, {}, 1, "nccl")
    assert what == "refuse" and "only 1 GPU(s) visible" in msg
    what, msg = bench.launch_plan(2, {"WORLD_SIZE": "2"}, 1, "nccl")
    assert what == "refuse"
    assert bench.launch_plan(0, {}, 1, "nccl")[0] == "refuse"
    assert bench.launch_plan(1, {}, 0, "nccl")[0] == "refuse"             # no GPU: there is no CPU fallback
