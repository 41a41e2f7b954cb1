"""N > 1 host logic with two gloo processes on CPU: sequence sharding (no data-path collective in forward),
the max-over-ranks timing rule of bench.py, and the bucketed gradient all-reduce."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import vpt_amd  # noqa: F401
from vpt_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # sharding: 5 sequences over 2 ranks -> 3 + 2, disjoint, covering
    img = torch.arange(5 * 2).view(5, 2, 1)
    first = torch.zeros(5, 2, dtype=torch.bool)
    mine, f = D.shard_batch(img, first, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine[:, 0, 0].tolist())
    assert sorted(sum(gathered, [])) == [0, 2, 4, 6, 8] and len(mine) == (3 if rank == 0 else 2) and len(f) == len(mine)
    # timing rule
    assert D.max_over_ranks(1.0 + rank) == float(world)
    # bucketed all-reduce (3 tensors, tiny buckets -> several collectives), averaged
    g = torch.Generator().manual_seed(7)
    base = [torch.randn(1000, generator=g), torch.randn(33, 7, generator=g), torch.randn(5, generator=g)]
    mine_g = [b * (rank + 1) for b in base]
    n = D.bucketed_all_reduce_(mine_g, bucket_bytes=2048, average=True)
    assert n >= 2
    for t, b in zip(mine_g, base):
        assert torch.allclose(t, b * (sum(range(1, world + 1)) / world), atol=1e-6)
    # the persistent flat arena of the data-parallel BC step: adopt (one copy in), in-place all-reduce of contiguous slices, views out
    names, shapes = ["w0", "b0", "w1", "unused"], [torch.Size([300, 7]), torch.Size([7]), torch.Size([1000]), torch.Size([3])]
    arena = D.GradArena(names, shapes, "cpu", bucket_bytes=4096)
    assert len(arena.buckets) >= 2 and arena.buckets[0][0] == 0 and arena.buckets[-1][1] == arena.flat.numel()
    for step in range(2):                      # the arena is reused step after step
        gs = {"w0": torch.ones(300, 7) * (rank + 1 + step), "b0": torch.arange(7.0) * (rank + 1), "w1": base[0].clone() * (rank + 1)}
        arena.adopt(gs)
        assert gs["w1"].data_ptr() == arena.view("w1").data_ptr() and float(gs["unused"].abs().sum()) == 0.0
        D.bucketed_all_reduce_finish(arena.all_reduce_start())
        tot = sum(range(1, world + 1))
        assert torch.allclose(gs["w1"], base[0] * tot, atol=1e-5) and torch.allclose(gs["b0"], torch.arange(7.0) * tot)
        assert torch.allclose(gs["w0"], torch.full((300, 7), float(tot + world * step)))
    out[rank] = True
    dist.destroy_process_group()


def test_two_rank_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert len(out) == world


def test_shard_range_properties():
    for n in (1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
