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


# ---------------------------------------------------------------------------------------------------------
# BCTrainer.reduced_loss_and_grads -- the REAL host logic of the data-parallel step -- on three gloo ranks with uneven shards (B = 5 -> 2, 2, 1)
# and a synthetic per-frame gradient in place of the HIP backward: the frame-weighted loss, the two arenas, the early / late exchange, and the path
# where one rank fails before or after its early exchange has started.
# ---------------------------------------------------------------------------------------------------------
_NAMES = ["net.lastlayer.layer.weight", "pi_head.buttons.linear_layer.bias", "net.img_process.cnn.stacks.0.firstconv.layer.weight", "net.img_process.cnn.dense.layer.weight"]
_SHAPES = [(64, 32), (17,), (8, 3, 3, 3), (16, 50)]


def _fake_trainer(fail=None):
    from vpt_amd.training import BCTrainer
    tr = object.__new__(BCTrainer)
    tr.params = {n: torch.nn.Parameter(torch.zeros(s)) for n, s in zip(_NAMES, _SHAPES)}
    tr.trainable, tr._arenas, tr.scaled, tr.loss_scale = list(_NAMES), None, False, 1.0
    g = torch.Generator().manual_seed(3)
    base = {n: torch.randn(s, generator=g) for n, s in zip(_NAMES, _SHAPES)}

    def loss_and_grads(img, first, state, ab, ac, global_frames=None, on_trunk_grads=None, unscaled=True, debug=None):
        if fail == "early":
            raise MemoryError("synthetic failure before the early exchange")
        w = float(img.double().sum())                     # frame f carries weight f: d loss / d theta = sum_f f * base / global_frames
        grads = {n: base[n] * (w / global_frames) for n in _NAMES if not n.startswith("net.img_process.cnn.")}
        on_trunk_grads(grads)
        if fail == "late":
            raise MemoryError("synthetic failure after the early exchange")
        grads.update({n: base[n] * (w / global_frames) for n in _NAMES if n.startswith("net.img_process.cnn.")})
        return torch.tensor(w / img.numel()), grads, state

    tr.loss_and_grads = loss_and_grads
    return tr, base


def _dp_worker(rank, world, port, out, fail_rank, fail):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    D.init_from_env("gloo")
    try:
        tr, base = _fake_trainer(fail if rank == fail_rank else None)
        b, t = 5, 2
        frames = torch.arange(1.0, b * t + 1).view(b, t)      # frame weights 1 .. 10
        b0, b1 = D.shard_range(b, rank, world)
        assert (b1 - b0) == (2, 2, 1)[rank]
        mine = frames[b0:b1]
        try:
            loss, grads, _ = tr.reduced_loss_and_grads(mine, torch.zeros(b1 - b0, t, dtype=torch.bool), None, None, None)
        except RuntimeError as e:
            out[rank] = str(e)
            return
        total = float(frames.sum())
        for n in _NAMES:                                        # every rank holds the gradient of the GLOBAL mean: sum over all 10 frames / 10
            assert torch.allclose(grads[n], base[n] * (total / (b * t)), rtol=1e-6, atol=1e-6), n
        assert abs(float(loss) - total / (b * t)) < 1e-9        # frame-weighted mean of the ranks' losses (shards of 4, 4 and 2 frames)
        assert tr._global_frames == b * t
        # second call: the arenas persist and are reused
        a0 = tr._arenas[1][0].flat.data_ptr()
        tr.reduced_loss_and_grads(mine, torch.zeros(b1 - b0, t, dtype=torch.bool), None, None, None)
        assert tr._arenas[1][0].flat.data_ptr() == a0
        out[rank] = "ok"
    finally:
        dist.destroy_process_group()


def _run_dp(fail_rank, fail):
    world = 3
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_dp_worker, args=(world, _free_port(), out, fail_rank, fail), nprocs=world, join=True)
        return dict(out)


def test_three_rank_uneven_shards_through_the_trainer_host_logic():
    assert _run_dp(-1, None) == {0: "ok", 1: "ok", 2: "ok"}


def test_three_rank_failure_on_one_rank_raises_everywhere():
    for fail in ("early", "late"):
        res = _run_dp(2, fail)
        assert "failed on this rank" in res[2] and "synthetic failure" in res[2], res
        assert "failed on another rank" in res[0] and "failed on another rank" in res[1], res
