"""Second diagnosis of the rare 2-rank discrepancy (tools/diag_two_rank.py reproduced it in 1 of 5 runs: 5e-3 on stack-0 tensors).
(A) ONE process, no process group: the two shards run one after the other through fresh trainers, their gradients summed, against the whole batch --
    is the shard-wise computation itself unstable from run to run?   (B) two ranks: call 1 and call 2 of the worker saved separately."""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_distributed as T  # noqa: E402


def worst(a, b):
    rows = []
    for k, g1 in a.items():
        nb = float(g1.norm())
        if nb > 0:
            rows.append((float((b[k].reshape(g1.shape).float() - g1.float()).norm()) / nb, k))
    rows.sort(reverse=True)
    return f"{rows[0][0]:.2e} {rows[0][1].replace('net.', '')} | median {rows[len(rows) // 2][0]:.2e}"


def worker2(rank, world, port, out_dir, b, precision):
    import torch.distributed as dist
    from vpt_amd import distributed as D
    from vpt_amd.training import BCTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pol = T._make(precision=precision)
        tr = BCTrainer(pol, train_cnn=True, weight_decay=0.0)
        img, first, ab, ac = T._batch(b)
        b0, b1 = D.shard_range(img.shape[0], rank, world)
        sl = slice(b0, b1)
        args = (img[sl].cuda(), first[sl].cuda(), pol.initial_state(b1 - b0), ab[sl].cuda(), ac[sl].cuda())
        calls = []
        for _ in range(3):
            _, grads, _ = tr.reduced_loss_and_grads(*args)
            torch.cuda.synchronize()
            calls.append({k: v.cpu().clone() for k, v in grads.items()})
        torch.save(calls, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    from vpt_amd.training import BCTrainer
    precision, b = "bf16", 4
    img, first, ab, ac = T._batch(b)
    pol = T._make(precision=precision)
    tr = BCTrainer(pol, train_cnn=True, weight_decay=0.0)
    _, g, _ = tr.reduced_loss_and_grads(img.cuda(), first.cuda(), pol.initial_state(b), ab.cuda(), ac.cuda())
    torch.cuda.synchronize()
    ref = {k: v.cpu().clone() for k, v in g.items()}
    m_global = b * img.shape[1]
    for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):          # (A)
        total = None
        for sl in (slice(0, 2), slice(2, 4)):
            p2 = T._make(precision=precision)
            t2 = BCTrainer(p2, train_cnn=True, weight_decay=0.0)
            _, gs, _ = t2.loss_and_grads(img[sl].cuda(), first[sl].cuda(), p2.initial_state(2), ab[sl].cuda(), ac[sl].cuda(), global_frames=m_global, unscaled=False)
            torch.cuda.synchronize()
            gs = {k: v.cpu().clone() for k, v in gs.items()}
            total = gs if total is None else {k: total[k] + gs[k] for k in total}
            del p2, t2
        print(f"(A)[{r}] shards summed in one process vs whole batch: {worst(ref, total)}", flush=True)
    for r in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):           # (B)
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(worker2, args=(2, 29700 + r, d, b, precision), nprocs=2, join=True)
            c0 = torch.load(os.path.join(d, "rank0.pt"))
        print(f"(B)[{r}] 2-rank call 1 / 2 / 3 vs whole batch: " + " || ".join(worst(ref, c) for c in c0), flush=True)


if __name__ == "__main__":
    main()
