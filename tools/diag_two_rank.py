"""Diagnosis of test_two_rank_bc_step_matches_single_process: how far do two single-process runs of the SAME batch differ (order of fp32 atomics),
and how far the 2-rank all-reduced gradients from them, per tensor?   python tools/diag_two_rank.py [reps] [precision]      (needs an MI355X)"""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_distributed as T  # noqa: E402


def top(a, b, n=4):
    rows = []
    for k, g1 in a.items():
        nb = float(g1.norm())
        if nb == 0:
            continue
        rows.append((float((b[k].reshape(g1.shape).float() - g1.float()).norm()) / nb, k, nb, float(g1.abs().sum())))
    rows.sort(reverse=True)
    return "; ".join(f"{k.replace('net.', '')} {e:.2e} (norm {nb:.2e}, abs-sum {ab:.2e})" for e, k, nb, ab in rows[:n])


def main():
    import torch.multiprocessing as mp
    from vpt_amd.training import BCTrainer
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    precision = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    b = 4
    pol = T._make(precision=precision)
    tr = BCTrainer(pol, train_cnn=True, weight_decay=0.0)
    img, first, ab, ac = T._batch(b)
    args = (img.cuda(), first.cuda(), pol.initial_state(b), ab.cuda(), ac.cuda())
    ref = None
    for r in range(reps):
        _, g, _ = tr.reduced_loss_and_grads(*args)
        torch.cuda.synchronize()
        g = {k: v.cpu().clone() for k, v in g.items()}
        if ref is None:
            ref = g
        else:
            print(f"[{r}] single vs single(0): {top(ref, g)}", flush=True)
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(T._worker, args=(2, 29600 + r, d, b, precision), nprocs=2, join=True)
            r0 = torch.load(os.path.join(d, "rank0.pt"))
        print(f"[{r}] 2-rank vs single(0): {top(ref, r0['grads'])}", flush=True)


if __name__ == "__main__":
    main()
