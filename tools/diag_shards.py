"""Third diagnosis: how often does the shard-wise BC gradient (two shards of 2 sequences x 5 frames, fresh trainer each, ONE process, no process group) deviate from the
whole-batch gradient by more than the order-of-additions noise (5e-7)?   python tools/diag_shards.py [reps]    -- run under VPT_HIP_LIB=... for another library build."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_distributed as T  # noqa: E402


def main():
    from vpt_amd.training import BCTrainer
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    precision, b = "bf16", 4
    img, first, ab, ac = T._batch(b)
    pol = T._make(precision=precision)
    tr = BCTrainer(pol, train_cnn=True, weight_decay=0.0)
    m_global = b * img.shape[1]
    whole = []
    for _ in range(6):                       # the whole batch, repeated in ONE trainer: run-to-run noise at 20 frames
        _, g, _ = tr.reduced_loss_and_grads(img.cuda(), first.cuda(), pol.initial_state(b), ab.cuda(), ac.cuda())
        torch.cuda.synchronize()
        whole.append({k: v.cpu().clone() for k, v in g.items()})
    ref = whole[0]

    def cmp(x):
        rows = sorted(((float((x[k].reshape(g1.shape).float() - g1.float()).norm()) / float(g1.norm()), k) for k, g1 in ref.items() if float(g1.norm()) > 0), reverse=True)
        return rows

    for i, w in enumerate(whole[1:]):
        r = cmp(w)
        print(f"whole[{i + 1}] worst {r[0][0]:.2e} {r[0][1][4:]} median {r[len(r) // 2][0]:.2e} n>1e-5: {sum(e > 1e-5 for e, _ in r)}", flush=True)
    events = 0
    p2 = T._make(precision=precision)
    t2 = BCTrainer(p2, train_cnn=True, weight_decay=0.0)
    for rep in range(reps):
        total = None
        for sl in (slice(0, 2), slice(2, 4)):
            _, gs, _ = t2.loss_and_grads(img[sl].cuda(), first[sl].cuda(), p2.initial_state(2), ab[sl].cuda(), ac[sl].cuda(), global_frames=m_global, unscaled=False)
            torch.cuda.synchronize()
            gs = {k: v.cpu().clone() for k, v in gs.items()}
            total = gs if total is None else {k: total[k] + gs[k] for k in total}
        r = cmp(total)
        if r[0][0] > 5e-6:
            events += 1
            print(f"shards[{rep}] worst {r[0][0]:.2e} {r[0][1][4:]} | 2nd {r[1][0]:.2e} {r[1][1][4:]} | median {r[len(r) // 2][0]:.2e} n>1e-5: {sum(e > 1e-5 for e, _ in r)}", flush=True)
    print(f"shards: {events} of {reps} repetitions above 5e-6 (lib {os.environ.get('VPT_HIP_LIB', 'in-tree')})", flush=True)


if __name__ == "__main__":
    main()
