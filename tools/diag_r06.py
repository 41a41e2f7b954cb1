"""Round-6 diagnosis of the 2-ranks-on-one-GPU BC-gradient deviation: WHICH tensor differs first, and under which arrangement.

    python tools/diag_r06.py MODE [spawns] [calls] [KEY=VALUE ...]

The parent computes, in process, the reference of each rank's shard (2 sequences x 5 frames of tests/test_gpu_distributed._batch): every saved
activation of forward_saving (S), every intermediate backward_from records (debug) and the shard's gradients, and writes them to a temp dir.  Each worker
process recomputes its shard `calls` times and compares EVERYTHING with that reference (bit mismatches and rel-L2), in the order of computation.

MODE   pg      two workers, gloo process group, reduced_loss_and_grads (the arrangement of the test); S / debug captured (synchronises after the forward)
       pga     as pg, the capture by stream-ordered device clones only: no synchronisation inside a call
       sync    as nopg, the two ranks start every call together (file barrier): lock-step without a process group
       pgraw   as pg without the capture hooks: gradients only (all-reduced, against the sum of the two shard references)
       nopg    two workers, NO process group: loss_and_grads of the own shard, concurrently
       lock    as nopg, every call (incl. its synchronize) under an exclusive file lock: never two ranks' kernels at once
       hammer  ONE worker (rank 0's shard) beside an unrelated process that keeps the GPU busy (GEMMs + copies)
       solo    ONE worker alone (the control: a fresh process against the parent's reference)
       poison  in the PARENT process: torch.empty filled with NaN (VPT_POISON=big: 1e30 / 0x5A) -- an uninitialised read shows at once
       poisonlds  in the PARENT process: every CU's LDS filled with NaN patterns before each launch (ops.POISON_LDS, vpt_debug_poison_lds)
KEY=VALUE pairs are exported to the workers' environment before the spawn (AMD_SERIALIZE_KERNEL=3, VPT_BC_GATED_DGRAD=0, ...)."""
import fcntl
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_distributed as T  # noqa: E402

B, PRECISION = 4, os.environ.get("DIAG_PRECISION", "bf16")
THRESH = 5e-6


def flat(prefix, obj, out):
    if torch.is_tensor(obj):
        out[prefix] = obj.detach().clone()          # on the device, stream-ordered: no synchronisation (the capture must not change the timing)
    elif isinstance(obj, dict):
        for k, v in obj.items():
            flat(f"{prefix}.{k}" if prefix else str(k), v, out)
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            flat(f"{prefix}[{i}]", v, out)
    return out


def compare(ref, got, thresh=THRESH):
    """-> list of (key, rel-L2, number of differing elements, nan count) for keys that differ by more than `thresh`, in insertion (= computation) order."""
    rows = []
    for k, r in ref.items():
        g = got.get(k)
        if g is not None and g.shape == r.shape and g.dtype == r.dtype and torch.equal(g, r):
            continue
        if g is None:
            rows.append((k, float("inf"), -1, 0))
            continue
        if g.shape != r.shape:
            rows.append((k, float("inf"), -2, 0))
            continue
        if r.dtype in (torch.bool, torch.uint8, torch.int16, torch.int32, torch.int64):
            nd = int((g != r).sum())
            if nd:
                rows.append((k, float(nd) / max(1, r.numel()), nd, 0))
            continue
        rd, gd = r.double(), g.double()
        nan = int(torch.isnan(gd).sum()) - int(torch.isnan(rd).sum())
        nd = int((gd != rd).sum()) - int((torch.isnan(gd) & torch.isnan(rd)).sum())
        if nd == 0 and nan == 0:
            continue
        nr = float(rd.nan_to_num().norm())
        rel = float((gd.nan_to_num(nan=1e30) - rd.nan_to_num(nan=1e30)).norm()) / (nr if nr > 0 else 1.0)
        if rel > thresh or nan:
            rows.append((k, rel, nd, nan))
    return rows


def fmt_rows(rows, n=10):
    if not rows:
        return "clean"
    worst = max(rows, key=lambda r: r[1])
    head = " ; ".join(f"{k} {rel:.2e} ({nd} el{', %d nan' % nan if nan else ''})" for k, rel, nd, nan in rows[:n])
    return f"{len(rows)} tensors differ, worst {worst[0]} {worst[1]:.2e} | FIRST: {head}"


class Capture:
    """Hooks on a BCTrainer that record S (after the forward; synchronises) and the backward's debug intermediates."""

    def __init__(self, tr, capture=True):
        self.tr, self.rec = tr, {}
        if not capture:
            return
        fs, bf = tr.forward_saving, tr.backward_from

        def forward_saving(*a, **k):
            S = fs(*a, **k)
            if self.sync:
                torch.cuda.synchronize()
            flat("S", {k_: v for k_, v in S.items() if k_ not in ("dev",)}, self.rec)
            return S

        def backward_from(S, dz, **k):
            self.rec["dz"] = dz.detach().clone()
            dbg = {}
            k["debug"] = dbg
            g = bf(S, dz, **k)
            if self.sync:
                torch.cuda.synchronize()
            flat("dbg", dbg, self.rec)
            return g

        tr.forward_saving, tr.backward_from = forward_saving, backward_from
        self.sync = True

    def take(self, grads):
        out = dict(self.rec)
        flat("g", {k: v for k, v in grads.items()}, out)
        self.rec = {}
        return out


def shard_args(pol, rank, world=2):
    from vpt_amd import distributed as D
    img, first, ab, ac = T._batch(B)
    b0, b1 = D.shard_range(B, rank, world)
    sl = slice(b0, b1)
    return (img[sl].cuda(), first[sl].cuda(), pol.initial_state(b1 - b0), ab[sl].cuda(), ac[sl].cuda())


def _barrier(out_dir, tag, rank, world=2):
    """File barrier (no process group): every rank drops a file and spins until all are there."""
    open(os.path.join(out_dir, f"bar_{tag}_{rank}"), "w").close()
    while not all(os.path.exists(os.path.join(out_dir, f"bar_{tag}_{r}")) for r in range(world)):
        pass


def worker(rank, world, port, out_dir, mode, calls):
    import torch.distributed as dist
    from vpt_amd.training import BCTrainer
    pg = mode in ("pg", "pga", "pgraw")
    if pg:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pol = T._make(precision=PRECISION)
        tr = BCTrainer(pol, train_cnn=True, weight_decay=0.0)
        cap = Capture(tr, capture=mode != "pgraw")
        cap.sync = mode in ("pg",)
        ref = {k: v.cuda() for k, v in torch.load(os.path.join(out_dir, f"ref{rank}.pt")).items()}
        if pg:       # gradients are all-reduced: against the sum of the two shard references; S / debug against the own shard's
            ref = {k: v for k, v in ref.items() if not k.startswith("g.")} if mode != "pgraw" else {}
            ref.update({k: v.cuda() for k, v in torch.load(os.path.join(out_dir, "refsum.pt")).items()})
        args = shard_args(pol, rank)
        m_global = B * args[0].shape[1]
        lock = open(os.path.join(out_dir, "lock"), "w") if mode == "lock" else None
        lines, bad = [], 0
        for c in range(calls):
            if lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
            if mode == "sync":
                _barrier(out_dir, c, rank)
            if pg:
                _, grads, _ = tr.reduced_loss_and_grads(*args)
            else:
                _, grads, _ = tr.loss_and_grads(*args, global_frames=m_global, unscaled=False)
            got = cap.take(grads)
            torch.cuda.synchronize()
            if lock:
                fcntl.flock(lock, fcntl.LOCK_UN)
            rows = compare(ref, got, thresh=0.0)
            if rows:
                bad += 1
                lines.append(f"  rank {rank} call {c}: {fmt_rows(rows, n=16)}")
        lines.append(f"  rank {rank}: {bad} of {calls} calls differ from the reference in at least one bit")
        with open(os.path.join(out_dir, f"out{rank}.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    finally:
        if pg:
            dist.destroy_process_group()


def hammer(rank, out_dir):
    """An unrelated process that keeps the device busy until the stop file appears."""
    x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    y = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    host = torch.empty(1 << 26, dtype=torch.uint8).pin_memory()
    side = torch.cuda.Stream()
    n = 0
    while not os.path.exists(os.path.join(out_dir, "stop")):
        for _ in range(20):
            z = x @ y
        with torch.cuda.stream(side):
            big[: 1 << 26].copy_(host, non_blocking=True)
            big[1 << 27:].copy_(big[: 1 << 27])
        torch.cuda.synchronize()
        n += 1
    print(f"  hammer: {n} rounds", flush=True)


def reference(out_dir, poison=None):
    """Shard references computed in THIS process (twice: the second must reproduce the first)."""
    from vpt_amd.training import BCTrainer
    pol = T._make(precision=PRECISION)
    tr = BCTrainer(pol, train_cnn=True, weight_decay=0.0)
    cap = Capture(tr)
    refs = []
    for rank in range(2):
        args = shard_args(pol, rank)
        m_global = B * args[0].shape[1]
        runs = []
        for rep in range(3):
            _, grads, _ = tr.loss_and_grads(*args, global_frames=m_global, unscaled=False)
            torch.cuda.synchronize()
            runs.append(cap.take(grads))
        for rep in (1, 2):
            print(f"reference shard {rank} repetition {rep} vs 0 (bitwise): {fmt_rows(compare(runs[0], runs[rep], thresh=0.0))}", flush=True)
        torch.save({k: v.cpu() for k, v in runs[0].items()}, os.path.join(out_dir, f"ref{rank}.pt"))
        refs.append(runs[0])
    torch.save({k: (refs[0][k] + refs[1][k]).cpu() for k in refs[0] if k.startswith("g.")}, os.path.join(out_dir, "refsum.pt"))
    return pol, tr, cap, refs


def install_poison(kind):
    orig = torch.empty

    def empty(*a, **k):
        t = orig(*a, **k)
        if t.is_cuda and t.numel():
            if t.is_floating_point():
                t.fill_(float("nan") if kind == "nan" else (6e4 if t.dtype == torch.float16 else 1e30))
            elif t.dtype == torch.bool:
                t.fill_(True)
            else:
                t.fill_(0x5A)
        return t

    torch.empty = empty
    return orig


def main():
    import torch.multiprocessing as mp
    mode = sys.argv[1]
    nums = [a for a in sys.argv[2:] if "=" not in a]
    spawns = int(nums[0]) if len(nums) > 0 else 4
    calls = int(nums[1]) if len(nums) > 1 else 4
    envs = dict(a.split("=", 1) for a in sys.argv[2:] if "=" in a)
    print(f"=== diag_r06 mode {mode} spawns {spawns} calls {calls} env {envs} precision {PRECISION}", flush=True)
    with tempfile.TemporaryDirectory() as d:
        pol, tr, cap, refs = reference(d)
        if mode in ("poison", "poisonlds"):
            kind = os.environ.get("VPT_POISON", "nan")
            if mode == "poison":
                install_poison(kind)
            else:
                from vpt_amd import ops
                ops.POISON_LDS, kind = True, "lds"
            for rank in range(2):
                args = shard_args(pol, rank)
                for c in range(calls):
                    _, grads, _ = tr.loss_and_grads(*args, global_frames=B * args[0].shape[1], unscaled=False)
                    torch.cuda.synchronize()
                    print(f"  poison({kind}) shard {rank} call {c}: {fmt_rows(compare(refs[rank], cap.take(grads), thresh=0.0), n=14)}", flush=True)
            return
        os.environ.update(envs)
        for s in range(spawns):
            t0 = time.time()
            for f in os.listdir(d):
                if f == "stop" or f.startswith("out") or f.startswith("bar_"):
                    os.remove(os.path.join(d, f))
            if mode in ("pg", "pga", "pgraw", "nopg", "lock", "sync"):
                mp.spawn(worker, args=(2, 29800 + s, d, mode, calls), nprocs=2, join=True)
            elif mode == "solo":
                mp.spawn(worker, args=(1, 0, d, "nopg", calls), nprocs=1, join=True)
            elif mode == "hammer":
                ctx = mp.spawn(hammer, args=(d,), nprocs=1, join=False)
                time.sleep(8)
                mp.spawn(worker, args=(1, 0, d, "nopg", calls), nprocs=1, join=True)
                open(os.path.join(d, "stop"), "w").close()
                ctx.join()
            else:
                raise SystemExit(f"unknown mode {mode}")
            print(f"[{mode} spawn {s}] {time.time() - t0:.0f} s", flush=True)
            for r in range(2):
                p = os.path.join(d, f"out{r}.txt")
                if os.path.exists(p):
                    print(open(p).read().rstrip(), flush=True)


if __name__ == "__main__":
    main()
