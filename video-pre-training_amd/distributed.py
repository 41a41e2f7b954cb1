"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X, "gloo" in the CPU tests).  The forward path shards sequences across ranks with NO data-path
collective (each sequence's KV memory stays on its owner); the only collectives are the timing reduction
used by bench.py and -- once the BC step exists -- the gradient all-reduce below."""
import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = "nccl"):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver
    if not dist.is_initialized():
        dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of `n_items` sequences owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_batch(obs_img: torch.Tensor, first: torch.Tensor, rank: int, world: int):
    """Slice a [B, T, ...] batch of frame sequences (and its `first` flags) for this rank."""
    b, e = shard_range(obs_img.shape[0], rank, world)
    return obs_img[b:e], first[b:e]


def max_over_ranks(value: float, device="cpu") -> float:
    """bench.py's timing rule: the job is as slow as its slowest rank."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def bucketed_all_reduce_start(tensors: Sequence[torch.Tensor], bucket_bytes: int = 64 << 20):
    """Launch the (sum) all-reduce of `tensors` in flat buckets of ~bucket_bytes asynchronously; returns the pending list
    for bucketed_all_reduce_finish.  The collectives are ordered after the work already enqueued on the current stream
    and run on the backend's own stream, i.e. concurrently with whatever the caller enqueues next."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return []
    buckets: List[List[torch.Tensor]] = [[]]
    size = 0
    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if size and size + nbytes > bucket_bytes:
            buckets.append([])
            size = 0
        buckets[-1].append(t)
        size += nbytes
    works = []
    for bucket in buckets:
        if not bucket:
            continue
        flat = torch.cat([t.reshape(-1) for t in bucket])
        works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, bucket))
    return works


class GradArena:
    """One preallocated flat fp32 buffer for a fixed list of gradient tensors (names, shapes), reused step after step: the data-parallel
    step copies each gradient in ONCE when it is final (`adopt`), hands the caller views of the buffer in its place, and all-reduces
    contiguous slices of the buffer in place (`all_reduce_start`) -- no torch.cat of ~1 GB per step, no copy back (the optimiser reads the
    views).  Buckets are slices of ~bucket_bytes cut at tensor boundaries, so every tensor lies in exactly one collective."""

    def __init__(self, names: Sequence[str], shapes: Sequence[torch.Size], device, bucket_bytes: int = 64 << 20):
        self.names = list(names)
        self.offsets, off = {}, 0
        for n, shp in zip(self.names, shapes):
            numel = 1
            for d in shp:
                numel *= int(d)
            self.offsets[n] = (off, numel, tuple(shp))
            off += (numel + 63) // 64 * 64                    # 256-byte aligned views
        self.flat = torch.zeros(max(off, 1), dtype=torch.float32, device=device)
        self.buckets: List[Tuple[int, int]] = []              # [begin, end) element ranges
        begin, limit = 0, max(1, bucket_bytes // 4)
        for n in self.names:
            o, numel, _ = self.offsets[n]
            end = o + (numel + 63) // 64 * 64
            if end - begin > limit and o > begin:
                self.buckets.append((begin, o))
                begin = o
        if off > begin:
            self.buckets.append((begin, off))

    def view(self, name: str) -> torch.Tensor:
        o, numel, shp = self.offsets[name]
        return self.flat[o:o + numel].view(shp)

    def adopt(self, grads: dict) -> None:
        """Copy grads[name] into the arena for every name it holds (missing names: zeros) and re-point grads[name] at the view."""
        dst, src = [], []
        for n in self.names:
            v = self.view(n)
            g = grads.get(n)
            if g is None:
                v.zero_()
            else:
                dst.append(v)
                src.append(g.reshape(v.shape))
            grads[n] = v
        if dst:
            torch._foreach_copy_(dst, src)

    def all_reduce_start(self, force: bool = False):
        """Asynchronous in-place sum of every bucket; returns the work handles (wait() on each, or all_reduce_finish).
        force: issue the collectives in a one-rank group too (a test that the transport -- RCCL on a 1-GPU box -- takes the arena's slices)."""
        if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
            return []
        return [(dist.all_reduce(self.flat[b:e], op=dist.ReduceOp.SUM, async_op=True), None, None) for b, e in self.buckets]


def bucketed_all_reduce_finish(works, scale: float = 1.0) -> int:
    """Wait for the pending buckets and scatter the reduced values back into the original tensors (x scale)."""
    for work, flat, bucket in works:
        work.wait()
        if flat is None:            # a GradArena slice, reduced in place: nothing to scatter back
            continue
        if scale != 1.0:
            flat.mul_(scale)
        off = 0
        for t in bucket:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    return len(works)


def bucketed_all_reduce_(tensors: Sequence[torch.Tensor], bucket_bytes: int = 64 << 20, average: bool = True) -> int:
    """Sum (or average) `tensors` in place across ranks in flat buckets of ~bucket_bytes: the gradient
    exchange of the data-parallel BC step (994 MB fp32 for the 2x model).  Returns the number
    of collectives issued.  64 MB buckets keep each of the 7 xGMI links busy without serialising behind one
    giant tensor; the reduce is asynchronous per bucket and waited at the end."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    works = bucketed_all_reduce_start(tensors, bucket_bytes)
    return bucketed_all_reduce_finish(works, 1.0 / dist.get_world_size() if average else 1.0)
