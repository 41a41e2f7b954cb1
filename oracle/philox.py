"""Philox4x32-10 restated in numpy  --  TEST INFRASTRUCTURE (only tests/ may import this).

The reference draws the uniforms of CategoricalActionHead.sample with `th.rand_like(logits)` (lib/action_head.py:200), i.e. from
torch's generator -- on a GPU that generator is Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
1, 2, 3", SC'11; the Random123 library).  The HIP head kernel draws its own uniforms with the same published algorithm from a
device-resident {seed, step} (csrc/vpt_common.h: vpt_philox_uniform) so that a captured acting step samples afresh at every
replay.  This file restates the algorithm and the kernel's counter layout; tests/test_philox_cpu.py pins it on Random123's
known-answer vectors, tests/test_gpu_sampling.py compares the kernel with it bit for bit.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: uint32 [..., 4], key: uint32 [..., 2] (broadcastable) -> uint32 [..., 4]."""
    c = [np.asarray(ctr[..., i], dtype=np.uint32) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32)
    k1 = np.asarray(key[..., 1], dtype=np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[1] ^ k0
            n1 = (p1 & MASK32).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[3] ^ k1
            n3 = (p0 & MASK32).astype(np.uint32)
            c = [n0, n1, n2, n3]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(np.broadcast_arrays(*c), axis=-1)


def head_uniforms(seed: int, step: int, stream: int, rows: int, n: int) -> np.ndarray:
    """fp32 [rows, n]: the uniforms the head kernel draws for generator state {seed, step} and head `stream`:
    counter = (element // 4, row, step lo, step hi ^ (stream << 24)), key = (seed lo, seed hi), word element % 4,
    u = (word >> 8) * 2^-24  (in [0, 1), torch.rand's float32 construction)."""
    el = np.arange(n, dtype=np.uint64)[None, :].repeat(rows, 0)
    row = np.arange(rows, dtype=np.uint64)[:, None].repeat(n, 1)
    ctr = np.stack([(el >> np.uint64(2)).astype(np.uint32), row.astype(np.uint32),
                    np.full((rows, n), step & 0xFFFFFFFF, dtype=np.uint32),
                    np.full((rows, n), ((step >> 32) & 0xFFFFFFFF) ^ ((stream << 24) & 0xFFFFFFFF), dtype=np.uint32)], axis=-1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    out = philox4x32_10(ctr, key)                                   # [rows, n, 4]
    word = np.take_along_axis(out, (el & np.uint64(3)).astype(np.int64)[..., None], axis=-1)[..., 0]
    return ((word >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def gumbel_argmax(logp: np.ndarray, u: np.ndarray) -> np.ndarray:
    """CategoricalActionHead.sample (lib/action_head.py:198-207) on log-probs [rows, n] with uniforms u: argmax(logp - log(-log u)),
    u == 1 -> 0.999 as the reference guards; first maximum.  float32 arithmetic like the kernel's."""
    u = np.where(u == 1.0, np.float32(0.999), u).astype(np.float32)
    with np.errstate(divide="ignore"):
        g = -np.log(-np.log(u, dtype=np.float32), dtype=np.float32)
    return np.argmax(logp.astype(np.float32) + g, axis=-1)
