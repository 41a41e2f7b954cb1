"""The sampler's generator, on the CPU: oracle/philox.py (the numpy restatement the GPU test compares the head kernel with) against
Random123's published known-answer vectors for philox4x32-10, plus the distributional sanity of the uniforms built from it."""
import numpy as np

from oracle import philox as PH


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors, "philox4x32 10": counter, key -> output
    kat = [
        ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = PH.philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert tuple(int(x) for x in got) == want, (ctr, [hex(int(x)) for x in got])


def test_head_uniforms_layout_and_distribution():
    u = PH.head_uniforms(seed=0x1234567890ABCDEF, step=5, stream=1, rows=3, n=8641)
    assert u.shape == (3, 8641) and u.dtype == np.float32
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
    # words of one Philox block are consecutive elements; a different row / step / stream / seed gives a different stream
    blk = PH.philox4x32_10(np.array([0, 2, 5, (1 << 24)], dtype=np.uint32), np.array([0x90ABCDEF, 0x12345678], dtype=np.uint32))
    assert np.array_equal(u[2, :4], (blk >> np.uint32(8)).astype(np.float32) / np.float32(16777216.0))
    for other in (PH.head_uniforms(0x1234567890ABCDEF, 6, 1, 3, 64), PH.head_uniforms(0x1234567890ABCDEF, 5, 0, 3, 64), PH.head_uniforms(7, 5, 1, 3, 64)):
        assert not np.array_equal(other, u[:, :64])
    assert not np.array_equal(u[0, :64], u[1, :64])
    # uniformity: 20 equal bins over 25 923 draws (chi-square, 19 degrees of freedom: 43.8 is the 0.1 % point)
    counts, _ = np.histogram(u.ravel(), bins=20, range=(0.0, 1.0))
    exp = u.size / 20.0
    assert float(((counts - exp) ** 2 / exp).sum()) < 43.8
    assert abs(float(u.mean()) - 0.5) < 0.01


def test_gumbel_argmax_follows_the_distribution():
    p = np.array([0.5, 0.25, 0.125, 0.125], dtype=np.float64)
    rows = 4000
    u = PH.head_uniforms(seed=11, step=0, stream=0, rows=rows, n=4)
    a = PH.gumbel_argmax(np.log(p)[None, :].repeat(rows, 0), u)
    counts = np.bincount(a, minlength=4)
    chi2 = float(((counts - rows * p) ** 2 / (rows * p)).sum())
    assert chi2 < 16.27, (counts, chi2)        # 3 degrees of freedom, 0.1 % point
