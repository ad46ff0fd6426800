"""GPU: the stable LSD radix sort behind the table build and the vote (shz_prims.hip), against numpy's stable
argsort: 8- and 9-bit digit plans, payloads of 0 / 4 / 8 bytes, partial tiles, tiny inputs, heavy duplicates and
already-sorted / constant digits (every lane of a row in one bin)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _digits(bit_lo, bit_hi):
    """the sort compares exactly the bits [bit_lo, bit_hi): the last digit of a plan is masked to what is left"""
    return bit_lo, bit_hi


@pytest.mark.parametrize("n", [1, 63, 64, 4095, 4096, 4097, 3 * 4096 + 17, 300001])
@pytest.mark.parametrize("bits", [(0, 8), (0, 9), (1, 36), (0, 27), (5, 23), (0, 62), (32, 41), (0, 64)])
@pytest.mark.parametrize("vb", [0, 4, 8])
def test_sort_matches_stable_argsort(n, bits, vb):
    import shazam_amd as S
    ctx = S.get_context(0)
    if n > 5000 and vb == 4 and bits not in ((1, 36), (0, 62)):
        pytest.skip("large cases: a subset")
    rng = np.random.default_rng(n * 131 + bits[1] * 7 + vb)
    keys = rng.integers(0, 1 << 63, n, dtype=np.uint64) * 2 + rng.integers(0, 2, n, dtype=np.uint64)
    if n > 100:
        keys[n // 3: n // 3 + n // 4] = keys[n // 3]                       # a long run of one key
        keys[-(n // 5):] &= np.uint64(0xFFFF)                              # many small keys: few distinct high digits
    lo, hi = _digits(*bits)
    mask = np.uint64(((1 << (hi - lo)) - 1) << lo) if hi - lo < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    order = np.argsort((keys & mask) >> np.uint64(lo), kind="stable")
    if vb == 0:
        got = ctx.sort_pairs(keys, None, *bits)
        assert np.array_equal(got, keys[order])
    else:
        vals = np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64) * (3 if vb == 4 else 0x100000001)
        gk, gv = ctx.sort_pairs(keys, vals, *bits)
        assert np.array_equal(gk, keys[order])
        assert np.array_equal(gv, vals[order])       # stability: equal digits keep their input order


def test_sorted_and_constant_inputs():
    import shazam_amd as S
    ctx = S.get_context(0)
    n = 50000
    asc = np.arange(n, dtype=np.uint64) << np.uint64(3)
    assert np.array_equal(ctx.sort_pairs(asc, None, 0, 20), asc)
    assert np.array_equal(ctx.sort_pairs(asc[::-1], None, 0, 20), asc)
    const = np.full(n, 0x1234567, np.uint64)
    k, v = ctx.sort_pairs(const, np.arange(n, dtype=np.uint32), 0, 35)
    assert np.array_equal(k, const) and np.array_equal(v, np.arange(n, dtype=np.uint32))
    assert ctx.sort_pairs(np.zeros(0, np.uint64), None, 0, 64).size == 0


@pytest.mark.parametrize("n", [1, 5, 4095, 4096, 4097, 70000, 3_000_001])
@pytest.mark.parametrize("bits", [(0, 32), (1, 31), (1, 28), (3, 12), (0, 9), (5, 5), (1, 19)])
def test_sort_keys32_against_numpy(n, bits):
    """The 4-byte sort of the vote: stable on the chosen bit range, last pass widens and adds."""
    import shazam_amd as S
    ctx = S.get_context(0)
    lo, hi = bits
    rng = np.random.default_rng(n * 131 + lo * 7 + hi)
    k = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    if n > 10:
        k[: n // 3] &= np.uint32(0xFF0)            # many equal digits: stability matters
    add = int(rng.integers(0, 1 << 40)) << 8
    got = ctx.sort_keys32(k, lo, hi, add)
    field = (k >> np.uint32(lo)) & np.uint32((1 << (hi - lo)) - 1) if hi > lo else np.zeros(n, np.uint32)
    want = k[np.argsort(field, kind="stable")].astype(np.uint64) + np.uint64(add)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("sizes", [[1], [5, 0, 7], [4096, 4097, 1, 0, 0, 12289], [3, 70001, 2, 4095, 64, 64, 8193],
                                   [1000] * 128, [0, 0, 9], [300017]])
@pytest.mark.parametrize("bits", [(15, 31), (0, 8), (11, 27), (3, 32), (14, 32)])
def test_segmented_sort32(sizes, bits):
    """shz_sort_u32_seg (the vote passes): every segment ordered among its own keys, stably, on the given bits; segments of
    any length (empty, one key, not a multiple of four: their blocks start at unaligned addresses)."""
    import shazam_amd as S
    ctx = S.get_context(0)
    rng = np.random.default_rng(sum(sizes) * 31 + bits[0])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(off[-1])
    keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    if n > 100:
        keys[n // 2: n // 2 + n // 5] = keys[n // 2]          # a long run of one key
    lo, hi = bits
    got = ctx.sort_keys32_seg(keys, off, lo, hi)
    mask = np.uint32((((1 << (hi - lo)) - 1) << lo) & 0xFFFFFFFF)
    for i in range(len(sizes)):
        seg = keys[int(off[i]):int(off[i + 1])]
        order = np.argsort((seg & mask) >> np.uint32(lo), kind="stable")
        assert np.array_equal(got[int(off[i]):int(off[i + 1])], seg[order]), (i, sizes[i])


def test_segmented_sort32_large_blocks():
    """2^24 keys and more: the sort cuts its segments into blocks of 8,192 keys (shz_seg_tile) -- same order."""
    import shazam_amd as S
    ctx = S.get_context(0)
    sizes = [9_000_000, 0, 5, 7_800_001, 8193]
    rng = np.random.default_rng(2024)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(off[-1])
    assert n >= 1 << 24
    keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    keys[1000:3_000_000] &= np.uint32(0x7FFF)                 # three million keys with every sorted bit equal: one long run
    lo, hi = 15, 31
    got = ctx.sort_keys32_seg(keys, off, lo, hi)
    mask = np.uint32((((1 << (hi - lo)) - 1) << lo) & 0xFFFFFFFF)
    for i in range(len(sizes)):
        seg = keys[int(off[i]):int(off[i + 1])]
        order = np.argsort((seg & mask) >> np.uint32(lo), kind="stable")
        assert np.array_equal(got[int(off[i]):int(off[i + 1])], seg[order]), (i, sizes[i])
