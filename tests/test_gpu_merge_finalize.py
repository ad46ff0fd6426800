"""GPU: incremental finalize = radix sort of the new rows + merge-path merge into the sorted active segment, written
into the segment's own buffers.  After every step the table must equal numpy's sorted unique rows: duplicates of rows
already in the table and inside the batch disappear (INSERT IGNORE, mysql_database.py:62-68), batches smaller and
larger than the table, song ids / offsets that outgrow the packing of the earlier rows, tile edges of the merge."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch(rng, n, max_sid, max_off, nkeys=3000):
    key = (rng.integers(0, nkeys, n).astype(np.uint32) << 8) | rng.integers(0, 3, n).astype(np.uint32)
    return key, rng.integers(1, max_sid + 1, n).astype(np.uint32), rng.integers(0, max_off + 1, n).astype(np.uint32)


def test_incremental_finalize_equals_sorted_unique():
    import shazam_amd as S
    ctx = S.get_context(0)
    rng = np.random.default_rng(42)
    t = S.Table(ctx)
    have = np.zeros((0, 3), np.uint64)
    steps = [(5, 40, 300), (70000, 40, 300), (2047, 40, 300), (2049, 41, 300), (1, 41, 300), (30000, 5000, 300),
             (4096, 5000, 70000), (200000, 5000, 70000), (8, 5000, 70000)]
    for n, max_sid, max_off in steps:
        k, s, o = _batch(rng, n, max_sid, max_off)
        if len(have):   # re-insert some rows the table already holds
            again = have[rng.integers(0, len(have), min(len(have), n // 2 + 1))]
            k = np.concatenate([k, again[:, 0].astype(np.uint32)])
            s = np.concatenate([s, again[:, 1].astype(np.uint32)])
            o = np.concatenate([o, again[:, 2].astype(np.uint32)])
        t.insert(k, s, o)
        t.finalize()
        have = np.unique(np.concatenate([have, np.stack([k, s, o], 1).astype(np.uint64)]), axis=0)
        ek, es, eo = t.export()
        assert t.rows() == (len(have), 0)
        assert np.array_equal(np.stack([ek, es, eo], 1).astype(np.uint64), have), (n, max_sid, max_off)
    # the bucket index follows: every key of the last batch is found with all its rows
    probe = np.unique(k)[:200]
    lk, ls, lo = t.lookup(probe)
    want = np.concatenate([have[have[:, 0] == kk] for kk in probe.astype(np.uint64)])
    assert np.array_equal(np.stack([lk, ls, lo], 1).astype(np.uint64), want)
    t.close()
