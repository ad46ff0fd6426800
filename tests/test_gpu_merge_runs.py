"""GPU: building the table from sorted runs (the merge half of shz_table_allgather, SURVEY 8e: "every rank merges 8
sorted runs") gives exactly the table one sort of everything gives -- sorted multiset of (key, song_id, offset)
triples, UNIQUE applied (mysql_database.py:54-55) -- for ragged and empty runs, duplicates inside and across runs,
several segments; and the match on it returns the same results.  The RCCL transfer itself is covered by the 1-rank
communicator tests; here the runs are blocks of one GPU's staged rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import shazam_amd as S
    from shazam_amd import _ffi
    return S, _ffi, S.get_context(0)


def _rows(rng, n, nsid=300, noff=4000):
    k = rng.integers(0, 1 << 22, n).astype(np.uint32) << np.uint32(8) | rng.integers(0, 6, n).astype(np.uint32)
    return k, rng.integers(1, nsid, n).astype(np.uint32), rng.integers(0, noff, n).astype(np.uint32)


@pytest.mark.parametrize("runs, seg_rows", [([5000], None), ([3000, 4100], None), ([1200, 0, 800, 5, 3000, 1, 0, 2500], None),
                                             ([4000] * 8, 9000), ([70000, 50000, 90000], 60000)])
def test_runs_equal_one_sort(env, runs, seg_rows):
    S, F, ctx = env
    rng = np.random.default_rng(sum(runs))
    n = sum(runs)
    k, s, o = _rows(rng, n)
    if n > 100:   # duplicates across runs and inside one
        k[-50:], s[-50:], o[-50:] = k[:50], s[:50], o[:50]
        k[10:20], s[10:20], o[10:20] = k[0], s[0], o[0]
    a, b = F.Table(ctx), F.Table(ctx)
    for t in (a, b):
        if seg_rows:
            t.set_segment_rows(seg_rows)
        t.insert(k, s, o)
    a.finalize()
    b.finalize_runs(runs)
    ea, eb = a.export(), b.export()
    # both are sorted inside their segments; as multisets of rows they are the same set, each row once
    ra = np.unique(np.stack(ea, 1), axis=0)
    rb = np.stack(eb, 1)
    assert len(rb) == len(ra) == a.rows()[0] == b.rows()[0]
    assert np.array_equal(np.unique(rb, axis=0), ra)
    if not seg_rows:
        assert all(np.array_equal(x, y) for x, y in zip(ea, eb))   # one segment: same order too
    st = b.build_stats()
    assert st["sort_s"] > 0 and st["segments_s"] > 0
    # the match sees the same table
    qk, qo = k[::7][:500], (o[::7][:500] + 5) % 4000
    qoff = np.array([0, 200, 200, len(qk)], np.uint64)
    res_a, res_b = a.match(qk, qo, qoff, 3), b.match(qk, qo, qoff, 3)
    for key in res_a:
        assert np.array_equal(res_a[key], res_b[key]), key
    a.close()
    b.close()


def test_runs_into_nonempty_table_take_the_general_path(env):
    S, F, ctx = env
    rng = np.random.default_rng(1)
    k, s, o = _rows(rng, 4000)
    a, b = F.Table(ctx), F.Table(ctx)
    for t in (a, b):
        t.insert(k[:1500], s[:1500], o[:1500])
        t.finalize()
        t.insert(k[1500:], s[1500:], o[1500:])
    a.finalize()
    b.finalize_runs([1000, 1500])
    assert all(np.array_equal(x, y) for x, y in zip(a.export(), b.export()))
    a.close()
    b.close()
