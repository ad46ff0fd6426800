"""GPU: the bulk build -- staged rows sealed into sorted runs, runs turned into segments by ONE k-way merge
(shz_build.hip: seal_rows / kw_tile_kernel) -- leaves exactly the table the reference's schema defines: the set of
(hash, song_id, offset) rows, UNIQUE(song_id, offset, hash) + INSERT IGNORE (mysql_database.py:54-55, 62-68), sorted
by (key, song_id, offset) inside every segment.  Checked against numpy on: rows that arrive ordered by (song id, offset)
(the 4-pass sort) and rows that do not, duplicates inside a run / across runs / of rows in a frozen segment, more runs
than one merge takes, tiny segments, a reserved slab, clear + rebuild, and the match on the result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import shazam_amd as S
    from shazam_amd import _ffi
    return S, _ffi, S.get_context(0)


def _rows(rng, n, sid_lo, sid_hi, noff=600, ordered=False):
    k = (rng.integers(0, 300, n).astype(np.uint32) << np.uint32(20)) | (rng.integers(0, 300, n).astype(np.uint32) << np.uint32(8)) | \
        rng.integers(0, 6, n).astype(np.uint32)
    s = rng.integers(sid_lo, sid_hi, n).astype(np.uint32)
    o = rng.integers(0, noff, n).astype(np.uint32)
    if ordered:   # the order ingest produces: song after song, offsets ascending inside a song
        idx = np.lexsort((o, s))
        k, s, o = k[idx], s[idx], o[idx]
    return k, s, o


def _uniq(parts):
    rows = np.concatenate([np.stack([k, s, o], 1) for k, s, o in parts]).astype(np.uint64)
    return np.unique(rows, axis=0)


def _check_table(t, want):
    k, s, o = t.export()
    got = np.stack([k, s, o], 1).astype(np.uint64)
    assert len(got) == len(want) == t.rows()[0]
    assert np.array_equal(np.unique(got, axis=0), want)          # each row once
    return got


@pytest.mark.parametrize("ordered", [True, False])
@pytest.mark.parametrize("n_runs, per, seg_rows", [(1, 5000, None), (3, 40000, None), (8, 9000, 20000), (5, 30000, 16000),
                                                    (19, 3000, None), (2, 700, 256)])
def test_sealed_runs_equal_numpy(env, ordered, n_runs, per, seg_rows):
    S, F, ctx = env
    rng = np.random.default_rng(n_runs * 1000 + per + int(ordered))
    t = F.Table(ctx)
    if seg_rows:
        t.set_segment_rows(seg_rows)
    parts = []
    for r in range(n_runs):
        k, s, o = _rows(rng, per, 1 + 50 * r, 51 + 50 * r, ordered=ordered)   # disjoint song ids per run: the no-dedup merge
        t.insert(k, s, o)
        if r + 1 < n_runs:
            t.seal_run()
            assert t.rows()[1] > 0                                    # sealed rows are not visible yet
        parts.append((k, s, o))
    t.finalize()
    want = _uniq(parts)
    got = _check_table(t, want)
    assert t.rows()[1] == 0
    if not seg_rows:
        assert t.segments() == 1 and np.array_equal(got, want)       # one segment: globally sorted
    else:
        assert t.segments() >= len(want) // seg_rows
    # lookup of a few keys returns exactly their rows
    probe = np.unique(want[::997, 0]).astype(np.uint32)
    lk, ls, lo = t.lookup(probe)
    sel = want[np.isin(want[:, 0], probe)]
    assert np.array_equal(np.unique(np.stack([lk, ls, lo], 1).astype(np.uint64), axis=0), sel)
    t.close()


def test_duplicates_inside_across_runs_and_of_frozen_rows(env):
    S, F, ctx = env
    rng = np.random.default_rng(5)
    t = F.Table(ctx)
    t.set_segment_rows(12000)
    a = _rows(rng, 20000, 1, 40)
    b = _rows(rng, 9000, 20, 70)                  # song ids overlap a's: the merge must look for duplicates
    b[0][:3000], b[1][:3000], b[2][:3000] = a[0][:3000], a[1][:3000], a[2][:3000]      # across runs
    b[0][3000:3500], b[1][3000:3500], b[2][3000:3500] = b[0][5000], b[1][5000], b[2][5000]   # inside a run
    t.insert(*a)
    t.seal_run()                                   # 20,000 rows >= the segment limit: one full segment is cut here
    assert t.segments() >= 1
    t.insert(*b)
    t.seal_run()
    c = _rows(rng, 4000, 1, 70)
    c[0][:2000], c[1][:2000], c[2][:2000] = a[0][-2000:], a[1][-2000:], a[2][-2000:]   # rows that sit in frozen segments by now
    t.insert(*c)
    t.finalize()
    _check_table(t, _uniq([a, b, c]))
    t.close()


def test_reserved_build_clear_and_rebuild_keeps_memory(env):
    S, F, ctx = env
    rng = np.random.default_rng(9)
    t = F.Table(ctx)
    t.set_segment_rows(50000)
    t.reserve(200000, 40000)
    free0 = None
    for rep in range(4):
        parts = []
        for r in range(5):
            p = _rows(rng, 36000, 1 + 100 * r, 101 + 100 * r, ordered=True)
            t.insert(*p)
            t.seal_run()
            parts.append(p)
        t.finalize()
        _check_table(t, _uniq(parts))
        assert t.segments() >= 3
        free_now = ctx.mem_info()[0]
        if rep == 1:
            free0 = free_now
        if rep > 1:
            assert abs(free_now - free0) < (8 << 20), (free0, free_now)   # nothing is leaked per rebuild (segments_from_sorted)
        t.clear()
        assert t.rows() == (0, 0)
    t.close()


def test_bulk_table_matches_like_incremental_table(env):
    """The same rows through the bulk path (seal + k-way merge) and through the column path (finalize into a non-empty
    active segment, the mixed-ingest path) answer every query alike."""
    S, F, ctx = env
    rng = np.random.default_rng(21)
    parts = [_rows(rng, 30000, 1 + 40 * r, 41 + 40 * r, noff=3000, ordered=bool(r & 1)) for r in range(4)]
    bulk, inc = F.Table(ctx), F.Table(ctx)
    for i, p in enumerate(parts):
        bulk.insert(*p)
        bulk.seal_run()
        inc.insert(*p)
        inc.finalize()             # first call: bulk path with one run; later calls: merge into the active segment
    bulk.finalize()
    want = _uniq(parts)
    ga, gb = _check_table(bulk, want), _check_table(inc, want)
    assert np.array_equal(ga, gb)
    allk = np.concatenate([p[0] for p in parts])
    qk, qo, qoff = [], [], [0]
    for q in range(16):
        m = int(rng.integers(50, 400))
        qk.append(allk[rng.integers(0, len(allk), m)])
        qo.append(rng.integers(0, 100, m).astype(np.uint32))
        qoff.append(qoff[-1] + m)
    qk, qo, qoff = np.concatenate(qk), np.concatenate(qo), np.array(qoff, np.uint64)
    ra, rb = bulk.match(qk, qo, qoff, 3), inc.match(qk, qo, qoff, 3)
    for f in ra:
        assert np.array_equal(ra[f], rb[f]), f
    bulk.close()
    inc.close()


def test_wide_ids_fall_back_to_the_column_path(env):
    S, F, ctx = env
    rng = np.random.default_rng(3)
    k, s, o = _rows(rng, 5000, 1, 50)
    s = s.astype(np.uint32) + np.uint32(1 << 24)      # 25 bits of song id + 10 of offset: no 32-bit packing
    o = o + np.uint32(1 << 9)
    t = F.Table(ctx)
    t.insert(k, s, o)
    t.seal_run()                                      # = finalize on this table
    assert t.rows() == (len(_uniq([(k, s, o)])), 0)
    _check_table(t, _uniq([(k, s, o)]))
    t.close()


def test_clear_then_finalize_runs_does_not_leak(env):
    """a rebuild from runs into a cleared table used to overwrite the active columns without freeing them."""
    S, F, ctx = env
    rng = np.random.default_rng(2)
    t = F.Table(ctx)
    free_ref = None
    for rep in range(5):
        k, s, o = _rows(rng, 300000, 1, 500)
        t.insert(k, s, o)
        t.finalize_runs([100000, 150000, 50000])
        _check_table(t, _uniq([(k, s, o)]))
        ctx.sync()
        free_now = ctx.mem_info()[0]
        if rep == 1:
            free_ref = free_now
        if rep > 1:
            assert abs(free_now - free_ref) < (8 << 20), (free_ref, free_now)
        t.clear()
    t.close()
