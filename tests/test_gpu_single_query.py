"""GPU: ONE query handed over in host memory.  Its vote kernels are queued behind the probe before the host knows how
many votes there are (shz_table.hip: m_spec_plan_kernel / m_expand_spec_kernel, then the one-workgroup fold and rank);
with more than 32,768 votes those kernels do nothing and the call goes on through the vote passes.  Either way the
answer is align_matches' (recognizer.py:289-338: count descending, ties -> smaller song id, smallest offset difference
among a song's best), checked here against oracle/cpu_ref.py and against the same query matched inside a batch of two
(which never takes the queued path)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _table(rng, n_songs, rows, hot_rows=0):
    """rows random (key, song, offset) rows over a small key alphabet + hot_rows rows under one popular key"""
    key = ((rng.integers(0, 600, rows) << 20) | (rng.integers(0, 40, rows) << 8) | rng.integers(0, 4, rows)).astype(np.uint32)
    sid = rng.integers(1, n_songs + 1, rows).astype(np.uint32)
    off = rng.integers(0, 900, rows).astype(np.uint32)
    hot = np.uint32((700 << 20) | (5 << 8) | 1)
    if hot_rows:
        key = np.concatenate([key, np.full(hot_rows, hot)])
        sid = np.concatenate([sid, rng.integers(1, n_songs + 1, hot_rows).astype(np.uint32)])
        off = np.concatenate([off, rng.integers(0, 900, hot_rows).astype(np.uint32)])
    return key, sid, off, hot


def _oracle_db(key, sid, off, n_songs):
    from oracle import cpu_ref as O
    odb = O.DictDB()
    for s in range(1, n_songs + 1):
        odb.insert_song(str(s), "00", 1)
    by_song = {}
    for k, s, o in zip(key.tolist(), sid.tolist(), off.tolist()):
        by_song.setdefault(s, []).append((k, o))
    for s, hs in by_song.items():
        odb.insert_hashes(s, hs)
    return odb


def _check(res, q, qk, qo, odb, topn):
    from oracle import cpu_ref as O
    hs = set(zip(qk.tolist(), qo.tolist()))
    m, dd = O.return_matches(hs, odb)
    want = O.vote(m, topn)
    got = [(int(res["sid"][q, i]), int(res["delta"][q, i]), int(res["aligned"][q, i])) for i in range(int(res["nres"][q]))]
    assert got == [tuple(w) for w in want]
    assert [int(res["dedup"][q, i]) for i in range(len(got))] == [dd[w[0]] for w in want]
    assert int(res["npairs"][q]) == len(m) and int(res["nhash"][q]) == len(hs)
    return len(m)


def _same(a, b, qa=0, qb=0):
    n = int(a["nres"][qa])
    assert n == int(b["nres"][qb])
    for f in ("sid", "delta", "aligned", "dedup"):
        assert np.array_equal(a[f][qa, :n], b[f][qb, :n]), f
    assert int(a["npairs"][qa]) == int(b["npairs"][qb]) and int(a["nhash"][qa]) == int(b["nhash"][qb])


@pytest.mark.parametrize("topn", [1, 3, 8])
def test_queued_votes_equal_oracle_and_the_batched_match(topn):
    import shazam_amd as S
    ctx = S.get_context(0)
    rng = np.random.default_rng(100 + topn)
    n_songs = 300
    key, sid, off, _ = _table(rng, n_songs, 40000)
    t = S.Table(ctx)
    t.insert(key, sid, off)
    t.finalize()
    odb = _oracle_db(key, sid, off, n_songs)
    # the query: 500 of song 17's rows moved by a constant offset (the true match) + random hashes
    own = np.flatnonzero((sid == 17) & (off >= 40))[:500]
    qk = np.concatenate([key[own], key[rng.integers(0, len(key), 700)]])
    qo = np.concatenate([off[own] - 40, rng.integers(0, 300, 700).astype(np.uint32)]).astype(np.uint32)
    qoff = np.array([0, len(qk)], np.uint64)
    q0, u0 = ctx.spec_stats()
    res = t.match(qk, qo, qoff, topn)
    q1, u1 = ctx.spec_stats()
    assert (q1 - q0, u1 - u0) == (1, 1)                                  # queued, and its results were the answer
    votes = _check(res, 0, qk, qo, odb, topn)
    assert 0 < votes <= 32768
    assert int(res["sid"][0, 0]) == 17 and int(res["delta"][0, 0]) == 40
    # the same query as the second of two: the vote passes
    two = t.match(np.concatenate([qk[:5], qk]), np.concatenate([qo[:5], qo]), np.array([0, 5, 5 + len(qk)], np.uint64), topn)
    assert ctx.spec_stats() == (q1, u1)
    _same(res, two, 0, 1)
    t.close()


def test_more_votes_than_the_queued_kernels_take_continue_through_the_passes():
    import shazam_amd as S
    ctx = S.get_context(0)
    rng = np.random.default_rng(7)
    n_songs = 2000
    key, sid, off, hot = _table(rng, n_songs, 60000, hot_rows=50000)
    t = S.Table(ctx)
    t.insert(key, sid, off)
    t.finalize()
    odb = _oracle_db(key, sid, off, n_songs)
    qk = np.concatenate([np.array([hot, hot], np.uint32), key[rng.integers(0, 60000, 300)]])
    qo = np.concatenate([np.array([3, 11], np.uint32), rng.integers(0, 200, 300).astype(np.uint32)])
    qoff = np.array([0, len(qk)], np.uint64)
    q0, u0 = ctx.spec_stats()
    res = t.match(qk, qo, qoff, 5)
    q1, u1 = ctx.spec_stats()
    assert (q1 - q0, u1 - u0) == (1, 0)                                  # queued, did nothing
    assert _check(res, 0, qk, qo, odb, 5) > 32768
    t.close()


def test_no_vote_and_no_hash_queries():
    import shazam_amd as S
    ctx = S.get_context(0)
    rng = np.random.default_rng(9)
    key, sid, off, _ = _table(rng, 50, 5000)
    t = S.Table(ctx)
    t.insert(key, sid, off)
    t.finalize()
    absent = np.array([(900 << 20) | (1 << 8) | 1, (901 << 20) | (2 << 8) | 2], np.uint32)    # no such key in the table
    res = t.match(absent, np.array([0, 4], np.uint32), np.array([0, 2], np.uint64), 3)
    assert int(res["nres"][0]) == 0 and int(res["npairs"][0]) == 0 and int(res["nhash"][0]) == 2
    res = t.match(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.array([0, 0], np.uint64), 3)
    assert int(res["nres"][0]) == 0 and int(res["npairs"][0]) == 0 and int(res["nhash"][0]) == 0
    t.close()


def test_paths_that_do_not_queue_give_the_same_answer():
    """more results than the vote tiles hold (topn 9), and a query already on the device: the count is read first, as
    before; same arrays"""
    import shazam_amd as S
    ctx = S.get_context(0)
    rng = np.random.default_rng(21)
    n_songs = 120
    key, sid, off, _ = _table(rng, n_songs, 30000)
    t = S.Table(ctx)
    t.insert(key, sid, off)
    t.finalize()
    odb = _oracle_db(key, sid, off, n_songs)
    pick = rng.integers(0, len(key), 900)
    qk, qo = key[pick], (off[pick] + 5).astype(np.uint32)
    qoff = np.array([0, len(qk)], np.uint64)
    q0, _ = ctx.spec_stats()
    res9 = t.match(qk, qo, qoff, 9)
    assert ctx.spec_stats()[0] == q0
    _check(res9, 0, qk, qo, odb, 9)
    res8 = t.match(qk, qo, qoff, 8)
    assert ctx.spec_stats()[0] == q0 + 1
    n = int(res8["nres"][0])
    for f in ("sid", "delta", "aligned", "dedup"):
        assert np.array_equal(res8[f][0, :n], res9[f][0, :n]), f
    t.close()


def test_a_probe_that_gives_up_in_the_queued_fold_is_voted_again():
    """SHZ_DEBUG_VT_PROBE1: the queued one-workgroup fold flags its table as full; the call repeats the query through
    the full sort (which does not queue) and returns the same answer"""
    import shazam_amd as S
    ctx = S.get_context(0)
    rng = np.random.default_rng(33)
    n_songs = 400
    key, sid, off, _ = _table(rng, n_songs, 50000)
    t = S.Table(ctx)
    t.insert(key, sid, off)
    t.finalize()
    pick = rng.integers(0, len(key), 1500)
    qk, qo = key[pick], (off[pick] + 2).astype(np.uint32)
    qoff = np.array([0, len(qk)], np.uint64)
    want = t.match(qk, qo, qoff, 4)
    r0 = ctx.vt_redo_count()
    ctx.set_debug(2)   # SHZ_DEBUG_VT_PROBE1
    try:
        got = t.match(qk, qo, qoff, 4)
    finally:
        ctx.set_debug(0)
    assert ctx.vt_redo_count() == r0 + 1
    _same(want, got)
    t.close()
