"""GPU: schema semantics of the fingerprints table beyond one sorted segment.

  * UNIQUE(song_id, offset, hash) + INSERT IGNORE (mysql_database.py:54-55, 62-68) across segments: a row that already
    sits in a frozen segment is ignored when it arrives again, so `dedup_hashes` (DB rows per song, recognizer.py:261-264)
    and the aligned counts do not double;
  * ON DELETE CASCADE (mysql_database.py:57-58) behind DELETE_UNFINGERPRINTED (:132-134, called at __init__.py:424):
    a song that never reached set_song_fingerprinted leaves with all its rows;
  * DROP TABLE (mysql_database.py:122-124) empties the table.
Checked against numpy set arithmetic and against the oracle's DictDB."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import shazam_amd as S
    from shazam_amd import _ffi
    return S, _ffi, S.get_context(0)


def _rows(rng, n, nsid=40):
    k = rng.integers(0, 1 << 24, n).astype(np.uint32) << np.uint32(8) | rng.integers(0, 6, n).astype(np.uint32)
    return k, rng.integers(1, nsid, n).astype(np.uint32), rng.integers(0, 5000, n).astype(np.uint32)


def _triples(k, s, o):
    return set(zip(k.tolist(), s.tolist(), o.tolist()))


def test_unique_across_segments(env):
    S, F, ctx = env
    rng = np.random.default_rng(3)
    t = F.Table(ctx)
    t.set_segment_rows(3000)
    want = set()
    batches = [_rows(rng, 2500) for _ in range(4)]
    for i, (k, s, o) in enumerate(batches):
        t.insert(k, s, o)
        if i >= 1:   # the previous batch again: it sits in a frozen segment by now
            pk, ps, po = batches[i - 1]
            t.insert(pk[::2], ps[::2], po[::2])
        t.finalize()
        want |= _triples(k, s, o)
    k, s, o = t.export()
    assert len(k) == len(want) and _triples(k, s, o) == want
    # a whole song inserted twice, the copies two finalize calls apart: match counts its rows once
    t2 = F.Table(ctx)
    t2.set_segment_rows(2000)
    keys = (np.arange(1500, dtype=np.uint32) * 977 + 5) << np.uint32(8)
    offs = np.arange(1500, dtype=np.uint32)
    t2.insert(keys, np.full(1500, 7, np.uint32), offs)
    t2.finalize()
    fk, fs, fo = _rows(rng, 1900)
    t2.insert(fk, fs + 100, fo)
    t2.finalize()                                   # freezes the first segment
    t2.insert(keys, np.full(1500, 7, np.uint32), offs)
    t2.finalize()
    assert t2.rows()[0] == len(_triples(keys, np.full(1500, 7), offs) | _triples(fk, fs + 100, fo))
    res = t2.match(keys[:400], offs[:400] + 3, np.array([0, 400], np.uint64), 2)
    assert int(res["sid"][0, 0]) == 7 and int(res["delta"][0, 0]) == -3
    assert int(res["dedup"][0, 0]) == 400 and int(res["aligned"][0, 0]) == 400
    t.close()
    t2.close()


def test_delete_songs_everywhere(env):
    S, F, ctx = env
    rng = np.random.default_rng(5)
    t = F.Table(ctx)
    t.set_segment_rows(4000)
    allrows = set()
    for _ in range(3):
        k, s, o = _rows(rng, 3500)
        t.insert(k, s, o)
        t.finalize()
        allrows |= _triples(k, s, o)
    k, s, o = _rows(rng, 500)
    t.insert(k, s, o)                               # staged, not finalized: deleted there too
    allrows |= _triples(k, s, o)
    gone = [3, 17, 39, 1000]
    n_del = t.delete_songs(gone)
    t.finalize()
    want = {r for r in allrows if r[1] not in gone}
    k, s, o = t.export()
    assert _triples(k, s, o) == want and len(k) == len(want)
    assert n_del >= len(allrows) - len(want)        # staged duplicates may add to the count
    for sid in gone:
        assert t.song_rows(sid) == 0
    # the table still answers: every remaining row is found through its bucket
    some = np.array(sorted(want)[:200], np.uint32)
    lk, ls, lo = t.lookup(np.unique(some[:, 0]))
    assert _triples(lk, ls, lo) >= {tuple(r) for r in some.tolist()}
    # deleting everything leaves an empty, usable table
    t.delete_songs(np.arange(0, 200, dtype=np.uint32))
    t.finalize()
    assert t.rows()[0] == 0
    t.insert(np.array([5 << 8], np.uint32), np.array([1], np.uint32), np.array([2], np.uint32))
    t.finalize()
    assert t.rows()[0] == 1
    t.clear()
    assert t.rows() == (0, 0)
    t.close()


def test_unfingerprinted_song_never_matches(env):
    """insert_song + insert_hashes without set_song_fingerprinted (a crashed ingest), then setup(): the reference's
    DELETE_UNFINGERPRINTED + CASCADE.  The query is a crop of exactly that song."""
    S, F, ctx = env
    from oracle import cpu_ref as O, synth
    db = S.get_database("hip")(ctx=ctx)
    odb = O.DictDB()
    clips = [synth.synth_clip(31, c, 2048 * 80, 4000, 1500) for c in range(4)]
    for c, x in enumerate(clips):
        fp = set(S.fingerprint(x))
        sid = db.insert_song(f"s{c}", "AB" * 20, len(fp))
        db.insert_hashes(sid, fp)
        if c != 2:
            db.set_song_fingerprinted(sid)
            osid = odb.insert_song(f"s{c}", "AB" * 20, len(fp))
            assert osid == sid or c == 3
            odb.insert_hashes(osid, fp)
    n_before = db.num_fingerprints()
    db.setup()
    assert db.num_fingerprints() < n_before and db.table.song_rows(3) == 0
    q = clips[2][2048 * 11:2048 * 50]
    got, *_ = S.recognize(q, db=db, topn=3)
    assert all(r["song_id"] != 3 for r in got)
    for r in got:                                     # whatever matches by chance is a live song
        assert db.get_song_by_id(r["song_id"])["song_name"] in ("s0", "s1", "s3")
    # a live song is still found, and the reference's empty() clears everything
    got, *_ = S.recognize(clips[1][2048 * 5:2048 * 40], db=db, topn=2)
    assert got[0]["song_id"] == 2 and got[0]["offset"] == 5
    with db.cursor() as cur:
        cur.execute("DROP TABLE IF EXISTS `fingerprints`;")
    assert db.num_fingerprints() == 0
    db.close()


def test_sharded_table_delete(env):
    S, F, ctx = env
    from shazam_amd.shard import ShardedTable
    rng = np.random.default_rng(9)
    k, s, o = _rows(rng, 6000)
    st, t = ShardedTable(ctx, nshards=3), F.Table(ctx)
    for tb in (st, t):
        tb.insert(k, s, o)
        tb.finalize()
        tb.delete_songs([4, 5, 6])
        tb.finalize()
    assert st.rows()[0] == t.rows()[0] == len({r for r in _triples(k, s, o) if r[1] not in (4, 5, 6)})
    st.close()
    t.close()


def test_segments_are_topped_up_before_they_freeze(env):
    """Batches of 0.55 x the segment limit: the active segment takes the slices (by key) of the next batch that still fit
    before it is frozen, so 12 batches end in ~7 segments of ~limit rows, not in 12 half-empty ones -- and the rows are
    exactly the union of the batches (duplicates across batches included)."""
    S, F, ctx = env
    rng = np.random.default_rng(17)
    limit, per = 20000, 11000
    t = F.Table(ctx)
    t.set_segment_rows(limit)
    want = set()
    prev = None
    for i in range(12):
        k, s, o = _rows(rng, per, nsid=4000)
        t.insert(k, s, o)
        if prev is not None:                     # part of the previous batch again: some of it sits in a frozen segment
            t.insert(prev[0][::7], prev[1][::7], prev[2][::7])
        t.finalize()
        want |= _triples(k, s, o)
        prev = (k, s, o)
    k, s, o = t.export()
    assert len(k) == len(want) and _triples(k, s, o) == want
    assert t.segments() <= -(-len(want) // limit) + 2, t.segments()
    # a query still finds every copy: lookup of a key that went into different segments
    probe = np.unique(k[:200])
    lk, ls, lo = t.lookup(probe)
    assert _triples(lk, ls, lo) == {x for x in want if x[0] in set(probe.tolist())}
