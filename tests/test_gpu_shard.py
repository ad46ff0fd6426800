"""GPU: a key-sharded table must answer exactly like the unsharded one.

The shards vote separately (shz_match_votes), the records are concatenated / all-gathered and merged
(shz_votes_merge).  Parity bar: every output array of Table.match, bit for bit, for several shard counts,
including tie cases (equal counts for two songs / two offset differences) and queries without matches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("sid", "delta", "aligned", "dedup", "nres", "nhash", "npairs")


def _rows(rng, n, nsongs=60, noff=400):
    key = (rng.integers(0, 50, n).astype(np.uint32) << 20) | (rng.integers(0, 50, n).astype(np.uint32) << 8) | \
        rng.integers(0, 5, n).astype(np.uint32)
    return key, rng.integers(1, nsongs, n).astype(np.uint32), rng.integers(0, noff, n).astype(np.uint32)


def _queries(rng, allk, nq, maxlen=200, maxoff=60):
    qk, qo, qoff = [], [], [0]
    for q in range(nq):
        m = 0 if q == 3 else int(rng.integers(1, maxlen))          # query 3 is empty
        kk = allk[rng.integers(0, len(allk), m)] if q != 5 else np.full(m, 0xFFFFFFF0, np.uint32)   # query 5 matches nothing
        qk.append(kk.astype(np.uint32))
        qo.append(rng.integers(0, maxoff, m).astype(np.uint32))
        qoff.append(qoff[-1] + m)
    return np.concatenate(qk), np.concatenate(qo), np.array(qoff, np.uint64)


@pytest.mark.parametrize("nshards", [1, 2, 3, 8])
def test_sharded_match_equals_unsharded(nshards):
    import shazam_amd as S
    from shazam_amd.shard import ShardedTable
    ctx = S.get_context(0)
    rng = np.random.default_rng(100 + nshards)
    k, s, o = _rows(rng, 120000)
    one, sh = S.Table(ctx), ShardedTable(ctx, nshards=nshards)
    one.insert(k, s, o)
    sh.insert(k, s, o)
    one.finalize()
    sh.finalize()
    assert one.rows()[0] == sh.rows()[0]            # duplicates share a key, hence a shard: same dedup
    qk, qo, qoff = _queries(rng, k, 40)
    for topn in (1, 3):
        ra, rb = one.match(qk, qo, qoff, topn), sh.match(qk, qo, qoff, topn)
        for f in FIELDS:
            assert np.array_equal(ra[f], rb[f]), (f, nshards, topn)
    one.close()
    sh.close()


def test_sharded_fingerprint_db_and_ties():
    """Real fingerprints through insert_clips + keep_shard, with duplicated songs so that two songs tie."""
    import shazam_amd as S
    from shazam_amd.shard import ShardedTable
    from oracle import synth
    ctx = S.get_context(0)
    n = 5 * 44100
    clips = [synth.synth_clip(77, c, n) for c in range(6)]
    clips.append(clips[2].copy())                                  # song 7 == song 3: every vote ties
    key, t1, hoff = S.fingerprint_batch(clips, Fs=44100)
    one, sh = S.Table(ctx), ShardedTable(ctx, nshards=3)
    one.insert_clips(key, t1, hoff, 1)
    sh.insert_clips(key, t1, hoff, 1)
    one.finalize()
    sh.finalize()
    assert one.rows()[0] == sh.rows()[0]
    # queries: crops of songs 3, 5 and a mix
    qs = [clips[2][20 * 2048:20 * 2048 + 3 * 44100], clips[4][7 * 2048:7 * 2048 + 2 * 44100]]
    qkey, qt1, qhoff = S.fingerprint_batch(qs, Fs=44100)
    ra, rb = one.match(qkey, qt1, qhoff, 3), sh.match(qkey, qt1, qhoff, 3)
    for f in FIELDS:
        assert np.array_equal(ra[f], rb[f]), f
    assert int(rb["sid"][0, 0]) == 3 and int(rb["sid"][0, 1]) == 7      # tie -> smaller song id first
    assert int(rb["delta"][0, 0]) == 20 and int(rb["aligned"][0, 0]) == int(rb["aligned"][0, 1])
    assert int(rb["sid"][1, 0]) == 5 and int(rb["delta"][1, 0]) == 7
    one.close()
    sh.close()


def test_votes_capacity_and_merge_inputs():
    import ctypes as C
    import shazam_amd as S
    from shazam_amd import _ffi
    from shazam_amd.shard import match_votes, votes_merge
    ctx = S.get_context(0)
    rng = np.random.default_rng(5)
    k, s, o = _rows(rng, 20000)
    t = S.Table(ctx)
    t.insert(k, s, o)
    t.finalize()
    qk, qo, qoff = _queries(rng, k, 8)
    cols, n, nhash, npairs = match_votes(t, qk, qo, qoff)
    assert n > 0 and all(len(c) == n for c in cols)
    assert int(cols[3].sum()) == int(npairs.sum())           # every match is in exactly one record
    assert np.all(cols[4] <= cols[3])
    # too small a capacity reports the size needed and writes nothing past it
    cnt = C.c_uint64()
    small = [np.full(4, 0xAB, dt) for dt in (np.uint32, np.uint32, np.int32, np.uint32, np.uint32)]
    rc = _ffi.lib().shz_match_votes(ctx.h, t.h, _ffi.ptr(qk), _ffi.ptr(qo), qoff.ctypes.data_as(_ffi.u64p), len(qoff) - 1, 0,
                                    *[_ffi.ptr(c) for c in small], 4, C.byref(cnt), None, None)
    assert rc == _ffi.E_CAPACITY and cnt.value == n
    # merging the records of one table reproduces match(); record order does not matter
    ref = t.match(qk, qo, qoff, 2)
    perm = rng.permutation(n)
    got = votes_merge(ctx, [c[perm] for c in cols], n, len(qoff) - 1, 2)
    for f in ("sid", "delta", "aligned", "dedup", "nres"):
        assert np.array_equal(ref[f], got[f]), f
    # no records at all
    empty = votes_merge(ctx, [], 0, 3, 2)
    assert not empty["nres"].any()
    # a query index outside [0, n_queries) is rejected
    with pytest.raises(_ffi.ShzError):
        votes_merge(ctx, cols, n, 2, 2)
    t.close()


def test_one_rank_communicator_paths():
    """shard exchange + votes all-gather over a 1-rank RCCL communicator (the N>1 wiring with N = 1)."""
    import shazam_amd as S
    from shazam_amd import _ffi
    from shazam_amd.shard import ShardedTable
    ctx = S.get_context(0)
    try:
        uid = _ffi.comm_unique_id()
    except _ffi.ShzError:
        pytest.skip("librccl.so not loadable")
    comm = _ffi.Comm(ctx, uid, 0, 1)
    rng = np.random.default_rng(9)
    k, s, o = _rows(rng, 50000)
    one, sh = S.Table(ctx), ShardedTable(ctx, comm=comm)
    one.insert(k, s, o)
    sh.insert(k, s, o)
    one.finalize()
    sh.finalize()
    assert one.rows()[0] == sh.rows()[0] and sh.bytes_received == 0
    qk, qo, qoff = _queries(rng, k, 20)
    ra, rb = one.match(qk, qo, qoff, 2), sh.match(qk, qo, qoff, 2)
    for f in FIELDS:
        assert np.array_equal(ra[f], rb[f]), f
    sh.close()
    one.close()
    comm.close()
