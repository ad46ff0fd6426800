"""GPU: a key-sharded table must answer exactly like the unsharded one.

Every shard emits the packed votes of its rows (shz_match_pairs); they are concatenated / all-gathered and ranked
once (shz_pairs_vote).  Parity bar: every output array of Table.match, bit for bit, for several shard counts,
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


def test_pairs_capacity_layout_and_vote_inputs():
    import ctypes as C
    import shazam_amd as S
    from shazam_amd import _ffi
    from shazam_amd.shard import pairs_vote, table_maxima, vote_layout
    ctx = S.get_context(0)
    rng = np.random.default_rng(5)
    k, s, o = _rows(rng, 20000)
    t = S.Table(ctx)
    t.insert(k, s, o)
    t.finalize()
    assert table_maxima(t) == (int(s.max()), int(o.max()))
    qk, qo, qoff = _queries(rng, k, 8)
    nq = len(qoff) - 1
    ref = t.match(qk, qo, qoff, 2)
    lay = vote_layout(*table_maxima(t), qo, nq)

    def pairs(cap, layout=lay, shard=0, nshards=1):
        buf, cnt = ctx.alloc(max(cap, 1) * 8), C.c_uint64()
        nh, npr = np.zeros(nq, np.uint32), np.zeros(nq, np.uint64)
        rc = _ffi.lib().shz_match_pairs(ctx.h, t.h, _ffi.ptr(qk), _ffi.ptr(qo), qoff.ctypes.data_as(_ffi.u64p), nq, 0, shard,
                                        nshards, *layout, _ffi.ptr(buf), cap, C.byref(cnt), _ffi.ptr(nh), _ffi.ptr(npr))
        return rc, buf, int(cnt.value), nh, npr

    total = int(ref["npairs"].sum())
    rc, buf, n, nh, npr = pairs(total)
    assert rc == _ffi.OK and n == total and np.array_equal(nh, ref["nhash"]) and np.array_equal(npr, ref["npairs"])
    # too small a capacity reports the size needed
    rc2, small, n2, _, _ = pairs(7)
    assert rc2 == _ffi.E_CAPACITY and n2 == total
    small.free()
    # the votes of one table, in any order, rank to exactly match()
    v = buf.download(np.uint64, n)
    buf.upload(v[rng.permutation(n)])
    got = pairs_vote(ctx, buf, n, nq, lay, 2)
    for f in ("sid", "delta", "aligned", "dedup", "nres"):
        assert np.array_equal(ref[f], got[f]), f
    buf.free()
    # a layout that cannot hold this table's song ids or these queries' offsets is refused, not truncated
    for bad in ((lay[0] - 1, lay[1], lay[2]), (lay[0], lay[1], max(lay[2] - 1, 0)) if lay[2] else (lay[0], 1, 0)):
        rcb, b2, _, _, _ = pairs(total, layout=bad)
        assert rcb == _ffi.E_INVALID
        b2.free()
    # a shard index outside [0, nshards) too
    rcb, b2, _, _, _ = pairs(total, shard=2, nshards=2)
    assert rcb == _ffi.E_INVALID
    b2.free()
    # shards split the query hashes: the per-query counts of the parts add up to the whole
    parts = [pairs(total, shard=i, nshards=3) for i in range(3)]
    assert all(p[0] == _ffi.OK for p in parts)
    assert np.array_equal(sum(p[3] for p in parts), ref["nhash"])
    for p in parts:
        p[1].free()
    # no votes at all
    empty = pairs_vote(ctx, None, 0, 3, lay, 2)
    assert not empty["nres"].any()
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


def test_one_pass_partition_via_keep_shard():
    """shz_table_keep_shard runs the one-pass partition of the shard exchange (pack, one radix pass on the shard bits,
    unpack) and keeps one slice: over all shards the slices are disjoint, complete and agree with shard_of_keys."""
    import shazam_amd as S
    from shazam_amd import _ffi
    from shazam_amd.shard import shard_of_keys
    ctx = S.get_context(0)
    rng = np.random.default_rng(77)
    k, s, o = _rows(rng, 90001, nsongs=5000, noff=70000)
    want = np.unique(np.stack([k, s, o], 1).astype(np.uint64), axis=0)
    for nsh in (2, 5, 8):
        got = []
        for sh in range(nsh):
            t = S.Table(ctx)
            t.insert(k, s, o)
            ctx.check(_ffi.lib().shz_table_keep_shard(t.h, sh, nsh))
            kept = t.rows()[1]
            assert kept == int((shard_of_keys(k, nsh) == sh).sum())
            t.finalize()
            ek, es, eo = t.export()
            assert np.all(shard_of_keys(ek, nsh) == sh)
            got.append(np.stack([ek, es, eo], 1).astype(np.uint64))
            t.close()
        allrows = np.concatenate(got)
        assert len(allrows) == len(want)
        assert np.array_equal(np.unique(allrows, axis=0), want)
