"""GPU: a table split into several segments (tiny segment size) must answer lookups and matches exactly
like the single-segment table -- the mechanism that lifts the 2^32-row limit of one radix sort."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(rng, n):
    key = (rng.integers(0, 60, n).astype(np.uint32) << 20) | (rng.integers(0, 60, n).astype(np.uint32) << 8) | \
        rng.integers(0, 4, n).astype(np.uint32)
    return key, rng.integers(1, 80, n).astype(np.uint32), rng.integers(0, 500, n).astype(np.uint32)


def test_segmented_table_equals_single():
    import shazam_amd as S
    from oracle import cpu_ref as O
    ctx = S.get_context(0)
    rng = np.random.default_rng(8)
    parts = [_mk(rng, n) for n in (30000, 5000, 41000, 900)]
    # songs must not repeat across inserts (segments dedup only internally): give each part its own sid range
    for i, (k, s, o) in enumerate(parts):
        s += np.uint32(100 * i)
    one, seg = S.Table(ctx), S.Table(ctx)
    seg.set_segment_rows(16000)
    for k, s, o in parts:
        one.insert(k, s, o)
        seg.insert(k, s, o)
        seg.finalize()                     # incremental: opens new segments as rows arrive
    one.finalize()
    assert one.rows()[0] == seg.rows()[0]
    a = np.stack(one.export(), 1).astype(np.uint64)
    b = np.stack(seg.export(), 1).astype(np.uint64)
    assert np.array_equal(a, np.unique(b, axis=0)) and len(a) == len(b)
    assert one.song_rows(105) == seg.song_rows(105)
    keys = np.concatenate([parts[0][0][:40], np.array([0xFFFFFFFF, 7], np.uint32)])
    la = np.stack(one.lookup(keys), 1).astype(np.uint64)
    lb = np.stack(seg.lookup(keys), 1).astype(np.uint64)
    assert len(la) == len(lb)
    # same rows per key (order inside a key differs: segment order vs global order)
    pa, pb = 0, 0
    for kk in keys.astype(np.uint64):
        na = int((la[pa:, 0] == kk).cumprod().sum())
        nb = int((lb[pb:, 0] == kk).cumprod().sum())
        assert na == nb
        assert np.array_equal(np.unique(la[pa:pa + na], axis=0), np.unique(lb[pb:pb + nb], axis=0))
        pa, pb = pa + na, pb + nb
    # match: identical top-n, counts, dedup, pair totals
    nq = 12
    qk, qo, qoff = [], [], [0]
    allk = np.concatenate([p[0] for p in parts])
    for q in range(nq):
        m = int(rng.integers(1, 150))
        qk.append(allk[rng.integers(0, len(allk), m)])
        qo.append(rng.integers(0, 40, m).astype(np.uint32))
        qoff.append(qoff[-1] + m)
    qk, qo, qoff = np.concatenate(qk), np.concatenate(qo), np.array(qoff, np.uint64)
    ra, rb = one.match(qk, qo, qoff, 4), seg.match(qk, qo, qoff, 4)
    for f in ("sid", "delta", "aligned", "dedup", "nres", "nhash", "npairs"):
        assert np.array_equal(ra[f], rb[f]), f
    # and both equal the oracle's vote
    odb = O.DictDB()
    for s_ in range(1, 400):
        odb.insert_song(str(s_), "00", 1)
    for k, s, o in parts:
        for kk, ss, oo in zip(k.tolist(), s.tolist(), o.tolist()):
            odb.insert_hashes(ss, [(kk, oo)])
    for q in range(nq):
        hs = set(zip(qk[qoff[q]:qoff[q + 1]].tolist(), qo[qoff[q]:qoff[q + 1]].tolist()))
        m, dd = O.return_matches(hs, odb)
        want = O.vote(m, 4)
        got = [(int(rb["sid"][q, i]), int(rb["delta"][q, i]), int(rb["aligned"][q, i])) for i in range(int(rb["nres"][q]))]
        assert got == [tuple(w) for w in want] and int(rb["npairs"][q]) == len(m)
    one.close()
    seg.close()


def test_sparse_hits_over_many_segments():
    """Few matching rows under thousands of empty (hash, segment) sub-groups: one expand tile then spans far more
    sub-groups than its LDS tables hold (the global-memory branch of m_expand_kernel), with query hashes that occur at
    several offsets (n_offsets > 1 per group) in both branches."""
    import shazam_amd as S
    from oracle import cpu_ref as O
    ctx = S.get_context(0)
    rng = np.random.default_rng(31)
    seg, one, odb = S.Table(ctx), S.Table(ctx), O.DictDB()
    seg.set_segment_rows(700)
    for s_ in range(1, 300):
        odb.insert_song(str(s_), "00", 1)
    present = []
    for part in range(18):                                  # 18 finalizes of 600 rows -> ~16 segments
        key = ((rng.integers(300, 900, 600) << 20) | (rng.integers(0, 2049, 600) << 8) | rng.integers(0, 201, 600)).astype(np.uint32)
        sid = (rng.integers(1, 17, 600) + 16 * part).astype(np.uint32)
        off = rng.integers(0, 90, 600).astype(np.uint32)
        for t in (seg, one):
            t.insert(key, sid, off)
        seg.finalize()
        present.append(key)
        for kk, ss, oo in zip(key.tolist(), sid.tolist(), off.tolist()):
            odb.insert_hashes(ss, [(kk, oo)])
    one.finalize()
    present = np.concatenate(present)
    qk, qo, qoff = [], [], [0]
    for q in range(3):
        miss = ((rng.integers(0, 290, 3500) << 20) | (rng.integers(0, 2049, 3500) << 8) | rng.integers(0, 201, 3500)).astype(np.uint32)
        hit = present[rng.integers(0, len(present), 40)]
        hit = np.concatenate([hit, hit[:15], hit[:5]])      # the same hash at two and three query offsets
        k = np.concatenate([miss, hit])
        o = rng.integers(0, 30, len(k)).astype(np.uint32)
        p = rng.permutation(len(k))
        qk.append(k[p]); qo.append(o[p]); qoff.append(qoff[-1] + len(k))
    qk, qo, qoff = np.concatenate(qk), np.concatenate(qo), np.array(qoff, np.uint64)
    ra, rb = one.match(qk, qo, qoff, 6), seg.match(qk, qo, qoff, 6)
    for f in ("sid", "delta", "aligned", "dedup", "nres", "nhash", "npairs"):
        assert np.array_equal(ra[f], rb[f]), f
    for q in range(3):
        hs = set(zip(qk[qoff[q]:qoff[q + 1]].tolist(), qo[qoff[q]:qoff[q + 1]].tolist()))
        m, dd = O.return_matches(hs, odb)
        want = O.vote(m, 6)
        got = [(int(rb["sid"][q, i]), int(rb["delta"][q, i]), int(rb["aligned"][q, i])) for i in range(int(rb["nres"][q]))]
        assert got == [tuple(w) for w in want] and int(rb["npairs"][q]) == len(m) > 40
        assert [int(rb["dedup"][q, i]) for i in range(len(got))] == [dd[w[0]] for w in want]
    one.close()
    seg.close()
