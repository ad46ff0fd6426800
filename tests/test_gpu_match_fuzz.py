"""GPU: randomized differential test of the match (probe -> expand -> sort -> fold -> top-n) against a vectorised numpy
statement of return_matches / align_matches (recognizer.py:222-338): many table shapes (one or many segments, hot keys,
duplicate rows), query shapes (repeated hashes, several offsets per hash, empty queries) and vote counts around the
kernels' tile sizes (8 votes per thread in the fold, 2,048 per expand tile, 4,096 per sort tile)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _expected(tk, ts, to, qk, qo, qoff, topn):
    """per query: list of (sid, delta, aligned, dedup) ranked like align_matches, plus n_pairs and n_hashes"""
    rows = np.unique(np.stack([tk, ts, to], 1).astype(np.int64), axis=0)          # UNIQUE(sid, offset, hash)
    order = np.argsort(rows[:, 0], kind="stable")
    rows = rows[order]
    out = []
    for q in range(len(qoff) - 1):
        a, b = int(qoff[q]), int(qoff[q + 1])
        hs = np.unique(np.stack([qk[a:b], qo[a:b]], 1).astype(np.int64), axis=0)  # set of (hash, offset)
        if len(hs) == 0:
            out.append(([], 0, 0))
            continue
        lo = np.searchsorted(rows[:, 0], hs[:, 0], "left")
        hi = np.searchsorted(rows[:, 0], hs[:, 0], "right")
        cnt = hi - lo
        idx = np.repeat(lo, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
        sid = rows[idx, 1]
        delta = rows[idx, 2] - np.repeat(hs[:, 1], cnt)
        npairs = len(sid)
        # dedup_hashes[sid]: DB rows whose hash is queried, once per row
        keys_q = np.unique(hs[:, 0])
        l2 = np.searchsorted(rows[:, 0], keys_q, "left")
        h2 = np.searchsorted(rows[:, 0], keys_q, "right")
        c2 = h2 - l2
        idx2 = np.repeat(l2, c2) + (np.arange(c2.sum()) - np.repeat(np.cumsum(c2) - c2, c2))
        ds, dc = np.unique(rows[idx2, 1], return_counts=True)
        dedup = dict(zip(ds.tolist(), dc.tolist()))
        if npairs == 0:
            out.append(([], 0, len(hs)))
            continue
        pk = (sid << 32) | (delta + (1 << 24))
        u, c = np.unique(pk, return_counts=True)
        us, ud = u >> 32, (u & 0xFFFFFFFF) - (1 << 24)
        best = {}
        for s_, d_, c_ in zip(us.tolist(), ud.tolist(), c.tolist()):     # ascending (sid, delta): first max wins
            if s_ not in best or c_ > best[s_][1]:
                best[s_] = (d_, c_)
        ranked = sorted(best.items(), key=lambda kv: (-kv[1][1], kv[0]))[:topn]
        out.append(([(s_, d_, c_, dedup[s_]) for s_, (d_, c_) in ranked], npairs, len(hs)))
    return out


@pytest.mark.parametrize("seed", range(24))
def test_match_equals_numpy_reference(seed):
    import shazam_amd as S
    ctx = S.get_context(0)
    rng = np.random.default_rng(1000 + seed)
    n_rows = int(rng.choice([50, 3000, 40000, 250000]))
    n_keys = int(rng.choice([3, 40, 2000, 60000]))
    n_sid = int(rng.choice([2, 17, 300, 5000]))
    max_off = int(rng.choice([5, 200, 3000]))
    keyspace = ((rng.integers(0, 2049, n_keys) << 20) | (rng.integers(0, 2049, n_keys) << 8) | rng.integers(0, 201, n_keys)).astype(np.uint32)
    tk = keyspace[rng.integers(0, n_keys, n_rows)]
    ts = rng.integers(1, n_sid + 1, n_rows).astype(np.uint32)
    to = rng.integers(0, max_off + 1, n_rows).astype(np.uint32)
    t = S.Table(ctx)
    parts = int(rng.integers(1, 4))
    if seed % 3 == 1:
        # several segments; rows are unique inside a segment, not across segments (DESIGN.md limits: a song is ingested
        # once), so every insert gets song ids of its own
        t.set_segment_rows(max(16, n_rows // 5))
        for pi, part in enumerate(np.array_split(np.arange(n_rows), parts)):
            ts[part] += np.uint32(pi * (n_sid + 1))
    for part in np.array_split(np.arange(n_rows), parts):
        t.insert(tk[part], ts[part], to[part])
        t.finalize()
    nq = int(rng.choice([1, 2, 7, 40]))
    qk, qo, qoff = [], [], [0]
    for q in range(nq):
        m = int(rng.choice([0, 1, 9, 150, 2500]))
        k = keyspace[rng.integers(0, n_keys, m)] if m else np.zeros(0, np.uint32)
        if m and rng.random() < 0.5:                         # hashes that are not in the table at all
            k[rng.integers(0, m, max(1, m // 3))] = np.uint32(0xFFF00000) | rng.integers(0, 1 << 20, max(1, m // 3)).astype(np.uint32)
        qk.append(k)
        qo.append(rng.integers(0, int(rng.choice([1, 4, 120])), m).astype(np.uint32))
        qoff.append(qoff[-1] + m)
    qk, qo, qoff = np.concatenate(qk).astype(np.uint32), np.concatenate(qo).astype(np.uint32), np.array(qoff, np.uint64)
    topn = int(rng.choice([1, 2, 5, 16]))
    res = t.match(qk, qo, qoff, topn)
    want = _expected(tk, ts, to, qk, qo, qoff, topn)
    for q, (ranked, npairs, nhash) in enumerate(want):
        assert int(res["npairs"][q]) == npairs and int(res["nhash"][q]) == nhash, (seed, q)
        got = [(int(res["sid"][q, i]), int(res["delta"][q, i]), int(res["aligned"][q, i]), int(res["dedup"][q, i]))
               for i in range(int(res["nres"][q]))]
        assert got == ranked, (seed, q)
    t.close()
