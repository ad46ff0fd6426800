"""GPU: the vote fold's bar logic where it is most delicate -- MANY songs that tie at the top, spread over hundreds of
tiles (rows of 400,000 songs; 0.6-1.2 M votes a query; planted groups of songs with exactly equal best counts at low, middle
and high song ids), topn 1..8: the tiles' filter / query-wide bar / deferred batches must give the arrays of the exact
full sort (SHZ_MATCH_FULL_SORT), for queries alone and in a batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    for k in ("sid", "delta", "aligned", "dedup", "nres", "nhash", "npairs"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("seed,n_songs,rows_per_key,planted", [(1, 400000, 3000, 5), (2, 400000, 6000, 3), (3, 60000, 3000, 2), (4, 1000000, 4000, 4)])
def test_many_tied_songs_over_many_tiles(seed, n_songs, rows_per_key, planted):
    import shazam_amd as S
    ctx = S.get_context(0)
    rng = np.random.default_rng(seed)
    n_keys, nq = 600, 3
    keys = ((rng.permutation(2049 * 64)[:n_keys] // 64 << 20) | (rng.integers(0, 2049, n_keys) << 8) | rng.integers(0, 201, n_keys)).astype(np.uint32)
    keys = np.unique(keys)
    n_keys = len(keys)
    # table rows: every key in `rows_per_key` random songs at offsets 100 + {0, 1, 2} -> plenty of (song, delta) pairs of count 2-3
    tk = np.repeat(keys, rows_per_key)
    ts = rng.integers(1, n_songs + 1, len(tk)).astype(np.uint32)
    to = (100 + rng.integers(0, 3, len(tk))).astype(np.uint32)
    # planted: groups of 12 songs (ids low / middle / high) that get exactly `planted` + 3 votes at one delta from each query's hashes
    groups = [np.arange(5, 17), np.arange(n_songs // 2, n_songs // 2 + 12), np.arange(n_songs - 20, n_songs - 8)]
    t = S.Table(ctx)
    extra_k, extra_s, extra_o = [], [], []
    qsel = [rng.choice(n_keys, 200, replace=False) for _ in range(nq)]
    for q in range(nq):
        for g in groups:
            ks = keys[qsel[q][: planted + 8]]
            for s_ in g:
                extra_k.append(ks)
                extra_s.append(np.full(len(ks), s_, np.uint32))
                extra_o.append(np.full(len(ks), 700 + q, np.uint32))
    tk = np.concatenate([tk] + extra_k)
    ts = np.concatenate([ts] + extra_s)
    to = np.concatenate([to] + extra_o)
    t.insert(tk, ts, to)
    t.finalize()
    qk = np.concatenate([keys[s] for s in qsel])
    qo = np.concatenate([np.full(200, 10 + q, np.uint32) for q in range(nq)])
    qoff = np.arange(nq + 1, dtype=np.uint64) * 200
    for topn in (1, 2, 5, 8):
        fast = t.match(qk, qo, qoff, topn)
        _same(fast, t.match(qk, qo, qoff, topn, full_sort=True))
        assert fast["npairs"].min() > 500000
        for q in range(nq):
            one = t.match(qk[200 * q:200 * q + 200], qo[200 * q:200 * q + 200], np.array([0, 200], np.uint64), topn)
            for k in ("sid", "delta", "aligned", "dedup", "nres"):
                assert np.array_equal(one[k][0], fast[k][q]), (k, q, topn)
        if n_songs >= 400000 and rows_per_key <= 4000:   # (sparse enough that the planted 36 songs hold the best count: the lowest ids win)
            assert np.array_equal(fast["sid"][:, 0], np.full(nq, 5, np.uint32)) and np.all(fast["aligned"][:, 0] >= planted + 8)
            if topn == 8:
                assert np.array_equal(fast["sid"][0], np.arange(5, 13, dtype=np.uint32))
    t.close()
