"""GPU: BASELINE.json configs[2]-[4] AS WRITTEN on one MI355X, under -m gpu (not only in builder-side bench_db runs).

  configs[2]/[3]: 100,000 synthetic 3-minute tracks into one HBM table (6.8e9 rows, 81 GB), then 5 s queries against it --
                  clean crops must all come back with their track and offset; noisy ones (SNR 0 dB, the reference's
                  ADD_NOISE rule) must give the SAME result arrays in a batch of 500, one by one, and through the exact
                  full sort (SHZ_MATCH_FULL_SORT): size-independent properties, the oracle cannot hold 6.8e9 rows.
  configs[4]:     1,000,000 tracks of 30 s (1.13e10 rows, 136 GB + the arena that holds the runs), 10 s queries in
                  batches of 200 and alone, new songs joining between query batches (the single-GPU half; N > 1 is the
                  driver's).
Skipped when the GPU has less free memory than the tables need (another process on the card)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FS = 44100


def _free_gb(ctx):
    free, _total = ctx.mem_info()
    return free / 2 ** 30


def _same(a, b, rows=None):
    for k in ("sid", "delta", "aligned", "dedup", "nres", "nhash", "npairs"):
        x, y = (a[k], b[k]) if rows is None else (a[k][rows], b[k])
        assert np.array_equal(x, y), k


def _queries(ctx, bench_db, tids, starts, qn, snr):
    q, bufs = bench_db.make_queries(ctx, tids, starts, qn, snr)
    k, t1, ho, _ = ctx.fingerprint_batch(q, np.arange(len(tids) + 1, dtype=np.uint64) * qn, fs=FS, pcm_device=True)
    for b in {id(b): b for b in bufs}.values():
        b.free()
    return k, t1, ho


def test_configs_2_3_100k_tracks_of_3_minutes():
    import bench_db
    import shazam_amd as S
    ctx = S.get_context(0)
    ctx.release_workspace()          # (what earlier tests left in the context's cache)
    if _free_gb(ctx) < 200:
        pytest.skip("needs ~180 GB of free HBM")
    songs, seconds = 100000, 180.0
    n = int(seconds * FS)
    tbl, st, bufs = bench_db.build_table(ctx, songs, seconds, chunk=500, finalize_every=10000)
    try:
        rows, _ = tbl.rows()
        assert 6.5e9 < rows <= st["rows_inserted"] and st["key_range_segments"]
        rng = np.random.default_rng(17)
        # clean crops on the frame grid: every one is found, at its offset
        nq, qn = 400, 5 * FS
        tids, fr = rng.integers(0, songs, nq), rng.integers(0, (n - qn) // 2048, nq)
        k, t1, ho = _queries(ctx, bench_db, tids, fr * 2048, qn, 300.0)
        res = tbl.match(k, t1, ho, 2)
        assert np.array_equal(res["sid"][:, 0], (tids + 1).astype(np.uint32)) and np.array_equal(res["delta"][:, 0], fr.astype(np.int32))
        # configs[3]: noisy 5 s queries, SNR 0 dB: one batch == one by one == the exact full sort
        nq = 500
        tids, st_ = rng.integers(0, songs, nq), rng.integers(0, n - qn, nq)
        k, t1, ho = _queries(ctx, bench_db, tids, st_, qn, 0.0)
        res = tbl.match(k, t1, ho, 2)
        assert res["npairs"].min() > 0
        pick = rng.choice(nq, 12, replace=False)
        for i in pick:
            a, b = int(ho[i]), int(ho[i + 1])
            one = tbl.match(k[a:b], t1[a:b], np.array([0, b - a], np.uint64), 2)
            _same(res, one, rows=slice(i, i + 1))
        sub = np.sort(pick[:6])
        kk = np.concatenate([k[int(ho[i]):int(ho[i + 1])] for i in sub])
        tt = np.concatenate([t1[int(ho[i]):int(ho[i + 1])] for i in sub])
        hh = np.concatenate([[0], np.cumsum([int(ho[i + 1] - ho[i]) for i in sub])]).astype(np.uint64)
        _same(tbl.match(kk, tt, hh, 2), tbl.match(kk, tt, hh, 2, full_sort=True))
    finally:
        for b in bufs[:2]:
            b.free()
        tbl.close()
        ctx.release_workspace()


def test_config_4_one_million_tracks_single_gpu_half():
    import bench_db
    import shazam_amd as S
    ctx = S.get_context(0)
    ctx.release_workspace()
    if _free_gb(ctx) < 262:
        pytest.skip("needs ~255 GB of free HBM (136 GB of columns + the arena that holds the runs)")
    songs, seconds = 1000000, 30.0
    n = int(seconds * FS)
    tbl, st, bufs = bench_db.build_table(ctx, songs, seconds, chunk=1000, finalize_every=63000)
    try:
        rows, _ = tbl.rows()
        assert 1.1e10 < rows <= st["rows_inserted"]
        rng = np.random.default_rng(23)
        nq, qn = 200, 10 * FS
        tids, st_ = rng.integers(0, songs, nq), rng.integers(0, n - qn, nq)
        k, t1, ho = _queries(ctx, bench_db, tids, st_, qn, 10.0)
        res = tbl.match(k, t1, ho, 2)
        assert (res["sid"][:, 0] == tids + 1).mean() >= 0.99
        for i in rng.choice(nq, 8, replace=False):      # the deferred-batch fold of ONE query == the batch's == the full sort
            a, b = int(ho[i]), int(ho[i + 1])
            off1 = np.array([0, b - a], np.uint64)
            one = tbl.match(k[a:b], t1[a:b], off1, 2)
            _same(res, one, rows=slice(i, i + 1))
            _same(one, tbl.match(k[a:b], t1[a:b], off1, 2, full_sort=True))
        # mixed stream: 1,000 new songs join (column path into the key-range table), old and new are found
        new0 = songs
        pcm = ctx.synth_pcm(bench_db.SEED_TRACKS, new0, 1000, n, 4000, 1500)
        kk, tt, hh, _ = ctx.fingerprint_batch(pcm, np.arange(1001, dtype=np.uint64) * n, fs=FS, pcm_device=True)
        pcm.free()
        tbl.insert_clips(kk, tt, hh, sid0=1 + new0)
        tbl.finalize()
        assert tbl.rows()[0] > rows
        tids2 = np.concatenate([rng.integers(0, songs, 20), rng.integers(new0, new0 + 1000, 20)])
        k2, t2, h2 = _queries(ctx, bench_db, tids2, rng.integers(0, n - qn, 40), qn, 10.0)
        r2 = tbl.match(k2, t2, h2, 2)
        assert (r2["sid"][:, 0] == tids2 + 1).mean() >= 0.95
    finally:
        for b in bufs[:2]:
            b.free()
        tbl.close()
        ctx.release_workspace()
