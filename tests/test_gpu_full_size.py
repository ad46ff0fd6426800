"""GPU, BASELINE.json sizes, through size-independent properties.

configs[1] (1,000 x 30 s clips, one call): sampled clips bit-exact against the oracle; the result of a clip does not
depend on the batch around it; per-clip offsets are consistent.
configs[3]-shaped round trip (ingest -> crop -> match): every clean hop-aligned crop of an ingested track comes back
as that track at offset = the crop's first frame, with every query hash matched (`hashes_matched_in_input` >= the
query's hash count, the track holds all of them) -- 3,000 tracks, 600 queries in one batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 44100


def test_config2_1000_clips_sampled_against_oracle():
    import shazam_amd as S
    from oracle import cpu_ref as O, synth
    ctx = S.get_context(0)
    n_clips, n = 1000, 30 * FS
    pcm = ctx.synth_pcm(1234, 0, n_clips, n, 4000, 1500)
    off = np.arange(n_clips + 1, dtype=np.uint64) * n
    k, t1, ho, cnt = ctx.fingerprint_batch(pcm, off, pcm_device=True)
    assert cnt == len(k) == int(ho[-1]) > 10_000_000 and np.all(np.diff(ho.astype(np.int64)) > 5000)
    assert int(t1.max()) < 644                                    # offsets are frames of the 30 s clip
    for c in (0, 1, 499, 733, 999):
        x = synth.synth_clip(1234, c, n, 4000, 1500)              # numpy twin of the device generator
        ok, ot, _, _ = O.fingerprint_keys(x)
        a, b = int(ho[c]), int(ho[c + 1])
        assert np.array_equal(k[a:b], ok.astype(np.uint32)) and np.array_equal(t1[a:b], ot.astype(np.uint32)), c
    # the last 150 clips alone (short peak_pick segments instead of long ones): the same hashes
    sub = ctx.fingerprint_batch(pcm.ptr + 850 * n * 2, np.arange(151, dtype=np.uint64) * n, pcm_device=True)
    a = int(ho[850])
    assert np.array_equal(sub[0], k[a:]) and np.array_equal(sub[1], t1[a:])
    assert np.array_equal(sub[2].astype(np.int64), ho[850:].astype(np.int64) - a)
    pcm.free()


def test_ingest_crop_match_round_trip_3000_tracks():
    import shazam_amd as S
    ctx = S.get_context(0)
    n_tracks, n = 3000, 30 * FS
    tbl = S.Table(ctx)
    cap = 1000 * 700 * 40
    kb, tb = ctx.alloc(cap * 4), ctx.alloc(cap * 4)
    pcm = ctx.alloc(1000 * n * 2)
    off = np.arange(1001, dtype=np.uint64) * n
    for c0 in range(0, n_tracks, 1000):
        ctx.synth_pcm(4321, c0, 1000, n, 4000, 1500, out=pcm)
        _, _, ho, _ = ctx.fingerprint_batch(pcm, off, pcm_device=True, out_key=kb, out_t1=tb, cap=cap)
        tbl.insert_clips(kb, tb, ho, sid0=1 + c0, device=True)
    tbl.finalize()
    assert tbl.rows()[0] > 30_000_000
    rng = np.random.default_rng(3)
    nq, qn = 600, 5 * FS
    tids = rng.integers(0, n_tracks, nq)
    starts = rng.integers(0, (n - qn) // 2048, nq)
    q = ctx.alloc(nq * qn * 2)
    from shazam_amd import _ffi
    for i in range(nq):   # device-side crop: samples [start, start + qn) of track tids[i]
        ctx.check(_ffi.lib().shz_synth_pcm(ctx.h, 4321, int(tids[i]), 1, qn, 4000, 1500, int(starts[i]) * 2048,
                                           _ffi.vp(q.ptr + i * qn * 2)))
    k, t1, ho, _ = ctx.fingerprint_batch(q, np.arange(nq + 1, dtype=np.uint64) * qn, pcm_device=True)
    res = tbl.match(k, t1, ho, 2)
    assert np.all(res["nres"] >= 1)
    assert np.array_equal(res["sid"][:, 0], (1 + tids).astype(res["sid"].dtype))
    assert np.array_equal(res["delta"][:, 0], starts.astype(res["delta"].dtype))
    # interior frames of a crop see the same 21-frame neighbourhoods as in the track: most query hashes are the track's
    assert np.all(res["aligned"][:, 0] >= 0.6 * res["nhash"])
    assert np.all(res["dedup"][:, 0] >= res["aligned"][:, 0])
    for b in (kb, tb, pcm, q):
        b.free()
    tbl.close()
