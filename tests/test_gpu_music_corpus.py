"""GPU: the music-like corpus.
* the device generators (shz_synth_corpus: music tracks, traffic-like noise) equal their numpy twins bit for bit, at any
  start offset;
* on that corpus the device path equals the REFERENCE: hashes (hex20, t1) of every track in order, and the full
  result dicts of 24 queries -- crops at arbitrary sample offsets under traffic noise at 0 dB and -6 dB SNR mixed by the
  reference's rule (recognizer_test.py:426-435) -- from tests/golden/music_cases.* (made by make_golden_music.py, which
  runs the reference itself)."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import shazam_amd as S
    from oracle import synth
    return S, S.get_context(0), synth


def test_device_generators_equal_numpy_twins(env):
    S, ctx, synth = env
    for clip0, nc, n, start in ((0, 3, 50000, 0), (7, 2, 33333, 12345), (100000, 1, 70001, 999999)):
        d = ctx.synth_corpus(1, 4321, clip0, nc, n, 3000, 100, 1500, start)
        got = d.download(np.int16, nc * n).reshape(nc, n)
        d.free()
        for c in range(nc):
            assert np.array_equal(got[c], synth.music_clip(4321, clip0 + c, n, 3000, 100, start, 1500)), (clip0, c)
        d = ctx.synth_corpus(2, 777, clip0, nc, n, 2000, 0, 0, start)
        got = d.download(np.int16, nc * n).reshape(nc, n)
        d.free()
        for c in range(nc):
            assert np.array_equal(got[c], synth.traffic_noise(777, clip0 + c, n, 2000, start)), (clip0, c)
    # other amplitudes; no bed, no burst
    d = ctx.synth_corpus(1, 5, 2, 1, 40000, 6000, 0, 0, 0)
    assert np.array_equal(d.download(np.int16, 40000), synth.music_clip(5, 2, 40000, 6000, 0, 0, 0))
    d.free()


def _norm(res):
    out = []
    for r in res:
        r = dict(r)
        for k, v in r.items():
            if isinstance(v, bytes):
                r[k] = v.decode()
            elif isinstance(v, np.integer):
                r[k] = int(v)
        out.append(r)
    return out


def test_music_corpus_against_reference_goldens(env, golden_dir):
    S, ctx, synth = env
    g = np.load(os.path.join(golden_dir, "music_cases.npz"))
    meta = json.load(open(os.path.join(golden_dir, "music_cases.json")))
    p = meta["params"]
    db = S.get_database("hip")(ctx=ctx)
    for s in range(p["n_tracks"]):
        x = synth.music_clip(p["seed_tracks"], s, p["n_song"], p["amp"], p["bed"], burst=p["burst"])
        assert np.array_equal(np.frombuffer(hashlib.sha256(x.tobytes()).digest(), np.uint8), g[f"t{s}_pcm_sha256"])
        hs = S.fingerprint(x)
        assert [h.encode() for h, _ in hs] == list(g[f"t{s}_hash_hex"]) and [int(o) for _, o in hs] == list(g[f"t{s}_hash_t1"]), s
        fp = set(hs)
        sid = db.insert_song(f"m{s:04d}", hashlib.sha1(x.tobytes()).hexdigest().upper(), len(fp))
        assert sid == meta["songs"][s]["sid"] and len(fp) == meta["songs"][s]["total_hashes"]
        db.insert_hashes(sid, fp)
        db.set_song_fingerprinted(sid)
    for q in meta["queries"]:
        sig = synth.music_clip(p["seed_tracks"], q["song"], p["q_len"], p["amp"], p["bed"], start=q["start"], burst=p["burst"])
        if q["snr"] is not None:
            sig = synth.mix_query(sig, synth.traffic_noise(p["seed_noise"], q["q"], p["q_len"], p["traffic_amp"]), q["snr"])
        res, *_ = S.recognize(sig, db=db, topn=3)
        assert _norm(res) == q["results"], q["q"]
