"""CPU: the oracle (oracle/cpu_ref.py, oracle/thirdparty_ref.py) against the golden
vectors captured from the reference itself (tests/golden/make_golden.py).

Bar: peaks and hashes bit-exact (indices, hex strings, order); spectrogram dB
within 1e-5 relative (north star) -- the oracle actually lands ~1e-12.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import cpu_ref as C
from oracle import synth, thirdparty_ref as T

REL = 1e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _check_case(g, prefix, x, Fs, impl=C):
    A = impl.spectrogram_db(x, Fs)
    assert A.shape[1] == int(g[f"{prefix}n_frames"])
    pf, pt, pv = g[f"{prefix}probe_f"], g[f"{prefix}probe_t"], g[f"{prefix}probe_db"]
    got = A[pf, pt]
    np.testing.assert_allclose(got, pv, rtol=REL, atol=1e-9)
    np.testing.assert_allclose(A.sum(), float(g[f"{prefix}sum_db"]), rtol=1e-9)
    f, t = impl.peaks_2d(A)
    np.testing.assert_array_equal(f, g[f"{prefix}peaks_f"])
    np.testing.assert_array_equal(t, g[f"{prefix}peaks_t"])
    hashes = impl.fingerprint(x, Fs=Fs)
    assert [h.encode() for h, _ in hashes] == list(g[f"{prefix}hash_hex"])
    assert [int(o) for _, o in hashes] == list(g[f"{prefix}hash_t1"])
    return A


@pytest.mark.parametrize("impl", [C, T], ids=["numpy", "thirdparty"])
@pytest.mark.parametrize("fs", [22050, 44100])
def test_wav_known_answers(golden_dir, fs, impl):
    g = _load(golden_dir, "wav_kat.npz")
    x = g["pcm"]
    A = _check_case(g, f"fs{fs}_", x, fs, impl)
    np.testing.assert_allclose(A[:, 0], g[f"fs{fs}_col0_db"], rtol=REL, atol=1e-9)
    if fs == 22050:  # SURVEY 8c anchors
        hashes = impl.fingerprint(x, Fs=fs)
        assert len(hashes) == 1626 and hashes[0] == ("987a1bcc49e707cb9e6a", 0)
        assert hashes[-1] == ("fca775991d3b605dd017", 105)
        assert abs(A[100, 50] - 36.417874919345) < 1e-9


@pytest.mark.parametrize("name", ["white_5s", "white_30s", "tonal_5s", "tonal_30s", "tonal_list_input_2s"])
def test_synth_clips(golden_dir, name):
    g = _load(golden_dir, "synth_clips.npz")
    seed, clip, n, ta, na = (int(v) for v in g[f"{name}_params"])
    x = synth.synth_clip(seed, clip, n, ta, na)
    assert hashlib.sha256(x.tobytes()).hexdigest().encode() == bytes(g[f"{name}_pcm_sha256"])
    _check_case(g, f"{name}_", x, 44100)


EDGE = ["short_3000", "exact_4096", "ragged_6143", "two_frames_6144", "silence_20000", "square_p64",
        "gap_250_frames", "loud_fullscale", "dc_offset"]


@pytest.mark.parametrize("impl", [C, T], ids=["numpy", "thirdparty"])
@pytest.mark.parametrize("name", EDGE)
def test_edge_cases(golden_dir, name, impl):
    g = _load(golden_dir, "edge_cases.npz")
    _check_case(g, f"{name}_", g[f"{name}_pcm"], 44100, impl)


def test_packed_keys_match_hex(golden_dir):
    g = _load(golden_dir, "synth_clips.npz")
    seed, clip, n, ta, na = (int(v) for v in g["tonal_5s_params"])
    k, t1, f, t = C.fingerprint_keys(synth.synth_clip(seed, clip, n, ta, na))
    assert [h.encode() for h in C.sha1_hex20(k)] == list(g["tonal_5s_hash_hex"])
    assert C.sha1_prefix10(k[:50]).tobytes().hex() == "".join(h.decode() for h in g["tonal_5s_hash_hex"][:50])
    f1, f2, dt = C.unpack_key(k)
    assert np.array_equal(C.pack_key(f1, f2, dt), k) and dt.max() <= 200


def test_frame_count():
    for n in (0, 1, 3000, 4095, 4096, 4097, 6143, 6144, 8191, 8192, 220500, 1323000, 7938000):
        want = 1 if n < 4096 else (n - 2048) // 2048
        assert C.frame_count(n) == want


def _song_pcm(s, p):
    if s == 7:
        return synth.synth_clip(p["seed"], 3, p["n"], p["tone_amp"], p["noise_amp"])
    if s == 11:
        half = synth.synth_clip(p["seed"], 11, 2048 * 100, p["tone_amp"], p["noise_amp"])
        return np.concatenate([half, half])
    return synth.synth_clip(p["seed"], s, p["n"], p["tone_amp"], p["noise_amp"])


@pytest.fixture(scope="module")
def match_golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "match_cases.json")))


@pytest.fixture(scope="module")
def mini_db(match_golden):
    p = match_golden["song_params"]
    db = C.DictDB()
    pcm = {}
    for s in range(20):
        x = _song_pcm(s, p)
        pcm[s] = x
        fp = set(C.fingerprint(x))
        sid = db.insert_song(f"{s:06d}", hashlib.sha1(x.tobytes()).hexdigest().upper(), len(fp))
        assert sid == match_golden["songs"][s]["sid"] and len(fp) == match_golden["songs"][s]["total_hashes"]
        db.insert_hashes(sid, fp)
        db.set_song_fingerprinted(sid)
    return db, pcm


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


def test_match_align_audio_queries(match_golden, mini_db):
    db, pcm = mini_db
    for q in match_golden["queries"]:
        sig = pcm[q["song"]][q["start"]:q["start"] + 220500]
        if q["snr"] is not None:
            sig = synth.mix_query(sig, synth.synth_clip(777, q["q"], 220500, 0, 8000), q["snr"])
        hashes = set(C.fingerprint(sig))
        assert len(hashes) == q["n_hashes"]
        matches, dedup = C.return_matches(hashes, db)
        assert len(matches) == q["n_matches"]
        assert {str(k): v for k, v in sorted(dedup.items())} == q["dedup"]
        assert _norm(C.align_matches(matches, dedup, len(hashes), db, topn=3)) == q["results"]


def test_match_align_crafted_ties(match_golden):
    c = match_golden["crafted"]
    db = C.DictDB()
    for sid_s, rr in c["rows"].items():
        rr = [(h, o) for h, o in rr]
        sid = db.insert_song(f"c{sid_s}", "AB" * 20, len(set(rr)))
        assert sid == int(sid_s)
        db.insert_hashes(sid, rr)
    qh = set((h, o) for h, o in c["query"])
    matches, dedup = C.return_matches(qh, db)
    assert sorted([list(m) for m in matches]) == c["matches_sorted"]
    assert {str(k): v for k, v in sorted(dedup.items())} == c["dedup"]
    for topn in (1, 2, 3, 10):
        assert _norm(C.align_matches(matches, dedup, len(qh), db, topn=topn)) == c[f"results_top{topn}"]
    # keys recorded for the crafted hashes are the packed preimages
    for h, k in c["keys"].items():
        assert C.sha1_hex20(np.array([k], np.uint32))[0] == h
