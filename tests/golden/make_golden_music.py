#!/usr/bin/env python3
"""Goldens for the music-like corpus, made by RUNNING THE REFERENCE on it.

Build-container only (needs /root/reference).  Tracks: oracle/synth.music_clip (integer generator, device twin
shz_synth_corpus); queries: crops at arbitrary sample offsets mixed with oracle/synth.traffic_noise at SNR 0 dB and -6 dB by
the reference's rule (recognizer_test.py:426-435).  Written: per track the reference's hashes (hex20, t1) and peaks, per query
its full result dicts -- inputs are regenerated from the seeds, only outputs are stored.

    python tests/golden/make_golden_music.py     # rewrites tests/golden/music_cases.npz / music_cases.json
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (loaders of the reference's functions)
from oracle import synth  # noqa: E402

SEED_TRACKS, SEED_NOISE = 4321, 777
N_TRACKS, N_SONG, Q_LEN = 12, 15 * 44100, 5 * 44100
AMP, BED, BURST, TRAFFIC_AMP = 3000, 100, 1500, 2000


def main():
    ref = G.load_reference_extraction()
    db = G.StandInDB()
    m = G.load_reference_match(db)
    arrs = {}
    songs = []
    for s in range(N_TRACKS):
        x = synth.music_clip(SEED_TRACKS, s, N_SONG, AMP, BED, burst=BURST)
        hashes = ref.fingerprint(x, Fs=44100)
        arrs[f"t{s}_hash_hex"] = np.array([h for h, _ in hashes], dtype="S20")
        arrs[f"t{s}_hash_t1"] = np.array([int(o) for _, o in hashes], np.int64)
        arrs[f"t{s}_pcm_sha256"] = np.frombuffer(hashlib.sha256(x.tobytes()).digest(), np.uint8)
        fp = set(hashes)
        sid = db.insert_song(f"m{s:04d}", hashlib.sha1(x.tobytes()).hexdigest().upper(), len(fp))
        db.insert_hashes(sid, fp)
        db.set_song_fingerprinted(sid)
        songs.append({"song": s, "sid": sid, "total_hashes": len(fp)})
        print("track", s, len(hashes), "hashes")
    rng = np.random.default_rng(9)
    queries = []
    for q in range(24):
        s = int(rng.integers(0, N_TRACKS))
        start = int(rng.integers(0, N_SONG - Q_LEN))
        sig = synth.music_clip(SEED_TRACKS, s, Q_LEN, AMP, BED, start=start, burst=BURST)
        snr = (0.0, -6.0, None)[q % 3]
        if snr is not None:
            sig = synth.mix_query(sig, synth.traffic_noise(SEED_NOISE, q, Q_LEN, TRAFFIC_AMP), snr)
        hashes = set(ref.fingerprint(sig, Fs=44100))
        matches, dedup, _ = m["find_matches"](hashes)
        res = m["align_matches"](matches, dedup, len(hashes), topn=3)
        for r in res:
            for k, v in list(r.items()):
                if isinstance(v, bytes):
                    r[k] = v.decode()
                elif isinstance(v, np.integer):
                    r[k] = int(v)
        queries.append({"q": q, "song": s, "start": start, "snr": snr, "n_hashes": len(hashes), "n_matches": len(matches),
                        "results": res})
    np.savez_compressed(os.path.join(HERE, "music_cases.npz"), **arrs)
    json.dump({"params": {"seed_tracks": SEED_TRACKS, "seed_noise": SEED_NOISE, "n_tracks": N_TRACKS, "n_song": N_SONG, "q_len": Q_LEN,
                          "amp": AMP, "bed": BED, "burst": BURST, "traffic_amp": TRAFFIC_AMP},
               "songs": songs, "queries": queries}, open(os.path.join(HERE, "music_cases.json"), "w"), indent=1)
    ok = sum(1 for q in queries if q["results"] and q["results"][0]["song_id"] == q["song"] + 1)
    print("queries:", len(queries), "top-1 correct:", ok)


if __name__ == "__main__":
    main()
