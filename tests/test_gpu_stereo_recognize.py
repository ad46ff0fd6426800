"""GPU: recognise flow with multi-channel queries (`hashes |= set(fingerprints)` over the
channels, recognizer.py:378-382) against the oracle's restatement of the reference flow, result dict by result dict:
`input_total_hashes` is the size of the UNION, a hash present in both channels at the same offset counts once, at
different offsets twice."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_stereo_queries_union_of_channel_hashes():
    import shazam_amd as S
    from oracle import cpu_ref as O, synth
    ctx = S.get_context(0)
    n = 2048 * 90
    songs = [synth.synth_clip(55, i, n, 3500, 1800) for i in range(5)]
    db, odb = S.get_database("hip")(ctx=ctx), O.DictDB()
    for i, x in enumerate(songs):
        hs = set(S.fingerprint(x))
        assert hs == set(O.fingerprint(x))
        for d in (db, odb):
            sid = d.insert_song(f"song{i}", f"{i:040x}", len(hs))
            d.insert_hashes(sid, hs)
            d.set_song_fingerprinted(sid)
    db.finalize()
    rng = np.random.default_rng(2)
    queries = []
    for i, start in ((0, 7), (3, 21), (4, 2)):
        left = songs[i][start * 2048:start * 2048 + 2048 * 30]
        noise = rng.integers(-400, 400, len(left)).astype(np.int32)
        right = np.clip(left.astype(np.int32) + noise, -32768, 32767).astype(np.int16)
        queries.append([left, right])
    queries.append([songs[1][:2048 * 25], songs[1][2048 * 3:2048 * 28]])   # channels shifted by 3 frames: same hashes, other offsets
    queries.append([songs[2][:2048 * 20], songs[2][:2048 * 20].copy()])   # identical channels: the union is one channel's set
    got, _ = S.recognize_batch(queries, db, topn=3)
    for qi, chans in enumerate(queries):
        want = O.recognize(chans, odb, topn=3)
        assert len(got[qi]) == len(want) > 0
        for g, w in zip(got[qi], want):
            assert g == w, (qi, g, w)
    one = O.recognize([queries[4][0]], odb, topn=1)[0]
    assert got[4][0]["input_total_hashes"] == one["input_total_hashes"]          # identical channels add nothing
    both = got[0][0]["input_total_hashes"]
    assert both > len(set(O.fingerprint(queries[0][0])))                           # the noisy channel adds hashes
    # single-query entry point takes a list of channels too
    res, *_ = S.recognize(queries[1], db=db, topn=3)
    assert res == O.recognize(queries[1], odb, topn=3)
    db.close()
