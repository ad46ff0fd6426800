"""GPU: ingest orchestrator, table dump/load/export and the digital harness
against the oracle."""
import os
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _write_wav(path, chans, fs=44100):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(len(chans))
        w.setsampwidth(2)
        w.setframerate(fs)
        w.writeframes(np.stack(chans, 1).astype("<i2").tobytes())


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    from oracle import synth
    d = tmp_path_factory.mktemp("songs")
    songs = {}
    for i in range(6):
        left = synth.synth_clip(21, i, 2048 * 120 + 31 * i, 4000, 1200)
        if i % 2:   # stereo: right channel = left plus a little independent noise
            right = np.clip(left.astype(np.int32) + synth.synth_clip(22, i, len(left), 0, 300), -32768, 32767).astype(np.int16)
            chans = [left, right]
        else:
            chans = [left]
        _write_wav(d / f"{100 + i}.wav", chans, 44100 if i != 4 else 22050)
        songs[str(100 + i)] = chans
    return str(d), songs


def test_fingerprint_directory_matches_oracle(corpus):
    import shazam_amd as S
    from shazam_amd import ingest
    from oracle import cpu_ref as O
    d, songs = corpus
    db = S.get_database("hip")(ctx=S.get_context(0))
    done = ingest.fingerprint_directory(d, [".wav"], db, batch_files=4)
    assert len(done) == 6
    by_name = {name: (sid, n) for sid, name, n in done}
    want_rows = []
    for name, chans in songs.items():
        fp = set()
        for c in chans:
            k, t1, _, _ = O.fingerprint_keys(c)          # hashes do not depend on Fs
            fp |= set(zip(k.tolist(), t1.tolist()))
        sid, n = by_name[name]
        assert n == len(fp) == db.get_song_by_id(sid)["total_hashes"] == db.table.song_rows(sid)
        want_rows += [(k, sid, o) for k, o in fp]
    k, s, o = db.table.export()
    assert sorted(zip(k.tolist(), s.tolist(), o.tolist())) == sorted(want_rows)
    # second run: everything is skipped by file SHA-1 (__init__.py:346-348)
    assert ingest.fingerprint_directory(d, [".wav"], db) == []
    by_song = {row[1]: row for row in db.get_songs()}
    assert len(by_song) == 6 and by_song["100"][2] == ingest.unique_hash(os.path.join(d, "100.wav"))

    # dump / load round trip and the MySQL row export
    path = os.path.join(d, "table.npz")
    db.save(path)
    db2 = S.get_database("hip").load(path, ctx=db.ctx)
    for a, b in zip(db.table.export(), db2.table.export()):
        assert np.array_equal(a, b)
    assert db2.get_song_by_id(3) == db.get_song_by_id(3) and db2.insert_song("x", "00", 1) == 7
    rows = list(db.export_mysql_rows())
    assert len(rows) == len(k)
    assert rows[0] == (int(s[0]), bytes.fromhex(O.sha1_hex20(k[:1])[0]), int(o[0]))
    assert rows[-1][1] == bytes.fromhex(O.sha1_hex20(k[-1:])[0])

    # harness: clean 3 s crops of every song are recognised; CSV has the reference's columns
    from shazam_amd import harness
    mono = [(name, chans[0]) for name, chans in songs.items() if name != "104"]   # 104 is the 22.05 kHz file
    rows = harness.run(db, mono, record_seconds=3, topn=2, seed=1)
    assert harness.accuracy(rows) == 1.0
    noise = np.random.default_rng(0).integers(-8000, 8000, 44100 * 20).astype(np.int16)
    rows_n = harness.run(db, mono, record_seconds=3, add_noise=True, snr=10, noise=noise, seed=2)
    assert harness.accuracy(rows_n) >= 0.6
    out = os.path.join(d, "res.csv")
    harness.write_csv(rows, out)
    head = open(out).readline().strip().split(",")
    assert head == harness.CSV_COLUMNS
    # the oracle recognises the same crops identically
    odb = O.DictDB()
    for sid, name, n in sorted(done):
        assert odb.insert_song(name, db.get_song_by_id(sid)["file_sha1"], n) == sid
    for kk, ss, oo in zip(k.tolist(), s.tolist(), o.tolist()):
        odb.insert_hashes(ss, [(kk, oo)])
    from random import Random
    rnd = Random(1)
    for (name, pcm), row in zip(mono, rows):
        st = rnd.randrange(0, max(1, int(len(pcm) / 44100) - 3))
        sig = pcm[st * 44100: st * 44100 + 3 * 44100]
        kq, tq, _, _ = O.fingerprint_keys(sig)
        hs = set(zip(kq.tolist(), tq.tolist()))
        m, dd = O.return_matches(hs, odb)
        want = O.align_matches(m, dd, len(hs), odb, topn=2)
        assert str(want) == row["final_results"]
