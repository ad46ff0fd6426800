"""GPU: the storage plugin with `shards=N` (key-partitioned tables on one GPU) behaves like the default one:
ingest from files, SELECT through the cursor emulation, recognize, dump/load and the MySQL row export."""
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


def test_sharded_plugin_equals_default(tmp_path):
    import shazam_amd as S
    from shazam_amd import ingest
    from oracle import synth
    ctx = S.get_context(0)
    songs = []
    for i in range(5):
        x = synth.synth_clip(31, i, 2048 * 150 + 17 * i, 3000, 2000)
        _write_wav(tmp_path / f"s{i}.wav", [x])
        songs.append(x)
    a = S.get_database("hip")(ctx=ctx)
    b = S.get_database("hip")(ctx=ctx, shards=3)
    da = ingest.fingerprint_directory(str(tmp_path), [".wav"], a)
    db_ = ingest.fingerprint_directory(str(tmp_path), [".wav"], b)
    assert sorted(da) == sorted(db_)
    assert a.num_fingerprints() == b.num_fingerprints()
    ka, kb = a.table.export(), b.table.export()
    assert all(np.array_equal(x, y) for x, y in zip(ka, kb))
    for sid in range(1, 6):
        assert a.table.song_rows(sid) == b.table.song_rows(sid)
    # reference-style SELECT ... WHERE hash IN (...) through the cursor (recognizer.py:251-259)
    hexes = [h.upper() for h in S.hex_of_keys(ctx, ka[0][::997])] + ["00" * 10]
    rows = []
    for d in (a, b):
        with d.cursor() as cur:
            cur.execute(d.SELECT_MULTIPLE % ", ".join([d.IN_MATCH] * len(hexes)), hexes)
            rows.append(sorted(cur))
    assert rows[0] == rows[1] and len(rows[0]) >= len(hexes) - 1
    # recognize: identical result dicts (all ten keys, recognizer.py:321-334)
    qs = [songs[1][9 * 2048:9 * 2048 + 2 * 44100], songs[4][30 * 2048 + 5:30 * 2048 + 5 + 3 * 44100]]
    ra, rb = S.recognize_batch(qs, a, topn=3), S.recognize_batch(qs, b, topn=3)
    assert ra[0] == rb[0]
    sid_of = {name: sid for sid, name, _ in da}          # ids follow the directory walk order
    assert ra[0][0][0]["song_id"] == sid_of["s1"] and ra[0][0][0]["offset"] == 9
    assert ra[0][1][0]["song_id"] == sid_of["s4"] and ra[0][1][0]["offset"] == 30
    # dump with shards, load without (and the other way round): same table
    b.save(str(tmp_path / "dump"))
    c = S.get_database("hip").load(str(tmp_path / "dump"), ctx=ctx)
    d2 = S.get_database("hip").load(str(tmp_path / "dump"), ctx=ctx, shards=2)
    for other in (c, d2):
        assert all(np.array_equal(x, y) for x, y in zip(ka, other.table.export()))
    assert list(a.export_mysql_rows())[:50] == list(d2.export_mysql_rows())[:50]
    for d in (a, b, c, d2):
        d.close()
