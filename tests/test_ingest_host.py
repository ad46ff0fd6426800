"""CPU: host-side ingest helpers that mirror __init__.py:70-113, 286-323 (no GPU, no compute)."""
import hashlib
import os
import wave

import numpy as np

from shazam_amd import ingest


def _write_wav(path, chans, fs=44100):
    data = np.stack(chans, 1).astype("<i2").tobytes()
    with wave.open(str(path), "wb") as w:
        w.setnchannels(len(chans))
        w.setsampwidth(2)
        w.setframerate(fs)
        w.writeframes(data)


def test_read_unique_hash_find_files(tmp_path):
    rng = np.random.default_rng(0)
    l, r = rng.integers(-3000, 3000, 50000).astype(np.int16), rng.integers(-3000, 3000, 50000).astype(np.int16)
    (tmp_path / "sub").mkdir()
    _write_wav(tmp_path / "a.wav", [l, r])
    _write_wav(tmp_path / "sub" / "b.wav", [l], fs=22050)
    (tmp_path / "c.txt").write_text("x")
    files = sorted(ingest.find_files(str(tmp_path), [".wav"]))
    assert [os.path.basename(f) for f, _ in files] == ["a.wav", "b.wav"] and files[0][1] == "wav"
    channels, fs, sha = ingest.read(str(tmp_path / "a.wav"))
    assert fs == 44100 and len(channels) == 2 and np.array_equal(channels[0], l) and np.array_equal(channels[1], r)
    assert sha == hashlib.sha1(open(tmp_path / "a.wav", "rb").read()).hexdigest().upper() == ingest.unique_hash(str(tmp_path / "a.wav"))
    channels, fs, _ = ingest.read(str(tmp_path / "a.wav"), limit=1)
    assert len(channels[0]) == 44100 and np.array_equal(channels[1], r[:44100])
    channels, fs, _ = ingest.read(str(tmp_path / "sub" / "b.wav"))
    assert fs == 22050 and len(channels) == 1
