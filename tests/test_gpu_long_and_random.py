"""GPU: a 3-minute track (BASELINE config 3 unit: 3,874 frames, 16 time segments of peak_pick) and
randomised ragged batches against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_three_minute_track_matches_oracle():
    import shazam_amd as S
    from oracle import cpu_ref as O, synth
    x = synth.synth_clip(77, 0, 7938000, 4000, 1500)
    k, t1, ho = S.fingerprint_batch([x, x[:2048 * 300 + 17]])
    ok, ot1, pf, pt = O.fingerprint_keys(x)
    assert len(ok) > 60000 and int(pt.max()) <= 3873
    assert np.array_equal(k[:ho[1]], ok) and np.array_equal(t1[:ho[1]], ot1)
    ok2, ot2, _, _ = O.fingerprint_keys(x[:2048 * 300 + 17])
    assert np.array_equal(k[ho[1]:], ok2) and np.array_equal(t1[ho[1]:], ot2)


def test_random_ragged_batches():
    import shazam_amd as S
    from oracle import cpu_ref as O
    rng = np.random.default_rng(2026)
    for trial in range(3):
        clips = []
        for _ in range(int(rng.integers(3, 9))):
            n = int(rng.choice([0, 1, 100, 4095, 4096, 4097, 6143, 6144, 10000, 50001, 131072, 200003]))
            kind = rng.integers(0, 4)
            if kind == 0:
                x = rng.integers(-20000, 20000, n)
            elif kind == 1:
                x = (8000 * np.sin(np.arange(n) * rng.uniform(0.01, 1.0)) + rng.integers(-200, 200, n)).round()
            elif kind == 2:
                x = np.repeat(rng.integers(-30000, 30000, n // 37 + 1), 37)[:n]   # staircase: broadband steps
            else:
                x = np.zeros(n)
                x[n // 3: n // 2] = rng.integers(-9000, 9000, max(0, n // 2 - n // 3))
            clips.append(np.clip(x, -32768, 32767).astype(np.int16))
        k, t1, ho = S.fingerprint_batch(clips)
        for i, x in enumerate(clips):
            ok, ot1, _, _ = O.fingerprint_keys(x)
            assert np.array_equal(k[ho[i]:ho[i + 1]], ok), (trial, i, len(x))
            assert np.array_equal(t1[ho[i]:ho[i + 1]], ot1), (trial, i, len(x))
