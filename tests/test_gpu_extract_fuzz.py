"""GPU: randomised differential test of the extraction pass.

Every batch is fingerprinted three ways -- fp32 staging + verification (the default), fp64 staging, and clip by clip --
and a sample of its clips by the oracle.  All must agree bit for bit.  The inputs are chosen to sit where the fp32
pass has decisions to hand over: amplitudes from a few counts to full scale (window maxima near the amp_min
threshold), amp_min anywhere in the distribution, lengths from a fraction of a window to thousands of frames,
silence, clipping, repeated material, odd sample offsets inside the packed PCM buffer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import shazam_amd as S
    from oracle import cpu_ref as O, synth
    return S, S.get_context(0), O, synth


def _clip(rng, synth, kind, n):
    amp = int(10 ** rng.uniform(0.5, 4.5))
    if kind == 0:
        return synth.synth_clip(int(rng.integers(1 << 30)), int(rng.integers(1000)), n, 0, max(amp, 2))
    if kind == 1:
        return synth.synth_clip(int(rng.integers(1 << 30)), int(rng.integers(1000)), n, min(amp, 8000), max(amp // 8, 1))
    if kind == 2:   # clipped
        x = synth.synth_clip(int(rng.integers(1 << 30)), 3, n, 0, 30000).astype(np.int32) * 3
        return np.clip(x, -32768, 32767).astype(np.int16)
    if kind == 3:   # noise with silent stretches
        x = synth.synth_clip(int(rng.integers(1 << 30)), 5, n, 0, max(amp, 2)).copy()
        for _ in range(3):
            a = int(rng.integers(0, max(n - 1, 1)))
            x[a:a + int(rng.integers(1, 40000))] = 0
        return x
    x = synth.synth_clip(int(rng.integers(1 << 30)), 7, max(n // 3, 1), 2000, 300)   # the same material three times
    return np.concatenate([x, x, x])[:n] if n >= 3 else x[:n]


def test_random_batches_three_ways(env):
    S, ctx, O, synth = env
    rng = np.random.default_rng(20261004)
    checked_oracle = 0
    for trial in range(14):
        nc = int(rng.integers(1, 24))
        clips = [_clip(rng, synth, int(rng.integers(0, 5)), int(rng.choice([rng.integers(1, 4096), rng.integers(4096, 30000),
                                                                            rng.integers(30000, 400000)])))
                 for _ in range(nc)]
        amp_min = float(rng.choice([10.0, 0.0, rng.uniform(0, 70), rng.uniform(20, 50)]))
        fan = int(rng.choice([5, 5, 2, 9]))
        off = np.concatenate([[0], np.cumsum([len(x) for x in clips])]).astype(np.uint64)
        x = np.concatenate(clips)
        ctx.set_stage_f64(False)
        a = ctx.fingerprint_batch(x, off, amp_min=amp_min, fan_value=fan)
        pa = ctx.peaks(x, off, amp_min=amp_min)
        ctx.set_stage_f64(True)
        try:
            b = ctx.fingerprint_batch(x, off, amp_min=amp_min, fan_value=fan)
            pb = ctx.peaks(x, off, amp_min=amp_min)
        finally:
            ctx.set_stage_f64(False)
        for u, v in zip(a + pa, b + pb):
            assert np.array_equal(u, v), (trial, amp_min, fan)
        k, t1, ho, _ = a
        for c in rng.choice(nc, size=min(nc, 3), replace=False):
            xc = np.ascontiguousarray(clips[c])
            k1, t11, _, _ = ctx.fingerprint_batch(xc, np.array([0, len(xc)], np.uint64), amp_min=amp_min, fan_value=fan)
            assert np.array_equal(k[ho[c]:ho[c + 1]], k1) and np.array_equal(t1[ho[c]:ho[c + 1]], t11), (trial, c)
            if len(xc) < 150000:
                ok, ot1, _, _ = O.fingerprint_keys(xc, fan_value=fan, amp_min=amp_min)
                # the oracle's logarithm is numpy's: a window maximum within one rounding of amp_min may fall either way
                if not (np.array_equal(k1, ok) and np.array_equal(t11, ot1)):
                    A = O.spectrogram_db(xc)
                    assert np.any(np.abs(A - amp_min) < 1e-9), (trial, c, len(k1), len(ok))
                checked_oracle += 1
    assert checked_oracle >= 20
    print("extract stats after the fuzz:", ctx.extract_stats())


def test_quiet_clips_around_the_threshold(env):
    """Window maxima spread across amp_min = 10 dB: noise of a few counts."""
    S, ctx, O, synth = env
    for na in (2, 3, 4, 6, 9, 14):
        clips = [synth.synth_clip(808, c, 90000, 0, na) for c in range(12)]
        off = np.concatenate([[0], np.cumsum([len(x) for x in clips])]).astype(np.uint64)
        x = np.concatenate(clips)
        ctx.set_stage_f64(False)
        a = ctx.fingerprint_batch(x, off)
        ctx.set_stage_f64(True)
        try:
            b = ctx.fingerprint_batch(x, off)
        finally:
            ctx.set_stage_f64(False)
        assert all(np.array_equal(u, v) for u, v in zip(a, b)), na
        ok, ot1, _, _ = O.fingerprint_keys(clips[0])
        assert np.array_equal(a[0][a[2][0]:a[2][1]], ok), na


def test_two_identical_frames_are_decided_on_fp64_values(env):
    """Exact ties between exactly two cells of a window: two frames of a noise clip carry the same samples, so cell
    (t, f) and cell (t + 5, f) hold the same power.  fp32 cannot rank them and there are only two of them: the
    verification kernel has to recompute both in fp64 (no fp64 pass), find them equal, and mark BOTH where they are
    their window's maximum -- every tied cell is a peak."""
    S, ctx, O, synth = env
    clips = []
    for c in range(6):
        x = synth.synth_clip(515, c, 2048 * 70, 0, 6000).copy()
        a, b = 10 + c, 15 + c
        x[2048 * b:2048 * b + 4096] = x[2048 * a:2048 * a + 4096]
        clips.append(x)
    off = np.concatenate([[0], np.cumsum([len(x) for x in clips])]).astype(np.uint64)
    x = np.concatenate(clips)
    s0 = ctx.extract_stats()
    ctx.set_stage_f64(False)
    a = ctx.fingerprint_batch(x, off)
    pa = ctx.peaks(x, off)
    s1 = ctx.extract_stats()
    assert s1["f64_passes"] == s0["f64_passes"], "two tied cells per window must not need the fp64 pass"
    assert s1["decided_f64"] - s0["decided_f64"] >= 20 and s1["frames_recomputed"] > s0["frames_recomputed"]
    ctx.set_stage_f64(True)
    try:
        b = ctx.fingerprint_batch(x, off)
        pb = ctx.peaks(x, off)
    finally:
        ctx.set_stage_f64(False)
    for u, v in zip(a + pa, b + pb):
        assert np.array_equal(u, v)
    k, t1, ho, _ = a
    n_pairs = 0
    for c, xc in enumerate(clips):
        ok, ot1, of, ot = O.fingerprint_keys(xc)
        assert np.array_equal(k[ho[c]:ho[c + 1]], ok) and np.array_equal(t1[ho[c]:ho[c + 1]], ot1), c
        pk = set(zip(of.tolist(), ot.tolist()))
        n_pairs += sum(1 for (f, t) in pk if t == 10 + c and (f, 15 + c) in pk)
    assert n_pairs >= 3, "the construction must produce peaks that tie across the two frames"
