"""GPU parity (run with -m gpu on the MI355X box): the HIP extraction path, called through the
C ABI, against the golden vectors captured from the reference and against the oracle.

Bars: peaks / keys / hex hashes bit-exact incl. order; dB spectrogram within 1e-5 relative in
power (= 4.3e-5 dB absolute), the north-star tolerance.
"""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DB_TOL = 10 * np.log10(1 + 1e-5)  # 1e-5 relative on the power spectrum, expressed in dB


@pytest.fixture(scope="module")
def S():
    import shazam_amd
    return shazam_amd


@pytest.fixture(scope="module")
def ctx(S):
    return S.get_context(0)


@pytest.fixture(scope="module")
def O():
    from oracle import cpu_ref
    return cpu_ref


def _golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _cases(golden_dir):
    """(name, pcm, Fs, golden dict, prefix) for every extraction fixture."""
    from oracle import synth
    out = []
    g = _golden(golden_dir, "wav_kat.npz")
    for fs in (22050, 44100):
        out.append((f"wav{fs}", g["pcm"], fs, g, f"fs{fs}_"))
    g = _golden(golden_dir, "synth_clips.npz")
    for name in ("white_5s", "white_30s", "tonal_5s", "tonal_30s", "tonal_list_input_2s"):
        seed, clip, n, ta, na = (int(v) for v in g[f"{name}_params"])
        out.append((name, synth.synth_clip(seed, clip, n, ta, na), 44100, g, f"{name}_"))
    g = _golden(golden_dir, "edge_cases.npz")
    for name in ("short_3000", "exact_4096", "ragged_6143", "two_frames_6144", "silence_20000", "square_p64",
                 "gap_250_frames", "loud_fullscale", "dc_offset"):
        out.append((name, g[f"{name}_pcm"], 44100, g, f"{name}_"))
    return out


def test_synth_matches_numpy_twin(ctx):
    from oracle import synth
    for (ta, na, n, start) in ((0, 8000, 50001, 0), (4000, 1500, 70000, 0), (6000, 0, 40000, 12345), (0, 32768, 4099, 7),
                               (4000, 1500, 200003, 16381), (10000, 32768, 33000, 8), (1, 0, 9, 16383)):
        buf = ctx.synth_pcm(1234, 5, 3, n, ta, na, start)
        got = buf.download(np.int16, 3 * n).reshape(3, n)
        for c in range(3):
            want = synth.synth_clip(1234, 5 + c, n, ta, na, start)
            assert np.array_equal(got[c], want), (ta, na, n, c)
        buf.free()


def test_stages_against_reference_goldens(golden_dir, S, ctx, O):
    worst = 0.0
    for name, x, fs, g, p in _cases(golden_dir):
        x = np.ascontiguousarray(x, np.int16)
        off = np.array([0, len(x)], np.uint64)
        A = ctx.stft_db(x if len(x) else np.zeros(1, np.int16), off, fs=fs)[0]
        assert A.shape == (2049, int(g[f"{p}n_frames"])), name
        got = A[g[f"{p}probe_f"], g[f"{p}probe_t"]]
        err = np.abs(got - g[f"{p}probe_db"]).max()
        worst = max(worst, err)
        assert err <= DB_TOL, (name, err)
        # full-array check against the oracle as well
        Ao = O.spectrogram_db(x, fs)
        assert np.abs(A - Ao).max() <= DB_TOL, name
        assert abs(A.sum() - float(g[f"{p}sum_db"])) <= 1e-6 * max(1.0, abs(float(g[f"{p}sum_db"]))), name
        # peaks: device order is (t, f); golden is np.where order (f, t)
        pf, pt, po = ctx.peaks(x, off, fs=fs)
        order = np.lexsort((pt, pf))
        assert np.array_equal(pf[order], g[f"{p}peaks_f"]) and np.array_equal(pt[order], g[f"{p}peaks_t"]), name
        assert np.all(np.diff(pt.astype(np.int64)) >= 0), name
        # end to end through the reference-shaped API
        hashes = S.fingerprint(x, Fs=fs)
        assert [h.encode() for h, _ in hashes] == list(g[f"{p}hash_hex"]), name
        assert [o for _, o in hashes] == list(g[f"{p}hash_t1"]), name
    print("max |dB - reference| over probes:", worst)


def test_list_input_and_python_ints(S, golden_dir):
    g = _golden(golden_dir, "wav_kat.npz")
    x = g["pcm"][:30000]
    assert S.fingerprint(list(x)) == S.fingerprint(x) == S.fingerprint([int(v) for v in x])
    assert S.fingerprint(np.zeros(0, np.int16)) == []
    with pytest.raises(NotImplementedError):
        S.fingerprint(x.astype(np.float64) / 3.0)


def test_ragged_batch_equals_single_calls(golden_dir, S, ctx, O):
    cases = [c for c in _cases(golden_dir) if c[2] == 44100 and c[0] != "white_30s"]
    clips = [np.ascontiguousarray(c[1], np.int16) for c in cases]
    k, t1, ho = S.fingerprint_batch(clips, ctx=ctx)
    assert ho[0] == 0 and ho[-1] == len(k)
    for i, (name, x, fs, g, p) in enumerate(cases):
        ok, ot1, _, _ = O.fingerprint_keys(x)
        assert np.array_equal(k[ho[i]:ho[i + 1]], ok), name
        assert np.array_equal(t1[ho[i]:ho[i + 1]], ot1), name
        assert [h.encode() for h in S.hex_of_keys(ctx, k[ho[i]:ho[i + 1]])] == list(g[f"{p}hash_hex"]), name
    # odd sample offsets inside the packed PCM buffer (clips of odd length precede others) are covered above;
    # force tiny sub-batches: same answer
    ctx.set_workspace_limit(700 * 2056 * 8)  # one 30 s clip (644 frames) at a time
    try:
        k2, t2, ho2 = S.fingerprint_batch(clips, ctx=ctx)
    finally:
        ctx.set_workspace_limit(0)
    assert np.array_equal(k, k2) and np.array_equal(t1, t2) and np.array_equal(ho, ho2)


def test_device_resident_io(ctx, O):
    from oracle import synth
    n, nc = 2048 * 50 + 77, 5
    pcm = ctx.synth_pcm(42, 0, nc, n, 3000, 2000)
    off = np.arange(nc + 1, dtype=np.uint64) * n
    kb, tb = ctx.alloc(4 * 200000), ctx.alloc(4 * 200000)
    _, _, ho, cnt = ctx.fingerprint_batch(pcm, off, pcm_device=True, out_key=kb, out_t1=tb)
    k, t1 = kb.download(np.uint32, cnt), tb.download(np.uint32, cnt)
    for c in range(nc):
        ok, ot1, _, _ = O.fingerprint_keys(synth.synth_clip(42, c, n, 3000, 2000))
        assert np.array_equal(k[ho[c]:ho[c + 1]], ok) and np.array_equal(t1[ho[c]:ho[c + 1]], ot1)
    # capacity error reports the needed count and does not write past the buffer
    from shazam_amd import _ffi
    with pytest.raises(_ffi.ShzError) as e:
        ctx.fingerprint_batch(pcm, off, pcm_device=True, out_key=kb, out_t1=tb, cap=100)
    assert e.value.code == _ffi.E_CAPACITY
    assert np.array_equal(ctx.sha1_prefix(kb, device=True, n=50), O.sha1_prefix10(k[:50]))
    for b in (pcm, kb, tb):
        b.free()


def test_get_2D_peaks_arbitrary_arrays(S, O):
    rng = np.random.default_rng(3)
    for shape in ((1, 1), (5, 7), (30, 300), (229, 40), (457, 25), (2049, 64), (700, 3)):
        A = rng.normal(20, 15, shape)
        A[rng.random(shape) < 0.05] = 0.0
        if shape[0] > 20:
            A[3:15, :] = np.round(A[3:15, :])  # plateaus: exact ties
        got = S.get_2D_peaks(A)
        f, t = O.peaks_2d(A)
        assert [(int(a), int(b)) for a, b in got] == list(zip(f.tolist(), t.tolist())), shape
        got5 = S.get_2D_peaks(A, amp_min=35.5)
        f, t = O.peaks_2d(A, amp_min=35.5)
        assert [(int(a), int(b)) for a, b in got5] == list(zip(f.tolist(), t.tolist())), shape
    const = np.full((40, 50), 12.0)
    assert len(S.get_2D_peaks(const)) == 2000  # every cell of a plateau is a peak, as in the reference


def test_generate_hashes_parity(S, O, golden_dir):
    g = _golden(golden_dir, "synth_clips.npz")
    f, t = g["tonal_5s_peaks_f"].astype(np.int64), g["tonal_5s_peaks_t"].astype(np.int64)
    peaks = list(zip(f.tolist(), t.tolist()))        # (freq asc, time asc) like get_2D_peaks returns
    got = S.generate_hashes(peaks)
    assert [h.encode() for h, _ in got] == list(g["tonal_5s_hash_hex"])
    assert peaks == sorted(peaks, key=lambda p: p[1])  # sorted in place like the reference
    for fan in (1, 2, 9):
        fs_, ts_ = O.sort_peaks(f, t)
        k, t1 = O.pair_keys(fs_, ts_, fan)
        got = S.generate_hashes(list(zip(f.tolist(), t.tolist())), fan_value=fan)
        assert [h for h, _ in got] == O.sha1_hex20(k) and [o for _, o in got] == t1.tolist()
    assert S.generate_hashes([]) == []


def test_full_size_properties(ctx):
    """BASELINE config 2 shape at reduced clip count: 64 x 30 s clips resident in HBM.
    Size-independent properties: identical clips give identical hashes; per-clip hash counts are
    in the oracle-measured band; keys decode to valid (f1, f2, dt)."""
    n, nc = 1323000, 64
    pcm = ctx.synth_pcm(1234, 0, nc, n, 0, 8000)
    # make clip 63 a copy of clip 1 (golden white_30s: 11942 hashes)
    from shazam_amd import _ffi
    import ctypes as C
    one = pcm.download(np.int16, n, offset_bytes=1 * n * 2)
    pcm.upload(one, offset_bytes=63 * n * 2)
    off = np.arange(nc + 1, dtype=np.uint64) * n
    k, t1, ho, cnt = ctx.fingerprint_batch(pcm, off, pcm_device=True)
    per = np.diff(ho.astype(np.int64))
    assert per[1] == 11942 and per[63] == 11942
    assert np.array_equal(k[ho[1]:ho[2]], k[ho[63]:ho[64]]) and np.array_equal(t1[ho[1]:ho[2]], t1[ho[63]:ho[64]])
    assert per.min() > 11000 and per.max() < 13000
    assert (k >> 20).max() <= 2048 and ((k >> 8) & 0xFFF).max() <= 2048 and (k & 0xFF).max() <= 200
    assert t1.max() < 644
    for c in range(nc):
        assert np.all(np.diff(t1[ho[c]:ho[c + 1]].astype(np.int64)) >= 0)
    pcm.free()


def test_device_snr_mix_matches_twin(ctx):
    """shz_sumsq_i16 + shz_mix_i16 (query preparation) == oracle/synth.mix_query bit for bit."""
    from oracle import synth
    n, nc = 30011, 4
    sig = ctx.synth_pcm(5, 0, nc, n, 4000, 1500)
    noi = ctx.synth_pcm(6, 10, nc, n, 0, 8000)
    for snr in (0.0, 10.0, -3.5):
        out = ctx.mix_snr(sig, noi, nc, n, snr)
        got = out.download(np.int16, nc * n).reshape(nc, n)
        for c in range(nc):
            want = synth.mix_query(synth.synth_clip(5, c, n, 4000, 1500), synth.synth_clip(6, 10 + c, n, 0, 8000), snr)
            assert np.array_equal(got[c], want), (snr, c)
        out.free()
    sig.free()
    noi.free()
