"""GPU: fp32 staging + fp64 verification (the default) gives the peaks and hashes of fp64 staging bit for bit.

fp32 staging is an optimisation of HBM traffic, not a change of arithmetic: every decision fp32 values cannot make
(cells that share the top two fp32 steps of their 21x21 window, window maxima within 1e-7 of the amp_min threshold) is
made on fp64 values recomputed by the same FFT (peak_verify_kernel), and stationary / plateau material sends the whole
pass to fp64 staging.  This file pins that on noise, music-like, edge-case and near-tie inputs, and checks through the
counters (shz_extract_stats) that each mechanism actually ran."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import shazam_amd as S
    from oracle import synth
    return S, S.get_context(0), synth


def _both(ctx, x, off, **kw):
    ctx.set_stage_f64(False)
    a = ctx.fingerprint_batch(x, off, **kw)
    pa = ctx.peaks(x, off, amp_min=kw.get("amp_min", 10.0))
    ctx.set_stage_f64(True)
    try:
        b = ctx.fingerprint_batch(x, off, **kw)
        pb = ctx.peaks(x, off, amp_min=kw.get("amp_min", 10.0))
    finally:
        ctx.set_stage_f64(False)
    return a, b, pa, pb


def _same(a, b, pa, pb, what):
    for u, v in zip(a, b):
        assert np.array_equal(u, v), what
    for u, v in zip(pa, pb):
        assert np.array_equal(u, v), what


def test_noise_and_tonal_batches(env):
    S, ctx, synth = env
    for ta, na, n in ((0, 8000, 1323000), (4000, 1500, 441000), (6000, 200, 300000), (0, 30, 100000)):
        xs = [synth.synth_clip(77, c, n + 1000 * c, ta, na) for c in range(6)]
        off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.uint64)
        s0 = ctx.extract_stats()
        a, b, pa, pb = _both(ctx, np.concatenate(xs), off)
        _same(a, b, pa, pb, (ta, na, n))
        s1 = ctx.extract_stats()
        assert s1["f64_passes"] == s0["f64_passes"], "noisy input must not need the fp64 pass"
        assert len(a[0]) > 0 or na <= 30


def test_near_threshold_amp_min(env):
    """amp_min chosen INSIDE the distribution of window maxima: the threshold band is exercised."""
    S, ctx, synth = env
    xs = [synth.synth_clip(5, c, 200000, 0, 200) for c in range(4)]
    off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.uint64)
    x = np.concatenate(xs)
    P = ctx.stft_db(xs[0], np.array([0, len(xs[0])], np.uint64))[0]
    for amp in (float(np.median(P)), float(np.percentile(P, 99)), float(P.max()) - 1e-9, 0.0, 33.3):
        a, b, pa, pb = _both(ctx, x, off, amp_min=amp)
        _same(a, b, pa, pb, amp)


def test_tie_inputs_fall_back_and_agree(env):
    S, ctx, synth = env
    inputs = synth.tie_inputs()
    for name, x in inputs.items():
        off = np.array([0, len(x)], np.uint64)
        a, b, pa, pb = _both(ctx, x, off)
        _same(a, b, pa, pb, name)
    # which of them the fp32 pass hands to a whole fp64 pass (windows with more than 32 near-maximum cells: ties across
    # frames AND bins) and which the verification kernel settles cell by cell
    fell_back = []
    for name, x in inputs.items():
        s0 = ctx.extract_stats()
        ctx.fingerprint_batch(x, np.array([0, len(x)], np.uint64))
        s1 = ctx.extract_stats()
        if s1["f64_passes"] > s0["f64_passes"] or s1["f64_clips"] > s0["f64_clips"]:
            fell_back.append(name)
    print("fp64 pass needed for:", fell_back)
    assert "sine_1k_10s" not in fell_back and "click_train_30s" not in fell_back   # <= 32 tied cells: settled by verification
    # a click per hop: every frame carries the same samples and its spectrum ripples through few distinct values --
    # hundreds of cells of a window tie, which is what the fp64 pass is for
    x = np.zeros(2048 * 60, np.int16)
    x[1024::2048] = 20000
    off = np.array([0, len(x)], np.uint64)
    s0 = ctx.extract_stats()
    a, b, pa, pb = _both(ctx, x, off)
    _same(a, b, pa, pb, "click per hop")
    s1 = ctx.extract_stats()
    assert s1["f64_clips"] > s0["f64_clips"] and s1["f64_passes"] == s0["f64_passes"]   # the clip is redone, not a pass


def test_edge_cases_and_mixed_batch(env, golden_dir):
    S, ctx, synth = env
    g = np.load(os.path.join(golden_dir, "edge_cases.npz"))
    names = ("short_3000", "exact_4096", "ragged_6143", "two_frames_6144", "silence_20000", "square_p64",
             "gap_250_frames", "loud_fullscale", "dc_offset")
    xs = [g[f"{n}_pcm"] for n in names] + [synth.synth_clip(3, 1, 150000, 0, 8000)]
    for x in xs:
        a, b, pa, pb = _both(ctx, np.ascontiguousarray(x), np.array([0, len(x)], np.uint64))
        _same(a, b, pa, pb, len(x))
    off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.uint64)
    a, b, pa, pb = _both(ctx, np.concatenate(xs), off)
    _same(a, b, pa, pb, "mixed")


def test_verification_decides_shared_fp32_maxima(env):
    """Construct shared window maxima WITHOUT stationarity: a noise clip repeated after a gap shorter than the window
    would tie with itself, so instead take many clips and count: over ~2e6 cells in windows some share the top fp32
    steps by chance; the counters must show that verification ran, and results equal fp64 staging."""
    S, ctx, synth = env
    xs = [synth.synth_clip(4242, c, 1323000, 0, 8000) for c in range(40)]
    off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.uint64)
    x = np.concatenate(xs)
    s0 = ctx.extract_stats()
    a, b, pa, pb = _both(ctx, x, off)
    _same(a, b, pa, pb, "40 noise clips")
    s1 = ctx.extract_stats()
    assert s1["f64_passes"] == s0["f64_passes"]
    print("undecided cells over 40 x 30 s noise (2 passes):", {k: s1[k] - s0[k] for k in s1})


def test_one_tied_clip_among_many_is_redone_alone(env):
    """a click-per-hop clip (hundreds of tied cells per window) among noise clips is the only clip
    fingerprinted again with fp64 staging; the batch's hashes and peaks equal fp64 staging of everything, for device
    and host outputs, with the clip first, in the middle and last."""
    S, ctx, synth = env
    click = np.zeros(2048 * 60, np.int16)
    click[1024::2048] = 20000
    for pos in (0, 5, 11):
        xs = [synth.synth_clip(91, c, 2048 * (50 + 3 * c), 0, 8000) for c in range(12)]
        xs[pos] = click
        off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.uint64)
        x = np.concatenate(xs)
        s0 = ctx.extract_stats()
        a, b, pa, pb = _both(ctx, x, off)
        _same(a, b, pa, pb, pos)
        s1 = ctx.extract_stats()
        assert s1["f64_passes"] == s0["f64_passes"]
        redone = s1["f64_clips"] - s0["f64_clips"]
        assert 2 <= redone <= 4                                             # hashes and peaks (a call that first reports SHZ_E_CAPACITY runs twice)
        assert s1["f64_clip_frames"] - s0["f64_clip_frames"] == redone * 59   # the click clip's frames, nobody else's
        # device in, device out: the same entries
        d_pcm = ctx.alloc(x.nbytes)
        d_pcm.upload(x)
        cap = len(a[0]) + 1000
        kb, tb = ctx.alloc(cap * 4), ctx.alloc(cap * 4)
        _, _, ho, cnt = ctx.fingerprint_batch(d_pcm, off, pcm_device=True, out_key=kb, out_t1=tb, cap=cap)
        assert cnt == len(a[0]) and np.array_equal(ho, a[2])
        assert np.array_equal(kb.download(np.uint32, cnt), a[0]) and np.array_equal(tb.download(np.uint32, cnt), a[1])
        for buf in (d_pcm, kb, tb):
            buf.free()


def test_several_redone_clips_with_exactly_the_final_capacity(env):
    """the host splice of re-run clips moved entries in place from the back, and an intermediate total could
    outgrow arrays that hold exactly the final count (a late clip grows, an early one shrinks).  Several tied clips among
    noise clips, host outputs, capacity == what the first call reported: the same arrays as with room to spare."""
    S, ctx, synth = env
    from shazam_amd import _ffi
    import ctypes as C
    ties = synth.tie_inputs()
    click = np.zeros(2048 * 60, np.int16)
    click[1024::2048] = 20000
    xs = [synth.synth_clip(17, c, 2048 * (40 + 5 * c), 0, 8000) for c in range(9)]
    xs[1], xs[4], xs[7] = ties["two_tone_10s"][:2048 * 80], click, ties["sine_1k_10s"][:2048 * 70]
    off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.uint64)
    x = np.concatenate(xs)
    s0 = ctx.extract_stats()
    k_ref, t_ref, ho_ref, n_ref = ctx.fingerprint_batch(x, off)      # generous capacity
    assert ctx.extract_stats()["f64_clips"] - s0["f64_clips"] >= 2   # several clips were re-run and spliced
    L = _ffi.lib()
    for cap in (n_ref, n_ref + 1):
        k, t1 = np.full(cap + 64, 0xDEADBEEF, np.uint32), np.full(cap + 64, 0xDEADBEEF, np.uint32)   # 64 guard entries behind `cap`
        ho, cnt = np.zeros(len(xs) + 1, np.uint64), C.c_uint64()
        rc = L.shz_fingerprint_batch(ctx.h, _ffi.ptr(x), off.ctypes.data_as(_ffi.u64p), len(xs), 44100, 10.0, 5, 0,
                                     _ffi.ptr(k), _ffi.ptr(t1), ho.ctypes.data_as(_ffi.u64p), cap, C.byref(cnt))
        assert rc == 0 and cnt.value == n_ref
        assert np.array_equal(k[:n_ref], k_ref) and np.array_equal(t1[:n_ref], t_ref) and np.array_equal(ho, ho_ref)
        assert np.all(k[cap:] == 0xDEADBEEF) and np.all(t1[cap:] == 0xDEADBEEF)       # nothing written past the capacity
    # one entry too few: SHZ_E_CAPACITY and the count to provide, nothing past the capacity
    k, t1 = np.full(n_ref + 63, 0xDEADBEEF, np.uint32), np.full(n_ref + 63, 0xDEADBEEF, np.uint32)
    ho, cnt = np.zeros(len(xs) + 1, np.uint64), C.c_uint64()
    rc = L.shz_fingerprint_batch(ctx.h, _ffi.ptr(x), off.ctypes.data_as(_ffi.u64p), len(xs), 44100, 10.0, 5, 0,
                                 _ffi.ptr(k), _ffi.ptr(t1), ho.ctypes.data_as(_ffi.u64p), n_ref - 1, C.byref(cnt))
    assert rc == _ffi.E_CAPACITY and cnt.value >= n_ref
    assert np.all(k[n_ref - 1:] == 0xDEADBEEF) and np.all(t1[n_ref - 1:] == 0xDEADBEEF)
