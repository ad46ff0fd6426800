"""GPU: error behaviour of the C ABI (include/shz.h): bad arguments come back as SHZ_E_* codes with a message in
shz_last_error, never as a crash or a silent wrong answer; the two-call capacity idiom reports the size needed;
objects used in the wrong state say so.  The Python layer turns the codes into ShzError / the reference's
exceptions (ValueError for a window the reference's mlab call rejects, mlab:242)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import shazam_amd as S
    from shazam_amd import _ffi
    from oracle import synth
    ctx = S.get_context(0)
    x = synth.synth_clip(5, 0, 3 * 44100, 3000, 2000)
    return S, _ffi, ctx, x


def _call(_ffi, fn, *args):
    return getattr(_ffi.lib(), fn)(*args)


def test_extraction_argument_errors(env):
    S, _ffi, ctx, x = env
    off = np.array([0, len(x)], np.uint64)
    k, t1, ho = np.empty(100000, np.uint32), np.empty(100000, np.uint32), np.zeros(2, np.uint64)
    cnt = C.c_uint64()
    good = [ctx.h, _ffi.ptr(x), off.ctypes.data_as(_ffi.u64p), 1, 44100, 10.0, 5, 0, _ffi.ptr(k), _ffi.ptr(t1),
            ho.ctypes.data_as(_ffi.u64p), len(k), C.byref(cnt)]
    assert _call(_ffi, "shz_fingerprint_batch", *good) == _ffi.OK and cnt.value > 1000
    n_hashes = cnt.value

    def bad(i, v):
        a = list(good)
        a[i] = v
        return _call(_ffi, "shz_fingerprint_batch", *a)

    assert bad(0, None) == _ffi.E_INVALID                     # no context
    assert bad(1, None) == _ffi.E_INVALID                     # NULL pcm
    assert bad(2, None) == _ffi.E_INVALID                     # NULL clip offsets
    assert bad(4, 0) == _ffi.E_INVALID                        # Fs = 0
    assert bad(6, 0) == _ffi.E_INVALID and bad(6, 65) == _ffi.E_INVALID    # fan_value outside [1, 64]
    assert b"fan_value" in _ffi.lib().shz_last_error(ctx.h)
    # decreasing clip offsets
    rev = np.array([len(x), 0], np.uint64)
    assert bad(2, rev.ctypes.data_as(_ffi.u64p)) == _ffi.E_INVALID
    # capacity: nothing written past cap, *count = hashes needed, then the second call succeeds
    small_k, small_t = np.full(16, 0xABCDEF01, np.uint32), np.full(16, 0xABCDEF01, np.uint32)
    a = list(good)
    a[8], a[9], a[11] = _ffi.ptr(small_k), _ffi.ptr(small_t), 8
    assert _call(_ffi, "shz_fingerprint_batch", *a) == _ffi.E_CAPACITY and cnt.value == n_hashes
    assert np.all(small_k[8:] == 0xABCDEF01) and np.all(small_t[8:] == 0xABCDEF01)
    # zero clips is not an error
    a = list(good)
    a[3] = 0
    assert _call(_ffi, "shz_fingerprint_batch", *a) == _ffi.OK and cnt.value == 0
    # the Python API: a window the reference's specgram call rejects raises ValueError (mlab:242); window sizes of 8192 and
    # above and sizes that are not powers of two are not implemented on the HIP path
    with pytest.raises(ValueError):
        S.fingerprint(x, wsize=4096, wratio=1.0)
    with pytest.raises(ValueError):
        S.fingerprint(x, wsize=1024, wratio=1.0)
    for bad_size in (8192, 3000, 32):
        with pytest.raises(NotImplementedError):
            S.fingerprint(x, wsize=bad_size)


def test_peaks_from_array_errors(env):
    S, _ffi, ctx, x = env
    a = np.random.default_rng(1).normal(0, 20, (64, 50))
    pf, pt, cnt = np.empty(4096, np.uint16), np.empty(4096, np.uint32), C.c_uint64()
    args = [ctx.h, _ffi.ptr(a), 64, 50, 10.0, _ffi.ptr(pf), _ffi.ptr(pt), 4096, C.byref(cnt)]
    assert _call(_ffi, "shz_peaks_from_db", *args) == _ffi.OK
    n = cnt.value
    args[7] = 1
    assert _call(_ffi, "shz_peaks_from_db", *args) == (_ffi.E_CAPACITY if n > 1 else _ffi.OK) and cnt.value == n
    args[1] = None
    assert _call(_ffi, "shz_peaks_from_db", *args) == _ffi.E_INVALID
    # an empty array has no peaks (get_2D_peaks on a 0-column array returns [])
    assert S.get_2D_peaks(np.zeros((2049, 0))) == []


def test_table_state_and_match_errors(env):
    S, _ffi, ctx, x = env
    t = S.Table(ctx)
    k = np.array([(5 << 20) | (7 << 8) | 3] * 4, np.uint32)
    t.insert(k, np.array([1, 1, 2, 2], np.uint32), np.array([10, 10, 4, 5], np.uint32))
    qk, qo, qoff = k[:1].copy(), np.array([2], np.uint32), np.array([0, 1], np.uint64)
    # staged rows not finalized yet: the table refuses to answer
    with pytest.raises(_ffi.ShzError) as e:
        t.match(qk, qo, qoff, 2)
    assert e.value.code == _ffi.E_STATE
    with pytest.raises(_ffi.ShzError):
        t.lookup(qk)
    t.finalize()
    assert t.rows() == (3, 0)                                   # INSERT IGNORE on (song_id, offset, hash)
    r = t.match(qk, qo, qoff, 2)
    assert r["nres"][0] == 2 and r["sid"][0].tolist() == [1, 2] and r["delta"][0].tolist() == [8, 2]
    for topn in (0, 65):
        with pytest.raises(_ffi.ShzError) as e:
            t.match(qk, qo, qoff, topn)
        assert e.value.code == _ffi.E_INVALID
    # query offsets are frame indices < 2^20
    with pytest.raises(_ffi.ShzError) as e:
        t.match(qk, np.array([1 << 20], np.uint32), qoff, 2)
    assert e.value.code == _ffi.E_UNSUPPORTED
    # a table is bound to the context that made it
    other = _ffi.Context(0)
    res = [np.zeros(2, dt) for dt in (np.uint32, np.int32, np.uint32, np.uint32)]
    rc = _ffi.lib().shz_match_batch(other.h, t.h, _ffi.ptr(qk), _ffi.ptr(qo), qoff.ctypes.data_as(_ffi.u64p), 1, 2, 0,
                                    *[_ffi.ptr(a) for a in res], _ffi.ptr(np.zeros(1, np.uint32)), None, None)
    assert rc == _ffi.E_INVALID
    other.close()
    # lookup capacity idiom
    kk, ss, oo, cnt = np.empty(1, np.uint32), np.empty(1, np.uint32), np.empty(1, np.uint32), C.c_uint64()
    rc = _ffi.lib().shz_table_lookup(t.h, _ffi.ptr(qk), 1, _ffi.ptr(kk), _ffi.ptr(ss), _ffi.ptr(oo), 1, C.byref(cnt))
    assert rc == _ffi.E_CAPACITY and cnt.value == 3
    # shard arguments
    assert _ffi.lib().shz_table_keep_shard(t.h, 3, 3) == _ffi.E_INVALID
    assert _ffi.lib().shz_table_stage_from(t.h, t.h, 0, 2) == _ffi.E_INVALID
    t.close()


def test_workspace_limit_is_enforced_not_ignored(env):
    S, _ffi, ctx, x = env
    from oracle import synth
    y = synth.synth_clip(5, 1, 10 * 44100, 3000, 2000)          # 214 frames
    off = np.array([0, len(y)], np.uint64)
    ctx.set_workspace_limit(1 << 16)                # the floor is 64 frames per sub-batch: this clip cannot be staged
    try:
        with pytest.raises(_ffi.ShzError) as e:
            ctx.fingerprint_batch(y, off)
        assert e.value.code == _ffi.E_UNSUPPORTED and "frames" in str(e.value)
        k3, _, _, n3 = ctx.fingerprint_batch(x, np.array([0, len(x)], np.uint64))   # 63 frames still fit
        assert n3 == len(k3) > 1000
    finally:
        ctx.set_workspace_limit(0)
    k, t1, ho, n = ctx.fingerprint_batch(y, off)    # and the context is usable again afterwards
    assert n == len(k) > 3000


def test_round4_entry_points_refuse_bad_arguments(env):
    """The entry points added in round 4: NULL handles, out-of-range values and objects of different contexts come back as
    SHZ_E_* codes; a refused shz_set_overlap leaves the window as it was."""
    S, _ffi, ctx, x = env
    L = _ffi.lib()
    p = _ffi.vp()
    assert L.shz_host_alloc(None, 64, C.byref(p)) == _ffi.E_INVALID
    assert L.shz_host_alloc(ctx.h, 64, None) == _ffi.E_INVALID
    assert L.shz_host_free(None, None) == _ffi.OK                    # nothing to free
    assert L.shz_upload_stats(None, None, None, None, None) == _ffi.E_INVALID
    assert L.shz_comm_warmup(None) == _ffi.E_INVALID
    assert L.shz_table_exchange_run(None, None) == _ffi.E_INVALID
    assert L.shz_table_exchange_stats(None, None, None, None, None) == _ffi.E_INVALID
    assert L.shz_table_set_run_rows(None, 100) == _ffi.E_INVALID
    assert L.shz_frame_count_hop(10 * 4096, 0) == 0 and L.shz_frame_count_hop(10 * 4096, 4097) == 0
    assert L.shz_frame_count_hop(10 * 4096, 4096) == 10 and L.shz_frame_count_hop(4095, 17) == 1
    # overlap: mlab's rule (noverlap < NFFT); a refusal changes nothing
    before = S.fingerprint(x)
    assert L.shz_set_overlap(ctx.h, 4096) == _ffi.E_INVALID and L.shz_set_overlap(None, 2048) == _ffi.E_INVALID
    assert S.fingerprint(x) == before
    with pytest.raises(ValueError):
        S.fingerprint(x, wratio=-0.1)
    # the generic spectrogram: NULL handle / buffers, capacity (reported through n_frames), no samples
    xs = np.ascontiguousarray(x[:5000], np.int16)
    out, nf = np.empty(257 * 18, np.float64), C.c_uint64()
    assert L.shz_stft_db_any(None, _ffi.ptr(xs), 5000, 44100, 512, 256, 0, _ffi.ptr(out), out.size, C.byref(nf)) == _ffi.E_INVALID
    assert L.shz_stft_db_any(ctx.h, None, 5000, 44100, 512, 256, 0, _ffi.ptr(out), out.size, C.byref(nf)) == _ffi.E_INVALID
    assert L.shz_stft_db_any(ctx.h, _ffi.ptr(xs), 0, 44100, 512, 256, 0, _ffi.ptr(out), out.size, C.byref(nf)) == _ffi.E_INVALID
    assert L.shz_stft_db_any(ctx.h, _ffi.ptr(xs), 5000, 0, 512, 256, 0, _ffi.ptr(out), out.size, C.byref(nf)) == _ffi.E_INVALID
    assert L.shz_stft_db_any(ctx.h, _ffi.ptr(xs), 5000, 44100, 512, 256, 0, _ffi.ptr(out), 100, C.byref(nf)) == _ffi.E_CAPACITY
    assert nf.value == 18
    assert L.shz_stft_db_any(ctx.h, _ffi.ptr(xs), 5000, 44100, 512, 256, 0, None, 0, C.byref(nf)) == _ffi.E_CAPACITY
    assert L.shz_stft_db_any(ctx.h, _ffi.ptr(xs), 5000, 44100, 512, 256, 0, _ffi.ptr(out), out.size, None) == _ffi.OK
    # the numpy window of the fp64 path: NULL, a sum that is no sum, values that are no window -- and the right one again
    w = np.hanning(4096)
    assert L.shz_set_numpy_window(None, _ffi.ptr(w), 1535.625) == _ffi.E_INVALID
    assert L.shz_set_numpy_window(ctx.h, None, 1535.625) == _ffi.E_INVALID
    assert L.shz_set_numpy_window(ctx.h, _ffi.ptr(w), 0.0) == _ffi.E_INVALID
    assert L.shz_set_numpy_window(ctx.h, _ffi.ptr(np.full(4096, np.nan)), 1535.625) == _ffi.E_INVALID
    assert L.shz_set_numpy_window(ctx.h, _ffi.ptr(w), float((w ** 2).sum())) == _ffi.OK
    # run rows: 0 (the limit of a sort) or >= 16
    tbl = S.Table(ctx)
    assert L.shz_table_set_run_rows(tbl.h, 5) == _ffi.E_INVALID and L.shz_table_set_run_rows(tbl.h, 0) == _ffi.OK
    # a table and a communicator of different contexts do not go together
    ctx2 = _ffi.Context(0)
    comm2 = _ffi.Comm.local(ctx2, 4711, 0, 1)
    assert L.shz_table_exchange_run(tbl.h, comm2.h) == _ffi.E_INVALID
    b = C.c_uint64()
    assert L.shz_table_allgather(tbl.h, comm2.h, C.byref(b)) == _ffi.E_INVALID
    # membw: the host link modes exist, a mode beyond them does not
    gb = C.c_float()
    assert L.shz_membw(ctx.h, 6, 1 << 20, 1, C.byref(gb)) == _ffi.E_INVALID
    assert L.shz_membw(ctx.h, 3, 8 << 20, 1, C.byref(gb)) == _ffi.OK and gb.value > 1.0
    comm2.close()
    ctx2.close()
    tbl.close()
