"""GPU: host PCM fed in chunks beside the kernels (shz_extract.hip: extract_streamed) -- what every caller of the reference
does hand over (__init__.py:248-268; recognizer.py:377-382).  Results must be those of one pass over device-resident PCM:
same (key32, t1) in the same order, same per-clip offsets; from pageable memory and from pinned memory (Context.host_array);
and the two-call capacity idiom still reports what the caller has to provide."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_streamed_host_pcm_equals_one_pass_over_device_pcm():
    from shazam_amd import _ffi
    ctx = _ffi.Context(0)
    n_samples, nc = 30 * 44100, 90                    # 238 MB of PCM: two chunks of whole clips
    lens = np.full(nc, n_samples, np.uint64)
    lens[7] = 5000                                    # ragged clips keep their places
    lens[41] = 44100 * 12 + 13
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    dev = ctx.synth_pcm(99, 0, nc, n_samples, 3000, 1500)
    full = dev.download(np.int16, nc * n_samples)
    host = np.concatenate([full[c * n_samples:c * n_samples + int(lens[c])] for c in range(nc)])
    dev.free()
    dbuf = ctx.alloc(host.nbytes)
    dbuf.upload(host)
    before = ctx.upload_stats()
    k0, t0, ho0, n0 = ctx.fingerprint_batch(dbuf, off, pcm_device=True, out_key=None)   # one pass, device PCM, host outputs
    assert ctx.upload_stats() == before               # device PCM does not go through the upload pipeline
    k1, t1, ho1, n1 = ctx.fingerprint_batch(host, off)                                  # pageable host PCM
    st = ctx.upload_stats()
    assert st["chunks"] - before["chunks"] >= 2 and st["bytes"] - before["bytes"] == host.nbytes
    pin = ctx.host_array(len(host), np.int16)
    pin[:] = host
    k2, t2, ho2, n2 = ctx.fingerprint_batch(pin, off)                                   # pinned host PCM
    for k, t, ho, n in ((k1, t1, ho1, n1), (k2, t2, ho2, n2)):
        assert n == n0 and np.array_equal(ho, ho0) and np.array_equal(k, k0) and np.array_equal(t, t0)
    # capacity too small: SHZ_E_CAPACITY and the count the caller has to provide
    small = ctx.alloc(1000 * 4), ctx.alloc(1000 * 4)
    with pytest.raises(_ffi.ShzError) as ei:
        ctx.fingerprint_batch(host, off, out_key=small[0], out_t1=small[1], cap=1000)
    assert ei.value.code == _ffi.E_CAPACITY
    for b in (*small, dbuf):
        b.free()
    ctx.close()
