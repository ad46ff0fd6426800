"""Non-default fingerprint() parameters (amp_min, fan_value, Fs, wratio, wsize) against outputs of the reference
(tests/golden/param_variants.npz): oracle on CPU, HIP path on the GPU.  wsize: 4096 and the powers of two 64 .. 2048 (a
generic GPU spectrogram composed with the GPU peak and hash stages); anything else is refused."""
import os

import numpy as np
import pytest

VARIANTS = {"amp0": dict(amp_min=0), "amp25": dict(amp_min=25), "ampneg5": dict(amp_min=-5), "amp33p3": dict(amp_min=33.3),
            "fan2": dict(fan_value=2), "fan10": dict(fan_value=10), "fan1": dict(fan_value=1),
            "fs8000": dict(Fs=8000), "fs48000": dict(Fs=48000),
            "wr075": dict(wratio=0.75), "wr025": dict(wratio=0.25), "wr0": dict(wratio=0.0), "wr08999": dict(wratio=0.8999),
            "ws2048": dict(wsize=2048), "ws1024": dict(wsize=1024), "ws512_wr075": dict(wsize=512, wratio=0.75),
            "ws256_wr0": dict(wsize=256, wratio=0.0), "ws64": dict(wsize=64), "ws2048_fs8000": dict(wsize=2048, Fs=8000)}
DB_CASES = ("ws1024", "ws256", "ws2048short")


def _load(golden_dir):
    from oracle import synth
    g = np.load(os.path.join(golden_dir, "param_variants.npz"))
    seed, clip, n, ta, na = (int(v) for v in g["pcm_params"])
    return g, synth.synth_clip(seed, clip, n, ta, na)


@pytest.mark.parametrize("tag", sorted(VARIANTS))
def test_oracle_variants(golden_dir, tag):
    from oracle import cpu_ref as C
    g, x = _load(golden_dir)
    hs = C.fingerprint(x, **VARIANTS[tag])
    assert [h.encode() for h, _ in hs] == list(g[f"{tag}_hash_hex"])
    assert [o for _, o in hs] == list(g[f"{tag}_hash_t1"])


@pytest.mark.parametrize("tag", DB_CASES)
def test_oracle_window_size_db(golden_dir, tag):
    from oracle import cpu_ref as C
    g, x = _load(golden_dir)
    nfft, nov, n = (int(v) for v in g[f"{tag}_db_args"])
    a = C.spectrogram_db(x[:n], 44100, nfft, nov / nfft)
    assert a.shape == g[f"{tag}_db"].shape
    np.testing.assert_allclose(a, g[f"{tag}_db"], rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_gpu_window_size_db(golden_dir):
    """The generic spectrogram against the reference's own dB arrays: float64, so to a tolerance here (1e-9 dB; the peak and
    hash outputs below are the exact check), shape and the where != 0 rule exactly; and its refusals."""
    from shazam_amd import _ffi
    import shazam_amd as S
    ctx = S.get_context()
    g, x = _load(golden_dir)
    for tag in DB_CASES:
        nfft, nov, n = (int(v) for v in g[f"{tag}_db_args"])
        a = ctx.stft_db_any(x[:n], 44100, nfft, nov)
        assert a.shape == g[f"{tag}_db"].shape, tag
        np.testing.assert_allclose(a, g[f"{tag}_db"], rtol=0, atol=1e-9, err_msg=tag)
    z = ctx.stft_db_any(np.zeros(4096, np.int16), 44100, 512, 256)
    assert z.shape == (257, 15) and not z.any()                 # log of 0 -> 0, as the reference's where= leaves it
    p = ctx.stft_db_any(x[:5000], 44100, 512, 256, power=True)
    d = ctx.stft_db_any(x[:5000], 44100, 512, 256)
    np.testing.assert_allclose(10 * np.log10(p[p != 0]), d[p != 0], rtol=0, atol=1e-9)
    for bad in (4096, 8192, 1000, 32, 0):
        with pytest.raises(_ffi.ShzError) as e:
            ctx.stft_db_any(x[:5000], 44100, bad, 0)
        assert e.value.code == _ffi.E_UNSUPPORTED
    with pytest.raises(_ffi.ShzError) as e:
        ctx.stft_db_any(x[:5000], 44100, 512, 512)
    assert e.value.code == _ffi.E_INVALID


@pytest.mark.gpu
def test_gpu_variants(golden_dir):
    import shazam_amd as S
    g, x = _load(golden_dir)
    for tag, kw in VARIANTS.items():
        hs = S.fingerprint(x, **kw)
        assert [h.encode() for h, _ in hs] == list(g[f"{tag}_hash_hex"]), tag
        assert [o for _, o in hs] == list(g[f"{tag}_hash_t1"]), tag
    assert S.fingerprint(x) == S.fingerprint(x, wratio=0.5)      # the context's overlap is back at its default after a variant
    for bad in (8192, 3000, 32):
        with pytest.raises(NotImplementedError):
            S.fingerprint(x, wsize=bad)
    with pytest.raises(ValueError):
        S.fingerprint(x, wratio=1.0)                             # mlab: noverlap must be less than NFFT
