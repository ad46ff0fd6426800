"""GPU: the fp64 path computes the spectrogram with the reference's own arithmetic (numpy's pocketfft passes, its complex
product, mlab's scaling: csrc np_fft4096 / stft_np_kernel) -- every value of every fixture equals what the reference's
`mlab.specgram` call (__init__.py:232-237) returned, bit for bit: sha256 over the whole array against
tests/golden/psd_digests.json (made by tests/golden/make_golden.py --only-psd from the reference's call itself)."""
import hashlib

import numpy as np
import pytest

from test_numpy_tables import psd_cases

pytestmark = pytest.mark.gpu


def _power(ctx, x, wratio, fs, nfft=4096):
    if nfft != 4096:   # the generic spectrogram (other window sizes)
        P = ctx.stft_db_any(np.ascontiguousarray(x, np.int16), fs, nfft, int(nfft * wratio), power=True)
        return np.where(P == 0, 1.0, P)
    ctx.set_overlap(int(4096 * wratio))
    try:
        return ctx.stft_db(np.ascontiguousarray(x, np.int16), np.array([0, len(x)], np.uint64), fs=fs, power=True)[0]
    finally:
        ctx.set_overlap(2048)


def test_power_spectrogram_is_the_references_bit_for_bit(golden_dir):
    import shazam_amd as S
    ctx = S.get_context(0)
    for name, (x, d) in psd_cases(golden_dir).items():
        P = _power(ctx, x, d["wratio"], d["Fs"], d["nfft"])
        assert list(P.shape) == d["shape"], name
        assert int((P == 1.0).sum()) >= d["zeros"], name
        for a, b, v in d["probe"]:
            assert float(P[a, b]).hex() == v, (name, a, b)
        assert hashlib.sha256(np.ascontiguousarray(P).tobytes()).hexdigest() == d["sha256"], name


def test_batch_members_and_oracle(golden_dir):
    """Clips of one call get the values they get alone; a clip outside the fixtures equals the oracle's restatement."""
    import shazam_amd as S
    from oracle import np_exact as E, synth
    ctx = S.get_context(0)
    xs = [synth.synth_clip(77, c, n, 3000, 2500) for c, n in ((0, 4096 * 3 + 17), (1, 1000), (2, 2048 * 11), (3, 4096))]
    off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.uint64)
    together = ctx.stft_db(np.concatenate(xs), off, power=True)
    for x, P in zip(xs, together):
        alone = ctx.stft_db(x, np.array([0, len(x)], np.uint64), power=True)[0]
        assert np.array_equal(P, alone)
        want = E.psd_exact(x, 44100, 2048)
        assert np.array_equal(P, np.where(want == 0, 1.0, want))


def test_unfused_product_variant(golden_dir):
    """A host whose numpy has no FMA3 forms re*re + im*im: the library follows (shz_set_numpy_product), the oracle says what
    that gives, and it is NOT the fixtures' digest."""
    import shazam_amd as S
    from oracle import np_exact as E
    ctx = S.get_context(0)
    x, d = psd_cases(golden_dir)["variant_wr0"]
    ctx.set_numpy_product(False)
    try:
        P = _power(ctx, x, 0.0, 44100)
        P512 = _power(ctx, x[:20000], 0.5, 44100, 512)
    finally:
        ctx.set_numpy_product(True)
    want = E.psd_exact(x, 44100, 0, 4096, fused=False)
    assert np.array_equal(P, np.where(want == 0, 1.0, want))
    assert hashlib.sha256(np.ascontiguousarray(P).tobytes()).hexdigest() != d["sha256"]
    want = E.psd_exact(x[:20000], 44100, 256, 512, fused=False)
    assert np.array_equal(P512, np.where(want == 0, 1.0, want))
    assert hashlib.sha256(np.ascontiguousarray(_power(ctx, x, 0.0, 44100)).tobytes()).hexdigest() == d["sha256"]

