"""GPU: amp_min < 0 with regions of exact zeros.  The reference's erosion/XOR term (__init__.py:147-151) removes
zero-valued local maxima whose whole 21x21 window is zero; peak_zero_plateau_kernel clears exactly those mask bits.
Checked against the scipy call sequence on arrays, and against the oracle on PCM with digital silence."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def zero_region_arrays():   # same arrays as tests/test_amp_min_domain.py (kept local: test files do not import each other)
    rng = np.random.default_rng(0)
    A = rng.normal(0, 5, (60, 80))
    A[10:40, 20:60] = 0.0
    A[5, 5] = 0.0
    B = -np.abs(rng.normal(0, 5, (70, 90)))
    B[:25, :30] = 0.0
    B[40:48, 50:58] = 0.0
    return {"A": A, "B": B, "C": np.zeros((30, 40))}


def test_array_api_zero_plateaus():
    import shazam_amd as S
    from oracle import thirdparty_ref as T
    for name, X in zero_region_arrays().items():
        for amp_min in (10, 0, -0.5, -20):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                f, t = T.peaks_2d(X, amp_min)
            want = sorted(zip(np.asarray(f).tolist(), np.asarray(t).tolist()))
            got = sorted(S.get_2D_peaks(X, amp_min=amp_min))
            assert got == want, (name, amp_min, len(got), len(want))


def test_pcm_with_digital_silence_negative_threshold():
    import shazam_amd as S
    from oracle import cpu_ref as O, synth
    x = synth.synth_clip(3, 0, 2048 * 120, 3000, 1500)
    x[2048 * 30:2048 * 75] = 0            # 45 frames of digital silence: a zero plateau with an interior
    x[2048 * 100:] = 0                    # and silence running into the end of the clip
    for amp_min in (-5, -40):
        want = O.fingerprint(x, amp_min=amp_min)
        got = S.fingerprint(x, amp_min=amp_min)
        assert got == want and len(got) > 100
    # a clip that is silence from start to end has no peaks at any threshold
    assert S.fingerprint(np.zeros(2048 * 40, np.int16), amp_min=-10) == O.fingerprint(np.zeros(2048 * 40, np.int16), amp_min=-10) == []
