"""CPU: the peak rule against the reference's get_2D_peaks call sequence (scipy maximum_filter / binary_erosion,
__init__.py:116-177) on arrays with regions of exact zeros (digital silence maps to 0 dB, __init__.py:241).
amp_min >= 0: "21x21 local maximum and value > amp_min".  amp_min < 0: zero-valued cells pass the threshold and the
erosion/XOR term removes those whose whole window is zero (peak_zero_plateau_kernel on the GPU)."""
import warnings

import numpy as np
import pytest

from oracle import cpu_ref as O, thirdparty_ref as T


def _peaks(fn, A, amp_min):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f, t = fn(A, amp_min)
    return sorted(zip(np.asarray(f).tolist(), np.asarray(t).tolist()))


def zero_region_arrays():
    rng = np.random.default_rng(0)
    A = rng.normal(0, 5, (60, 80))
    A[10:40, 20:60] = 0.0          # a plateau of exact zeros with an interior
    A[5, 5] = 0.0                  # an isolated zero
    B = -np.abs(rng.normal(0, 5, (70, 90)))   # everything below zero: zeros are the maxima
    B[:25, :30] = 0.0              # plateau touching two array edges (outside counts as zero)
    B[40:48, 50:58] = 0.0          # plateau narrower than the window: no interior
    C = np.zeros((30, 40))         # silence everywhere
    return {"A": A, "B": B, "C": C}


@pytest.mark.parametrize("amp_min", [10, 3, 0, -0.5, -1, -20])
def test_restated_rule_equals_reference_sequence(amp_min):
    for name, X in zero_region_arrays().items():
        assert _peaks(T.peaks_2d, X, amp_min) == _peaks(O.peaks_2d, X, amp_min), (name, amp_min)


def test_pcm_with_digital_silence_equals_reference_sequence():
    from oracle import synth
    x = synth.synth_clip(3, 0, 2048 * 60, 3000, 1500)
    x[2048 * 15:2048 * 45] = 0
    for amp_min in (10, -5):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = T.fingerprint(x, amp_min=amp_min)
        assert O.fingerprint(x, amp_min=amp_min) == want and len(want) > 50
