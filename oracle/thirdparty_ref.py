"""Second CPU restatement that drives the SAME third-party routines the reference
calls (matplotlib.mlab.specgram, scipy.ndimage, hashlib) at the reference's call
sites -- CHECKER / CPU-BASELINE ONLY (see ``oracle/__init__.py``).

Why it exists: the reference's arithmetic lives in these libraries
(SURVEY.md 8c), so this module is the closest thing to "the reference's CPU
path" that can travel to the GPU box (the reference's files cannot).  It
cross-checks ``cpu_ref`` (an independent numpy formulation) and is what
``bench.py`` times as ``cpu_baseline`` (kind "port").
"""
from __future__ import annotations

import hashlib

import numpy as np

from . import cpu_ref as C


def spectrogram_db(x, Fs=C.RATE, wsize=C.DEFAULT_WINDOW_SIZE, wratio=C.DEFAULT_OVERLAP_RATIO):
    """__init__.py:232-241."""
    import matplotlib.mlab as mlab
    P = mlab.specgram(x, NFFT=wsize, Fs=Fs, window=mlab.window_hanning, noverlap=int(wsize * wratio))[0]
    return 10 * np.log10(P, out=np.zeros_like(P), where=(P != 0))


def peaks_2d(A, amp_min=C.DEFAULT_AMP_MIN):
    """__init__.py:130-177 including the erosion/XOR term."""
    from scipy.ndimage import binary_erosion, generate_binary_structure, iterate_structure, maximum_filter
    nb = iterate_structure(generate_binary_structure(2, 2), C.PEAK_NEIGHBORHOOD_SIZE)
    local_max = maximum_filter(A, footprint=nb) == A
    eroded = binary_erosion(A == 0, structure=nb, border_value=1)
    det = local_max != eroded
    amps = A[det]
    f, t = np.where(det)
    keep = amps > amp_min
    return f[keep].astype(np.int64), t[keep].astype(np.int64)


def generate_hashes(f, t, fan_value=C.DEFAULT_FAN_VALUE):
    """__init__.py:194-210, python loop + hashlib like the reference."""
    peaks = sorted(zip(f.tolist(), t.tolist()), key=lambda p: p[1])
    n = len(peaks)
    out = []
    for i in range(n):
        f1, t1 = peaks[i]
        for j in range(1, fan_value):
            if i + j < n:
                f2, t2 = peaks[i + j]
                dt = t2 - t1
                if C.MIN_HASH_TIME_DELTA <= dt <= C.MAX_HASH_TIME_DELTA:
                    h = hashlib.sha1(f"{f1}|{f2}|{dt}".encode("utf-8"))
                    out.append((h.hexdigest()[: C.FINGERPRINT_REDUCTION], t1))
    return out


def fingerprint(x, Fs=C.RATE, wsize=C.DEFAULT_WINDOW_SIZE, wratio=C.DEFAULT_OVERLAP_RATIO,
                fan_value=C.DEFAULT_FAN_VALUE, amp_min=C.DEFAULT_AMP_MIN):
    A = spectrogram_db(x, Fs, wsize, wratio)
    f, t = peaks_2d(A, amp_min)
    return generate_hashes(f, t, fan_value)
