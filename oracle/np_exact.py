"""TEST INFRASTRUCTURE (oracle): the arithmetic of the reference's spectrogram call, operation by operation.

``fingerprint()`` (reference ``__init__.py:232-237``) calls ``mlab.specgram(x, NFFT=4096, Fs, window=window_hanning,
noverlap)``; on a host with numpy 2.x that is (matplotlib ``mlab._spectral_helper``; numpy and matplotlib are third-party
code that is not in /root/reference, SURVEY 8c -- this file restates their published algorithms and is pinned by
``tests/golden/psd_digests.json``, digests of what the reference's own call returned):

    result = frames * np.hanning(4096)            one product a sample
    np.fft.fft(result, axis=0)[:2049]             pocketfft's COMPLEX transform of the real frame; 4096 = 8 x 8 x 8 x 8:
                                                  four passes of its radix-8 butterfly (`pass8`), twiddles from
                                                  `sincos_2pibyn` (two short libm tables whose entries are multiplied),
                                                  every product and sum rounded on its own
    np.conj(result) * result                      numpy's complex product: real part fma(re, re, im * im) on a host with
                                                  FMA3 (x86-64 AVX2 / AVX-512), re*re + im*im otherwise
    result[1:-1] *= 2; result /= Fs; result /= (window ** 2).sum()
                                                  complex / real in numpy = times the rounded reciprocal

Only tests may import this.  The device's fp64 path (csrc/shz_extract.hip: np_fft4096, stft_np_kernel) is compared
with it and with the digests; `fft_pow8` is compared with `np.fft.fft` bit for bit in tests/test_numpy_tables.py."""
from __future__ import annotations

import ctypes
import ctypes.util

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.fma.restype = ctypes.c_double
_libm.fma.argtypes = [ctypes.c_double] * 3
_fma = np.frompyfunc(_libm.fma, 3, 1)


def hanning(n: int) -> np.ndarray:
    """np.hanning(n) spelled out (numpy/lib/_function_base_impl.py): 0.5 + 0.5 cos(pi k / (n - 1)), k = 1-n, 3-n, ..."""
    k = np.arange(1 - n, n, 2)
    return 0.5 + 0.5 * np.cos(np.pi * k / (n - 1))


def _calc(x: int, n: int, ang: float):
    """pocketfft sincos_2pibyn::calc: cos/sin of 2 pi x / n from an argument reduced to the first octant."""
    from math import cos, sin
    x <<= 3
    if x < 4 * n:
        if x < 2 * n:
            if x < n:
                return cos(x * ang), sin(x * ang)
            return sin((2 * n - x) * ang), cos((2 * n - x) * ang)
        x -= 2 * n
        if x < n:
            return -sin(x * ang), cos(x * ang)
        return -cos((2 * n - x) * ang), sin((2 * n - x) * ang)
    x = 8 * n - x
    if x < 2 * n:
        if x < n:
            return cos(x * ang), -sin(x * ang)
        return sin((2 * n - x) * ang), -cos((2 * n - x) * ang)
    x -= 2 * n
    if x < n:
        return -sin(x * ang), -cos(x * ang)
    return -cos((2 * n - x) * ang), -sin((2 * n - x) * ang)


def sincos_2pibyn(n: int):
    """(re, im)[i] = pocketfft's comp[i], i < n: two tables (i & mask, i >> shift) of first-octant values, multiplied."""
    ang = float(np.longdouble(0.25) * np.longdouble("3.141592653589793238462643383279502884197") / np.longdouble(n))
    nval = (n + 2) // 2
    shift = 1
    while (1 << shift) * (1 << shift) < nval:
        shift += 1
    mask = (1 << shift) - 1
    v1 = [(1.0, 0.0)] + [_calc(i, n, ang) for i in range(1, mask + 1)]
    v2 = [(1.0, 0.0)] + [_calc(i * (mask + 1), n, ang) for i in range(1, (nval + mask) // (mask + 1))]
    re, im = np.empty(n), np.empty(n)
    for idx in range(n):
        m = idx if 2 * idx <= n else n - idx
        (a, b), (c, d) = v1[m & mask], v2[m >> shift]
        re[idx] = a * c - b * d
        im[idx] = (a * d + b * c) if 2 * idx <= n else -(a * d + b * c)
    return re, im


_HSQT2 = 0.707106781186547524400844362104849


def _pass8(cr, ci, l1, ido, wr, wi):
    """pocketfft pass8<fwd=true>: cc[i + ido (b + 8 k)] -> ch[i + ido (k + l1 c)], arrays as [k][b][i] / [c][k][i]."""
    cr, ci = cr.reshape(l1, 8, ido), ci.reshape(l1, 8, ido)
    c = [(cr[:, b, :], ci[:, b, :]) for b in range(8)]
    add = lambda p, q: (p[0] + q[0], p[1] + q[1])
    sub = lambda p, q: (p[0] - q[0], p[1] - q[1])
    rot90 = lambda p: (p[1], -p[0])
    a1, a5 = add(c[1], c[5]), sub(c[1], c[5])
    a3, a7 = add(c[3], c[7]), sub(c[3], c[7])
    a1, a3 = add(a1, a3), sub(a1, a3)
    a3, a7 = rot90(a3), rot90(a7)
    a5, a7 = add(a5, a7), sub(a5, a7)
    a5 = (_HSQT2 * (a5[0] + a5[1]), _HSQT2 * (a5[1] - a5[0]))
    a7 = (_HSQT2 * (a7[1] - a7[0]), _HSQT2 * (-a7[0] - a7[1]))
    a0, a4 = add(c[0], c[4]), sub(c[0], c[4])
    a2, a6 = add(c[2], c[6]), sub(c[2], c[6])
    a0, a2 = add(a0, a2), sub(a0, a2)
    a6 = rot90(a6)
    a4, a6 = add(a4, a6), sub(a4, a6)
    o = [add(a0, a1), add(a4, a5), add(a2, a3), add(a6, a7), sub(a0, a1), sub(a4, a5), sub(a2, a3), sub(a6, a7)]
    outr, outi = np.empty((8, l1, ido)), np.empty((8, l1, ido))
    i = np.arange(ido)
    for cc in range(8):
        vr, vi = o[cc]
        if cc and ido > 1:
            w_r, w_i = wr[cc * l1 * i], wi[cc * l1 * i]
            tr, ti = vr * w_r + vi * w_i, vi * w_r - vr * w_i   # special_mul<fwd>: v * conj(w)
            tr[:, 0], ti[:, 0] = vr[:, 0], vi[:, 0]             # i = 0: no product at all
            vr, vi = tr, ti
        outr[cc], outi[cc] = vr, vi
    return outr.reshape(-1), outi.reshape(-1)


def fft_pow8(x):
    """np.fft.fft of one real or complex vector whose length is a power of 8, as numpy 2.x computes it."""
    x = np.asarray(x)
    n = len(x)
    assert n >= 8 and 8 ** round(np.log(n) / np.log(8)) == n
    cr, ci = np.array(x.real, np.float64), np.array(x.imag, np.float64)
    wr, wi = sincos_2pibyn(n)
    l1 = 1
    while l1 < n:
        ido = n // (8 * l1)
        cr, ci = _pass8(cr, ci, l1, ido, wr, wi)
        l1 *= 8
    return cr + 1j * ci


def psd_exact(x, Fs=44100, noverlap=2048, nfft=4096) -> np.ndarray:
    """mlab.specgram(x, NFFT=4096, Fs, window_hanning, noverlap)[0], float64 [2049][F]."""
    x = np.asarray(x)
    if len(x) < nfft:
        xp = np.zeros(nfft, x.dtype if x.size else np.int16)
        xp[: len(x)] = x
        x = xp
    w = hanning(nfft)
    r_fs, r_s = 1.0 / float(Fs), 1.0 / float((w ** 2).sum())
    frames = np.lib.stride_tricks.sliding_window_view(x, nfft)[:: nfft - noverlap]
    out = np.empty((nfft // 2 + 1, frames.shape[0]))
    for f, fr in enumerate(frames):
        z = fft_pow8(fr * w)[: nfft // 2 + 1]
        re, im = z.real.copy(), z.imag.copy()
        p = _fma(re, re, im * im).astype(np.float64)
        p[1:-1] *= 2.0
        out[:, f] = (p * r_fs) * r_s
    return out
