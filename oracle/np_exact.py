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


def _pass4(cr, ci, l1, ido, wr, wi):
    """pocketfft pass4<fwd=true>: cc[i + ido (b + 4 k)] -> ch[i + ido (k + l1 c)]."""
    cr, ci = cr.reshape(l1, 4, ido), ci.reshape(l1, 4, ido)
    c = [(cr[:, b, :], ci[:, b, :]) for b in range(4)]
    add = lambda p, q: (p[0] + q[0], p[1] + q[1])
    sub = lambda p, q: (p[0] - q[0], p[1] - q[1])
    t2, t1 = add(c[0], c[2]), sub(c[0], c[2])
    t3, t4 = add(c[1], c[3]), sub(c[1], c[3])
    t4 = (t4[1], -t4[0])                                  # ROTX90<fwd>
    o = [add(t2, t3), add(t1, t4), sub(t2, t3), sub(t1, t4)]
    return _twiddle_out(o, l1, ido, wr, wi)


def _pass2(cr, ci, l1, ido, wr, wi):
    """pocketfft pass2<fwd=true>."""
    cr, ci = cr.reshape(l1, 2, ido), ci.reshape(l1, 2, ido)
    a, b = (cr[:, 0, :], ci[:, 0, :]), (cr[:, 1, :], ci[:, 1, :])
    o = [(a[0] + b[0], a[1] + b[1]), (a[0] - b[0], a[1] - b[1])]
    return _twiddle_out(o, l1, ido, wr, wi)


def _twiddle_out(o, l1, ido, wr, wi):
    """outputs c >= 1 times conj(comp[c l1 i]) for i > 0 (special_mul<fwd>), no product at i = 0; layout [c][k][i]"""
    ip = len(o)
    outr, outi = np.empty((ip, l1, ido)), np.empty((ip, l1, ido))
    i = np.arange(ido)
    for cc in range(ip):
        vr, vi = o[cc]
        if cc and ido > 1:
            w_r, w_i = wr[cc * l1 * i], wi[cc * l1 * i]
            tr, ti = vr * w_r + vi * w_i, vi * w_r - vr * w_i
            tr[:, 0], ti[:, 0] = vr[:, 0], vi[:, 0]
            vr, vi = tr, ti
        outr[cc], outi[cc] = vr, vi
    return outr.reshape(-1), outi.reshape(-1)


def factors_pow2(n: int):
    """pocketfft cfftp::factorize for a power of two: 8s, then 4s, then one 2 -- which goes to the FRONT of the list."""
    f, ln = [], n
    while ln & 7 == 0:
        f.append(8)
        ln >>= 3
    while ln & 3 == 0:
        f.append(4)
        ln >>= 2
    if ln & 1 == 0:
        ln >>= 1
        f.append(2)
        f[0], f[-1] = f[-1], f[0]
    assert ln == 1, "powers of two only"
    return f


def fft_pow2(x):
    """np.fft.fft of one real or complex vector whose length is a power of two, as numpy 2.x computes it."""
    x = np.asarray(x)
    n = len(x)
    cr, ci = np.array(x.real, np.float64), np.array(x.imag, np.float64)
    if n == 1:
        return cr + 1j * ci
    wr, wi = sincos_2pibyn(n)
    l1 = 1
    for ip in factors_pow2(n):
        ido = n // (ip * l1)
        cr, ci = {8: _pass8, 4: _pass4, 2: _pass2}[ip](cr, ci, l1, ido, wr, wi)
        l1 *= ip
    return cr + 1j * ci


def fft_pow8(x):
    """(the 4096-point case and its smaller relatives: radix-8 passes only)"""
    return fft_pow2(x)


def psd_exact(x, Fs=44100, noverlap=2048, nfft=4096, fused=True) -> np.ndarray:
    """mlab.specgram(x, NFFT=nfft, Fs, window_hanning, noverlap)[0], float64 [nfft/2 + 1][F].  fused: numpy's complex product
    has FMA3 on the host (the fixtures' hosts); False: re*re + im*im."""
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
        z = fft_pow2(fr * w)[: nfft // 2 + 1]
        re, im = z.real.copy(), z.imag.copy()
        p = _fma(re, re, im * im).astype(np.float64) if fused else re * re + im * im
        p[1:-1] *= 2.0
        out[:, f] = (p * r_fs) * r_s
    return out
