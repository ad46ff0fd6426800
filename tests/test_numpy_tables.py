"""CPU: the pieces of the reference's spectrogram arithmetic that the device's fp64 path follows (csrc: shz_numpy_tables,
np_fft4096) against numpy itself on this host, and the oracle's restatement (oracle/np_exact.py) against the digests of
the reference's own spectrograms (tests/golden/psd_digests.json)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest


def _tables(n):
    from shazam_amd import _ffi
    w, tw, s = np.empty(n), np.empty(2 * n), C.c_double()
    assert _ffi.lib().shz_numpy_tables(n, _ffi.ptr(w), _ffi.ptr(tw), C.byref(s)) == 0
    return w, tw[0::2].copy(), tw[1::2].copy(), s.value


@pytest.mark.parametrize("n", [4096, 2048, 1024, 512, 256, 128, 64])
def test_library_tables_are_numpys(n):
    from oracle import np_exact as E
    w, tr, ti, s = _tables(n)
    assert np.array_equal(w, np.hanning(n)) and np.array_equal(w, E.hanning(n))
    assert s == float((np.hanning(n) ** 2).sum())
    er, ei = E.sincos_2pibyn(n)
    assert np.array_equal(tr, er) and np.array_equal(ti, ei)
    k = np.arange(n)
    assert np.abs(tr - np.cos(2 * np.pi * k / n)).max() < 1e-15 and np.abs(ti - np.sin(2 * np.pi * k / n)).max() < 1e-15   # (the comparison value carries the error of its argument)


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_restated_passes_are_numpys_fft_bit_for_bit(n):
    from oracle import np_exact as E
    rng = np.random.default_rng(n)
    for x in (rng.standard_normal(n), rng.integers(-32768, 32767, n) * np.hanning(n), rng.standard_normal(n) + 1j * rng.standard_normal(n),
              np.zeros(n), np.eye(1, n, 3)[0]):
        got, want = E.fft_pow2(x), np.fft.fft(x)
        assert np.array_equal(got.real, want.real) and np.array_equal(got.imag, want.imag)


def _digest(P):
    return hashlib.sha256(np.ascontiguousarray(np.where(P == 0, 1.0, P), np.float64).tobytes()).hexdigest()


def psd_cases(golden_dir):
    """name -> (pcm, digest entry): inputs regenerated from integers, checked against the fixture's pcm digest"""
    import importlib.util
    from oracle import synth  # noqa: F401
    spec = importlib.util.spec_from_file_location("make_golden_inputs", os.path.join(golden_dir, "make_golden.py"))   # (imports nothing of the reference until its main() runs)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    dig = json.load(open(os.path.join(golden_dir, "psd_digests.json")))
    out = {}
    for name, case in mod.psd_inputs().items():
        x, fs, wr = case[:3]
        d = dig[name]
        assert hashlib.sha256(x.tobytes()).hexdigest() == d["pcm_sha256"] and d["Fs"] == fs and d["wratio"] == wr
        assert d["nfft"] == (case[3] if len(case) > 3 else 4096)
        out[name] = (x, d)
    return out


@pytest.mark.parametrize("name", ["short_1500", "edge_exact_4096", "edge_silence_20000", "variant_wr0", "edge_loud_fullscale",
                                  "ws2048", "ws1024", "ws512", "ws256", "ws128", "ws64", "ws2048_short"])
def test_oracle_spectrogram_hits_the_references_digest(golden_dir, name):
    from oracle import np_exact as E
    x, d = psd_cases(golden_dir)[name]
    P = E.psd_exact(x, d["Fs"], int(d["nfft"] * d["wratio"]), d["nfft"])
    assert list(P.shape) == d["shape"]
    for a, b, v in d["probe"]:
        assert float(np.where(P == 0, 1.0, P)[a, b]).hex() == v
    assert _digest(P) == d["sha256"]


def test_this_hosts_numpy_product_is_what_the_fixtures_hosts_was():
    """The digests were made on a host whose numpy multiplies complex numbers with FMA3; the probe the Python layer runs for
    every context must say so here (the oracle hits the digests with fused=True above, and misses with fused=False)."""
    from shazam_amd import _ffi
    assert _ffi.numpy_product_is_fused() is True

