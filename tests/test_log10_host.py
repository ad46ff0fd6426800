"""CPU: the logarithm that decides ties in the peak test (`maximum_filter(A) == A` on A = 10*log10(P),
__init__.py:143 after :241) is the correctly rounded one.  `shz_db_values` is the host build of the very
function the kernels call (csrc/shz_log10.h); checked against `decimal` at 60 digits."""
import ctypes as C
from decimal import Decimal, getcontext

import numpy as np

from shazam_amd import _ffi


def _db(p):
    p = np.ascontiguousarray(p, np.float64)
    out = np.empty_like(p)
    rc = _ffi.lib().shz_db_values(p.ctypes.data_as(C.c_void_p), p.size, out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def _cr_db(p):
    getcontext().prec = 60
    return np.array([10.0 * float(Decimal(float(v)).log10()) if v != 0 else 0.0 for v in p])


def test_db_values_correctly_rounded():
    rng = np.random.default_rng(11)
    p = np.concatenate([np.exp(rng.uniform(np.log(1e-30), np.log(1e20), 20000)),
                        1.0 + rng.uniform(-1e-3, 1e-3, 2000),
                        [0.0, 1.0, 2.0, 0.5, 10.0, 1e5, 1e-5, 2.0 ** 0.5, 0.5 ** 0.5, 5e-324, 1e-310, 1.7e308]])
    got = _db(p)
    assert np.array_equal(got, _cr_db(p))
    # what numpy's vendor routine does with the same inputs on this host (informative; SVML: ~0.05 % differ)
    nz = p != 0
    assert (10.0 * np.log10(p[nz]) != got[nz]).mean() < 0.06


def test_adjacent_powers_collapse_to_one_db_value():
    """The property that makes the dB-domain test differ from a power-domain one: runs of adjacent doubles share a dB
    value.  Around 1e5 the runs hold 5-14 doubles."""
    v = [1e5]
    for _ in range(60):
        v.append(np.nextafter(v[-1], np.inf))
    d = _db(np.array(v))
    assert np.all(np.diff(d) >= 0)
    _, counts = np.unique(d, return_counts=True)
    assert counts[1:-1].min() >= 4 and counts.max() <= 16
