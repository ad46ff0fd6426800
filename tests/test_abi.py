"""CPU: the C-ABI library loads and exports every symbol include/shz.h declares (no compute)."""
import os
import re

from shazam_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "shz.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(shz_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    L = _ffi.lib()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), f"{n} declared in shz.h but not exported by libshz.so"
    # and the ctypes table binds exactly the declared set
    assert set(_ffi.SIGNATURES) == names


def test_frame_count_needs_no_gpu():
    L = _ffi.lib()
    for n, want in ((0, 1), (3000, 1), (4096, 1), (6143, 1), (6144, 2), (220500, 106), (1323000, 644), (7938000, 3874)):
        assert L.shz_frame_count(n) == want
    assert L.shz_version().startswith(b"shz")


def test_registry_mirrors_reference():
    import pytest
    import shazam_amd as S
    assert S.get_database("hip").__name__ == "HipFingerprintDB"
    with pytest.raises(TypeError, match="Unsupported database type supplied."):
        S.get_database("nope")
    with pytest.raises(NotImplementedError):
        S.fingerprint([0, 1, 2], wsize=8192)
