"""GPU: table finalize with song ids / offsets too wide for the single-u64 fast path (sid + off bits > 32)
goes through the generic two-sort path; both must give the same sorted unique rows as numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sid_hi,off_hi", [(2 ** 31, 2 ** 20), (2 ** 10, 2 ** 31), (2 ** 17, 2 ** 12)])
def test_finalize_wide_and_narrow(sid_hi, off_hi):
    import shazam_amd as S
    ctx = S.get_context(0)
    rng = np.random.default_rng(sid_hi % 97)
    n = 150000
    key = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    key[: n // 3] = key[n // 3: 2 * (n // 3)]
    sid = rng.integers(1, sid_hi, n).astype(np.uint32)
    off = rng.integers(0, off_hi, n).astype(np.uint32)
    sid[:100], off[:100], key[:100] = sid[100:200], off[100:200], key[100:200]     # exact duplicate rows
    t = S.Table(ctx)
    t.insert(key, sid, off)
    t.finalize()
    want = np.unique(np.stack([key, sid, off], 1).astype(np.uint64), axis=0)
    k, s, o = t.export()
    assert np.array_equal(np.stack([k, s, o], 1).astype(np.uint64), want)
    # the match path packs (q, sid, delta) into 64 bits: wide ids must either work or fail loudly
    qk, qo = key[:50].copy(), (off[:50] % 1000).astype(np.uint32)
    try:
        r = t.match(qk, qo, np.array([0, 50], np.uint64), 3)
        assert int(r["nhash"][0]) == len(set(zip(qk.tolist(), qo.tolist())))
    except S.ShzError as e:
        assert e.code == -5
    t.close()
