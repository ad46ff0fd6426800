"""CPU: the match-side CPU baseline of bench.py (the oracle's return_matches + align_matches over a dict-indexed table, and the
numpy sorted-array matcher beside it) on a small host table: both must name the planted song."""
import numpy as np


class _FakeTable:
    def __init__(self, k, s, o):
        idx = np.lexsort((o, s, k))
        self.cols = k[idx], s[idx], o[idx]

    def export(self):
        return self.cols


def test_match_cpu_baseline_agrees_with_planted_songs():
    import bench
    rng = np.random.default_rng(3)
    songs, per = 40, 600
    k = rng.integers(0, 1 << 28, songs * per).astype(np.uint32)
    s = np.repeat(np.arange(1, songs + 1), per).astype(np.uint32)
    o = np.tile(np.sort(rng.integers(0, 640, per)), songs).astype(np.uint32)
    tbl = _FakeTable(k, s, o)
    # queries: 120 rows of song q + 1 shifted by 7 frames, plus a few rows of other songs
    qk, qo, ho = [], [], [0]
    nq = 12
    for q in range(nq):
        rows = np.arange(q * per + 50, q * per + 170)
        noise = rng.integers(0, songs * per, 15)
        qk.append(np.r_[k[rows], k[noise]])
        qo.append(np.r_[o[rows] - np.minimum(o[rows], 7), o[noise]])
        ho.append(ho[-1] + len(qk[-1]))
    qk, qo, ho = np.concatenate(qk), np.concatenate(qo).astype(np.uint32), np.array(ho, np.uint64)
    gpu = {"sid": np.arange(1, nq + 1, dtype=np.uint32)[:, None].repeat(2, 1), "nres": np.full(nq, 2, np.uint32)}
    out = bench.match_cpu_baseline(tbl, qk, qo, ho, gpu, songs, budget_s=0.0)
    assert out["queries"] == nq and out["cores"] == 1 and out["table_rows"] == songs * per
    assert out["dict_table"]["top1_equals_gpu"] == nq and out["numpy_sorted"]["top1_equals_gpu"] == nq
    assert out["dict_table"]["p50_ms"] > 0 and out["numpy_sorted"]["p50_ms"] > 0
