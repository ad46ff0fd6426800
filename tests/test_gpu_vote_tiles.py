"""GPU: the vote-tile fold (two radix passes over the 4-byte votes' upper bits, then per tile an LDS hash count -> one
entry per song -> top-n; shz_table.hip: vt_fold_kernel) returns exactly what the full sort + record chain returns, on
tables that exercise each of its cases:

  small   1,500 songs: the ordered bits reach down to whole song ids (no song bits left inside a tile group)
  wide    70,000 songs: 5 song-id bits stay unordered inside a tile group, passes of up to 16 queries
  tonal   64 copies of a stationary eight-tone song under consecutive ids far up the id range: one tile group carries
          thousands of distinct (song, delta) pairs, so the tile must be swept in several parts
  gaps    one pass of 24 queries of which every fourth has no hashes, every fourth only hashes the table does not hold and
          every fourth seven hashes: empty segments and blocks of a few votes between ordinary ones
  topn20  top-20 (beyond the tile path's limit) over several vote passes with two-level top-n: regression test for the
          fold's candidate buffers, which used to live in the workspace slots of the probe's group tables

The switches (SHZ_VOTE32, SHZ_VOTE_TILES) are read once per process: every mode runs in a child process on the same
table and queries and prints a digest of all result arrays."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, hashlib
import numpy as np
sys.path.insert(0, %r)
import shazam_amd as S
from shazam_amd import _ffi, Table
case = sys.argv[1]
ctx = _ffi.Context(0)
tbl = Table(ctx)
rng = np.random.default_rng(11)
h = hashlib.sha256()

def add_synth(seed, nc, n, sid0):
    pcm = ctx.synth_pcm(seed, 0, nc, n, 4000, 1500)
    k, t1, ho, _ = ctx.fingerprint_batch(pcm, np.arange(nc + 1, dtype=np.uint64) * n, pcm_device=True)
    tbl.insert_clips(k, t1, ho, sid0=sid0)
    pcm.free()

def synth_queries(seed, tids, n, qn):
    nq = len(tids)
    q = ctx.alloc(nq * qn * 2)
    for i in range(nq):
        ctx.check(_ffi.lib().shz_synth_pcm(ctx.h, seed, int(tids[i]), 1, qn, 4000, 1500, int(rng.integers(0, n - qn)),
                                           _ffi.vp(q.ptr + i * qn * 2)))
    out = ctx.fingerprint_batch(q, np.arange(nq + 1, dtype=np.uint64) * qn, pcm_device=True)
    q.free()
    return out[:3]

def run(qk, qt, qo, topn, expect=None):
    res = tbl.match(qk, qt, qo, topn)
    if expect is not None:
        assert (res["sid"][:, 0] == expect).all(), (res["sid"][:, 0], expect)
    for name in sorted(res):
        h.update(np.ascontiguousarray(res[name]).tobytes())

if case == "small":
    n, nc = 20 * 44100, 1500
    add_synth(4321, nc, n, 1)
    tbl.finalize()
    for nq in (1, 2, 3, 37, 300):
        tids = rng.integers(0, nc, nq)
        run(*synth_queries(4321, tids, n, 6 * 44100), 3, 1 + tids)
elif case == "wide":
    n, nc = 8 * 44100, 70000
    add_synth(99, nc, n, 1)
    tbl.finalize()
    for nq, topn in ((1, 1), (2, 5), (37, 8), (90, 2)):
        tids = rng.integers(0, nc, nq)
        run(*synth_queries(99, tids, n, 5 * 44100), topn, 1 + tids)
elif case == "tonal":
    n = 30 * 44100
    t = np.arange(n) / 44100.0
    fr = [440.0, 1318.5, 3520.0, 700.3, 2217.4, 5587.6, 260.7, 9000.1]
    amp = [5000, 4000, 3000, 3500, 2500, 2000, 3000, 1500]
    tone = sum(a * np.sin(2 * np.pi * f * t) for a, f in zip(amp, fr)).astype(np.int16)
    k, t1, ho = S.fingerprint_batch([tone], ctx=ctx)
    assert len(k) > 300   # ~1,000 hashes of ~500 keys: a 10 s excerpt votes ~1,000 times per copy at ~240 deltas
    ncopy = 64
    kk, tt = np.tile(k, ncopy), np.tile(t1, ncopy)
    hh = np.arange(ncopy + 1, dtype=np.uint64) * len(k)
    tbl.insert_clips(kk, tt, hh, sid0=100000)       # 17 song-id bits without 100,000 songs
    add_synth(7, 300, 10 * 44100, 1)
    tbl.finalize()
    qa = tone[5 * 44100: 15 * 44100]
    qk, qt, qo = S.fingerprint_batch([qa, qa, tone[: 8 * 44100]], ctx=ctx)
    for topn in (1, 8):
        res = tbl.match(qk, qt, qo, topn)
        assert (res["sid"][:, 0] == 100000).all(), res["sid"][:, 0]   # equal counts: the smallest id wins
        for name in sorted(res):
            h.update(np.ascontiguousarray(res[name]).tobytes())
    tids = rng.integers(0, 300, 5)
    sq = synth_queries(7, tids, 10 * 44100, 5 * 44100)
    # one batch of tonal and ordinary queries: tiles of both kinds in one pass
    k2 = np.concatenate([qk, sq[0]]); t2 = np.concatenate([qt, sq[1]])
    o2 = np.concatenate([qo, qo[-1] + sq[2][1:]])
    run(k2, t2, o2, 4)
elif case == "gaps":
    # a pass whose segments include empty ones: queries without hashes, queries whose hashes the table does not hold,
    # queries of a handful of hashes -- between ordinary ones (the expand runs by the sort's blocks: blocks of a few votes,
    # segments without blocks)
    n, nc = 8 * 44100, 3000
    add_synth(55, nc, n, 1)
    tbl.finalize()
    tids = rng.integers(0, nc, 24)
    qk, qt, qo = synth_queries(55, tids, n, 5 * 44100)
    ks, ts, offs, expect = [], [], [0], []
    for i in range(24):
        a, b = int(qo[i]), int(qo[i + 1])
        kind = i %% 4
        if kind == 1:      # no hashes at all
            kq, tq = qk[:0], qt[:0]
        elif kind == 2:    # hashes of keys no track has (f1 = f2 = 2047 never occurs: dt > 0 there is no such pair)
            kq = np.full(40, (2047 << 20) | (2047 << 8) | 3, np.uint32) + np.arange(40, dtype=np.uint32) %% 5
            tq = np.arange(40, dtype=np.uint32)
        elif kind == 3:    # a handful of the query's hashes
            kq, tq = qk[a:a + 7], qt[a:a + 7]
        else:
            kq, tq = qk[a:b], qt[a:b]
            expect.append((len(offs) - 1, 1 + int(tids[i])))
        ks.append(kq); ts.append(tq); offs.append(offs[-1] + len(kq))
    k2, t2, o2 = np.concatenate(ks), np.concatenate(ts), np.array(offs, np.uint64)
    for topn in (1, 3):
        res = tbl.match(k2, t2, o2, topn)
        for q, sid in expect:
            assert int(res["sid"][q, 0]) == sid, (q, sid)
        assert int(res["nres"][1]) == 0 and int(res["npairs"][2]) == 0
        for name in sorted(res):
            h.update(np.ascontiguousarray(res[name]).tobytes())
elif case == "topn20":
    n, nc = 30 * 44100, 12000
    add_synth(5, nc, n, 1)
    tbl.finalize()
    tids = rng.integers(0, nc, 40)
    run(*synth_queries(5, tids, n, 10 * 44100), 20, 1 + tids)
print(h.hexdigest(), tbl.match_stats()["pairs"])
"""


def _run(case, vote32, tiles):
    env = dict(os.environ)
    for name, val in (("SHZ_VOTE32", vote32), ("SHZ_VOTE_TILES", tiles)):
        env.pop(name, None)
        if val is not None:
            env[name] = str(val)
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT, case], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout.strip().split()


@pytest.mark.parametrize("case", ["small", "wide", "tonal", "gaps"])
def test_vote_tiles_equal_full_sort(case):
    base = _run(case, 0, 0)                      # 8-byte votes, one pass, full sort
    assert int(base[1]) > 8192
    assert _run(case, 1, 0) == base              # 4-byte votes, full sort with the widening last pass
    assert _run(case, 1, 1) == base              # 4-byte votes, tiles
    assert _run(case, None, None) == base        # the defaults


def test_many_results_over_several_passes():
    base = _run("topn20", 0, 0)
    assert int(base[1]) > 40 * 32768, "two-level top-n needs more than 32,768 votes per query"
    assert _run("topn20", 1, None) == base


def test_vote_paths_agree_at_200k_songs():
    """One table of 200,000 x 30 s tracks (2.3e9 rows), 300 ten-second queries at 10 dB SNR and 60 at 0 dB in batches,
    plus single queries: the shipped vote path (segmented 4-byte passes, vote tiles) against the full sort of 8-byte
    votes (SHZ_MATCH_FULL_SORT) in the same process, every output array."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench_db
    from shazam_amd import _ffi
    ctx = _ffi.Context(0)
    songs, n, qn = 200000, 30 * 44100, 10 * 44100
    tbl, _stats, bufs = bench_db.build_table(ctx, songs, 30.0, 1000, finalize_every=100000)
    for b in bufs[:2]:
        b.free()
    rng = np.random.default_rng(21)
    for nq, snr, topn in ((300, 10.0, 2), (60, 0.0, 5), (1, 10.0, 8), (1, 0.0, 1)):
        tids = rng.integers(0, songs, nq)
        starts = rng.integers(0, n - qn, nq)
        q, qb = bench_db.make_queries(ctx, tids, starts, qn, snr)
        k, t1, ho, _ = ctx.fingerprint_batch(q, np.arange(nq + 1, dtype=np.uint64) * qn, fs=44100, pcm_device=True)
        fast = tbl.match(k, t1, ho, topn)
        full = tbl.match(k, t1, ho, topn, full_sort=True)
        for name in sorted(full):
            assert np.array_equal(fast[name], full[name]), (nq, snr, name)
        if snr >= 10.0:
            assert (fast["sid"][:, 0] == 1 + tids).mean() > 0.97
        for b in {id(b): b for b in qb}.values():
            b.free()


def test_fuzz_and_schema_suites_under_forced_4_byte_votes():
    """The randomized differential test of the match (many table shapes incl. several segments, hot keys, empty queries)
    and the match golden tests, once more in a child process with SHZ_VOTE32=1: every pass whose layout fits takes the
    4-byte votes, the expand by sort blocks and the vote tiles, however few votes it has."""
    env = dict(os.environ, SHZ_VOTE32="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_match_fuzz.py"),
                          os.path.join(ROOT, "tests", "test_gpu_match.py"), "-x", "-q", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
