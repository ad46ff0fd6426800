"""GPU: the vote tiles never hang and never drop votes silently, and they handle tracks longer
than 2^12 frames (next #4).

* Every LDS hash probe is bounded; an exhausted probe, a full hand-over list or a range that fits no sweep sets the
  pass's flag word, and the host votes the sub-batch again through the full sort.  Forced here with the test switches
  of shz_set_debug: a hand-over list with room for ONE range, probes that give up after one round.
* A stationary 10- or 14-minute track makes a 10 s query vote at more than 2^12 distinct offsets differences for one
  song: vt_fold_kernel sweeps the delta range in parts.

In every case all result arrays equal those of SHZ_MATCH_FULL_SORT (align_matches is a function of the multiset of
votes, recognizer.py:289-338).  The 4-byte vote path is forced (SHZ_VOTE32=1, read once per process), so each case runs in
a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
import shazam_amd as S
from shazam_amd import _ffi, Table
case = sys.argv[1]
ctx = _ffi.Context(0)
tbl = Table(ctx)
rng = np.random.default_rng(3)
FS = 44100

def tone(seconds, shift=0.0):
    t = np.arange(int(seconds * FS)) / FS
    fr = [440.0, 1318.5, 3520.0, 700.3, 2217.4, 5587.6, 260.7, 9000.1]
    amp = [5000, 4000, 3000, 3500, 2500, 2000, 3000, 1500]
    return sum(a * np.sin(2 * np.pi * (f + shift) * t) for a, f in zip(amp, fr)).astype(np.int16)

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

def same(qk, qt, qo, topn, expect_redo=None):
    r0 = ctx.vt_redo_count()
    fast = tbl.match(qk, qt, qo, topn)
    redo = ctx.vt_redo_count() - r0
    full = tbl.match(qk, qt, qo, topn, full_sort=True)
    for name in sorted(full):
        assert np.array_equal(fast[name], full[name]), (case, name)
    if expect_redo is not None:
        assert (redo > 0) == expect_redo, (case, redo)
    return fast

if case == "forced":
    # 64 copies of a stationary track: batches of vt_stream overflow and are handed to vt_fold by the dozen
    tn = tone(30)
    k, t1, ho = S.fingerprint_batch([tn], ctx=ctx)
    ncopy = 64
    tbl.insert_clips(np.tile(k, ncopy), np.tile(t1, ncopy), np.arange(ncopy + 1, dtype=np.uint64) * len(k), sid0=5000)
    add_synth(7, 400, 10 * FS, 1)
    tbl.finalize()
    qa = tn[5 * FS: 15 * FS]
    tids = rng.integers(0, 400, 6)
    sq = synth_queries(7, tids, 10 * FS, 5 * FS)
    qk, qt, qo = S.fingerprint_batch([qa, qa, tn[: 8 * FS]], ctx=ctx)
    k2 = np.concatenate([qk, sq[0]]); t2 = np.concatenate([qt, sq[1]]); o2 = np.concatenate([qo, qo[-1] + sq[2][1:]])
    base = same(k2, t2, o2, 4, expect_redo=False)          # the shipped path: nothing overflows
    assert (base["sid"][:3, 0] == 5000).all()
    ctx.set_debug(1)                                        # room for one handed-over range: the second one sets the flag
    same(k2, t2, o2, 4, expect_redo=True)
    ctx.set_debug(2)                                        # probes give up after one round: any collision sets the flag
    same(k2, t2, o2, 4, expect_redo=True)
    same(qk[: int(qo[1])], qt[: int(qo[1])], qo[:2], 3, expect_redo=True)   # one query: the one-workgroup fold, same switch
    ctx.set_debug(0)
    same(k2, t2, o2, 4, expect_redo=False)
elif case in ("long14", "long15"):
    minutes = 10 if case == "long14" else 14
    long_tone = tone(60 * minutes)
    k, t1, ho = S.fingerprint_batch([long_tone, tone(60 * minutes, shift=37.0)], ctx=ctx)
    frames = int(t1.max()) + 1
    assert frames > (1 << 12) * (2 if case == "long14" else 3), frames
    add_synth(11, 300, 20 * FS, 1)
    tbl.insert_clips(k, t1, ho, sid0=301)                   # two long tracks among 300 short ones
    tbl.finalize()
    qa = long_tone[100 * FS: 110 * FS]
    qk, qt, qo = S.fingerprint_batch([qa], ctx=ctx)
    r = same(qk, qt, qo, 3, expect_redo=False)              # one query, one range: sweeps over delta parts
    assert r["sid"][0, 0] == 301 and r["npairs"][0] > 8192, (r["sid"][0], r["npairs"][0])
    tids = rng.integers(0, 300, 40)
    sq = synth_queries(11, tids, 20 * FS, 6 * FS)
    qb = S.fingerprint_batch([qa, long_tone[300 * FS: 305 * FS], tone(8, shift=37.0)], ctx=ctx)
    k2 = np.concatenate([qb[0], sq[0]]); t2 = np.concatenate([qb[1], sq[1]]); o2 = np.concatenate([qb[2], qb[2][-1] + sq[2][1:]])
    r = same(k2, t2, o2, 5, expect_redo=False)              # a batch: tiles, hand-overs, sweeps
    assert list(r["sid"][:3, 0]) == [301, 301, 302] and (r["sid"][3:, 0] == 1 + tids).all()
print("ok")
"""


def _run(case):
    env = dict(os.environ, SHZ_VOTE32="1")
    env.pop("SHZ_VOTE_TILES", None)
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT, case], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert out.stdout.strip().endswith("ok")


def test_forced_overflows_fall_back_to_the_full_sort():
    _run("forced")


@pytest.mark.parametrize("case", ["long14", "long15"])
def test_tracks_longer_than_4096_frames_keep_the_tiles(case):
    _run(case)
