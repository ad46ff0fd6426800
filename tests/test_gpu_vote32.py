"""GPU: vote passes with 4-byte votes (a pass's query index + song id + biased delta in 31 bits, the last radix pass
widening to the 64-bit layout) return exactly what the 8-byte single pass returns.  The switch (SHZ_VOTE32: 0 never,
1 whenever it fits, unset: when passes stay large) is read once per process: each mode runs in a child process on the
same table and queries, all result arrays are hashed."""
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
from shazam_amd import _ffi, Table
ctx = _ffi.Context(0)
n, nc = 20 * 44100, 1500
tbl = Table(ctx)
pcm = ctx.synth_pcm(4321, 0, nc, n, 4000, 1500)
k, t1, ho, _ = ctx.fingerprint_batch(pcm, np.arange(nc + 1, dtype=np.uint64) * n, pcm_device=True)
tbl.insert_clips(k, t1, ho, sid0=1)
tbl.finalize()
rng = np.random.default_rng(8)
h = hashlib.sha256()
for nq in (1, 2, 3, 37, 300):
    qn = 6 * 44100
    tids = rng.integers(0, nc, nq)
    q = ctx.alloc(nq * qn * 2)
    for i in range(nq):
        ctx.check(_ffi.lib().shz_synth_pcm(ctx.h, 4321, int(tids[i]), 1, qn, 4000, 1500, int(rng.integers(0, n - qn)),
                                           _ffi.vp(q.ptr + i * qn * 2)))
    qk, qt, qo, _ = ctx.fingerprint_batch(q, np.arange(nq + 1, dtype=np.uint64) * qn, pcm_device=True)
    res = tbl.match(qk, qt, qo, 3)
    assert (res["sid"][:, 0] == 1 + tids).all()
    for name in sorted(res):
        h.update(np.ascontiguousarray(res[name]).tobytes())
    q.free()
print(h.hexdigest(), tbl.match_stats()["pairs"])
"""


def _run(mode):
    env = dict(os.environ)
    env.pop("SHZ_VOTE32", None)
    if mode is not None:
        env["SHZ_VOTE32"] = str(mode)
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.strip().split()


def test_four_byte_votes_equal_eight_byte_votes():
    base = _run(0)
    assert int(base[1]) > 8192, "the last batch must be large enough to leave the one-workgroup path"
    assert _run(1) == base
    assert _run(None) == base
