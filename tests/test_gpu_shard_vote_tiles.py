"""GPU: the gathered votes of a key-sharded match go through the vote tiles as well (shz_table.hip: pairs_vote_tiles -- one
stable sort of the 8-byte votes by their query bits, the query bits dropped, then the passes of the unsharded match) when
their layout fits 31 bits.  Same arrays as the unsharded table, and as the full sort of the 8-byte votes:

  * small tables with SHZ_VOTE32=1 (the switch that takes the 4-byte paths whatever the vote count) in a child process --
    the switch is read once per process -- incl. an empty query, a query that matches nothing, ties;
  * a table large enough for the default path (> 2^22 votes), in this process."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import shazam_amd as S
from shazam_amd.shard import ShardedTable
from test_gpu_shard import _rows, _queries, FIELDS
ctx = S.get_context(0)
r0 = ctx.vt_redo_count()
for nshards, n, nq in ((2, 120000, 40), (5, 400000, 150), (8, 60000, 7)):
    rng = np.random.default_rng(7 * nshards)
    k, s, o = _rows(rng, n, nsongs=3000 if nshards == 5 else 60)
    one, sh = S.Table(ctx), ShardedTable(ctx, nshards=nshards)
    one.insert(k, s, o); sh.insert(k, s, o)
    one.finalize(); sh.finalize()
    qk, qo, qoff = _queries(rng, k, nq)
    for topn in (1, 3, 8):
        ra, rb = one.match(qk, qo, qoff, topn, full_sort=True), sh.match(qk, qo, qoff, topn)
        for f in FIELDS:
            assert np.array_equal(ra[f], rb[f]), (f, nshards, topn)
    one.close(); sh.close()
assert ctx.vt_redo_count() == r0
print("ok", sh.last_votes)
"""


def test_small_sharded_votes_through_the_tiles():
    env = dict(os.environ, SHZ_VOTE32="1")
    out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "tests"))], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stdout.split()[0] == "ok"


def test_large_sharded_votes_take_the_tiles_by_default():
    """4,000 songs x 20 s, 400 six-second queries: ~6 M votes in one call, 3 shards on this GPU"""
    import shazam_amd as S
    from shazam_amd import _ffi
    from shazam_amd.shard import ShardedTable
    ctx = S.get_context(0)
    n, nc, qn = 20 * 44100, 4000, 6 * 44100
    pcm = ctx.synth_pcm(77, 0, nc, n, 4000, 1500)
    k, t1, ho, _ = ctx.fingerprint_batch(pcm, np.arange(nc + 1, dtype=np.uint64) * n, pcm_device=True)
    pcm.free()
    one, sh = S.Table(ctx), ShardedTable(ctx, nshards=3)
    for t in (one, sh):
        t.insert_clips(k, t1, ho, 1)
        t.finalize()
    rng = np.random.default_rng(5)
    nq = 400
    tids = rng.integers(0, nc, nq)
    q = ctx.alloc(nq * qn * 2)
    for i in range(nq):
        ctx.check(_ffi.lib().shz_synth_pcm(ctx.h, 77, int(tids[i]), 1, qn, 4000, 1500, int(rng.integers(0, n - qn)),
                                           _ffi.vp(q.ptr + i * qn * 2)))
    qk, qt, qo, _ = ctx.fingerprint_batch(q, np.arange(nq + 1, dtype=np.uint64) * qn, pcm_device=True)
    q.free()
    for topn in (2, 5):
        ra, rb = one.match(qk, qt, qo, topn), sh.match(qk, qt, qo, topn)
        assert sh.last_votes > 1 << 22
        for f in ("sid", "delta", "aligned", "dedup", "nres", "nhash", "npairs"):
            assert np.array_equal(ra[f], rb[f]), (f, topn)
        assert (ra["sid"][:, 0] == 1 + tids).mean() > 0.98
    one.close()
    sh.close()
