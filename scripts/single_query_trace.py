"""One-query-at-a-time loop for a kernel trace: python scripts/single_query_trace.py [songs] [iters]
(run under rocprofv3 --kernel-trace; scripts/trace_gaps.py summarises the last iteration)."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from shazam_amd import _ffi, Table  # noqa: E402

songs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = _ffi.Context(0)
n = 30 * 44100
tbl = Table(ctx)
for c0 in range(0, songs, 1000):
    nc = min(1000, songs - c0)
    pcm = ctx.synth_pcm(1234, c0, nc, n, 0, 8000)
    k, t1, ho, _ = ctx.fingerprint_batch(pcm, np.arange(nc + 1, dtype=np.uint64) * n, pcm_device=True)
    pcm.free()
    tbl.insert_clips(k, t1, ho, sid0=1 + c0)
tbl.finalize()
_trk = ctx.synth_pcm(1234, 7, 1, n, 0, 8000)
q = _trk.download(np.int16, n)[13 * 2048 + 77:13 * 2048 + 77 + 5 * 44100].copy()
qoff = np.array([0, len(q)], np.uint64)
lat = []
for i in range(iters):
    t0 = time.perf_counter()
    k, t1, ho, _ = ctx.fingerprint_batch(q, qoff)
    t1_ = time.perf_counter()
    res = tbl.match(k, t1, ho, 2)
    t2 = time.perf_counter()
    lat.append((t1_ - t0, t2 - t1_))
lat = np.array(lat[5:]) * 1e3
print("fingerprint p50 %.3f ms, match p50 %.3f ms; top1 %d %d" % (np.median(lat[:, 0]), np.median(lat[:, 1]), res["sid"][0, 0], res["delta"][0, 0]))
