"""One 10 s query at a time against a 1M x 30 s table, for a kernel trace:
rocprofv3 --kernel-trace ... -- python3 scripts/single_query_1m_trace.py [songs] [iters]; scripts/trace_gaps.py <csv> 40"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_db  # noqa: E402
from shazam_amd import _ffi  # noqa: E402

songs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = _ffi.Context(0)
tbl, build, _bufs = bench_db.build_table(ctx, songs, 30.0, 1000, finalize_every=50000)
n, qn = 30 * 44100, 10 * 44100
rng = np.random.default_rng(3)
lat = []
for i in range(iters):
    tid, st = int(rng.integers(0, songs)), int(rng.integers(0, n - qn))
    q, bufs = bench_db.make_queries(ctx, np.array([tid]), np.array([st]), qn, 10.0)
    k, t1, ho, _ = ctx.fingerprint_batch(q, np.array([0, qn], np.uint64), fs=44100, pcm_device=True)
    ctx.sync()
    t0 = time.perf_counter()
    res = tbl.match(k, t1, ho, 2)
    lat.append(time.perf_counter() - t0)
    assert res["sid"][0, 0] == tid + 1
    for b in {id(b): b for b in bufs}.values():
        b.free()
print("match p50 %.3f ms" % (np.median(np.array(lat[3:])) * 1e3))
