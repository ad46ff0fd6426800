"""EXPERIMENT: what a bar known before the tiles start would be worth for ONE query at 1M songs.  The same query is matched
several times; under SHZ_VT_KEEPBAR=1 the query's bar survives from the previous call (= the final n-th best: the ideal bar).
rocprofv3 --kernel-trace ... -- python3 scripts/prebar_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_db  # noqa: E402
from shazam_amd import _ffi  # noqa: E402

songs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctx = _ffi.Context(0)
tbl, build, _bufs = bench_db.build_table(ctx, songs, 30.0, 1000, finalize_every=50000)
n, qn = 30 * 44100, 10 * 44100
rng = np.random.default_rng(3)
for qi in range(4):
    tid, st = int(rng.integers(0, songs)), int(rng.integers(0, n - qn))
    q, bufs = bench_db.make_queries(ctx, np.array([tid]), np.array([st]), qn, 10.0)
    k, t1, ho, _ = ctx.fingerprint_batch(q, np.array([0, qn], np.uint64), fs=44100, pcm_device=True)
    lat = []
    for i in range(6):
        ctx.sync()
        t0 = time.perf_counter()
        res = tbl.match(k, t1, ho, 2)
        lat.append((time.perf_counter() - t0) * 1e3)
        assert res["sid"][0, 0] == tid + 1
    print("query", qi, "match ms:", " ".join("%.3f" % v for v in lat), flush=True)
