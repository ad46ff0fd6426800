"""Latency of ONE query through the hot path (host PCM in, result dicts out): the serving-side number next to the
batched throughput of bench.py / bench_db.py.  python scripts/single_query_latency.py [songs]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import shazam_amd as S  # noqa: E402

songs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ctx = S.get_context(0)
db = S.get_database("hip")(ctx=ctx)
n = 30 * 44100
for c0 in range(0, songs, 500):
    nc = min(500, songs - c0)
    pcm = ctx.synth_pcm(4321, c0, nc, n, 4000, 1500)
    k, t1, ho, _ = ctx.fingerprint_batch(pcm, np.arange(nc + 1, dtype=np.uint64) * n, pcm_device=True)
    pcm.free()
    for i in range(nc):
        db.songs[c0 + i + 1] = {"song_name": str(c0 + i), "file_sha1": "00", "total_hashes": int(ho[i + 1] - ho[i]),
                                "fingerprinted": 1, "date_created": None}
    db.insert_clips(k, t1, ho, c0 + 1)
db.finalize()
_trk = ctx.synth_pcm(4321, 7, 1, n, 4000, 1500)
q = _trk.download(np.int16, n)[13 * 2048 + 77:13 * 2048 + 77 + 5 * 44100].copy()
for _ in range(3):
    S.recognize(q, db=db)
lat = {"fingerprint": [], "match": [], "total": []}
for _ in range(50):
    t0 = time.perf_counter()
    res, tf, tq, ta = S.recognize(q, db=db)
    lat["total"].append(time.perf_counter() - t0)
    lat["fingerprint"].append(tf)
    lat["match"].append(tq)
print("rows", db.num_fingerprints(), "top1", res[0]["song_id"], res[0]["offset"])
for k_, v in lat.items():
    v = np.array(v) * 1e3
    print(f"{k_:12s} p50 {np.median(v):.3f} ms  p99 {np.percentile(v, 99):.3f} ms")
