#!/usr/bin/env python3
"""One-GPU proxy of the per-rank post-exchange cost of the sharded build (VERDICT r02 next #1d): the rows of `songs`
30 s tracks as k blocks (what k ranks would have staged) -> shz_table_finalize_runs.  Prints one JSON line per k.

    python scripts/kway_bench.py [--songs 96000] [--ks 1,2,4,8,16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=96000)
    ap.add_argument("--ks", default="1,2,4,8,16")
    ap.add_argument("--seconds", type=float, default=30.0)
    a = ap.parse_args()
    from shazam_amd import _ffi, Table
    import bench_db
    ctx = _ffi.Context(0)
    chunk = 1000
    n_samples = int(a.seconds * 44100)
    frames = int(_ffi.lib().shz_frame_count(n_samples))
    cap = chunk * frames * 24 + 1024
    kbuf, tbuf, pcm = ctx.alloc(cap * 4), ctx.alloc(cap * 4), ctx.alloc(chunk * n_samples * 2)
    off = np.arange(chunk + 1, dtype=np.uint64) * n_samples
    for k in [int(x) for x in a.ks.split(",")]:
        per = a.songs // k // chunk * chunk
        tbl = Table(ctx)
        tbl.reserve(int(per * k * frames * bench_db.ROWS_PER_FRAME_HINT), 0, gather=True)
        run_rows = []
        for r in range(k):
            rows = 0
            for c0 in range(r * per, (r + 1) * per, chunk):
                ctx.synth_pcm(bench_db.SEED_TRACKS, c0, chunk, n_samples, 4000, 1500, out=pcm)
                _, _, ho, cnt = ctx.fingerprint_batch(pcm, off, fs=44100, pcm_device=True, out_key=kbuf, out_t1=tbuf, cap=cap)
                tbl.insert_clips(kbuf, tbuf, ho, sid0=1 + c0, device=True)
                rows += cnt
            run_rows.append(rows)
        ctx.sync()
        tbl.phase_stats(reset=True)
        t0 = time.perf_counter()
        tbl.finalize_runs(run_rows)
        ctx.sync()
        dt = time.perf_counter() - t0
        ph = {n: round(v, 4) for n, v in tbl.phase_stats().items() if v > 2e-4}
        print(json.dumps({"k": k, "rows": int(sum(run_rows)), "seconds": round(dt, 4), "build_stats": tbl.build_stats(), "phases": ph,
                          "segments": tbl.segments(), "post_exchange_ms_per_1e9_rows": round((ph.get("kway_plan", 0) + ph.get("kway_merge", 0) +
                                                                                            ph.get("bucket", 0)) * 1e12 / sum(run_rows), 2)}), flush=True)
        tbl.close()


if __name__ == "__main__":
    main()
