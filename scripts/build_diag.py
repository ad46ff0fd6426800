#!/usr/bin/env python3
"""Where the seconds of the single-GPU database build go: device allocation cost on this box,
per-finalize phase seconds of the 1M x 30 s build, and the one-GPU proxy of the per-rank post-exchange cost
(8 sorted runs through shz_table_finalize_runs).  Prints JSON lines; run on the GPU box.

    python scripts/build_diag.py [--songs 1000000] [--every 100000] [--proxy-songs 96000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def alloc_cost(ctx):
    out = {}
    for gb in (1, 4, 8, 16, 32):
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            b = ctx.alloc(gb << 30)
            t1 = time.perf_counter()
            b.free()
            t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1))
        out[f"{gb}GB"] = {"alloc_s": [round(a, 4) for a, _ in ts], "free_s": [round(f, 4) for _, f in ts]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=1000000)
    ap.add_argument("--every", type=int, default=100000)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--skip-alloc", action="store_true")
    ap.add_argument("--proxy-songs", type=int, default=96000)
    ap.add_argument("--proxy-runs", type=int, default=8)
    ap.add_argument("--skip-build", action="store_true")
    a = ap.parse_args()
    from shazam_amd import _ffi, Table
    import bench_db
    ctx = _ffi.Context(0)
    if not a.skip_alloc:
        print(json.dumps({"alloc_cost": alloc_cost(ctx)}), flush=True)

    if a.proxy_songs:
        # proxy of the per-rank post-exchange work: the rows of `runs` blocks of songs (what `runs` ranks would have
        # staged), fingerprinted here, -> finalize_runs
        runs, chunk = a.proxy_runs, 1000
        per = a.proxy_songs // runs // chunk * chunk
        n_samples = 30 * 44100
        frames = int(_ffi.lib().shz_frame_count(n_samples))
        cap = chunk * frames * 24 + 1024
        kbuf, tbuf, pcm = ctx.alloc(cap * 4), ctx.alloc(cap * 4), ctx.alloc(chunk * n_samples * 2)
        off = np.arange(chunk + 1, dtype=np.uint64) * n_samples
        tbl = Table(ctx)
        run_rows = []
        for r in range(runs):
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
        t_fin = time.perf_counter() - t0
        print(json.dumps({"proxy_finalize_runs": {"rows": int(sum(run_rows)), "runs": runs, "seconds": t_fin,
                                                  "build_stats": tbl.build_stats(), "rows_after": tbl.rows()[0],
                                                  "segments": tbl.segments()}}), flush=True)
        tbl.close()
        for b_ in (kbuf, tbuf, pcm):
            b_.free()
        ctx.release_workspace()

    if not a.skip_build:
        tbl_box = {}
        recs = []
        t_last = [time.perf_counter()]

        def progress(done):
            # called after every intermediate finalize
            pass

        # build with per-finalize phase records: wrap Table.finalize
        orig = Table.finalize

        def timed_finalize(self):
            t0 = time.perf_counter()
            orig(self)
            ctx.sync()
            dt = time.perf_counter() - t0
            ph = self.phase_stats(reset=True)
            recs.append({"finalize_s": round(dt, 4), "rows": self.rows()[0], "segments": self.segments(),
                         "phases": {k: round(v, 4) for k, v in ph.items() if v > 5e-4}})
            print(json.dumps({"finalize": recs[-1]}), flush=True)

        orig_seal = Table.seal_run

        def timed_seal(self):
            t0 = time.perf_counter()
            orig_seal(self)
            ctx.sync()
            dt = time.perf_counter() - t0
            ph = self.phase_stats(reset=True)
            print(json.dumps({"seal_run": {"seal_s": round(dt, 4), "rows": self.rows(), "segments": self.segments(),
                                           "phases": {k: round(v, 4) for k, v in ph.items() if v > 5e-4}}}), flush=True)

        Table.finalize = timed_finalize
        Table.seal_run = timed_seal
        tbl, build, bufs = bench_db.build_table(ctx, a.songs, a.seconds, 1000, 4000, 1500, finalize_every=a.every)
        Table.finalize = orig
        Table.seal_run = orig_seal
        print(json.dumps({"build": build}), flush=True)
        tbl.close()


if __name__ == "__main__":
    main()
