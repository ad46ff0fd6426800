#!/usr/bin/env python3
"""The db_build_scaling leg of bench.py on its own (BASELINE configs[2]: 100,000 x 3 min tracks into one table).

    python scripts/build_scaling.py --songs 100000 --seconds 180                 # one GPU, what bench.py --gpus 1 reports
    python scripts/build_scaling.py --songs 4000 --seconds 180 --local-ranks 4   # the N > 1 code path with thread ranks
                                                                                 # on ONE GPU (logic rehearsal, not a scaling number:
                                                                                 # the ranks share the GPU and every rank holds the whole table)
Prints one JSON line per run.
"""
import argparse
import json
import os
import sys
import threading
import time
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run_local(songs, seconds, world):
    """bench.db_build_scaling with `world` thread ranks on one GPU (in-process transport); returns every rank's result."""
    import bench
    from shazam_amd import _ffi
    args = types.SimpleNamespace(scaling_songs=songs, scaling_seconds=seconds)
    bar = threading.Barrier(world)
    outs, errs = [None] * world, [None] * world
    gid = 31337 + int(time.time() * 1e3) % 100000

    class Dist:   # what db_build_scaling needs of a process group, between threads -- WITHOUT torch: a process that has loaded
        # libshz.so (the system's HIP runtime) must not load torch's bundled HIP runtime and RCCL afterwards
        vals = [None] * world

        def __init__(self, r):
            self.r = r

        def barrier(self):
            bar.wait()

        def max_floats(self, v):
            Dist.vals[self.r] = list(v)
            bar.wait()
            m = [max(x[i] for x in Dist.vals) for i in range(len(v))]
            bar.wait()
            return m

    def go(r):
        try:
            ctx = _ffi.Context(0)
            comm = _ffi.Comm.local(ctx, gid, r, world)
            outs[r] = bench.db_build_scaling(args, ctx, Dist(r), comm, r, world)
            comm.close()
            ctx.close()
        except BaseException as e:  # noqa: BLE001
            errs[r] = e
            bar.abort()

    ths = [threading.Thread(target=go, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for e in errs:
        if e is not None:
            raise e
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=100000)
    ap.add_argument("--seconds", type=float, default=180.0)
    ap.add_argument("--local-ranks", type=int, default=1)
    a = ap.parse_args()
    import bench
    from shazam_amd import _ffi
    args = types.SimpleNamespace(scaling_songs=a.songs, scaling_seconds=a.seconds)
    if a.local_ranks <= 1:
        ctx = _ffi.Context(0)
        t0 = time.perf_counter()
        o = bench.db_build_scaling(args, ctx, None, None, 0, 1)
        o["wall_incl_setup_s"] = time.perf_counter() - t0
        print(json.dumps(o))
        return
    outs = run_local(a.songs, a.seconds, a.local_ranks)
    o = outs[0]
    o["transport"] = f"{a.local_ranks} thread ranks on one GPU (in-process transport)"
    o["rows_all_ranks"] = [x["rows"] for x in outs]
    print(json.dumps(o))


if __name__ == "__main__":
    main()
