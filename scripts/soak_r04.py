"""Soak of round 4's new paths: host PCM through the upload pipeline (pageable and pinned), tables that hold their runs, the
gathered build with thread ranks sending runs on the way, a wratio other than the default -- repeated, watching device memory
(Context.mem_info) and that results stay what they were."""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
import shazam_amd as S  # noqa: E402
from shazam_amd import _ffi  # noqa: E402

ctx = S.get_context(0)
n, nc = 30 * 44100, 90
dev = ctx.synth_pcm(5, 0, nc, n, 3000, 1500)
host = dev.download(np.int16, nc * n)
dev.free()
pin = ctx.host_array(len(host), np.int16)
pin[:] = host
off = np.arange(nc + 1, dtype=np.uint64) * n
k0, t0_, ho0, _ = ctx.fingerprint_batch(host, off)
k75 = S.fingerprint_batch([host[:n]], wratio=0.75)[0]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
marks, t_start = [], time.time()


def gathered(world, gid):
    outs = [None] * world

    def go(r):
        c = _ffi.Context(0)
        comm = _ffi.Comm.local(c, gid, r, world)
        tbl = S.Table(c)
        tbl.set_segment_rows(200000)
        tbl.reserve(0, 0, gather=True)
        lo, hi = r * nc // world, (r + 1) * nc // world
        for i, cidx in enumerate(range(lo, hi)):
            a, b = int(ho0[cidx]), int(ho0[cidx + 1])
            tbl.insert(k0[a:b], np.full(b - a, cidx + 1, np.uint32), t0_[a:b])
            if i % 10 == 9:
                tbl.exchange_run(comm)
        tbl.allgather(comm)
        outs[r] = tbl.rows()[0]
        tbl.close(); comm.close(); c.close()

    ths = [threading.Thread(target=go, args=(r,)) for r in range(world)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    return outs


want_rows = None
for it in range(iters):
    for arr in (host, pin):
        k, t1, ho, _ = ctx.fingerprint_batch(arr, off)
        assert np.array_equal(k, k0) and np.array_equal(ho, ho0)
    assert np.array_equal(S.fingerprint_batch([host[:n]], wratio=0.75)[0], k75)
    tbl = S.Table(ctx)
    tbl.reserve(len(k0) + 1000, len(k0) // 3 + 1000, gather=True)
    for c0 in range(0, nc, 30):
        a, b = int(ho0[c0]), int(ho0[c0 + 30])
        tbl.insert_clips(k0[a:b], t0_[a:b], ho0[c0:c0 + 31] - ho0[c0], sid0=c0 + 1)
        tbl.seal_run()
    tbl.finalize()
    q = slice(int(ho0[5]), int(ho0[6]))
    res = tbl.match(k0[q], t0_[q], np.array([0, q.stop - q.start], np.uint64), 2)
    assert int(res["sid"][0, 0]) == 6 and int(res["delta"][0, 0]) == 0
    rows = tbl.rows()[0]
    tbl.close()
    got = gathered(3, 900000 + it)
    want_rows = want_rows or rows
    assert rows == want_rows and all(g == want_rows for g in got), (rows, got, want_rows)
    if it % 5 == 0:
        free_b, total_b = ctx.mem_info()
        marks.append((it, round((total_b - free_b) / 2**20, 1), round(time.time() - t_start, 1)))
print("iteration, device memory in use MiB, seconds:", marks)
drift = marks[-1][1] - marks[1][1] if len(marks) > 2 else 0.0
print("drift after warm-up: %.1f MiB; upload stats %s" % (drift, ctx.upload_stats()))
assert drift < 64, "device memory keeps growing"
