"""Soak: repeat build / match / sharded match / incremental finalize and watch device memory (hipMemGetInfo via
rocm-smi is not needed: the library reports allocation failures; here we just look for drift in free memory)."""
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import shazam_amd as S  # noqa: E402
from shazam_amd.shard import ShardedTable  # noqa: E402


def used_mb():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        p = line.split(",")
        if len(p) >= 3 and p[0].startswith("card"):
            return int(p[2]) / 2**20
    return -1.0


ctx = S.get_context(0)
n = 10 * 44100
rng = np.random.default_rng(0)
t0 = time.time()
marks = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    pcm = ctx.synth_pcm(1000 + it, 0, 64, n, 3000, 1500)
    k, t1, ho, _ = ctx.fingerprint_batch(pcm, np.arange(65, dtype=np.uint64) * n, pcm_device=True)
    pcm.free()
    tbl, sh = S.Table(ctx), ShardedTable(ctx, nshards=3)
    for t in (tbl, sh):
        t.insert_clips(k, t1, ho, 1)
        t.finalize()
    q = slice(int(ho[5]), int(ho[6]))
    ra = tbl.match(k[q], t1[q], np.array([0, q.stop - q.start], np.uint64), 2)
    rb = sh.match(k[q], t1[q], np.array([0, q.stop - q.start], np.uint64), 2)
    assert int(ra["sid"][0, 0]) == 6 == int(rb["sid"][0, 0]) and int(ra["delta"][0, 0]) == 0
    tbl.insert(k[:1000], np.full(1000, 99, np.uint32), t1[:1000])
    tbl.finalize()
    tbl.close()
    sh.close()
    if it % 20 == 0:
        marks.append((it, round(used_mb(), 1), round(time.time() - t0, 1)))
print("iteration, VRAM used MiB, seconds:", marks)
drift = marks[-1][1] - marks[1][1] if len(marks) > 2 else 0.0
print("drift after warm-up: %.1f MiB" % drift)
assert drift < 64, "device memory keeps growing"
