"""Time the synthetic-PCM generators on the device (bench infrastructure: 1M x 30 s tracks are synthesised inside the match_1M
build clock).  python scripts/synth_time.py"""
import json
import sys
import time

sys.path.insert(0, ".")
from shazam_amd import _ffi  # noqa: E402

ctx = _ffi.Context(0)
n, nc = 1323000, 1000
buf = ctx.alloc(nc * n * 2)
out = {}
for name, fn in (("tonal_4000_1500", lambda: ctx.synth_pcm(99, 0, nc, n, 4000, 1500, out=buf)),
                 ("noise_8000", lambda: ctx.synth_pcm(99, 0, nc, n, 0, 8000, out=buf)),
                 ("music", lambda: ctx.synth_corpus(1, 99, 0, nc, n, out=buf))):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    ctx.sync()
    out[name] = {"ms_per_1000x30s": (time.perf_counter() - t0) / 5 * 1e3}
print(json.dumps(out))
