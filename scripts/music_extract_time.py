"""Extraction of 1,000 x 30 s music-like clips (tonal material: more near-ties for peak_verify than noise): ms per step, the
kernels' shares and the verification counters.  python scripts/music_extract_time.py"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from shazam_amd import _ffi  # noqa: E402

ctx = _ffi.Context(0)
n, nc = 1323000, 1000
for name, mk in (("music", lambda: ctx.synth_corpus(1, 4321, 0, nc, n)), ("tonal+noise", lambda: ctx.synth_pcm(4321, 0, nc, n, 4000, 1500)),
                 ("noise", lambda: ctx.synth_pcm(4321, 0, nc, n, 0, 8000))):
    pcm = mk()
    off = np.arange(nc + 1, dtype=np.uint64) * n
    cap = nc * 700 * 40
    kb, tb = ctx.alloc(cap * 4), ctx.alloc(cap * 4)
    ctx.fingerprint_batch(pcm, off, pcm_device=True, out_key=kb, out_t1=tb, cap=cap)
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(5):
        _, _, _, cnt = ctx.fingerprint_batch(pcm, off, pcm_device=True, out_key=kb, out_t1=tb, cap=cap)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    k = {nm: round(v[0] / 5, 3) for nm, v in ctx.kernel_ms().items()}
    print(json.dumps({"corpus": name, "ms_per_step": round(ms, 3), "hashes": int(cnt), "kernels_ms": k, "stats": ctx.extract_stats()}))
    ctx.set_profiling(False)
    for b in (pcm, kb, tb):
        b.free()
