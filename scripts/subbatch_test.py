import sys, time, numpy as np
sys.path.insert(0, '.')
from shazam_amd import _ffi
ctx = _ffi.Context(0)
n, nc = 1323000, 1000
pcm = ctx.synth_pcm(1234, 0, nc, n, 0, 8000)
off = np.arange(nc + 1, dtype=np.uint64) * n
cap = nc * 644 * 24
kb, tb = ctx.alloc(cap * 4), ctx.alloc(cap * 4)
for frames in (0, 330000, 165000, 83000):
    ctx.set_workspace_limit(frames * 2056 * 8 if frames else 0)
    for _ in range(2):
        ctx.fingerprint_batch(pcm, off, pcm_device=True, out_key=kb, out_t1=tb, cap=cap)
    ctx.sync(); ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.fingerprint_batch(pcm, off, pcm_device=True, out_key=kb, out_t1=tb, cap=cap)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    k = ctx.kernel_ms(); ctx.set_profiling(False)
    print(frames, "frames/sub-batch: ms/step", round(dt, 3), {a: round(b[0] / 5, 3) for a, b in k.items()})
