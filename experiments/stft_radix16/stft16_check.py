"""radix 16.16.8 STFT experiment (SHZ_STFT16=1) against the shipped kernel: same batch, hashes compared, step timed.
python scripts/stft16_check.py  (spawns one child per mode: the switch is read once per process)"""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from shazam_amd import _ffi
    ctx = _ffi.Context(0)
    nc, n = 1000, 30 * 44100
    pcm = ctx.synth_pcm(1234, 0, nc, n, 0, 8000)
    off = np.arange(nc + 1, dtype=np.uint64) * n
    k, t1, ho, cnt = ctx.fingerprint_batch(pcm, off, pcm_device=True)
    ctx.sync()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.fingerprint_batch(pcm, off, pcm_device=True)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5
    kms = ctx.kernel_ms()
    np.save(os.path.join(ROOT, "gpurun_out", "stft16_%s_k.npy" % os.environ.get("SHZ_STFT16", "0")), k)
    np.save(os.path.join(ROOT, "gpurun_out", "stft16_%s_ho.npy" % os.environ.get("SHZ_STFT16", "0")), ho)
    print(len(k), hashlib.sha256(k.tobytes()).hexdigest()[:16], "step %.3f ms" % (dt * 1e3), {a: round(b[0] / 5, 3) for a, b in kms.items()})
    sys.exit(0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for mode in ("0", "1"):
    env = dict(os.environ, SHZ_STFT16=mode)
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=600)
    print("SHZ_STFT16=" + mode, r.stdout.strip(), r.stderr.strip()[-500:])
a = np.load(os.path.join(ROOT, "gpurun_out", "stft16_0_k.npy")); b = np.load(os.path.join(ROOT, "gpurun_out", "stft16_1_k.npy"))
print("hashes", len(a), len(b), "identical" if len(a) == len(b) and (a == b).all() else "different: %d common of %d" % (len(np.intersect1d(a, b)), len(np.union1d(a, b))))
