"""Randomised campaign for the extraction pass (not part of the suite): the default path (fp32 staging, ties handed to the
reference's arithmetic) against fp64 staging of everything (numpy's arithmetic throughout) -- peaks, hashes and offsets must be
the same arrays.  Inputs lean towards what makes ties: tones, clicks, tiny amplitudes (a few counts: many equal powers),
repeated material, clipping, silence, DC, mixtures.   python tests/campaign_extract_fuzz.py [seconds] [seed0]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")   # (run from the repo root; lives under tests/ because it draws its inputs from the oracle's generators)
import shazam_amd as S  # noqa: E402
from oracle import synth  # noqa: E402

ctx = S.get_context(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget


def clip(rng, n):
    kind = int(rng.integers(0, 9))
    amp = int(10 ** rng.uniform(0.0, 4.5))
    if kind == 0:
        return synth.synth_clip(int(rng.integers(1 << 30)), int(rng.integers(1000)), n, 0, max(amp, 1))
    if kind == 1:
        return synth.synth_clip(int(rng.integers(1 << 30)), int(rng.integers(1000)), n, min(amp, 8000), max(amp // 8, 0))
    if kind == 2:
        return synth.lut_tone(n, float(rng.uniform(50, 8000)), amp=min(max(amp, 10), 30000))
    if kind == 3:   # click train, random period
        x = np.zeros(n, np.int16)
        x[:: int(rng.integers(500, 9000))] = min(max(amp, 50), 30000)
        return x
    if kind == 4:   # a few counts of noise: many cells with exactly equal power
        return rng.integers(-2, 3, n).astype(np.int16)
    if kind == 5:   # DC + tiny noise
        return (rng.integers(-1, 2, n) + int(rng.integers(-20000, 20000))).astype(np.int16)
    if kind == 6:   # repeated material
        x = synth.synth_clip(int(rng.integers(1 << 30)), 7, max(n // 4, 1), 2000, 300)
        return np.tile(x, 5)[:n]
    if kind == 7:   # clipped
        x = synth.synth_clip(int(rng.integers(1 << 30)), 3, n, 0, 30000).astype(np.int32) * 3
        return np.clip(x, -32768, 32767).astype(np.int16)
    x = np.zeros(n, np.int16)      # sparse clicks
    for _ in range(int(rng.integers(1, 12))):
        x[int(rng.integers(0, n))] = int(rng.integers(-32768, 32767))
    return x


n_cases = n_clips = n_hashes = n_f64_clips = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    nc = int(rng.integers(1, 16))
    clips = [np.ascontiguousarray(clip(rng, int(rng.choice([rng.integers(1, 4096), rng.integers(4096, 60000), rng.integers(60000, 500000)]))), np.int16)
             for _ in range(nc)]
    amp_min = float(rng.choice([10.0, 10.0, 0.0, rng.uniform(0, 60)]))
    off = np.concatenate([[0], np.cumsum([len(x) for x in clips])]).astype(np.uint64)
    x = np.concatenate(clips)
    ctx.set_stage_f64(False)
    a = ctx.fingerprint_batch(x, off, amp_min=amp_min)
    pa = ctx.peaks(x, off, amp_min=amp_min)
    st = ctx.extract_stats() if hasattr(ctx, "extract_stats") else {}
    ctx.set_stage_f64(True)
    try:
        b = ctx.fingerprint_batch(x, off, amp_min=amp_min)
        pb = ctx.peaks(x, off, amp_min=amp_min)
    finally:
        ctx.set_stage_f64(False)
    ok = all(np.array_equal(u, v) for u, v in zip(a[:3], b[:3])) and all(np.array_equal(u, v) for u, v in zip(pa, pb))
    if not ok:
        print("MISMATCH seed", seed, "clips", nc, "amp_min", amp_min, [len(c) for c in clips], flush=True)
        sys.exit(1)
    n_cases += 1
    n_clips += nc
    n_hashes += int(a[3])
    seed += 1
print("extract fuzz:", n_cases, "batches,", n_clips, "clips,", n_hashes, "hashes, next seed", seed, "-- fp32 staging + verification == fp64 staging (the reference's arithmetic) everywhere")
