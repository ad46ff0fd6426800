#!/usr/bin/env python3
"""Per-clip fp64 fallback: a batch of 1,000 x 30 s noise clips against the same batch with one clip
replaced by a click per hop (hundreds of tied cells per window).  Prints the step times, the fallback counters and whether the
hashes equal those of fp64 staging of everything."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shazam_amd import _ffi  # noqa: E402

ctx = _ffi.Context(0)
nc, n = 1000, 30 * 44100
pcm = ctx.synth_pcm(1234, 0, nc, n, 0, 8000)
off = np.arange(nc + 1, dtype=np.uint64) * n
cap = nc * 644 * 24 + (4 << 20)
kb, tb = ctx.alloc(cap * 4), ctx.alloc(cap * 4)


def step_ms(reps=10):
    for _ in range(2):
        ctx.fingerprint_batch(pcm, off, pcm_device=True, out_key=kb, out_t1=tb, cap=cap)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = ctx.fingerprint_batch(pcm, off, pcm_device=True, out_key=kb, out_t1=tb, cap=cap)
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3, r


ms_noise, r0 = step_ms()
click = np.zeros(n, np.int16)
click[1024::2048] = 20000
pcm.upload(click, offset_bytes=500 * n * 2)          # clip 500 becomes the click train
s0 = ctx.extract_stats()
ms_mixed, r1 = step_ms()
s1 = ctx.extract_stats()
k1 = kb.download(np.uint32, r1[3]).copy()
ctx.set_stage_f64(True)
r2 = ctx.fingerprint_batch(pcm, off, pcm_device=True, out_key=kb, out_t1=tb, cap=cap)
ctx.set_stage_f64(False)
same = r2[3] == r1[3] and np.array_equal(kb.download(np.uint32, r2[3]), k1) and np.array_equal(r1[2], r2[2])
print(json.dumps({"ms_per_step_all_noise": ms_noise, "ms_per_step_one_click_clip": ms_mixed, "ratio": ms_mixed / ms_noise,
                  "hashes_all_noise": int(r0[3]), "hashes_with_click_clip": int(r1[3]),
                  "f64_clips_per_step": (s1["f64_clips"] - s0["f64_clips"]) / 12, "f64_passes": s1["f64_passes"] - s0["f64_passes"],
                  "f64_clip_frames_per_step": (s1["f64_clip_frames"] - s0["f64_clip_frames"]) / 12,
                  "hashes_equal_fp64_staging_of_everything": bool(same)}))
