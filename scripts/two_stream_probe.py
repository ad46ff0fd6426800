#!/usr/bin/env python3
"""Experiment: do two independent fingerprint pipelines (two contexts = two HIP streams, one host thread each) on ONE
GPU overlap -- stft_psd of one batch with peak_pick of the other -- or do they serialise?  Prints the aggregate
audio-seconds/s of 1 and of 2 concurrent pipelines (500 x 30 s clips per step each)."""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from shazam_amd import _ffi  # noqa: E402

FS, NCLIP, SEC, STEPS = 44100, 500, 30.0, 6
n = int(FS * SEC)


class Pipe:
    def __init__(self, dev=0):
        self.ctx = _ffi.Context(dev)
        self.pcm = self.ctx.alloc(NCLIP * n * 2)
        self.ctx.synth_pcm(1234, 0, NCLIP, n, 4000, 1500, out=self.pcm)
        self.off = np.arange(NCLIP + 1, dtype=np.uint64) * n
        self.cap = NCLIP * 700 * 40
        self.k, self.t = self.ctx.alloc(self.cap * 4), self.ctx.alloc(self.cap * 4)
        self.step()
        self.ctx.sync()

    def step(self):
        self.ctx.fingerprint_batch(self.pcm, self.off, fs=FS, pcm_device=True, out_key=self.k, out_t1=self.t, cap=self.cap)

    def run(self, steps):
        for _ in range(steps):
            self.step()
        self.ctx.sync()


def measure(pipes):
    ths = [threading.Thread(target=p.run, args=(STEPS,)) for p in pipes]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return len(pipes) * STEPS * NCLIP * SEC / dt, dt


a, b, c = Pipe(), Pipe(), Pipe()
for label, pipes in (("1 pipeline", [a]), ("2 pipelines", [a, b]), ("3 pipelines", [a, b, c]), ("2 pipelines", [b, c]), ("1 pipeline", [b])):
    v, dt = measure(pipes)
    print(f"{label}: {v / 1e6:.3f} M audio-s/s ({dt * 1e3 / STEPS:.2f} ms per round of {len(pipes)} x {NCLIP} clips)", flush=True)
