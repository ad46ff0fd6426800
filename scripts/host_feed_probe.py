#!/usr/bin/env python3
"""Host-fed fingerprint rate beside the link probes (what bench.py reports as pcie_inclusive), on its own."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from shazam_amd import _ffi
ctx = _ffi.Context(0)
nh, n_samples, FS = int(sys.argv[1]) if len(sys.argv) > 1 else 400, 30 * 44100, 44100
dev = ctx.synth_pcm(1234, 0, nh, n_samples, 0, 8000)
host = dev.download(np.int16, nh * n_samples)
dev.free()
off = np.arange(nh + 1, dtype=np.uint64) * n_samples
pin = ctx.host_array(len(host), np.int16)
pin[:] = host
o = {"clips": nh, "link": {"pinned_h2d_GBs": ctx.membw(3, 512 << 20, 4), "pageable_h2d_GBs": ctx.membw(4, 512 << 20, 4)}}
for name, arr in (("pageable", host), ("pinned", pin)):
    ctx.fingerprint_batch(arr, off)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.fingerprint_batch(arr, off); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    o[name] = {"audio_s_per_s": nh * 30 / t, "GBs": host.nbytes / t / 1e9, "frac_of_link": host.nbytes / t / 1e9 / o["link"][name + "_h2d_GBs"]}
o["upload"] = ctx.upload_stats()
print(json.dumps(o))
