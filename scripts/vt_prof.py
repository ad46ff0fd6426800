"""phase times of vt_fold_kernel (needs libshz.so built with EXTRA=-DVT_PROFILE): python scripts/vt_prof.py [songs]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_db
from shazam_amd import _ffi
songs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = _ffi.Context(0)
tbl, build, _bufs = bench_db.build_table(ctx, songs, 30.0, 1000)
n_samples = 30 * 44100
out = (C.c_ulonglong * 16)()
_ffi.lib().shz_debug_vt_prof.argtypes = [C.c_void_p]
_ffi.lib().shz_debug_vt_prof(out)
r = bench_db.run_queries(ctx, tbl, songs, n_samples, 800, 10 * 44100, 10, 200)
_ffi.lib().shz_debug_vt_prof(out)
v = list(out)
names = ["tile start", "A votes", "C top-n", "D clear", "write", "-"]
tiles = v[7]
print("tiles", tiles, "ms/query", r["t_match"] * 1e3 / 800)
for n, x in zip(names, v[:6]):
    print(f"{n:24s} {x / max(tiles, 1):10.1f} ticks/tile (100 MHz wall clock: x10 ns)")
