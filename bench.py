#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X fingerprint hot path.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): audio-seconds fingerprinted per second, 44.1 kHz mono int16; second half: query match
latency against a 1M-song table (the `match_1M` extra, N = 1 only).
Workload (BASELINE configs[1]): per GPU a batch of 1,000 synthetic 30 s clips resident in HBM,
fingerprint-only: PCM -> STFT -> dB -> 21x21 peaks -> pair hashes (key32, t1) compacted on
the device.  One "step" = one pass over that batch.  N > 1 is weak scaling: every rank fingerprints its own
1,000 clips; `value` = all ranks' audio seconds / max-over-ranks wall time of the K timed steps.  Launched by
torch.distributed.run (RANK / WORLD_SIZE in the environment) or, when --gpus N > 1 is given without that
environment, by this script itself: N child processes, one per GPU, started before any HIP call.

Beside the headline line the JSON carries
  roofline      dominant kernel: ALGORITHMIC bytes (4,096 B PCM + 8 B per hash, per frame) per launch over
                its HIP-event duration vs the HBM peak (`achieved`, `frac`); the bytes the kernel itself moves per
                launch (`staged_*`), the PMC-measured traffic (`traffic`) and the whole step's figure beside it
  cpu_baseline  the reference's numpy/scipy/mlab call sequence (oracle/thirdparty_ref.py)
                timed on this box's host cores over a bounded sample (rank 0, N = 1 only)
  db_build      fingerprints -> HBM table (with the RCCL all-gather when N > 1), untimed extra
  match         batched recognise against that table, untimed extra
  single_query  one 5 s query at a time through recognize(): p50 / p99 latency
  match_1M      1M x 30 s tracks in one HBM table (1.1e10 rows): per-batch latency at batch sizes 1 and 200
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 44100
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the measured stream rates of the box
#                        (copy ~5.0, read ~6.2, write ~4.0 TB/s) go into roofline.measured_stream_GBs
PCM_BYTES_PER_FRAME = 4096               # 2,048 new int16 samples per frame
# bytes each kernel itself moves per frame with fp32 staging (the default; fp64 staging doubles the 2049-bin rows)
STAGED_BYTES_PER_FRAME = {"stft_psd": 4096 + 2049 * 4, "peak_pick": 2049 * 4 + 288}
FP64_VALU_PEAK_TFLOPS = 78.6                    # MI355X vector fp64 peak (MI355X_MICROARCH.md)
FFT_FLOP_PER_FRAME = 2.5 * 4096 * 12            # 2.5 N log2 N of the real 4096-point transform
PMC_KERNELS = {"stft_psd": "stft_psd_kernel<float>", "peak_pick": "peak_pick32_kernel<2, 4>"}


def newest_profile(suffix):
    """The newest committed profiles/r<NN><x>_<suffix> (names sort by round, then by letter)."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", f"r*_{suffix}")):
        m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(f))
        if m:
            key = (int(m.group(1)), m.group(2))
            if best is None or key > best[0]:
                best = (key, f)
    return best[1] if best else None


def pmc_kernel(prof, kernel):
    for name, rec in prof["kernels"].items():
        if name.startswith(PMC_KERNELS[kernel].split("<")[0]):
            return rec
    return None


def pmc_traffic(kernel, frames_per_launch):
    """(HBM GB per launch of `kernel`, profile file) from the newest committed rocprofv3 PMC passes (profiles/), scaled to
    this run's frames per launch; (None, None) when no profile of that kernel is committed."""
    try:
        f = newest_profile("pmc_traffic.json")
        prof = json.load(open(f))
        rec = pmc_kernel(prof, kernel)
        return rec["hbm_bytes_corrected"] / 1e9 * frames_per_launch / prof.get("frames_per_launch", 644000), os.path.basename(f)
    except Exception:
        return None, None


def pmc_valu_busy(kernel, waves_per_simd=3.0):
    """(fraction of issue cycles in which a SIMD's VALU is busy, profile file) from the newest committed SQ counter passes:
    frac_valu = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES is the share of a WAVE's cycles spent issuing VALU work; stft_psd runs
    3 workgroups x 4 waves per CU = 3 waves per SIMD (__launch_bounds__(256, 3)), so the SIMD is busy 3 x that."""
    try:
        f = newest_profile("pmc_counters.json")
        rec = pmc_kernel(json.load(open(f)), kernel)
        return min(1.0, rec["frac_valu"] * waves_per_simd), os.path.basename(f)
    except Exception:
        return None, None


def cpu_worker(args):
    seed, clip, n = args
    from oracle import synth, thirdparty_ref
    x = synth.synth_clip(seed, clip, n, 0, 8000)
    t0 = time.perf_counter()
    h = thirdparty_ref.fingerprint(x, Fs=FS)
    return time.perf_counter() - t0, len(h)


def _cpu_budget():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    return aff, quota


def cpu_baseline(n_samples, budget_s=15.0):
    """Reference call sequence (mlab.specgram + scipy.ndimage + hashlib) over a Pool of the host cores this process
    may use, like __init__.py:335-357.  Bounded sample: pool x k clips.  Library threads are pinned to one per
    worker so that `cores` workers mean `cores` busy threads."""
    import multiprocessing as mp
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[v] = "1"   # inherited by the forked workers; numpy's FFT and scipy.ndimage are single-threaded anyway
    aff, quota = _cpu_budget()
    pool_n = max(1, min(aff, int(quota) if quota else aff))
    t1 = cpu_worker((1234, 0, n_samples))[0]          # single-core time per clip (also warms imports)
    per_proc = max(1, int(budget_s / max(t1 * 2.5, 1e-3)))
    per_proc = min(per_proc, 4)
    clips = pool_n * per_proc
    ctx = mp.get_context("fork")
    with ctx.Pool(pool_n) as pool:
        pool.map(cpu_worker, [(1234, c, n_samples // 8) for c in range(pool_n)])  # warm-up
        t0 = time.perf_counter()
        res = pool.map(cpu_worker, [(1234, c, n_samples) for c in range(clips)], chunksize=1)
        wall = time.perf_counter() - t0
    audio_s = clips * n_samples / FS
    busy = float(sum(r[0] for r in res))   # CPU-seconds the workers spent inside fingerprint()
    return {"value": audio_s / wall, "unit": "audio-s/s", "cores": pool_n, "kind": "port",
            "affinity_cpus": aff, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count(),
            "effective_parallelism": busy / wall,
            "single_core_value": (n_samples / FS) / t1,
            "sample": f"{clips} synthetic {n_samples / FS:.0f} s clips (same generator/seed as the GPU run) through "
                      f"oracle/thirdparty_ref.py (mlab.specgram + scipy.ndimage + hashlib, the reference's call sites) "
                      f"on a multiprocessing.Pool({pool_n}) = min(affinity {aff}, cgroup quota {quota}), one library "
                      f"thread per worker; synth time excluded; wall {wall:.1f} s; effective_parallelism = sum of the "
                      f"workers' in-call seconds / wall",
            "hashes_per_clip": float(np.mean([r[1] for r in res]))}


def match_cpu_baseline(tbl, k, t1, ho, gpu_res, songs, budget_s=12.0, topn=2):
    """CPU baseline of the match half of the metric ("p50/p99 ms per query with an in-memory dict table
    standing in for MySQL"): the reference's return_matches + align_matches (recognizer.py:222-338, restated in
    oracle/cpu_ref.py and pinned to the reference's goldens) over the SAME table and the SAME queries, one core -- the
    reference's matcher is a single-threaded Python loop over the rows MySQL returns.  The table is this step's rows exported
    from the GPU and indexed by a dict hash -> row range (the B-tree on `hash`, mysql_database.py:46-59).  Beside it a numpy
    sorted-array matcher (searchsorted + unique) as the fastest thing numpy does on one core.  Bounded: queries until the
    budget is spent."""
    from oracle import cpu_ref as O
    t0 = time.perf_counter()
    key, sid, off = tbl.export()
    if len(key) > 1 and not np.all(key[1:] >= key[:-1]):   # several segments: one sorted table for the host index
        order = np.lexsort((off, sid, key))
        key, sid, off = key[order], sid[order], off[order]
    uk, start, cnt = np.unique(key, return_index=True, return_counts=True)
    index = dict(zip(uk.tolist(), zip(start.tolist(), (start + cnt).tolist())))
    sid_l, off_l = sid.tolist(), off.tolist()
    per_song = np.bincount(sid, minlength=songs + 2)
    t_index = time.perf_counter() - t0

    class ArrayDB:   # the duck-typed surface the oracle's matcher uses
        def select_multiple(self, values):
            for h in values:
                se = index.get(h)
                if se:
                    for i in range(se[0], se[1]):
                        yield h, sid_l[i], off_l[i]

        def get_song_by_id(self, s_):
            return {"song_name": f"track{s_}", "total_hashes": int(per_song[s_]), "file_sha1": "0" * 40}

    db = ArrayDB()

    def numpy_match(qk, qo):
        q = np.unique((qk.astype(np.uint64) << np.uint64(32)) | qo.astype(np.uint64))   # set(hashes)
        qk, qo = (q >> np.uint64(32)).astype(np.uint32), (q & np.uint64(0xFFFFFFFF)).astype(np.int64)
        lo, hi = np.searchsorted(key, qk, "left"), np.searchsorted(key, qk, "right")
        n = hi - lo
        rows = np.repeat(lo, n) + (np.arange(n.sum()) - np.repeat(np.cumsum(n) - n, n))
        s_ = sid[rows].astype(np.int64)
        d = off[rows].astype(np.int64) - np.repeat(qo, n)
        pairs, c = np.unique((s_ << 32) | (d + (1 << 20)), return_counts=True)
        ps = pairs >> 32
        first = np.r_[True, ps[1:] != ps[:-1]]
        grp = np.cumsum(first) - 1
        best = np.zeros(grp[-1] + 1 if len(grp) else 0, np.int64)
        np.maximum.at(best, grp, c)
        order = np.argsort(-best, kind="stable")[:topn]
        return [int(ps[first][i]) for i in order]

    lat = {"dict": [], "numpy": []}
    agree = {"dict": 0, "numpy": 0}
    nq_done = 0
    t_begin = time.perf_counter()
    for q in range(len(ho) - 1):
        qk, qo = k[int(ho[q]):int(ho[q + 1])], t1[int(ho[q]):int(ho[q + 1])]
        hashes = set(zip(qk.tolist(), qo.tolist()))          # recognizer.py:378-382
        t0 = time.perf_counter()
        matches, dedup = O.return_matches(hashes, db)
        res = O.align_matches(matches, dedup, len(hashes), db, topn)
        t1_ = time.perf_counter()
        top_np = numpy_match(qk, qo)
        t2 = time.perf_counter()
        lat["dict"].append(t1_ - t0)
        lat["numpy"].append(t2 - t1_)
        want = int(gpu_res["sid"][q, 0]) if gpu_res["nres"][q] else None
        agree["dict"] += (res[0]["song_id"] if res else None) == want
        agree["numpy"] += (top_np[0] if top_np else None) == want
        nq_done += 1
        if nq_done >= 20 and time.perf_counter() - t_begin > budget_s:
            break
    o = {"kind": "port", "cores": 1, "queries": nq_done, "table_rows": int(len(key)), "songs": int(songs), "host_index_build_s": t_index,
         "sample": f"the first {nq_done} queries of the GPU's own batch against the same {len(key)} rows, one core; dict-table = the "
                   "oracle's return_matches + align_matches (the reference's Python loop) over a dict hash -> rows; numpy_sorted = "
                   "searchsorted + unique on the sorted columns"}
    for name in ("dict", "numpy"):
        v = np.array(lat[name]) * 1e3
        o[f"{'dict_table' if name == 'dict' else 'numpy_sorted'}"] = {"p50_ms": float(np.median(v)), "p99_ms": float(np.percentile(v, 99)),
                                                                     "mean_ms": float(v.mean()), "top1_equals_gpu": int(agree[name])}
    return o


def extras(a, ctx, dist, rank, world, nc, n_samples, kbuf, tbuf, hash_off, elapsed, pcm, out):
    from shazam_amd import _ffi

    def barrier():
        if dist:
            dist.barrier()

    # the measured streaming bandwidth of this GPU beside the vendor peak the fraction is quoted on
    try:
        bw = {m: ctx.membw(i, 2 << 30, 5) for i, m in enumerate(("copy", "read", "write"))}
        rf = out["roofline"]
        rf["measured_stream_GBs"] = bw
        rf["staged_frac_of_measured_copy"] = rf["staged_achieved"] / bw["copy"] if bw["copy"] > 0 else None
    except Exception as e:  # noqa: BLE001
        out["roofline"]["measured_stream_error"] = repr(e)

    from shazam_amd import Table
    tbl = Table(ctx)
    comm = None
    if world > 1 or a.force_dist:
        ids = [_ffi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = _ffi.Comm(ctx, ids[0], rank, world)
        comm.barrier()
    barrier()
    ctx.sync()
    t0 = time.perf_counter()
    tbl.insert_clips(kbuf, tbuf, hash_off, sid0=1 + rank * nc, device=True)
    t_ins = time.perf_counter() - t0
    recv = 0
    if comm:
        recv = tbl.allgather(comm)      # RCCL all-gather of every rank's rows + finalize
    else:
        tbl.finalize()
    ctx.sync()
    barrier()
    t_build = time.perf_counter() - t0
    rows, _ = tbl.rows()
    out["db_build"] = {"rows": int(rows), "songs": world * nc, "seconds_table_only": t_build,
                       "seconds_incl_fingerprint": t_build + elapsed / a.steps,
                       "songs_per_second_incl_fingerprint": world * nc / (t_build + elapsed / a.steps),
                       "allgather_bytes_received": int(recv),
                       "collective": ("exchange rounds over RCCL: every rank's sorted packed runs (8 B/row, pieces <= 1 GB, grouped "
                                      "ncclSend / ncclRecv on the communicator's own stream) to every peer, then one k-way merge of all runs") if comm else None,
                       "build_stats_s": tbl.build_stats() if comm else None}
    # STRONG scaling of the database build (north star: >= 6x at 8 GPUs): ONE fixed corpus, the same at every N, split in
    # contiguous blocks (ingest.shard_tracks), fingerprint -> RCCL all-gather of the sorted runs -> k-way merge -> table.
    # N = 1 runs the same code with one run and no exchange.  The reference's analogue is the file-level Pool of
    # fingerprint_directory (__init__.py:335-357).
    # (one GPU: run after the 1M-song table below -- that one reserves 250 GB beside its first chunks, and on memory nothing in
    # this process has freed yet the reservation costs nothing; the scaling build reserves outside its clock either way)
    if a.scaling_songs > 0 and (world > 1 or a.force_dist):
        try:
            out["db_build_scaling"] = db_build_scaling(a, ctx, dist, comm, rank, world)
        except Exception as e:  # noqa: BLE001
            out["db_build_scaling"] = {"error": repr(e)}
    # batched recognise: hop-aligned 5 s crops of this rank's own tracks (clean; SNR mixing is a test-side path)
    nq = min(a.queries, nc)
    qn = 220500
    rng = np.random.default_rng(7 + rank)
    starts = rng.integers(0, (n_samples - qn) // 2048, nq) * 2048
    qpcm = ctx.alloc(nq * qn * 2)
    for q in range(nq):   # device-side crop: generate samples [start, start+qn) of clip q
        _ffi.lib().shz_synth_pcm(ctx.h, 1234, rank * nc + q, 1, qn, 0, 8000, int(starts[q]), _ffi.vp(qpcm.ptr + q * qn * 2))
    qoff = np.arange(nq + 1, dtype=np.uint64) * qn
    ctx.sync()
    t0 = time.perf_counter()
    k, t1, ho, _ = ctx.fingerprint_batch(qpcm, qoff, pcm_device=True)
    t_fp = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = tbl.match(k, t1, ho, 2)
    t_match = time.perf_counter() - t0
    st = tbl.match_stats()
    correct = int(np.sum((res["nres"] > 0) & (res["sid"][:, 0] == 1 + rank * nc + np.arange(nq)) &
                         (res["delta"][:, 0] == starts // 2048)))
    out["match"] = {"queries": nq, "query_seconds": 5.0, "db_rows": int(rows), "ms_per_query_batched": t_match / nq * 1e3,
                    "qps": nq / t_match, "fingerprint_ms_per_query": t_fp / nq * 1e3, "top1_correct": correct,
                    "rows_scanned": st["rows_scanned"], "pairs": st["pairs"],
                    "alg_GBs": (8 * st["rows_scanned"] + 16 * st["distinct_keys"]) / t_match / 1e9}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            cb = match_cpu_baseline(tbl, k, t1, ho, res, world * nc)
            cb["gpu_ms_per_query_batched"] = out["match"]["ms_per_query_batched"]
            cb["gpu_over_cpu_dict_table"] = cb["dict_table"]["mean_ms"] / out["match"]["ms_per_query_batched"]
            out["match"]["cpu_baseline"] = cb
        except Exception as e:  # noqa: BLE001
            out["match"]["cpu_baseline"] = {"error": repr(e)}
    qpcm.free()
    # the same hot path fed from HOST memory -- what every caller of the reference hands over (__init__.py:248-268;
    # recognizer.py:377-382): PCIe-inclusive rates, never the headline value.  The call cuts the batch into chunks of whole
    # clips and uploads chunk i + 1 beside the kernels of chunk i (shz_extract.hip: extract_streamed), so its ceiling is the
    # link: the probes beside it say what the link gives from pinned and from pageable memory on this box.
    nh = min(400, nc)
    host_pcm = pcm.download(np.int16, nh * n_samples)
    hoff = np.arange(nh + 1, dtype=np.uint64) * n_samples
    bytes_per_audio_s = 2.0 * FS
    pi = {"clips": nh, "bytes": int(host_pcm.nbytes), "note": "host int16 PCM in, host (key32,t1) out; never the headline value"}
    try:
        probe = {"pinned_h2d_GBs": ctx.membw(3, 512 << 20, 4), "pageable_h2d_GBs": ctx.membw(4, 512 << 20, 4),
                 "pinned_d2h_GBs": ctx.membw(5, 512 << 20, 4)}
        pi["link_probe"] = probe
        pi["ceiling_audio_s_per_s"] = {"pinned": probe["pinned_h2d_GBs"] * 1e9 / bytes_per_audio_s,
                                       "pageable": probe["pageable_h2d_GBs"] * 1e9 / bytes_per_audio_s}
    except Exception as e:  # noqa: BLE001
        pi["link_probe"] = {"error": repr(e)}
    pinned_pcm = ctx.host_array(len(host_pcm), np.int16)
    pinned_pcm[:] = host_pcm
    for name, arr in (("pageable", host_pcm), ("pinned", pinned_pcm)):
        ctx.fingerprint_batch(arr, hoff)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            ctx.fingerprint_batch(arr, hoff)
            ts.append(time.perf_counter() - t0)
        t_host = float(np.median(ts))
        pi[name] = {"audio_s_per_s": nh * n_samples / FS / t_host, "GBs": host_pcm.nbytes / t_host / 1e9, "seconds": t_host}
        if "ceiling_audio_s_per_s" in pi:
            pi[name]["frac_of_link_ceiling"] = pi[name]["audio_s_per_s"] / pi["ceiling_audio_s_per_s"][name]
    pi["audio_s_per_s"] = pi["pinned"]["audio_s_per_s"]
    pi["upload_pipeline"] = ctx.upload_stats()
    out["pcie_inclusive"] = pi
    del pinned_pcm
    # One query at a time through the reference-shaped entry point (host PCM in, result dicts out): serving latency
    try:
        out["single_query"] = single_query_latency(ctx, tbl, rank, nc, n_samples)
    except Exception as e:  # noqa: BLE001
        out["single_query"] = {"error": repr(e)}
    tbl.close()
    if comm:
        comm.close()
    # Two contexts (two streams, one host thread each) sharing this GPU, each fingerprinting half of the step's clips:
    # stft_psd of one overlaps peak_pick of the other.  Reported beside the headline, which stays the
    # single pipeline whose kernels run one at a time (that is what `roofline` is measured on).
    import threading
    half = nc // 2
    if half >= 100:
        ctx2 = _ffi.Context(ctx.device_id)
        off_h = np.arange(half + 1, dtype=np.uint64) * n_samples
        cap_h = int(kbuf.nbytes // 4 // 2)
        k2, t2 = ctx2.alloc(cap_h * 4), ctx2.alloc(cap_h * 4)
        lanes = [(ctx, int(pcm.ptr), kbuf, tbuf), (ctx2, int(pcm.ptr) + half * n_samples * 2, k2, t2)]

        def lane(c, p, ko, to, steps):
            for _ in range(steps):
                c.fingerprint_batch(p, off_h, fs=FS, pcm_device=True, out_key=ko, out_t1=to, cap=cap_h)
            c.sync()

        rounds = []
        for steps in (1, a.steps, a.steps, a.steps):   # one warm-up round (second context's tables and workspace), then three timed
            ths = [threading.Thread(target=lane, args=(*ln, steps)) for ln in lanes]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            rounds.append((time.perf_counter() - t0) / steps)
        dt2 = float(np.median(rounds[1:]))
        out["two_contexts_one_gpu"] = {"audio_s_per_s": 2 * half * n_samples / FS / dt2, "ms_per_step": dt2 * 1e3,
                                       "ms_per_step_rounds": [r * 1e3 for r in rounds[1:]], "clips_per_context": half,
                                       "note": "not the headline: two independent fingerprint pipelines on this GPU, median of three rounds"}
        for b_ in (k2, t2):
            b_.free()
        ctx2.close()


def db_build_scaling(a, ctx, dist, comm, rank, world):
    """STRONG scaling of the database build on BASELINE configs[2] as written: ONE fixed corpus of a.scaling_songs x
    a.scaling_seconds tracks (100,000 x 3 min), the same at every N, split over the ranks in contiguous blocks and built
    by the product's own driver (shazam_amd.ingest.ShardedBuilder): fingerprint a chunk, stage its rows, every ~1e9 rows
    (or the rank's share) seal a sorted run and start it travelling to the peers while the next chunk is fingerprinted,
    one k-way merge of all runs at the end.  `seconds` = max over ranks of (first chunk .. table standing on the GPU)
    minus the PCM synthesis kernels' own time (events; the stand-in for audio decoding, which is not the product).  The
    table's arenas are reserved before the clock starts and the time that took is reported beside it."""
    import bench_db
    from shazam_amd import _ffi
    from shazam_amd.db import HipFingerprintDB
    from shazam_amd.ingest import ShardedBuilder, shard_tracks
    songs, seconds = a.scaling_songs, a.scaling_seconds
    note = None

    def plan(sec):
        n_samples = int(round(sec * FS))
        frames = int(_ffi.lib().shz_frame_count(n_samples))
        rows_total = int(songs * frames * bench_db.ROWS_PER_FRAME_HINT)
        # rows between two seals: <= 1e9 (a run is one radix sort), and with N > 1 a fraction of a rank's share -- what is still
        # to travel when a rank's fingerprinting ends is its LAST run only -- chosen so that the node has <= 16 runs in all
        # (the merge's small tiles, shz_build.hip: KW_TILE_SMALL): 4 runs a rank at N <= 4, 2 at N = 8
        per_rank = max(1, min(4, 16 // world)) if world > 1 else 1
        seal = min(1_000_000_000, -(-rows_total // (world * per_rank)))
        # columns + arena of all runs + staging + one sort scratch + the extraction workspace and PCM of a chunk
        need = rows_total * 12 * 1.03 + (rows_total * 1.02 + seal * 1.2) * 8 + seal * 1.1 * 20 + 24e9
        return n_samples, frames, rows_total, seal, need

    n_samples, frames, rows_total, seal, need = plan(seconds)
    free_b, total_b = ctx.mem_info()
    if need > 0.93 * total_b:
        note = f"{songs} x {seconds:.0f} s needs ~{need / 1e9:.0f} GB of HBM, {total_b / 1e9:.0f} GB here: 30 s tracks instead"
        seconds = 30.0
        n_samples, frames, rows_total, seal, need = plan(seconds)
    chunk = max(16, min(1000, int(700_000 // frames)))   # ~700k frames a fingerprint call (the headline step holds 644k)
    lo, hi = shard_tracks(songs, rank, world)
    db = HipFingerprintDB(ctx=ctx)
    pcm = ctx.alloc(chunk * n_samples * 2)
    synth_ms = [0.0]

    def source(c0, c1):   # PCM synthesis: its kernel time (events on the context's stream) is taken out of `seconds`
        ctx.timer_start(7)
        ctx.synth_pcm(bench_db.SEED_TRACKS, c0, c1 - c0, n_samples, 4000, 1500, out=pcm)
        synth_ms[0] += ctx.timer_stop(7)
        return pcm, n_samples

    # setup, outside the clock (reported): the arenas exist before the first chunk (device memory that went through hipFree
    # earlier in the process comes back scrubbed by the driver at ~40 GB/s, and that stalls kernel launches)
    t0 = time.perf_counter()
    db.table.reserve(rows_total, seal, gather=True, wait=True)
    t_reserve = time.perf_counter() - t0
    builder = ShardedBuilder(db, rank, world, comm, chunk_tracks=chunk, seal_rows=seal)
    if comm is not None:
        comm.warmup()   # RCCL connects two ranks on their first send / receive: not on this build's clock
    if dist:
        dist.barrier()
    ctx.sync()
    t_all0 = time.perf_counter()
    info = builder.build(songs, source, Fs=FS, rows_hint=rows_total)
    if dist:
        dist.barrier()
    t_wall = time.perf_counter() - t_all0
    t_synth = synth_ms[0] / 1e3
    vals = [t_wall - t_synth, info["fingerprint_s"], info["insert_s"], info["seal_exchange_s"], info["final_s"], t_synth, t_reserve]
    if dist and hasattr(dist, "max_floats"):   # (the thread-rank harness: no torch in that process)
        vals = dist.max_floats(vals)
    elif dist:
        import torch
        tt = torch.tensor(vals, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        vals = [float(x) for x in tt]
    rows, _ = db.table.rows()
    o = {"songs": songs, "clip_seconds": seconds, "n_gpus": world, "scaling": "strong", "seconds": vals[0],
         "songs_per_s": songs / vals[0], "audio_s_per_s": songs * seconds / vals[0], "fingerprint_s": vals[1],
         "insert_s": vals[2], "seal_and_send_s": vals[3], "final_rounds_and_merge_s": vals[4], "synth_s_excluded": vals[5],
         "reserve_s_outside_clock": vals[6], "rows": int(rows), "segments": int(db.table.segments()),
         "runs_sent_on_the_way_rank0": info["runs_sealed_on_the_way"], "chunk_tracks": chunk, "seal_rows": seal,
         "allgather_bytes_received": int(info["bytes_received"]), "build_stats_s": db.table.build_stats(),
         "exchange": db.table.exchange_stats(),
         "phases_s": {k: round(v, 4) for k, v in db.table.phase_stats().items() if v > 5e-4},
         "config": "BASELINE configs[2]" + ("" if note is None else " at 30 s tracks: " + note),
         "note": "seconds = max over ranks of (first chunk .. table standing) minus the synthesis kernels' event time; the same "
                 "corpus at every N; sealed runs travel on the communicator's stream beside the next chunk's fingerprinting"}
    db.close()
    pcm.free()
    return o


def single_query_latency(ctx, tbl, rank, nc, n_samples, n_iter=60):
    """p50 / p99 of ONE 5 s query: fingerprint (host int16 in) + match (top-2) per call, against the step's table."""
    track = ctx.synth_pcm(1234, rank * nc + 7, 1, n_samples, 0, 8000)   # the step's track 7, synthesised on the device
    q = track.download(np.int16, n_samples)[13 * 2048 + 77:13 * 2048 + 77 + 5 * FS].copy()
    track.free()
    qoff = np.array([0, len(q)], np.uint64)
    lat = {"fingerprint": [], "match": [], "total": []}
    top = None
    for i in range(n_iter + 5):
        t0 = time.perf_counter()
        k, t1, ho, _ = ctx.fingerprint_batch(q, qoff)
        t1_ = time.perf_counter()
        res = tbl.match(k, t1, ho, 2)
        t2 = time.perf_counter()
        if i >= 5:
            lat["fingerprint"].append(t1_ - t0)
            lat["match"].append(t2 - t1_)
            lat["total"].append(t2 - t0)
        top = (int(res["sid"][0, 0]), int(res["delta"][0, 0]))
    o = {"queries": n_iter, "query_seconds": 5.0, "top1": {"song_id": top[0], "offset": top[1]},
         "expected": {"song_id": rank * nc + 8, "offset": 13},
         "note": "host PCM in, host results out, one query per call; crop at sample 13*2048+77 of track 7"}
    for k_, v in lat.items():
        v = np.array(v) * 1e3
        o[f"{k_}_p50_ms"] = float(np.median(v))
        o[f"{k_}_p99_ms"] = float(np.percentile(v, 99))
    return o


def match_1m(ctx, songs, info):
    """Second half of BASELINE's metric: query match latency against a 1M-song table (1M x 30 s tracks, 1.1e10 rows,
    ~136 GB in one GPU's HBM).  10 s noisy crops (SNR 10 dB, the reference's mixing rule), true wall time per match call
    at batch sizes 1 and 200 (recognizer.py:273-286 `query_time` is this stage in the reference: p50 0.816 s at 13 M rows)."""
    import bench_db
    need = songs * 11300 * 12 * 1.35   # rows x 12 B x (sort scratch + headroom)
    if info["hbm_bytes"] < need:
        return {"skipped": f"needs ~{need / 1e9:.0f} GB of HBM"}
    t0 = time.perf_counter()
    # a run every 1/16 of the corpus: 16 runs are what the k-way merge takes with its small tiles (1.13e10 rows: plan + merge 0.13 s;
    # 20 runs: 0.39 s), and the batch (7.4e8 rows) still fits the arena beside the held runs
    tbl, build, bufs = bench_db.build_table(ctx, songs, 30.0, 1000, 4000, 1500, finalize_every=max(1000, -(-songs // 16000) * 1000))   # (rounded UP to whole chunks: 16 runs, not 17)
    n_samples = 30 * FS
    qn = 10 * FS
    o = {"songs": songs, "rows": build["rows"], "build_seconds": build["seconds_total"],
         "build": {k: build[k] for k in ("reserve_s", "fingerprint_s", "insert_s", "finalize_s", "synth_wait_s", "synth_overlapped",
                                         "songs_per_s", "segments", "key_range_segments", "phases_s")},
         "query_seconds": 10.0, "snr_db": 10.0}
    for bs, nq in ((1, 60), (200, 2000)):
        bench_db.run_queries(ctx, tbl, songs, n_samples, bs * 2, qn, 10.0, bs, 2, seed=5)   # warm the workspace
        r = bench_db.run_queries(ctx, tbl, songs, n_samples, nq, qn, 10.0, bs, 2, seed=99 + bs)
        alg = (8 * r["rows_scanned"] + 16 * r["distinct_keys"]) / r["t_match"] / 1e9
        o[f"batch{bs}"] = {"batches": int(len(r["batch_ms"])), "batch_ms_p50": float(np.percentile(r["batch_ms"], 50)),
                           "batch_ms_p99": float(np.percentile(r["batch_ms"], 99)),
                           "ms_per_query_p50": float(np.percentile(r["batch_ms"] / r["sizes"], 50)),
                           "qps": nq / r["t_match"], "top1_accuracy": r["correct"] / nq,
                           "rows_scanned_per_query": r["rows_scanned"] / nq, "pairs_per_query": r["pairs"] / nq,
                           "hashes_per_query": r["hashes"] / nq, "alg_GBs": alg, "alg_frac_of_hbm_peak": alg / HBM_PEAK_GBS,
                           "query_fingerprint_ms": r["query_fingerprint_s"] / nq * 1e3}
    o["p50_ms"] = o["batch1"]["batch_ms_p50"]
    o["p99_ms"] = o["batch1"]["batch_ms_p99"]
    o["seconds_total"] = time.perf_counter() - t0
    tbl.close()
    for b in bufs[:2]:
        b.free()
    return o


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) before this process
    touches HIP, relay their output, exit with the worst return code."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p_ in procs:
        p_.wait()
        rc = rc or p_.returncode
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=1000, help="clips per GPU per step")
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--match-songs", type=int, default=1000000, help="tracks of the match_1M extra (0 = skip)")
    ap.add_argument("--ws-limit-mb", type=float, default=0.0, help="cap on the extraction workspace (shz_set_workspace_limit): "
                    "smaller sub-batches, e.g. so that one sub-batch's staged spectrogram stays in the 256 MB Infinity Cache (experiment)")
    ap.add_argument("--scaling-songs", type=int, default=100000, help="fixed corpus of the db_build_scaling extra: the same "
                    "at every --gpus N (0 = skip)")
    ap.add_argument("--scaling-seconds", type=float, default=180.0, help="track length of that corpus (BASELINE configs[2]: 3 min)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal on one GPU: take the N > 1 branches (torch first, gloo group, "
                    "RCCL communicator, gathered builds, max over ranks) with ONE rank")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)   # never returns
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: one rank per GPU, launch with "
                         f"torch.distributed.run --nproc-per-node {a.gpus} or without a launcher")
    dist = None
    if world > 1 or a.force_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
        # torch.distributed is rendezvous plumbing only (barrier, max, id broadcast) on gloo/CPU;
        # the data path collective is RCCL called from libshz.so.
        # ORDER MATTERS: torch brings its own copy of the HIP runtime and of RCCL.  Imported BEFORE libshz.so is loaded, the whole
        # process -- libshz.so included -- binds to that one copy and everything works (scripts/rccl_probe.py).  Imported AFTER
        # it, the process holds two HIP runtimes and ncclCommInitRank fails ("unhandled cuda error").  So: torch first, here.
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if dist:
            dist.barrier()

    n_samples = int(round(a.seconds * FS))
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(n_samples)   # before any HIP call: the Pool forks from a GPU-free process

    from shazam_amd import _ffi
    ctx = _ffi.Context(int(os.environ.get("SHZ_BENCH_DEVICE", local)))
    info = ctx.device_info()
    if a.ws_limit_mb > 0:
        ctx.set_workspace_limit(int(a.ws_limit_mb * 1e6))
    nc = a.clips
    frames_per_clip = int(_ffi.lib().shz_frame_count(n_samples))
    pcm = ctx.synth_pcm(1234, rank * nc, nc, n_samples, 0, 8000)        # resident in HBM before timing
    off = np.arange(nc + 1, dtype=np.uint64) * n_samples
    cap = int(nc * frames_per_clip * 24) + 1024
    kbuf, tbuf = ctx.alloc(cap * 4), ctx.alloc(cap * 4)

    def step():
        return ctx.fingerprint_batch(pcm, off, fs=FS, pcm_device=True, out_key=kbuf, out_t1=tbuf, cap=cap)

    for _ in range(a.warmup):
        step()
    ctx.sync()
    ctx.set_profiling(True)          # event records only; no host sync inside the timed region
    barrier()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        _, _, hash_off, n_hashes = step()
    ctx.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    kms = ctx.kernel_ms()
    ctx.set_profiling(False)
    if dist:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])

    audio_s_per_step = nc * n_samples / FS
    value = world * audio_s_per_step * a.steps / elapsed
    frames_per_step = nc * frames_per_clip

    # roofline of the dominant kernel, from the HIP events recorded inside the timed region.
    # `achieved` / `frac` follow SURVEY 8(d): the implementation-independent bytes of the path (PCM in, hashes out) that
    # one launch of the kernel serves, over the kernel's own average duration; the bytes the kernel actually moves
    # (fp32 power rows staged through HBM) are the staged_* keys, the PMC-measured traffic is `traffic`.
    dom = max((k for k in kms if k in STAGED_BYTES_PER_FRAME), key=lambda k: kms[k][0])
    dom_ms, dom_launches = kms[dom]
    frames_per_launch = frames_per_step * a.steps / max(dom_launches, 1)
    avg_ms = dom_ms / max(dom_launches, 1)
    alg_bytes_per_frame = PCM_BYTES_PER_FRAME + 8.0 * n_hashes / frames_per_step
    achieved = frames_per_launch * alg_bytes_per_frame / (avg_ms * 1e-3) / 1e9
    staged = frames_per_launch * STAGED_BYTES_PER_FRAME[dom] / (avg_ms * 1e-3) / 1e9
    step_alg = frames_per_step * alg_bytes_per_frame / (elapsed / a.steps) / 1e9
    traffic, traffic_file = pmc_traffic(dom, frames_per_launch)
    valu_busy, valu_file = pmc_valu_busy(dom)
    tflops = frames_per_launch * FFT_FLOP_PER_FRAME / (avg_ms * 1e-3) / 1e12
    roofline = {"bound": "valu_fp64" if dom == "stft_psd" else "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_from_profile": True,   # a committed rocprofv3 PMC pass of this same command, not of this very run
                "traffic_unit": f"GB per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/{traffic_file})",
                "compute": {"bound": "valu_fp64", "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "compute_frac": tflops / FP64_VALU_PEAK_TFLOPS,
                            "accounting": f"{FFT_FLOP_PER_FRAME:.0f} flop/frame (2.5 N log2 N, N = 4096) x {frames_per_launch:.0f} "
                                          f"frames/launch / {avg_ms:.3f} ms; the kernel's own count is higher (window, split pass, |X|^2)",
                            "valu_busy_frac_pmc": valu_busy, "valu_busy_from_profile": True,
                            "valu_busy_source": f"SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x 3 waves per SIMD, profiles/{valu_file}"} if dom == "stft_psd" else None,
                "accounting": f"algorithmic bytes (SURVEY 8d): {alg_bytes_per_frame:.0f} B/frame (4096 B PCM + 8 B x "
                              f"{n_hashes / frames_per_step:.1f} hashes) x {frames_per_launch:.0f} frames/launch / "
                              f"{avg_ms:.3f} ms avg launch of {dom} (HIP events, {dom_launches} launches in the timed region)",
                "staged_achieved": staged, "staged_frac": staged / HBM_PEAK_GBS,
                "staged_accounting": f"bytes the kernel moves: {STAGED_BYTES_PER_FRAME[dom]} B/frame (PCM in + fp32 power row out)",
                "kernel_ms_per_step": {k: v[0] / a.steps for k, v in kms.items()},
                "step_achieved": step_alg, "step_frac": step_alg / HBM_PEAK_GBS,
                "host_overhead_ms_per_step": elapsed / a.steps * 1e3 - sum(v[0] for v in kms.values()) / a.steps,
                "pipelines": "two halves of the batch run as two passes on two streams (SHZ_DUAL=0: one): kernel durations "
                             "overlap, their sum exceeds the step and host_overhead goes negative" if os.environ.get("SHZ_DUAL", "0") not in ("", "0") else "one",
                "note": "the dominant kernel is bound by fp64 VALU issue + LDS exchange, not by HBM (SURVEY 8d: 30 flop/B fused): "
                        "achieved / frac stay the ALGORITHMIC bytes over the HBM peak as the contract defines them, "
                        "compute.compute_frac is the ceiling the kernel is actually up against; staged_frac is its own I/O rate"}

    out = {"metric": "audio_seconds_fingerprinted_per_second", "value": value, "unit": "audio-s/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"{nc} x {a.seconds:.0f} s synthetic 44.1 kHz mono int16 clips per GPU, fingerprint-only "
                                  "(BASELINE configs[1]); PCM resident in HBM, (key32,t1) compacted on device",
                      "clips_per_gpu": nc, "clip_seconds": a.seconds, "frames_per_step_per_gpu": frames_per_step,
                      "hashes_per_step_per_gpu": int(n_hashes), "parallelism": f"dp{world} (one process per GPU)",
                      "device": info["name"], "compute_units": info["compute_units"]},
           "x_realtime_per_gpu": value / world, "roofline": roofline}
    if a.ws_limit_mb > 0:
        out["config"]["workspace_limit_mb"] = a.ws_limit_mb

    # ---- extras (outside the timed region) -------------------------------------------------
    if cpu is not None:
        out["cpu_baseline"] = cpu
        out["gpu_over_cpu"] = value / cpu["value"]

    def emit():
        if rank == 0:
            print(json.dumps(out), flush=True)

    # ---- extras (outside the timed region).  A watchdog prints the headline line if the extras (RCCL init /
    # all-gather on an unknown node) hang, so the measured value is never lost -- and exits NON-ZERO: a hang is a failure.
    if not a.no_extras:
        import threading

        def on_timeout():   # runs in its own thread: a hung RCCL call inside ctypes cannot block it
            out["extras_error"] = "extras timed out after 420 s"
            emit()
            os._exit(3)

        dog = threading.Timer(420.0, on_timeout)
        dog.daemon = True
        dog.start()
        out["extract_stats"] = ctx.extract_stats()
        try:
            extras(a, ctx, dist, rank, world, nc, n_samples, kbuf, tbuf, hash_off, elapsed, pcm, out)
        except Exception as e:  # noqa: BLE001 -- extras must never cost the headline number
            out["extras_error"] = repr(e)
        if world == 1 and a.match_songs > 0:
            try:
                for b_ in (kbuf, tbuf, pcm):
                    b_.free()
                # (the workspace and the blocks earlier tables handed back stay with the context: memory that went through
                # hipFree comes back scrubbed by the driver at ~40 GB/s on its next use)
                out["match_1M"] = match_1m(ctx, a.match_songs, info)
            except Exception as e:  # noqa: BLE001
                out["match_1M"] = {"error": repr(e)}
        if world == 1 and a.scaling_songs > 0 and not a.force_dist:
            try:
                out["db_build_scaling"] = db_build_scaling(a, ctx, None, None, 0, 1)
            except Exception as e:  # noqa: BLE001
                out["db_build_scaling"] = {"error": repr(e)}
        dog.cancel()
    emit()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
