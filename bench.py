#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X fingerprint hot path.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): audio-seconds fingerprinted per second, 44.1 kHz mono int16.
Workload (BASELINE configs[1]): per GPU a batch of 1,000 synthetic 30 s clips resident in HBM,
fingerprint-only: PCM -> STFT -> dB -> 21x21 peaks -> pair hashes (key32, t1) compacted on
the device.  One "step" = one pass over that batch.  N > 1 (launched by torch.distributed.run)
is weak scaling: every rank fingerprints its own 1,000 clips; `value` = all ranks' audio
seconds / max-over-ranks wall time of the K timed steps.

Beside the headline line the JSON carries
  roofline      dominant kernel's bytes/launch over its HIP-event duration vs HBM peak
  cpu_baseline  the reference's numpy/scipy/mlab call sequence (oracle/thirdparty_ref.py)
                timed on this box's host cores over a bounded sample (rank 0, N = 1 only)
  db_build      fingerprints -> HBM table (with the RCCL all-gather when N > 1), untimed extra
  match         batched recognise against that table, untimed extra
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 44100
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the measured stream rates of the box
#                        (copy ~5.0, read ~6.2, write ~4.0 TB/s) go into roofline.measured_stream_GBs
STFT_BYTES_PER_FRAME = 4096 + 2049 * 8   # new PCM read + dB row written (staged kernel I/O)
COMPULSORY_BYTES_PER_FRAME = 4096 + 147  # PCM in + ~18.4 hashes x 8 B out


def pmc_traffic(kernel, frames_per_launch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/), scaled to this
    run's frames per launch; None when no profile of that kernel is committed."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r01g_pmc_traffic.json")))
        k = {"stft_psd": "stft_psd_kernel", "peak_pick": "peak_pick_kernel<true>"}[kernel]
        return prof["kernels"][k]["hbm_bytes_corrected"] / 1e9 * frames_per_launch / prof.get("frames_per_launch", 644000)
    except Exception:
        return None


def cpu_worker(args):
    seed, clip, n = args
    from oracle import synth, thirdparty_ref
    x = synth.synth_clip(seed, clip, n, 0, 8000)
    t0 = time.perf_counter()
    h = thirdparty_ref.fingerprint(x, Fs=FS)
    return time.perf_counter() - t0, len(h)


def cpu_baseline(n_samples, budget_s=15.0):
    """Reference call sequence (mlab.specgram + scipy.ndimage + hashlib) over a Pool of all
    host cores, like __init__.py:335-357.  Bounded sample: cores x k clips."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    t1 = cpu_worker((1234, 0, n_samples))[0]          # single-core time per clip (also warms imports)
    per_proc = max(1, int(budget_s / max(t1 * 2.5, 1e-3)))
    per_proc = min(per_proc, 4)
    clips = cores * per_proc
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(cpu_worker, [(1234, c, n_samples // 8) for c in range(cores)])  # warm-up
        t0 = time.perf_counter()
        res = pool.map(cpu_worker, [(1234, c, n_samples) for c in range(clips)], chunksize=1)
        wall = time.perf_counter() - t0
    audio_s = clips * n_samples / FS
    return {"value": audio_s / wall, "unit": "audio-s/s", "cores": cores, "kind": "port",
            "single_core_value": (n_samples / FS) / t1,
            "sample": f"{clips} synthetic {n_samples / FS:.0f} s clips (same generator/seed as the GPU run) through "
                      f"oracle/thirdparty_ref.py (mlab.specgram + scipy.ndimage + hashlib, the reference's call sites) "
                      f"on a multiprocessing.Pool({cores}); synth time excluded; wall {wall:.1f} s",
            "hashes_per_clip": float(np.mean([r[1] for r in res]))}


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
        rf["frac_of_measured_copy"] = rf["achieved"] / bw["copy"] if bw["copy"] > 0 else None
    except Exception as e:  # noqa: BLE001
        out["roofline"]["measured_stream_error"] = repr(e)

    from shazam_amd import Table
    tbl = Table(ctx)
    comm = None
    if world > 1:
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
                       "allgather_bytes_received": int(recv), "collective": "rccl grouped broadcast (all-gather-v)" if comm else None}
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
    qpcm.free()
    # the same hot path fed from HOST memory (pageable numpy -> hipMemcpy inside the call): PCIe-inclusive rate
    nh = min(200, nc)
    host_pcm = pcm.download(np.int16, nh * n_samples)
    hoff = np.arange(nh + 1, dtype=np.uint64) * n_samples
    ctx.fingerprint_batch(host_pcm, hoff)
    t0 = time.perf_counter()
    ctx.fingerprint_batch(host_pcm, hoff)
    t_host = time.perf_counter() - t0
    out["pcie_inclusive"] = {"clips": nh, "audio_s_per_s": nh * n_samples / FS / t_host,
                             "note": "host int16 PCM in, host (key32,t1) out, pageable memory; never the headline value"}
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
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        # torch.distributed is rendezvous plumbing only (barrier, max, id broadcast) on gloo/CPU;
        # the data path collective is RCCL called from libshz.so.
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

    # roofline of the dominant kernel, from the HIP events recorded inside the timed region
    dom = max(kms, key=lambda k: kms[k][0])
    dom_ms, dom_launches = kms[dom]
    frames_per_launch = frames_per_step * a.steps / max(dom_launches, 1)
    avg_ms = dom_ms / max(dom_launches, 1)
    bytes_per_frame = {"stft_psd": STFT_BYTES_PER_FRAME, "peak_pick": 2049 * 8 + 288}.get(dom, STFT_BYTES_PER_FRAME)
    achieved = frames_per_launch * bytes_per_frame / (avg_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, frames_per_launch), "traffic_unit": "GB per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01g_pmc_traffic.json)",
                "accounting": f"kernel I/O bytes: {bytes_per_frame} B/frame x {frames_per_launch:.0f} frames/launch / "
                              f"{avg_ms:.3f} ms avg launch (HIP events, {dom_launches} launches in the timed region)",
                "kernel_ms_per_step": {k: v[0] / a.steps for k, v in kms.items()},
                "compulsory_achieved_GBs": frames_per_step * COMPULSORY_BYTES_PER_FRAME / (elapsed / a.steps) / 1e9,
                "compulsory_frac": frames_per_step * COMPULSORY_BYTES_PER_FRAME / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS}

    out = {"metric": "audio_seconds_fingerprinted_per_second", "value": value, "unit": "audio-s/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"{nc} x {a.seconds:.0f} s synthetic 44.1 kHz mono int16 clips per GPU, fingerprint-only "
                                  "(BASELINE configs[1]); PCM resident in HBM, (key32,t1) compacted on device",
                      "clips_per_gpu": nc, "clip_seconds": a.seconds, "frames_per_step_per_gpu": frames_per_step,
                      "hashes_per_step_per_gpu": int(n_hashes), "parallelism": f"dp{world} (one process per GPU)",
                      "device": info["name"], "compute_units": info["compute_units"]},
           "x_realtime_per_gpu": value / world, "roofline": roofline}

    # ---- extras (outside the timed region) -------------------------------------------------
    if cpu is not None:
        out["cpu_baseline"] = cpu
        out["gpu_over_cpu"] = value / cpu["value"]

    def emit():
        if rank == 0:
            print(json.dumps(out), flush=True)

    # ---- extras (outside the timed region).  A watchdog prints the headline line and exits if the
    # extras (RCCL init / all-gather on an unknown node) hang, so the measured value is never lost.
    if not a.no_extras:
        import threading

        def on_timeout():   # runs in its own thread: a hung RCCL call inside ctypes cannot block it
            out["extras_error"] = "extras timed out after 240 s"
            emit()
            os._exit(0)

        dog = threading.Timer(240.0, on_timeout)
        dog.daemon = True
        dog.start()
        try:
            extras(a, ctx, dist, rank, world, nc, n_samples, kbuf, tbuf, hash_off, elapsed, pcm, out)
        except Exception as e:  # noqa: BLE001 -- extras must never cost the headline number
            out["extras_error"] = repr(e)
        dog.cancel()
    emit()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
