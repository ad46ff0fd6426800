#!/usr/bin/env python3
"""bench_db.py -- database build + batched noisy-query match on ONE MI355X (BASELINE configs 3/4,
scaled to what a replicated < 2^32-row table holds).  Not the driver's headline bench (that is
bench.py); this is the second half of the metric: "query match ms vs an N-song DB".  bench.py imports
build_table / make_queries / run_queries from here for its `match_1M` extra.

    python bench_db.py --songs 100000 --seconds 30 --queries 10000 --snr 0

Tracks: synthetic tonal+noise clips generated on the device (oracle twin: oracle/synth.py).
Queries: 5 s crops at arbitrary (not hop-aligned) sample offsets, mixed on the device with an
independent noise stream at the requested SNR using the reference's rule
(recognizer_test.py:426-435; ADD_NOISE / SNR at :39-40), fingerprinted and matched in batches.
Prints one JSON line.
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
SEED_TRACKS, SEED_NOISE = 4321, 777


ROWS_PER_FRAME_HINT = 18.5   # hashes per frame of the synthetic corpora (17.6-18.7 measured): sizes shz_table_reserve

# Corpora: "tonal" = six sinusoids + white noise per track, white query noise (rounds 1-2); "music" = music-like tracks (four
# voices of decaying harmonic notes, percussive onsets) under traffic-like low-passed query noise (oracle/synth.music_clip /
# traffic_noise; what the reference's accuracy was measured on is real music under street noise, recognizer_test.py:39-40)
MUSIC_AMP, MUSIC_BED, MUSIC_BURST, TRAFFIC_AMP = 3000, 100, 1500, 2000


def synth_tracks(ctx, corpus, c0, nc, n_samples, tone_amp, noise_amp, out, start=0):
    if corpus == "music":
        return ctx.synth_corpus(1, SEED_TRACKS, c0, nc, n_samples, MUSIC_AMP, MUSIC_BED, MUSIC_BURST, start, out=out)
    return ctx.synth_pcm(SEED_TRACKS, c0, nc, n_samples, tone_amp, noise_amp, start, out=out)


def build_table(ctx, songs, seconds=30.0, chunk=1000, tone_amp=4000, noise_amp=1500, finalize_every=0, shards=1,
                progress=None, reserve=True, corpus="tonal", hold=True, overlap_synth=False):
    """Synthesise `songs` tracks on the device, fingerprint them in chunks and build one HBM table.
    Every `finalize_every` songs the staged rows are sealed into a sorted run (bounded staging and sort scratch; rows become
    visible at the final finalize).  hold: the table keeps its runs until ONE merge at the end cuts the segments by key
    range (a query hash is then looked up in one segment) -- needs the arena to hold every row beside the columns
    (20 B a row); when that does not fit, full segments are cut on the way as before (every segment spans every key).
    The table's arenas are reserved on a helper thread beside the first chunks (shz_table_reserve).  Device memory that any
    process freed comes back scrubbed by the driver at ~43 GB/s (23 ms a GB), and the scrub shares the GPU with the kernels:
    reserved up front it adds its own time (1M x 30 s, 250 GB: reserve 5.7 s + fingerprint 6.4 s; seconds_total 20.0), beside
    the first chunks it stretches them instead (fingerprint 8.8 s, seconds_total 16.8) -- the second is what is done.
    overlap_synth (off): chunk i + 1 synthesised by a second context (own stream, second PCM buffer) while chunk i is
    fingerprinted and inserted.  Measured at 1M x 30 s (with the round-3 generator): 13.0 s against 6.43 + 6.88 s one after the other -- synthesis and
    the STFT are both bound by the vector ALU and only share the chip; left off so that fingerprint_s means fingerprinting.
    (The generator now takes 2.8 s for the same tracks.)
    Returns (table, stats); song ids are track index + 1 (mysql_database.py:34,200)."""
    from shazam_amd import _ffi, Table
    n_samples = int(round(seconds * FS))
    frames = int(_ffi.lib().shz_frame_count(n_samples))
    t_build0 = time.perf_counter()   # (the reservation counts: it is part of what a build costs)
    if shards > 1:
        from shazam_amd.shard import ShardedTable
        tbl = ShardedTable(ctx, nshards=shards)
    else:
        tbl = Table(ctx)
        held = False
        if reserve:   # the table's arenas in one go, before the first chunk
            per_batch = finalize_every if finalize_every else songs
            rows_hint, batch_hint = int(songs * frames * ROWS_PER_FRAME_HINT), int(min(per_batch, songs) * frames * ROWS_PER_FRAME_HINT)
            if hold:
                try:
                    tbl.reserve(rows_hint, batch_hint, gather=True)
                    held = True
                except _ffi.ShzError as e:
                    if e.code != _ffi.E_NOMEM:
                        raise
                    tbl.close()          # (the refused reservation left the hold flag set: start over)
                    tbl = Table(ctx)
            if not held:
                tbl.reserve(rows_hint, batch_hint)
    t_reserve = time.perf_counter() - t_build0
    t_mark = time.perf_counter()
    cap = chunk * frames * 24 + 1024
    kbuf, tbuf = ctx.alloc(cap * 4), ctx.alloc(cap * 4)
    overlap_synth = overlap_synth and songs > chunk
    ctx_s = _ffi.Context(ctx.device_id) if overlap_synth else ctx     # synthesis on its own stream
    pcms = [ctx.alloc(chunk * n_samples * 2) for _ in range(2 if overlap_synth else 1)]
    t_fp = t_ins = t_fin = t_synth_wait = 0.0
    n_rows_in = 0
    starts = list(range(0, songs, chunk))
    synth_tracks(ctx_s, corpus, 0, min(chunk, songs), n_samples, tone_amp, noise_amp, pcms[0])
    ctx.sync()
    t_setup = time.perf_counter() - t_mark   # buffers + the first chunk's synthesis
    t_loop0 = time.perf_counter()
    for i, c0 in enumerate(starts):
        nc = min(chunk, songs - c0)
        pcm = pcms[i % len(pcms)]
        off = np.arange(nc + 1, dtype=np.uint64) * n_samples
        t0 = time.perf_counter()
        ctx_s.sync()                      # chunk i is there
        t_synth_wait += time.perf_counter() - t0
        if overlap_synth and i + 1 < len(starts):   # chunk i + 1 into the other buffer (chunk i - 1, its last user, is done)
            synth_tracks(ctx_s, corpus, starts[i + 1], min(chunk, songs - starts[i + 1]), n_samples, tone_amp, noise_amp, pcms[(i + 1) % 2])
        t0 = time.perf_counter()
        _, _, ho, cnt = ctx.fingerprint_batch(pcm, off, fs=FS, pcm_device=True, out_key=kbuf, out_t1=tbuf, cap=cap)
        t_fp += time.perf_counter() - t0
        t0 = time.perf_counter()
        tbl.insert_clips(kbuf, tbuf, ho, sid0=1 + c0, device=True)
        t_ins += time.perf_counter() - t0
        n_rows_in += cnt
        if not overlap_synth and i + 1 < len(starts):
            synth_tracks(ctx, corpus, starts[i + 1], min(chunk, songs - starts[i + 1]), n_samples, tone_amp, noise_amp, pcm)
        if finalize_every and (c0 + nc) % finalize_every == 0 and c0 + nc < songs:
            t0 = time.perf_counter()
            tbl.seal_run()        # bounds staged rows + sort scratch (held runs wait for the one merge at the end)
            ctx.sync()
            t_fin += time.perf_counter() - t0
            if progress:
                progress(c0 + nc)
    t_loop = time.perf_counter() - t_loop0
    t0 = time.perf_counter()
    tbl.finalize()
    ctx.sync()
    t_fin += time.perf_counter() - t0
    t_build = time.perf_counter() - t_build0
    rows, _ = tbl.rows()
    for b_ in pcms:
        b_.free()
    if ctx_s is not ctx:
        ctx_s.close()
    stats = {"seconds_total": t_build, "reserve_s": t_reserve, "fingerprint_s": t_fp, "insert_s": t_ins, "finalize_s": t_fin,
             "synth_wait_s": t_synth_wait, "synth_overlapped": bool(overlap_synth), "setup_s": t_setup, "loop_s": t_loop,
             "rows_inserted": int(n_rows_in), "rows": int(rows), "songs_per_s": songs / t_build,
             "audio_s_per_s": songs * seconds / t_build, "segments": int(tbl.segments()) if shards == 1 else None,
             "key_range_segments": bool(shards == 1 and reserve and held),
             "phases_s": {k: round(v, 4) for k, v in tbl.phase_stats().items() if v > 5e-4} if shards == 1 else None}
    return tbl, stats, (kbuf, tbuf, cap)


def make_queries(ctx, tids, starts, qn, snr, tone_amp=4000, noise_amp=1500, noise_clip0=0, corpus="tonal"):
    """Device PCM of len(tids) queries: crop [start, start+qn) of track tid, mixed with an independent noise stream at
    `snr` dB by the reference's rule (snr >= 200: clean).  Returns (DevBuf, buffers to free)."""
    from shazam_amd import _ffi
    nb = len(tids)
    sig, noi = ctx.alloc(nb * qn * 2), ctx.alloc(nb * qn * 2)
    for i in range(nb):
        if corpus == "music":
            ctx.check(_ffi.lib().shz_synth_corpus(ctx.h, 1, SEED_TRACKS, int(tids[i]), 1, qn, MUSIC_AMP, MUSIC_BED, MUSIC_BURST,
                                                  int(starts[i]), _ffi.vp(sig.ptr + i * qn * 2)))
        else:
            ctx.check(_ffi.lib().shz_synth_pcm(ctx.h, SEED_TRACKS, int(tids[i]), 1, qn, tone_amp, noise_amp, int(starts[i]),
                                               _ffi.vp(sig.ptr + i * qn * 2)))
    if corpus == "music":
        ctx.synth_corpus(2, SEED_NOISE, noise_clip0, nb, qn, TRAFFIC_AMP, 0, 0, 0, out=noi)
    else:
        ctx.synth_pcm(SEED_NOISE, noise_clip0, nb, qn, 0, 8000, out=noi)
    q = ctx.mix_snr(sig, noi, nb, qn, snr) if snr < 200 else sig
    return q, [b for b in (sig, noi, q) if b is not None]


def run_queries(ctx, tbl, songs, n_samples, nq, qn, snr, match_batch, topn=2, tone_amp=4000, noise_amp=1500, seed=99,
                before_batch=None, corpus="tonal"):
    """nq queries in batches of match_batch: fingerprint + match, wall time of the match call per batch.
    Returns a dict with per-batch milliseconds (whole batch), accuracy and the rows/pairs the match touched."""
    rng = np.random.default_rng(seed)
    tids = rng.integers(0, songs, nq)
    starts = rng.integers(0, n_samples - qn, nq)
    batch_ms, sizes, correct, tot_pairs, tot_rows, tot_hash, tot_keys = [], [], 0, 0, 0, 0, 0
    t_qfp = 0.0
    for b0 in range(0, nq, match_batch):
        nb = min(match_batch, nq - b0)
        if before_batch:
            before_batch()
        q, bufs = make_queries(ctx, tids[b0:b0 + nb], starts[b0:b0 + nb], qn, snr, tone_amp, noise_amp, b0, corpus)
        qoff = np.arange(nb + 1, dtype=np.uint64) * qn
        ctx.sync()
        t0 = time.perf_counter()
        k, t1, ho, _ = ctx.fingerprint_batch(q, qoff, fs=FS, pcm_device=True)
        t_qfp += time.perf_counter() - t0
        t0 = time.perf_counter()
        res = tbl.match(k, t1, ho, topn)
        dt = time.perf_counter() - t0
        batch_ms.append(dt * 1e3)
        sizes.append(nb)
        st = tbl.match_stats()
        tot_pairs += st["pairs"]
        tot_rows += st["rows_scanned"]
        tot_keys += st["distinct_keys"]
        tot_hash += int(res["nhash"].sum())
        correct += int(np.sum((res["nres"] > 0) & (res["sid"][:, 0] == 1 + tids[b0:b0 + nb])))
        seen = set()
        for b in bufs:
            if id(b) not in seen:
                seen.add(id(b))
                b.free()
    batch_ms, sizes = np.array(batch_ms), np.array(sizes)
    t_match = float(batch_ms.sum() / 1e3)
    return {"batch_ms": batch_ms, "sizes": sizes, "t_match": t_match, "correct": correct, "pairs": tot_pairs,
            "rows_scanned": tot_rows, "distinct_keys": tot_keys, "hashes": tot_hash, "query_fingerprint_s": t_qfp}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--songs", type=int, default=20000)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--queries", type=int, default=10000)
    ap.add_argument("--query-seconds", type=float, default=5.0)
    ap.add_argument("--snr", type=float, default=0.0)
    ap.add_argument("--chunk", type=int, default=1000)
    ap.add_argument("--match-batch", type=int, default=1000)
    ap.add_argument("--tone-amp", type=int, default=4000)
    ap.add_argument("--noise-amp", type=int, default=1500)
    ap.add_argument("--topn", type=int, default=2)
    ap.add_argument("--finalize-every", type=int, default=0, help="songs between intermediate finalize calls (0 = once at the end)")
    ap.add_argument("--mixed-ingest", type=int, default=0, help="songs ingested (fingerprint + insert + finalize) before every "
                    "query batch: the mixed ingest + query stream of BASELINE configs[4]; 0 = queries only")
    ap.add_argument("--corpus", choices=("tonal", "music"), default="tonal", help="tonal: six sinusoids + white noise, white query "
                    "noise; music: music-like tracks under traffic-like query noise")
    ap.add_argument("--shards", type=int, default=1, help="partition the table by key into this many shards on the GPU "
                    "(shazam_amd/shard.py): measures the cost of per-shard voting + merge against the single table")
    ap.add_argument("--overlap-synth", action="store_true", help="synthesise the next chunk on a second context beside the fingerprint "
                    "kernels of this one (measured: no gain, both are bound by the vector ALU)")
    ap.add_argument("--no-hold", action="store_true", help="cut full segments on the way (bounded arena; every segment spans every "
                    "key) instead of holding all runs for one merge into key-range segments")
    a = ap.parse_args()

    from shazam_amd import _ffi
    ctx = _ffi.Context(int(os.environ.get("SHZ_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    n_samples = int(round(a.seconds * FS))
    tbl, build, (kbuf, tbuf, cap) = build_table(ctx, a.songs, a.seconds, a.chunk, a.tone_amp, a.noise_amp,
                                                a.finalize_every, a.shards, corpus=a.corpus, hold=not a.no_hold,
                                                overlap_synth=a.overlap_synth)
    rows = build["rows"]

    qn = int(round(a.query_seconds * FS))
    nq = a.queries
    mixed = {"songs": 0, "seconds": 0.0, "finalize_s": 0.0, "batches": 0}
    n_mix = min(a.mixed_ingest, a.chunk)
    mix_pcm = ctx.alloc(n_mix * n_samples * 2) if n_mix else None

    def ingest_batch():   # new songs arrive between the query batches; their ids continue after the base corpus
        c0 = a.songs + mixed["songs"]
        synth_tracks(ctx, a.corpus, c0, n_mix, n_samples, a.tone_amp, a.noise_amp, mix_pcm)
        ctx.sync()
        t0 = time.perf_counter()
        _, _, ho_m, _ = ctx.fingerprint_batch(mix_pcm, np.arange(n_mix + 1, dtype=np.uint64) * n_samples, fs=FS,
                                              pcm_device=True, out_key=kbuf, out_t1=tbuf, cap=cap)
        tbl.insert_clips(kbuf, tbuf, ho_m, sid0=1 + c0, device=True)
        t1_ = time.perf_counter()
        tbl.finalize()
        ctx.sync()
        mixed["finalize_s"] += time.perf_counter() - t1_
        mixed["seconds"] += time.perf_counter() - t0
        mixed["songs"] += n_mix
        mixed["batches"] += 1

    t_stream0 = time.perf_counter()
    r = run_queries(ctx, tbl, a.songs, n_samples, nq, qn, a.snr, a.match_batch, a.topn, a.tone_amp, a.noise_amp,
                    before_batch=ingest_batch if n_mix else None, corpus=a.corpus)
    lat = r["batch_ms"] / r["sizes"]
    t_match = r["t_match"]
    out = {"metric": "query_match_ms_per_query_batched", "value": float(np.median(lat)), "unit": "ms/query",
           "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)), "qps": nq / t_match,
           "batch_ms_p50": float(np.percentile(r["batch_ms"], 50)), "batch_ms_p99": float(np.percentile(r["batch_ms"], 99)),
           "ms_per_query_by_batch": [round(float(x), 5) for x in lat[:64]],
           "higher_is_better": False, "n_gpus": 1, "data": "synthetic",
           "config": {"corpus": a.corpus, "workload": f"{a.songs} x {a.seconds:.0f} s {'music-like' if a.corpus == 'music' else 'tonal+noise'} tracks in one HBM table; {nq} x "
                                  f"{a.query_seconds:.0f} s queries at arbitrary offsets, SNR {a.snr} dB, batches of {a.match_batch}",
                      "songs": a.songs, "rows": int(rows), "queries": nq, "snr_db": a.snr, "shards": a.shards},
           "top1_accuracy": r["correct"] / nq, "hashes_per_query": r["hashes"] / nq, "pairs_per_query": r["pairs"] / nq,
           "rows_scanned_per_query": r["rows_scanned"] / nq, "query_fingerprint_ms": r["query_fingerprint_s"] / nq * 1e3,
           "match_alg_GBs": (8 * r["rows_scanned"] + 16 * r["distinct_keys"]) / t_match / 1e9,
           "build": build, "extract_stats": ctx.extract_stats()}
    if n_mix:
        t_stream = time.perf_counter() - t_stream0   # includes query synthesis / mixing on the device
        out["mixed"] = {"ingest_batch_songs": n_mix, "batches": mixed["batches"], "songs_ingested": mixed["songs"],
                        "rows_after": int(tbl.rows()[0]), "ingest_s_per_batch": mixed["seconds"] / max(mixed["batches"], 1),
                        "finalize_s_per_batch": mixed["finalize_s"] / max(mixed["batches"], 1),
                        "stream_seconds": t_stream, "sustained_ingest_songs_per_s": mixed["songs"] / t_stream,
                        "sustained_qps": nq / t_stream,
                        "note": "one GPU alternating ingest batches and query batches; replicas multiply the QPS, the "
                                "ingest is repeated on every replica"}
        mix_pcm.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
