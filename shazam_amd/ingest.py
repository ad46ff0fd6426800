"""Sharded database build: tracks are partitioned over the ranks (one process per
GPU), every rank fingerprints its own block on its GPU and stages rows (key32, song_id, offset),
seals them into sorted runs that travel to every peer over RCCL/xGMI while the next batch is fingerprinted;
one k-way merge of all runs leaves the same node-global table on every GPU.

The reference's only parallelism is the file-level multiprocessing.Pool of
fingerprint_directory (__init__.py:335-357); song ids there are MySQL auto-increment values in
completion order.  Here ids are deterministic, ``song_id = global track index + 1``
(1-based like mysql_database.py:34,200), so the table is identical for every rank count.
"""
from __future__ import annotations

import numpy as np


def shard_tracks(n_tracks: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of track indices owned by ``rank``; sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(int(n_tracks), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def song_id_of_track(track_index: int) -> int:
    return int(track_index) + 1


def merge_rows(parts):
    """Reference semantics of the merged table: concatenation of every rank's rows, sorted by
    (key, song_id, offset), duplicates dropped (UNIQUE(song_id, offset, hash) + INSERT IGNORE,
    mysql_database.py:54-68).  Host-side statement of what shz_table_allgather leaves on the
    device; used by the multi-process tests."""
    rows = np.concatenate([np.stack([np.asarray(k, np.uint64), np.asarray(s, np.uint64), np.asarray(o, np.uint64)], 1)
                           for k, s, o in parts if len(k)] or [np.zeros((0, 3), np.uint64)])
    rows = np.unique(rows, axis=0)
    return rows[:, 0].astype(np.uint32), rows[:, 1].astype(np.uint32), rows[:, 2].astype(np.uint32)


def pack_rows(key32, sid, off, sid_bits: int, off_bits: int) -> np.ndarray:
    """key << (sid_bits + off_bits) | sid << off_bits | off: the 8-byte form in which rows are sorted, travel between
    the ranks and are merged (csrc/shz_build.hip: tbl_compose1_kernel); its numeric order is the table's order."""
    return ((np.asarray(key32, np.uint64) << np.uint64(sid_bits + off_bits)) | (np.asarray(sid, np.uint64) << np.uint64(off_bits))
            | np.asarray(off, np.uint64))


def merge_sorted_runs(runs, sid_bits: int, off_bits: int, segment_rows: int = 1 << 31):
    """Host-side statement of what shz_table_allgather does with the ranks' SORTED packed runs: pairwise
    merges until one run is left, duplicates dropped, the run cut into segments of <= segment_rows input rows.
    Returns a list of (key32, sid, off) segments."""
    runs = [np.asarray(r, np.uint64) for r in runs]
    for r in runs:
        assert np.all(r[1:] >= r[:-1]), "every rank sorts its own rows before the exchange"
    while len(runs) > 1:
        nxt = []
        for i in range(0, len(runs), 2):
            if i + 1 == len(runs):
                nxt.append(runs[i])
            else:   # merge of two sorted runs, ties keep the left run's element first (tbl_merge_kernel)
                a, b = runs[i], runs[i + 1]
                out = np.empty(len(a) + len(b), np.uint64)
                ia = np.searchsorted(b, a, side="left") + np.arange(len(a))    # elements of b strictly below a[i]
                ib = np.searchsorted(a, b, side="right") + np.arange(len(b))   # elements of a at or below b[j]
                out[ia], out[ib] = a, b
                nxt.append(out)
        runs = nxt
    g = runs[0] if runs else np.zeros(0, np.uint64)
    segs = []
    m_s, m_o = np.uint64((1 << sid_bits) - 1), np.uint64((1 << off_bits) - 1)
    for o in range(0, len(g), segment_rows):
        c = g[o:o + segment_rows]
        keep = np.ones(len(c), bool)
        keep[1:] = c[1:] != c[:-1]
        if o > 0:
            keep[0] = c[0] != g[o - 1]
        c = c[keep]
        if len(c):
            segs.append(((c >> np.uint64(sid_bits + off_bits)).astype(np.uint32), ((c >> np.uint64(off_bits)) & m_s).astype(np.uint32),
                         (c & m_o).astype(np.uint32)))
    return segs


# ---- the exchange rounds of the gathered build, stated on the host (csrc/shz_build.hip: gx_round) -------------------
GX_MAXR, GX_HDR = 16, 8
GX_BLOCK = GX_HDR + 2 * GX_MAXR      # u64 words a rank contributes to a round
GXF_GENERAL, GXF_MORE, GXF_FINISHING, GXF_CUT, GXF_HOLDS_RUNS, GXF_STAGED, GXF_BROKEN = 1, 2, 4, 8, 16, 32, 64
RUN_ROWS_MAX = (1 << 32) - 4096


def layout_for(max_sid: int, max_off: int, prev_off_bits: int = 0):
    """(sid_bits, off_bits) every rank packs in, or None when song id + offset need more than 32 bits: a pure function
    of the largest offset seen so far (run_layout) -- ranks that saw the same global maxima pack alike."""
    ob = max(int(max_off).bit_length() or 1, prev_off_bits)
    return (32 - ob, ob) if (int(max_sid).bit_length() or 1) + ob <= 32 else None


class GatherRounds:
    """Host-side statement of the exchange rounds shz_table_exchange_run / shz_table_allgather run on the device: the same
    320-byte blocks, the same verdicts, the same termination rule, with ``allgather(obj) -> list`` (any collective that
    returns every rank's object in rank order) in place of RCCL and numpy arrays in place of the run arena.  It is what
    the multi-process CPU tests run (gloo), and the specification the device code is read against.

    ``seal(key, sid, off)`` makes a sorted run of this rank's rows; ``exchange()`` is one round (runs not yet sent
    travel); ``finish()`` runs rounds until every rank is finishing and has nothing left, and returns the merged table.
    """

    def __init__(self, rank: int, world: int, allgather, run_rows: int = RUN_ROWS_MAX):
        self.rank, self.world, self.allgather = rank, world, allgather
        self.run_rows = min(int(run_rows), RUN_ROWS_MAX)
        self.max_sid = self.max_off = 0
        self.ob = 0
        self.local = []        # rows of runs made here and not yet sent: (key, sid, off) triples, sorted
        self.held = []         # every run this rank holds (its own that travelled, its peers'), as triples
        self.rounds = 0

    def seal(self, key, sid, off):
        key, sid, off = (np.asarray(a, np.uint32) for a in (key, sid, off))
        if len(key) == 0:
            return
        self.max_sid, self.max_off = max(self.max_sid, int(sid.max())), max(self.max_off, int(off.max()))
        for lo in range(0, len(key), self.run_rows):   # a run holds < 2^32 rows: more become several runs
            k, s_, o = key[lo:lo + self.run_rows], sid[lo:lo + self.run_rows], off[lo:lo + self.run_rows]
            idx = np.lexsort((o, s_, k))
            rows = np.unique(np.stack([k[idx], s_[idx], o[idx]], 1), axis=0)   # INSERT IGNORE inside the run
            self.local.append((rows[:, 0], rows[:, 1], rows[:, 2]))

    def _block(self, finishing: bool, general: bool, pending: bool, st_sid: int = 0, st_off: int = 0):
        ship = self.local[:GX_MAXR]
        blk = np.zeros(GX_BLOCK, np.uint64)
        blk[0] = len(self.local)
        blk[1], blk[2] = max(self.max_sid, st_sid), max(self.max_off, st_off)
        blk[4] = (GXF_GENERAL if general else 0) | (GXF_MORE if len(self.local) > len(ship) or pending else 0) | \
                 (GXF_FINISHING if finishing else 0) | (GXF_HOLDS_RUNS if self.local or self.held else 0)
        blk[5] = len(ship)
        for j, r in enumerate(ship):
            blk[GX_HDR + 2 * j] = len(r[0])
            blk[GX_HDR + 2 * j + 1] = int(r[1].min()) | (int(r[1].max()) << 32)
        return blk, ship

    def round(self, finishing: bool, general: bool = False, pending: bool = False, st_sid: int = 0, st_off: int = 0):
        blk, ship = self._block(finishing, general, pending, st_sid, st_off)
        got = self.allgather((blk, ship))          # (the device ships the block first, the runs after the verdict)
        self.rounds += 1
        blocks = [g[0] for g in got]
        v = {"any_general": any(int(b[4]) & GXF_GENERAL for b in blocks), "any_more": any(int(b[4]) & GXF_MORE for b in blocks),
             "any_runs": any(int(b[4]) & GXF_HOLDS_RUNS for b in blocks),
             "all_finishing": all(int(b[4]) & GXF_FINISHING for b in blocks),
             "rows_sent": sum(int(b[GX_HDR + 2 * j]) for b in blocks for j in range(int(b[5])))}
        gmax_sid, gmax_off = max(int(b[1]) for b in blocks), max(int(b[2]) for b in blocks)
        lay = layout_for(gmax_sid, gmax_off, self.ob)
        if lay is None:
            v["any_general"] = True
        if v["any_general"]:
            return v
        self.ob = lay[1]
        self.max_sid, self.max_off = max(self.max_sid, gmax_sid), max(self.max_off, gmax_off)
        for r, (b, runs) in enumerate(got):
            assert len(runs) == int(b[5]) and all(len(x[0]) == int(b[GX_HDR + 2 * j]) < RUN_ROWS_MAX for j, x in enumerate(runs))
            self.held.extend(runs)                  # a rank's own runs stay where they are; its peers' arrive
        self.local = self.local[len(ship):]
        return v

    def exchange(self):
        v = self.round(False)
        if v["any_general"]:
            raise RuntimeError("a rank's rows cannot travel as packed runs")
        return v

    def finish(self, staged=None):
        """staged: (key, sid, off) rows not yet sealed.  Returns the merged table (key, sid, off), or raises when the
        ranks' tables would differ (the device returns SHZ_E_STATE on every rank)."""
        n_st = 0 if staged is None else len(staged[0])
        st_sid = int(np.max(staged[1])) if n_st else 0
        st_off = int(np.max(staged[2])) if n_st else 0
        general = n_st > 0 and layout_for(max(self.max_sid, st_sid), max(self.max_off, st_off), self.ob) is None
        v = self.round(True, general, pending=n_st > 0, st_sid=st_sid, st_off=st_off)   # nothing sealed here yet
        if v["any_general"]:
            if v["any_runs"]:
                raise RuntimeError("column path needed while sealed runs wait on some rank")
            parts = self.allgather(staged if n_st else (np.zeros(0, np.uint32),) * 3)   # the column path: staged rows travel
            return merge_rows(parts)
        if n_st:
            self.seal(*staged)
        while not (v["all_finishing"] and not v["any_more"]):
            v = self.round(True)
            if v["any_general"]:
                raise RuntimeError("a rank turned to the column path after runs had travelled")
        sb, ob = 32 - self.ob, self.ob
        runs = [pack_rows(k, s_, o, sb, ob) for k, s_, o in self.held]
        segs = merge_sorted_runs(runs, sb, ob, segment_rows=1 << 62) if runs else []
        return segs[0] if segs else (np.zeros(0, np.uint32),) * 3


class ShardedBuilder:
    """Fingerprints this rank's block of tracks on its GPU and builds the node-global table.

    ``pcm_source(lo, hi)`` returns either a list of int16 arrays (host PCM) or a tuple
    ``(DevBuf, n_samples)`` of device-resident equal-length clips for tracks [lo, hi).

    The build is pipelined like the reference's (the parent inserts a song while the pool fingerprints the next,
    __init__.py:341, 357-386): whenever ``seal_rows`` rows are staged they are sealed into a sorted run, and with a
    communicator the run starts travelling to the peers (``Table.exchange_run``) while the next chunk is fingerprinted.
    Staging never holds more than ``seal_rows`` + one chunk of rows, a run is always < 2^32 rows, and the table holds its
    runs (``reserve(gather=True)``) so that ONE merge at the end cuts the segments by key range.  ``rows_hint``: expected
    rows of the WHOLE corpus (all ranks) -- the arenas are then allocated once, beside the first chunks.
    """

    def __init__(self, db, rank: int = 0, world: int = 1, comm=None, chunk_tracks: int = 256, seal_rows: int = 1_000_000_000):
        self.db, self.rank, self.world, self.comm, self.chunk = db, rank, world, comm, int(chunk_tracks)
        self.seal_rows = int(seal_rows)
        if world > 1 and comm is None:
            raise ValueError("world > 1 needs a communicator (shazam_amd._ffi.Comm)")

    def build(self, n_tracks: int, pcm_source, Fs: int = 44100, rows_hint: int = 0, reserve_wait: bool = False):
        import time

        import shazam_amd as S
        from . import _ffi
        lo, hi = shard_tracks(n_tracks, self.rank, self.world)
        ctx, tbl = self.db.ctx, self.db.table
        bulk = tbl.rows() == (0, 0)   # a fresh table: the bulk build (a table that holds rows takes the column path at the end)
        if bulk:
            batch = min(self.seal_rows, -(-int(rows_hint) // self.world)) if rows_hint else 0
            tbl.reserve(int(rows_hint), batch, gather=True, wait=reserve_wait)
        n_hashes = staged = runs = 0
        t_src = t_fp = t_ins = t_seal = 0.0
        kbuf = tbuf = None
        cap = 0
        t_begin = time.perf_counter()
        for c0 in range(lo, hi, self.chunk):
            c1 = min(c0 + self.chunk, hi)
            t0 = time.perf_counter()
            src = pcm_source(c0, c1)
            t1_ = time.perf_counter()
            if isinstance(src, tuple):   # device PCM: the fingerprints stay on the device too
                buf, n = src
                off = np.arange(c1 - c0 + 1, dtype=np.uint64) * int(n)
                need = (c1 - c0) * int(_ffi.lib().shz_frame_count(int(n))) * 24 + 1024
                while True:
                    if cap < need:
                        for b_ in (kbuf, tbuf):
                            if b_ is not None:
                                b_.free()
                        cap = need
                        kbuf, tbuf = ctx.alloc(cap * 4), ctx.alloc(cap * 4)
                    try:
                        _, _, ho, cnt = ctx.fingerprint_batch(buf, off, fs=Fs, pcm_device=True, out_key=kbuf, out_t1=tbuf, cap=cap)
                        break
                    except _ffi.ShzError as e:
                        if e.code != _ffi.E_CAPACITY:
                            raise
                        need = cap * 2
                t2 = time.perf_counter()
                self.db.insert_clips(kbuf, tbuf, ho, sid0=song_id_of_track(c0), device=True)
            else:
                k, t1, ho = S.fingerprint_batch(src, Fs, ctx=ctx)
                cnt = len(k)
                t2 = time.perf_counter()
                self.db.insert_clips(k, t1, ho, sid0=song_id_of_track(c0))
            t3 = time.perf_counter()
            n_hashes += cnt
            staged += cnt
            if bulk and staged >= self.seal_rows:
                if self.comm is not None:
                    tbl.exchange_run(self.comm)   # the run travels while the next chunk is fingerprinted
                else:
                    tbl.seal_run()
                staged, runs = 0, runs + 1
            t4 = time.perf_counter()
            t_src += t1_ - t0
            t_fp += t2 - t1_
            t_ins += t3 - t2
            t_seal += t4 - t3
        t0 = time.perf_counter()
        recv = 0
        if self.comm is not None:
            recv = tbl.allgather(self.comm)
            self.db._dirty = False
        else:
            self.db.finalize()
        ctx.sync()
        t_end = time.perf_counter()
        for b_ in (kbuf, tbuf):
            if b_ is not None:
                b_.free()
        return {"tracks": hi - lo, "hashes": n_hashes, "bytes_received": recv, "rows": tbl.rows()[0], "runs_sealed_on_the_way": runs,
                "seconds": t_end - t_begin, "source_s": t_src, "fingerprint_s": t_fp, "insert_s": t_ins, "seal_exchange_s": t_seal,
                "final_s": t_end - t0}


# ---------------------------------------------------------------------------------------------
# Ingest orchestration: fingerprint_directory with the reference's semantics
# (__init__.py:248-268, 286-393, 407-415), batched for the GPU.  mp3 decoding stays external
# (pydub/ffmpeg are not part of the hot path); WAV goes through the stdlib.
# ---------------------------------------------------------------------------------------------
import hashlib
import wave
from pathlib import Path


def unique_hash(file_path) -> str:
    """Identity of a file for the skip-if-already-ingested rule: SHA-1 of its bytes as upper-case hex, the form the
    songs table stores (__init__.py:305-323 defines the value; mysql_database.py:32-44 the column)."""
    h = hashlib.sha1()
    view = memoryview(bytearray(1 << 20))
    with open(file_path, "rb", buffering=0) as f:
        while n := f.readinto(view):
            h.update(view[:n])
    return h.hexdigest().upper()


def find_files(path, extensions):
    """[(file path, extension without dot)] for every file below ``path`` whose name ends in one of ``extensions``
    (given with or without the dot).  Same pairs as the reference's walk (__init__.py:286-303), in sorted order so
    that song ids do not depend on the directory's on-disk order."""
    wanted = {e.lstrip(".") for e in extensions}
    found = []
    for f in sorted(Path(path).rglob("*")):
        ext = f.suffix.lstrip(".")
        if ext in wanted and f.is_file():
            found.append((str(f), ext))
    return found


def read(file_name: str, limit: int = None):
    """(channels, frame_rate, file_sha1) like read() (__init__.py:70-113) for 16-bit PCM WAV files:
    channels de-interleaved as data[chn::n_channels], optional limit in seconds from the start."""
    with wave.open(file_name, "rb") as w:
        if w.getsampwidth() != 2:
            raise NotImplementedError(f"{file_name}: only 16-bit PCM WAV is read here ({8 * w.getsampwidth()}-bit given)")
        n = w.getnframes()
        if limit:
            n = min(n, int(limit * w.getframerate()))
        data = np.frombuffer(w.readframes(n), np.int16)
        nch, fr = w.getnchannels(), w.getframerate()
    return [np.ascontiguousarray(data[c::nch]) for c in range(nch)], fr, unique_hash(file_name)


def load_fingerprinted_audio_hashes(db, songhashes_set=None):
    """file SHA-1s of every fingerprinted song (__init__.py:407-415; FIELD_FILE_SHA1 = column 2)."""
    songhashes_set = set() if songhashes_set is None else songhashes_set
    for song in db.get_songs():
        songhashes_set.add(song[2])
    return songhashes_set


def fingerprint_directory(path: str, extensions, db, songhashes_set=None, limit: int = None, batch_files: int = 64,
                          reader=read):
    """Fingerprint every matching file not yet in ``db`` (by file SHA-1) and insert it, with the
    reference's bookkeeping (__init__.py:325-393): per file the fingerprints of all channels are
    united as a set (:254-265), ``insert_song(name, sha1, len(set))`` (:381), ``insert_hashes``,
    ``set_song_fingerprinted``.  Files are decoded on the host and fingerprinted ``batch_files`` at a
    time in one GPU batch instead of one process-pool task per file.  Returns [(song_id, name, n)]."""
    import shazam_amd as S
    songhashes_set = load_fingerprinted_audio_hashes(db, songhashes_set)
    todo = []
    for filename, _ in find_files(path, extensions):
        if unique_hash(filename) in songhashes_set:   # don't refingerprint (__init__.py:346-348)
            continue
        todo.append(filename)
    done = []
    for b0 in range(0, len(todo), batch_files):
        files, chans, owner, rates = todo[b0:b0 + batch_files], [], [], []
        meta = []
        for fi, fn in enumerate(files):
            try:
                channels, fs, file_hash = reader(fn, limit)
            except Exception as e:  # the reference prints and skips failed files (__init__.py:373-376)
                print(f"Failed fingerprinting {fn}: {e}")
                continue
            meta.append((fi, fn, fs, file_hash, len(chans), len(channels)))
            chans.extend(channels)
            rates.append(fs)
        # one GPU batch per distinct sample rate (Fs only scales the spectrogram; hashes do not depend on it)
        for fs in sorted(set(rates)):
            sel = [m for m in meta if m[2] == fs]
            flat = [c for m in sel for c in chans[m[4]:m[4] + m[5]]]
            if not flat:
                continue
            k, t1, ho = S.fingerprint_batch(flat, Fs=fs, ctx=db.ctx)
            pos = 0
            for (_fi, fn, _fs, file_hash, _c0, nch) in sel:
                lo, hi = int(ho[pos]), int(ho[pos + nch])
                pos += nch
                pairs = np.unique((k[lo:hi].astype(np.uint64) << np.uint64(32)) | t1[lo:hi].astype(np.uint64))
                song_name = Path(fn).stem
                sid = db.insert_song(song_name, file_hash, len(pairs))
                db.insert_keys(sid, (pairs >> np.uint64(32)).astype(np.uint32), (pairs & np.uint64(0xFFFFFFFF)).astype(np.uint32))
                db.set_song_fingerprinted(sid)
                songhashes_set.add(file_hash)
                done.append((sid, song_name, len(pairs)))
    db.finalize()
    return done
