"""Sharded database build: tracks are partitioned over the ranks (one process per
GPU), every rank fingerprints its own block on its GPU and stages rows (key32, song_id, offset),
then ONE exchange step -- an RCCL all-gather of the staged rows over xGMI -- leaves the same
node-global table on every GPU.

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


class ShardedBuilder:
    """Fingerprints this rank's block of tracks on its GPU and builds the node-global table.

    ``pcm_source(lo, hi)`` returns either a list of int16 arrays (host PCM) or a tuple
    ``(DevBuf, n_samples)`` of device-resident equal-length clips for tracks [lo, hi).
    """

    def __init__(self, db, rank: int = 0, world: int = 1, comm=None, chunk_tracks: int = 256):
        self.db, self.rank, self.world, self.comm, self.chunk = db, rank, world, comm, int(chunk_tracks)
        if world > 1 and comm is None:
            raise ValueError("world > 1 needs an RCCL communicator (shazam_amd._ffi.Comm)")

    def build(self, n_tracks: int, pcm_source, Fs: int = 44100):
        import shazam_amd as S
        lo, hi = shard_tracks(n_tracks, self.rank, self.world)
        ctx = self.db.ctx
        n_hashes = 0
        for c0 in range(lo, hi, self.chunk):
            c1 = min(c0 + self.chunk, hi)
            src = pcm_source(c0, c1)
            if isinstance(src, tuple):
                buf, n = src
                off = np.arange(c1 - c0 + 1, dtype=np.uint64) * int(n)
                k, t1, ho, cnt = ctx.fingerprint_batch(buf, off, fs=Fs, pcm_device=True)
            else:
                k, t1, ho = S.fingerprint_batch(src, Fs, ctx=ctx)
                cnt = len(k)
            self.db.insert_clips(k, t1, ho, sid0=song_id_of_track(c0))
            n_hashes += cnt
        recv = 0
        if self.comm is not None:
            recv = self.db.table.allgather(self.comm)
            self.db._dirty = False
        else:
            self.db.finalize()
        return {"tracks": hi - lo, "hashes": n_hashes, "bytes_received": recv, "rows": self.db.table.rows()[0]}
