"""Key-sharded fingerprints table: for databases larger than one GPU's HBM.

Rows are partitioned by a hash of key32 (``shard_of_keys``).  A DB row ``(hash, song_id, offset)`` lives on
exactly one shard, so what ``align_matches`` needs is a sum over shards: ``dedup_hashes[sid]`` counts DB rows
(`recognizer.py:261-264`) and ``counts[(sid, offset difference)]`` counts matches (`recognizer.py:305`).
Every shard votes on its own rows (``shz_match_votes``), the records are gathered, and one merge applies the
reference's ranking (``shz_votes_merge``) -- the result equals ``Table.match`` on the unsharded table bit for bit
(tests/test_gpu_shard.py).

Two deployments share the code:
* ``ShardedTable(ctx, nshards=S)``            -- S shards on one GPU (tests; a table split to bound sort scratch);
* ``ShardedTable(ctx, comm=Comm(...))``       -- one shard per rank, rows routed by an RCCL all-to-all
  (``shz_table_shard_exchange``), vote records collected by an all-gather (``shz_votes_allgather``).  All ranks
  call ``match`` with the same queries (SPMD) and every rank gets the full result.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import E_CAPACITY, IN_DEVICE, OUT_DEVICE, lib, ptr, u64p

VOTE_COLS = (("q", np.uint32), ("sid", np.uint32), ("delta", np.int32), ("cnt", np.uint32), ("dedup", np.uint32))


def shard_of_keys(key32, nshards: int) -> np.ndarray:
    """Shard index of each key: the library's function, evaluated on the host (no GPU needed)."""
    k = np.ascontiguousarray(key32, np.uint32)
    out = np.empty(len(k), np.uint32)
    rc = lib().shz_shard_of_keys(ptr(k), len(k), int(nshards), ptr(out))
    if rc != _ffi.OK:
        raise _ffi.ShzError(rc, "shz_shard_of_keys: invalid arguments")
    return out


def shard_of_keys_numpy(key32, nshards: int) -> np.ndarray:
    """numpy twin of ``shard_of`` in shz_table.hip (used by the CPU tests to pin the function)."""
    k = np.asarray(key32, np.uint32)
    h = ((k ^ (k >> np.uint32(15))).astype(np.uint64) * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    return ((h >> np.uint64(10)) % np.uint64(nshards)).astype(np.uint32)


def match_votes(table: "_ffi.Table", key32, q_off, query_off, device: bool = False):
    """Vote records of one (shard) table for CSR queries.  Returns (cols, n, nhash, npairs): host arrays, or
    DevBufs when ``device`` (for the all-gather)."""
    ctx = table.ctx
    k = np.ascontiguousarray(key32, np.uint32)
    o = np.ascontiguousarray(q_off, np.uint32)
    qo = np.ascontiguousarray(query_off, np.uint64)
    nq = len(qo) - 1
    nhash, npairs = np.zeros(nq, np.uint32), np.zeros(nq, np.uint64)
    cap = max(1024, 2 * len(k))
    while True:
        cnt = C.c_uint64()
        if device:
            cols = [ctx.alloc(cap * 4) for _ in VOTE_COLS]
        else:
            cols = [np.empty(cap, dt) for _, dt in VOTE_COLS]
        rc = lib().shz_match_votes(ctx.h, table.h, ptr(k), ptr(o), qo.ctypes.data_as(u64p), nq, OUT_DEVICE if device else 0,
                                   *[ptr(c) for c in cols], cap, C.byref(cnt), ptr(nhash), ptr(npairs))
        if rc == E_CAPACITY:
            if device:
                for c in cols:
                    c.free()
            cap = int(cnt.value)
            continue
        ctx.check(rc)
        n = int(cnt.value)
        if not device:
            cols = [c[:n] for c in cols]
        return cols, n, nhash, npairs


def votes_merge(ctx: "_ffi.Context", cols, n: int, n_queries: int, topn: int = 2, device: bool = False):
    """Sum the records of equal (query, song, difference) and rank like align_matches."""
    res = {"sid": np.zeros((n_queries, topn), np.uint32), "delta": np.zeros((n_queries, topn), np.int32),
           "aligned": np.zeros((n_queries, topn), np.uint32), "dedup": np.zeros((n_queries, topn), np.uint32),
           "nres": np.zeros(n_queries, np.uint32)}
    if n == 0:
        cols = [None] * len(VOTE_COLS)
    elif not device:
        cols = [np.ascontiguousarray(c[:n], dt) for c, (_, dt) in zip(cols, VOTE_COLS)]
    ctx.check(lib().shz_votes_merge(ctx.h, *[ptr(c) for c in cols], n, n_queries, topn, IN_DEVICE if device else 0,
                                    ptr(res["sid"]), ptr(res["delta"]), ptr(res["aligned"]), ptr(res["dedup"]),
                                    ptr(res["nres"])))
    return res


class ShardedTable:
    """Fingerprints table partitioned by key.  Mirrors ``Table``'s insert / finalize / match surface."""

    def __init__(self, ctx: "_ffi.Context", nshards: int = None, comm: "_ffi.Comm" = None):
        if (nshards is None) == (comm is None):
            raise ValueError("give either nshards (shards on this GPU) or comm (one shard per rank)")
        self.ctx, self.comm = ctx, comm
        self.nshards = comm.nranks if comm is not None else int(nshards)
        if self.nshards < 1:
            raise ValueError("nshards must be >= 1")
        # with a communicator this rank holds one shard; otherwise all of them
        self.tables = [_ffi.Table(ctx) for _ in range(1 if comm is not None else self.nshards)]
        self.bytes_received = 0
        self._stage = None       # staging table of insert_clips (shards on this GPU)
        self._vcap = 0           # capacity of the device vote columns kept between match() calls
        self._vcols = None

    def close(self):
        for t in self.tables + ([self._stage] if self._stage is not None else []):
            t.close()
        for b in self._vcols or []:
            b.free()
        self.tables, self._stage, self._vcols = [], None, None

    # -- build ---------------------------------------------------------------------------------
    def insert(self, key32, sid, off):
        """Stage rows (host arrays).  Local shards: routed now; with a communicator: routed at finalize."""
        k = np.ascontiguousarray(key32, np.uint32)
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(sid, np.uint32), k.shape))
        o = np.ascontiguousarray(off, np.uint32)
        if self.comm is not None:
            self.tables[0].insert(k, s, o)
            return
        sh = shard_of_keys(k, self.nshards)
        for i, t in enumerate(self.tables):
            m = sh == i
            if m.any():
                t.insert(k[m], s[m], o[m])

    def insert_clips(self, key32, t1, hash_off, sid0, device=False):
        """Stage the CSR output of fingerprint_batch (song id of clip c = sid0 + c)."""
        if self.comm is not None:
            self.tables[0].insert_clips(key32, t1, hash_off, sid0, device=device)
            return
        if self._stage is None:
            self._stage = _ffi.Table(self.ctx)
        self._stage.insert_clips(key32, t1, hash_off, sid0, device=device)   # expand once on the device ...
        for i, t in enumerate(self.tables):                                  # ... and deal the rows out by key
            self.ctx.check(lib().shz_table_stage_from(t.h, self._stage.h, i, self.nshards))
        self.ctx.check(lib().shz_table_clear_staged(self._stage.h))

    def finalize(self):
        if self.comm is not None:
            b = C.c_uint64()
            self.ctx.check(lib().shz_table_shard_exchange(self.tables[0].h, self.comm.h, C.byref(b)))
            self.bytes_received += b.value
        else:
            for t in self.tables:
                t.finalize()

    def rows(self):
        """(rows held here, staged rows held here)"""
        r = [t.rows() for t in self.tables]
        return sum(a for a, _ in r), sum(b for _, b in r)

    # -- query ---------------------------------------------------------------------------------
    def match(self, key32, q_off, query_off, topn=2):
        """Same result dict as ``Table.match`` on the union of all shards."""
        qo = np.ascontiguousarray(query_off, np.uint64)
        nq = len(qo) - 1
        ctx = self.ctx
        if self.comm is None:
            return self._match_local(key32, q_off, qo, nq, topn)
        cols, n, nhash, npairs = match_votes(self.tables[0], key32, q_off, qo, device=True)
        # counts first, then the records themselves
        tot = C.c_uint64()
        cap = None
        gathered = None
        while True:
            if cap is None:
                # a zero-capacity call returns the total through E_CAPACITY (or succeeds when there is nothing)
                rc = lib().shz_votes_allgather(self.comm.h, n, *[ptr(c) for c in cols], None, None, None, None, None, 0,
                                               C.byref(tot))
                if rc == E_CAPACITY:
                    cap = int(tot.value)
                    continue
                ctx.check(rc)
                gathered, cap = [], 0
                break
            gathered = [ctx.alloc(max(cap, 1) * 4) for _ in VOTE_COLS]
            ctx.check(lib().shz_votes_allgather(self.comm.h, n, *[ptr(c) for c in cols], *[ptr(g) for g in gathered], cap,
                                                C.byref(tot)))
            break
        res = votes_merge(ctx, gathered, int(tot.value), nq, topn, device=True) if cap else votes_merge(ctx, [], 0, nq, topn)
        for b in list(cols) + list(gathered):
            b.free()
        res["nhash"] = nhash
        res["npairs"] = npairs  # this rank's share; sum over ranks = matches against the whole table
        return res

    def _match_local(self, key32, q_off, qo, nq, topn):
        """Shards on this GPU: every shard appends its records to device columns, one merge reads them there."""
        ctx = self.ctx
        k = np.ascontiguousarray(key32, np.uint32)
        o = np.ascontiguousarray(q_off, np.uint32)
        nhash = np.zeros(nq, np.uint32)
        if self._vcap == 0:
            self._grow_votes(max(1 << 16, 4 * len(k)))
        while True:
            n, need, npairs = 0, 0, np.zeros(nq, np.uint64)
            for t in self.tables:
                cnt, np_s = C.c_uint64(), np.zeros(nq, np.uint64)
                room = max(self._vcap - n, 0)
                rc = lib().shz_match_votes(ctx.h, t.h, ptr(k), ptr(o), qo.ctypes.data_as(u64p), nq, OUT_DEVICE,
                                           *[ptr(b.ptr + 4 * n) for b in self._vcols], room, C.byref(cnt), ptr(nhash),
                                           ptr(np_s))
                if rc not in (_ffi.OK, E_CAPACITY):
                    ctx.check(rc)
                need += int(cnt.value)
                if rc == _ffi.OK:
                    n += int(cnt.value)
                npairs += np_s
            if need <= self._vcap:
                break
            self._grow_votes(need + need // 4)     # a shard did not fit: grow once and vote again
        res = votes_merge(ctx, self._vcols, n, nq, topn, device=True)
        res["nhash"] = nhash           # a property of the query alone
        res["npairs"] = npairs         # matches add up over shards
        return res

    def _grow_votes(self, cap):
        for b in self._vcols or []:
            b.free()
        self._vcols = [self.ctx.alloc(cap * 4) for _ in VOTE_COLS]
        self._vcap = cap

    # -- the rest of Table's surface (shards on this GPU only) ------------------------------------
    def _local_only(self, what):
        if self.comm is not None:
            raise NotImplementedError(f"{what} on a rank-sharded table needs a collective; only match() is distributed")

    def set_segment_rows(self, rows: int):
        for t in self.tables:
            t.set_segment_rows(rows)

    def lookup(self, keys):
        """Rows of the listed keys, grouped in key-list order (a key lives on one shard)."""
        self._local_only("lookup")
        kk = np.ascontiguousarray(keys, np.uint32)
        uniq = np.unique(kk)
        sh = shard_of_keys(uniq, self.nshards)
        rows = {}
        for i, t in enumerate(self.tables):
            mine = uniq[sh == i]
            if len(mine) == 0:
                continue
            k, s, o = t.lookup(mine)                       # grouped by key in the (sorted, distinct) order given
            cut = np.searchsorted(k, mine, side="left")    # k is non-decreasing here
            end = np.searchsorted(k, mine, side="right")
            for key, a, b in zip(mine.tolist(), cut.tolist(), end.tolist()):
                rows[key] = (s[a:b], o[a:b])
        ks, ss, os_ = [], [], []
        for key in kk.tolist():
            s, o = rows.get(key, (np.zeros(0, np.uint32), np.zeros(0, np.uint32)))
            ks.append(np.full(len(s), key, np.uint32))
            ss.append(s)
            os_.append(o)
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.uint32)  # noqa: E731
        return cat(ks), cat(ss), cat(os_)

    def export(self):
        """All rows sorted by (key, sid, off), like Table.export."""
        self._local_only("export")
        parts = [t.export() for t in self.tables]
        k, s, o = (np.concatenate([p[i] for p in parts]) for i in range(3))
        order = np.lexsort((o, s, k))
        return k[order], s[order], o[order]

    def song_rows(self, sid) -> int:
        self._local_only("song_rows")
        return sum(t.song_rows(sid) for t in self.tables)

    def match_stats(self):
        """Counters of the LAST shard's vote pass (rows scanned / pairs / distinct keys)."""
        return self.tables[-1].match_stats()
