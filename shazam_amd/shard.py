"""Key-sharded fingerprints table: for databases larger than one GPU's HBM.

Rows are partitioned by a hash of key32 (``shard_of_keys``).  A DB row ``(hash, song_id, offset)`` lives on
exactly one shard, so what ``align_matches`` needs is a sum over shards: ``dedup_hashes[sid]`` counts DB rows
(`recognizer.py:261-264`) and ``counts[(sid, offset difference)]`` counts matches (`recognizer.py:305`).
A shard looks up only the query hashes it owns and emits the votes of its rows, 8 bytes each, in a key layout all
shards share (``shz_match_pairs``); the votes are gathered and the normal tail of the match -- one sort, run
lengths, per-(query, song) fold, top-n -- runs once over all of them (``shz_pairs_vote``).  These are the kernels
of ``Table.match``, so the result equals the unsharded table's bit for bit (tests/test_gpu_shard.py).

Two deployments share the code:
* ``ShardedTable(ctx, nshards=S)``            -- S shards on one GPU (tests; a table split to bound sort scratch);
* ``ShardedTable(ctx, comm=Comm(...))``       -- one shard per rank, rows routed by an RCCL all-to-all
  (``shz_table_shard_exchange``), votes collected by an all-gather (``shz_pairs_allgather``).  All ranks call
  ``match`` with the same queries (SPMD) and every rank gets the full result.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import E_CAPACITY, IN_DEVICE, lib, ptr, u64p


def shard_of_keys(key32, nshards: int) -> np.ndarray:
    """Shard index of each key: the library's function, evaluated on the host (no GPU needed)."""
    k = np.ascontiguousarray(key32, np.uint32)
    out = np.empty(len(k), np.uint32)
    rc = lib().shz_shard_of_keys(ptr(k), len(k), int(nshards), ptr(out))
    if rc != _ffi.OK:
        raise _ffi.ShzError(rc, "shz_shard_of_keys: invalid arguments")
    return out


def shard_of_keys_numpy(key32, nshards: int) -> np.ndarray:
    """numpy twin of ``shard_of`` in shz_table_int.h (used by the CPU tests to pin the function)."""
    k = np.asarray(key32, np.uint32)
    h = ((k ^ (k >> np.uint32(15))).astype(np.uint64) * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    return ((h >> np.uint64(10)) % np.uint64(nshards)).astype(np.uint32)


def _bits(v: int) -> int:
    return max(1, int(v).bit_length())


def table_maxima(table: "_ffi.Table"):
    """(largest song id, largest offset) of a finalized table -- they size the packed vote key."""
    a, b = C.c_uint32(), C.c_uint32()
    table.ctx.check(lib().shz_table_maxima(table.h, C.byref(a), C.byref(b)))
    return a.value, b.value


def vote_layout(max_sid: int, max_off: int, q_off, n_queries: int):
    """(sid_bits, delta_bits, bias) every shard packs its votes with; raises if the key does not fit 64 bits."""
    q_off = np.asarray(q_off)
    bias = int(q_off.max()) if q_off.size else 0
    sid_bits, delta_bits = _bits(max_sid), _bits(max_off + bias)
    if _bits(max(n_queries - 1, 0)) + sid_bits + delta_bits + 1 > 64:
        raise _ffi.ShzError(_ffi.E_UNSUPPORTED, f"{n_queries} queries x {sid_bits} song-id bits x {delta_bits} offset bits "
                                                "do not fit the 64-bit vote key; match fewer queries per call")
    return sid_bits, delta_bits, bias


def pairs_vote(ctx: "_ffi.Context", d_pairs, n: int, n_queries: int, layout, topn: int = 2):
    """Rank the packed votes in a device buffer (overwritten) like align_matches."""
    res = {"sid": np.zeros((n_queries, topn), np.uint32), "delta": np.zeros((n_queries, topn), np.int32),
           "aligned": np.zeros((n_queries, topn), np.uint32), "dedup": np.zeros((n_queries, topn), np.uint32),
           "nres": np.zeros(n_queries, np.uint32)}
    ctx.check(lib().shz_pairs_vote(ctx.h, ptr(d_pairs) if n else None, n, n_queries, *layout, topn, ptr(res["sid"]),
                                   ptr(res["delta"]), ptr(res["aligned"]), ptr(res["dedup"]), ptr(res["nres"])))
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
        self._pbuf, self._pcap = None, 0    # device vote buffer kept between match() calls
        self.last_votes = 0      # votes ranked by the last match()

    def close(self):
        for t in self.tables + ([self._stage] if self._stage is not None else []):
            t.close()
        if self._pbuf is not None:
            self._pbuf.free()
        self.tables, self._stage, self._pbuf, self._pcap = [], None, None, 0

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

    def seal_run(self):
        """Bound the staged rows of a long build (shz_table_seal_run on every local shard; with a communicator the rows
        stay staged until the exchange)."""
        if self.comm is None:
            for t in self.tables:
                t.seal_run()

    def delete_songs(self, sids) -> int:
        """ON DELETE CASCADE on every shard held here (with a communicator: call it on every rank)."""
        return sum(t.delete_songs(sids) for t in self.tables)

    def clear(self):
        for t in self.tables:
            t.clear()

    def rows(self):
        """(rows held here, staged rows held here)"""
        r = [t.rows() for t in self.tables]
        return sum(a for a, _ in r), sum(b for _, b in r)

    # -- query ---------------------------------------------------------------------------------
    def _votes_into(self, k, o, qo, nq, layout, shard_ids):
        """Run shz_match_pairs for the listed (shard index, table) pairs, appending into the device buffer.
        Returns (n_votes, nhash, npairs) with the per-query counts summed over those shards."""
        ctx = self.ctx
        if self._pbuf is None:
            self._grow(max(1 << 16, 8 * len(k)))
        dk = do = None
        if len(shard_ids) > 1 and len(k):      # several passes over the same queries: upload them once
            dk, do = ctx.alloc(k.nbytes), ctx.alloc(o.nbytes)
            dk.upload(k)
            do.upload(o)
        try:
            return self._votes_passes(dk if dk is not None else k, do if do is not None else o, dk is not None, qo, nq,
                                      layout, shard_ids)
        finally:
            for b in (dk, do):
                if b is not None:
                    b.free()

    def _votes_passes(self, k, o, on_device, qo, nq, layout, shard_ids):
        ctx = self.ctx
        while True:
            n, need = 0, 0
            nhash, npairs = np.zeros(nq, np.uint32), np.zeros(nq, np.uint64)
            for i, t in shard_ids:
                cnt, nh_s, np_s = C.c_uint64(), np.zeros(nq, np.uint32), np.zeros(nq, np.uint64)
                rc = lib().shz_match_pairs(ctx.h, t.h, ptr(k), ptr(o), qo.ctypes.data_as(u64p), nq, IN_DEVICE if on_device else 0, i,
                                           self.nshards, *layout,
                                           ptr(self._pbuf.ptr + 8 * n), max(self._pcap - n, 0), C.byref(cnt), ptr(nh_s), ptr(np_s))
                if rc not in (_ffi.OK, E_CAPACITY):
                    ctx.check(rc)
                need += int(cnt.value)
                if rc == _ffi.OK:
                    n += int(cnt.value)
                nhash += nh_s
                npairs += np_s
            if need <= self._pcap:
                return n, nhash, npairs
            self._grow(need + need // 4)     # a shard did not fit: grow once and run the pass again

    def _grow(self, cap):
        if self._pbuf is not None:
            self._pbuf.free()
        self._pbuf, self._pcap = self.ctx.alloc(cap * 8), cap

    def match(self, key32, q_off, query_off, topn=2):
        """Same result dict as ``Table.match`` on the union of all shards."""
        ctx = self.ctx
        k = np.ascontiguousarray(key32, np.uint32)
        o = np.ascontiguousarray(q_off, np.uint32)
        qo = np.ascontiguousarray(query_off, np.uint64)
        nq = len(qo) - 1
        mx = [table_maxima(t) for t in self.tables]     # after a shard exchange: maxima of the whole table
        layout = vote_layout(max(a for a, _ in mx), max(b for _, b in mx), o, nq)
        if self.comm is None:
            n, nhash, npairs = self._votes_into(k, o, qo, nq, layout, list(enumerate(self.tables)))
            res = pairs_vote(ctx, self._pbuf, n, nq, layout, topn)
            self.last_votes = n
        else:
            n, nhash, npairs = self._votes_into(k, o, qo, nq, layout, [(self.comm.rank, self.tables[0])])
            tot = C.c_uint64()
            # a zero-capacity call returns the total through E_CAPACITY (or succeeds when there is nothing at all)
            rc = lib().shz_pairs_allgather(self.comm.h, n, ptr(self._pbuf), None, 0, C.byref(tot))
            if rc != E_CAPACITY:
                ctx.check(rc)
            total = int(tot.value)
            gathered = ctx.alloc(max(total, 1) * 8)
            if total:
                ctx.check(lib().shz_pairs_allgather(self.comm.h, n, ptr(self._pbuf), ptr(gathered), total, C.byref(tot)))
            res = pairs_vote(ctx, gathered, total, nq, layout, topn)
            gathered.free()
            self.last_votes = total
            # nhash / npairs: this rank's share; their sum over the ranks is the unsharded table's value
        res["nhash"], res["npairs"] = nhash, npairs
        return res

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
        """Counters of the last match(): votes ranked; rows scanned / distinct keys of the last shard's pass."""
        st = self.tables[-1].match_stats()
        st["pairs"] = self.last_votes
        return st
