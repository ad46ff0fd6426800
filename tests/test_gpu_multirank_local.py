"""GPU: the N > 1 logic of the sharded database build on ONE GPU ("configs[2] / [4] untested on hardware",
"the collectives have only ever run with nranks = 1").

Ranks are threads of this process, one context each, over the in-process transport (shz_comm_create_local: the same rank
protocol as over RCCL -- counts, maxima and path flags all-gathered, runs placed behind each other, one k-way merge -- with
rendezvous + device copies instead of ncclSend/ncclRecv).  What RCCL itself moves is not tested here; everything around it is.

* replicated table: every rank fingerprints its block of tracks (ingest.shard_tracks, song_id = track + 1),
  shz_table_allgather: all ranks end with the same table, equal to the one-rank build row for row, and answer queries alike;
  also: a rank without rows, a rank that sealed runs on the way, ids too wide to pack (every rank takes the column path),
  a rank whose table already holds rows (ditto);
* the gathered build at the shapes BASELINE's configs put it in: a rank that seals several times a
  segment's worth, lists of runs per rank (several rounds, more runs than one merge takes), runs that travel while the next
  batch is staged (exchange_run, ranks calling it different numbers of times), a layout that widens mid-build; and the two
  situations in which the tables WOULD differ -- wide ids on one rank with sealed runs on another, a table that was not
  reserved for gathering and has cut segments -- fail on every rank alike;
* ShardedBuilder (ingest.py) over thread ranks == its one-rank build;
* key-sharded table: all-to-all build + per-shard votes + all-gather of the votes == the unsharded table."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_ranks(world, fn):
    """fn(rank) on `world` threads; re-raises the first failure."""
    errs, outs = [None] * world, [None] * world

    def go(r):
        try:
            outs[r] = fn(r)
        except BaseException as e:  # noqa: BLE001
            errs[r] = e

    ths = [threading.Thread(target=go, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(600)
    for e in errs:
        if e is not None:
            raise e
    return outs


def _rows(rng, n, sid_lo, sid_hi, noff=600):
    k = (rng.integers(0, 400, n).astype(np.uint32) << np.uint32(20)) | (rng.integers(0, 400, n).astype(np.uint32) << np.uint32(8)) | \
        rng.integers(0, 6, n).astype(np.uint32)
    s = rng.integers(sid_lo, sid_hi, n).astype(np.uint32)
    o = rng.integers(0, noff, n).astype(np.uint32)
    idx = np.lexsort((o, s))
    return k[idx], s[idx], o[idx]


@pytest.mark.parametrize("world, case", [(2, "plain"), (3, "plain"), (4, "empty_rank"), (3, "sealed_on_the_way"), (3, "wide_ids"),
                                         (2, "one_table_holds_rows"), (8, "plain")])
def test_allgather_build_equals_one_rank_build(world, case):
    import shazam_amd as S
    from shazam_amd import _ffi
    from shazam_amd.ingest import shard_tracks
    n_tracks, per_track = 240, 900
    rng = np.random.default_rng(world * 10 + len(case))
    blocks = []
    for tr in range(n_tracks):
        sid = tr + 1 + ((1 << 24) if case == "wide_ids" else 0)
        k, s, o = _rows(rng, per_track, sid, sid + 1, noff=600 if case != "wide_ids" else 4000)
        blocks.append((k, s, o))
    pre = _rows(rng, 5000, 1000, 1100)          # rows one rank's table holds before the build ("one_table_holds_rows")
    want = np.unique(np.concatenate([np.stack(b, 1) for b in blocks] + ([np.stack(pre, 1)] if case == "one_table_holds_rows" else [])).astype(np.uint64), axis=0)
    qk = np.concatenate([b[0][:40] for b in blocks[::17]])
    qo = np.concatenate([b[2][:40] for b in blocks[::17]])
    qoff = np.arange(0, len(qk) + 1, 40).astype(np.uint64)
    gid = 1000 + world * 16 + sum(map(ord, case))

    def rank_fn(r):
        ctx = _ffi.Context(0)
        comm = _ffi.Comm.local(ctx, gid, r, world)
        tbl = S.Table(ctx)
        tbl.set_segment_rows(70000)
        lo, hi = shard_tracks(n_tracks, r, world)
        if case == "empty_rank" and r == 1:
            lo = hi = 0
        if case == "one_table_holds_rows":
            if r == 0:
                tbl.insert(*pre)
                tbl.finalize()
        for i, tr in enumerate(range(lo, hi)):
            if case == "empty_rank" and r == 0:
                pass
            tbl.insert(*blocks[tr])
            if case == "sealed_on_the_way" and r == 1 and i % 25 == 24:
                tbl.seal_run()
        if case == "empty_rank" and r == 0:     # rank 0 takes the tracks rank 1 gave up
            l1, h1 = shard_tracks(n_tracks, 1, world)
            for tr in range(l1, h1):
                tbl.insert(*blocks[tr])
        recv = tbl.allgather(comm)
        k, s, o = tbl.export()
        rows = np.stack([k, s, o], 1).astype(np.uint64)
        res = tbl.match(qk, qo, qoff, 3)
        st = tbl.build_stats()
        tbl.close()
        comm.close()
        ctx.close()
        return rows, res, recv, st

    outs = _run_ranks(world, rank_fn)
    for r, (rows, res, recv, st) in enumerate(outs):
        if case == "one_table_holds_rows" and r != 0:
            # the ranks whose tables were empty receive every staged row; rank 0's earlier rows stay rank 0's own
            exp = np.unique(np.concatenate([np.stack(b, 1) for b in blocks]).astype(np.uint64), axis=0)
            assert np.array_equal(np.unique(rows, axis=0), exp) and len(rows) == len(exp), r
            continue
        assert len(rows) == len(want) and np.array_equal(np.unique(rows, axis=0), want), (case, r)
        for name in outs[0][1]:
            assert np.array_equal(res[name], outs[0][1][name]), (case, r, name)
    if case == "plain":
        assert all(o[2] > 0 for o in outs) and all(o[3]["merge_s"] > 0 for o in outs)


def _build_ranks(world, gid, n_tracks, blocks, per_rank):
    """per_rank(r, tbl, comm, lo, hi, insert) stages / seals rank r's tracks; returns (rows, match result, recv, stats) per rank."""
    import shazam_amd as S
    from shazam_amd import _ffi
    from shazam_amd.ingest import shard_tracks
    qk = np.concatenate([b[0][:40] for b in blocks[::17]])
    qo = np.concatenate([b[2][:40] for b in blocks[::17]])
    qoff = np.arange(0, len(qk) + 1, 40).astype(np.uint64)

    def rank_fn(r):
        ctx = _ffi.Context(0)
        comm = _ffi.Comm.local(ctx, gid, r, world)
        tbl = S.Table(ctx)
        try:
            lo, hi = shard_tracks(n_tracks, r, world)
            per_rank(r, tbl, comm, lo, hi)
            recv = tbl.allgather(comm)
            k, s, o = tbl.export()
            rows = np.stack([k, s, o], 1).astype(np.uint64)
            return rows, tbl.match(qk, qo, qoff, 3), recv, tbl.build_stats(), tbl.segments(), tbl.exchange_stats()
        finally:
            tbl.close()
            comm.close()
            ctx.close()

    return _run_ranks(world, rank_fn)


def _check_equal_tables(outs, want):
    for r, out in enumerate(outs):
        rows = out[0]
        assert len(rows) == len(want) and np.array_equal(rows[np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0]))], want), r
        for name in outs[0][1]:
            assert np.array_equal(out[1][name], outs[0][1][name]), (r, name)


@pytest.mark.parametrize("world, case", [(3, "sealed_3x_segment"), (3, "many_runs"), (2, "two_rounds_of_runs"), (4, "pipelined"),
                                         (3, "pipelined_widening"), (2, "pipelined_one_rank_never")])
def test_gathered_build_with_lists_of_runs(world, case):
    n_tracks, per_track = 240, 900
    rng = np.random.default_rng(world * 7 + len(case))
    blocks = []
    for tr in range(n_tracks):
        # "widening": the last tracks of the LAST rank are long -- the layout every rank packs in grows mid-build
        noff = 5000 if case == "pipelined_widening" and tr >= n_tracks - 20 else 600
        blocks.append(_rows(rng, per_track, tr + 1, tr + 2, noff=noff))
    want = np.unique(np.concatenate([np.stack(b, 1) for b in blocks]).astype(np.uint64), axis=0)
    gid = 5000 + world * 16 + sum(map(ord, case))

    def per_rank(r, tbl, comm, lo, hi):
        tbl.set_segment_rows(20000)
        tbl.reserve(0, 0, gather=True)          # hold the runs: nothing is cut before the exchange
        if case == "many_runs":
            tbl.set_run_rows(5000)              # 72,000 rows a rank -> 15 runs a rank, 45 in all: more than one merge takes
        if case == "two_rounds_of_runs":
            tbl.set_run_rows(4000)              # 108,000 rows a rank -> 27 runs: two exchange rounds of <= 16
        for i, tr in enumerate(range(lo, hi)):
            tbl.insert(*blocks[tr])
            if case == "sealed_3x_segment" and r == 1 and i % 20 == 19:
                tbl.seal_run()                  # rank 1 seals 72,000 rows: 3.6 segments' worth
            if case.startswith("pipelined"):
                every = (7, 11, 13, 19)[r % 4]  # ranks call exchange_run different numbers of times
                if case == "pipelined_one_rank_never" and r == 1:
                    continue
                if i % every == every - 1:
                    tbl.exchange_run(comm)

    outs = _build_ranks(world, gid, n_tracks, blocks, per_rank)
    _check_equal_tables(outs, want)
    for out in outs:
        assert out[4] >= 2                      # several segments (cut by key range)
        assert out[2] > 0
    if case == "two_rounds_of_runs":
        assert all(o[5]["rounds"] >= 2 for o in outs)
    if case.startswith("pipelined"):
        assert all(o[5]["rounds"] >= 3 for o in outs)


@pytest.mark.parametrize("case", ["wide_ids_and_sealed_runs", "unreserved_table_cut_segments", "hold_mode_wide_seal"])
def test_gathered_build_fails_on_every_rank_instead_of_diverging(case):
    import shazam_amd as S
    from shazam_amd import _ffi
    from shazam_amd.ingest import shard_tracks
    world, n_tracks = 3, 90
    rng = np.random.default_rng(len(case))
    wide = lambda tr: case != "unreserved_table_cut_segments" and tr < 30   # noqa: E731  rank 0's tracks
    blocks = [_rows(rng, 900, tr + 1 + ((1 << 24) if wide(tr) else 0), tr + 2 + ((1 << 24) if wide(tr) else 0), noff=4000 if wide(tr) else 600)
              for tr in range(n_tracks)]
    gid = 9000 + sum(map(ord, case))

    def rank_fn(r):
        ctx = _ffi.Context(0)
        comm = _ffi.Comm.local(ctx, gid, r, world)
        tbl = S.Table(ctx)
        tbl.set_segment_rows(10000)
        out = {"seal_error": None, "gather_error": None}
        try:
            if case != "unreserved_table_cut_segments":
                tbl.reserve(0, 0, gather=True)
            lo, hi = shard_tracks(n_tracks, r, world)
            for i, tr in enumerate(range(lo, hi)):
                tbl.insert(*blocks[tr])
                seals = (case == "wide_ids_and_sealed_runs" and r == 1) or (case == "unreserved_table_cut_segments" and r == 1) or \
                        (case == "hold_mode_wide_seal" and r == 0)
                if seals and i % 10 == 9:
                    try:
                        tbl.seal_run()
                    except _ffi.ShzError as e:
                        out["seal_error"] = e.code
            try:
                tbl.allgather(comm)
            except _ffi.ShzError as e:
                out["gather_error"] = e.code
            out["rows"] = tbl.rows()
            return out
        finally:
            tbl.close()
            comm.close()
            ctx.close()

    outs = _run_ranks(world, rank_fn)
    if case == "hold_mode_wide_seal":
        # seal_run refuses (the rows stay staged) and says so; with no run sealed anywhere the column path still builds the table
        assert outs[0]["seal_error"] == _ffi.E_UNSUPPORTED
        assert all(o["gather_error"] is None for o in outs)
        want = len(np.unique(np.concatenate([np.stack(b, 1) for b in blocks]).astype(np.uint64), axis=0))
        assert all(o["rows"] == (want, 0) for o in outs)
    else:
        assert all(o["gather_error"] == _ffi.E_STATE for o in outs), outs


def test_sharded_builder_over_thread_ranks_equals_one_rank():
    """ingest.ShardedBuilder as a user drives it: device-resident synthetic tracks, runs sealed and sent on the way."""
    import shazam_amd as S
    from shazam_amd import _ffi
    from shazam_amd.db import HipFingerprintDB
    from shazam_amd.ingest import ShardedBuilder
    n_tracks, n_samples, world = 36, 2048 * 60, 3

    def run(world_, r, comm_of):
        ctx = _ffi.Context(0)
        comm = comm_of(ctx)
        db = HipFingerprintDB(ctx=ctx)
        db.table.set_segment_rows(9000)

        def source(lo, hi):
            return ctx.synth_pcm(4242, lo, hi - lo, n_samples, 3000, 1500), n_samples

        b = ShardedBuilder(db, r, world_, comm, chunk_tracks=4, seal_rows=3000)
        info = b.build(n_tracks, source, rows_hint=0)
        k, s, o = db.table.export()
        segs = db.table.segments()
        db.close()
        if comm is not None:
            comm.close()
        ctx.close()
        return np.stack([k, s, o], 1).astype(np.uint64), info, segs

    ref_rows, ref_info, ref_segs = run(1, 0, lambda ctx: None)
    assert ref_info["runs_sealed_on_the_way"] >= 2 and ref_segs >= 2
    # the same corpus handed over as HOST arrays (what read() yields, __init__.py:70-113): the same table
    hctx = _ffi.Context(0)
    hdb = HipFingerprintDB(ctx=hctx)
    hdb.table.set_segment_rows(9000)

    def host_source(lo, hi):
        buf = hctx.synth_pcm(4242, lo, hi - lo, n_samples, 3000, 1500)
        pcm = buf.download(np.int16, (hi - lo) * n_samples)
        buf.free()
        return [pcm[i * n_samples:(i + 1) * n_samples] for i in range(hi - lo)]

    ShardedBuilder(hdb, 0, 1, None, chunk_tracks=4, seal_rows=3000).build(n_tracks, host_source)
    hk, hs, ho_ = hdb.table.export()
    assert np.array_equal(np.stack([hk, hs, ho_], 1).astype(np.uint64), ref_rows)
    hdb.close()
    hctx.close()
    outs = _run_ranks(world, lambda r: run(world, r, lambda ctx: _ffi.Comm.local(ctx, 777001, r, world)))
    for rows, info, segs in outs:
        assert np.array_equal(rows, ref_rows)           # same rows in the same order: segments cut by key range concatenate to the sorted table
        assert info["runs_sealed_on_the_way"] >= 1 and info["bytes_received"] > 0


def test_key_sharded_table_over_thread_ranks():
    import shazam_amd as S
    from shazam_amd import _ffi
    from shazam_amd.ingest import shard_tracks
    from shazam_amd.shard import ShardedTable
    world, n_tracks = 3, 90
    rng = np.random.default_rng(77)
    blocks = [_rows(rng, 1500, tr + 1, tr + 2) for tr in range(n_tracks)]
    ref_ctx = _ffi.Context(0)
    ref = S.Table(ref_ctx)
    for b in blocks:
        ref.insert(*b)
    ref.finalize()
    qk = np.concatenate([b[0][:60] for b in blocks[::7]])
    qo = np.concatenate([(b[2][:60] + 3) % 600 for b in blocks[::7]])
    qoff = np.arange(0, len(qk) + 1, 60).astype(np.uint64)
    want = ref.match(qk, qo, qoff, 4)
    gid = 424242

    def rank_fn(r):
        ctx = _ffi.Context(0)
        comm = _ffi.Comm.local(ctx, gid, r, world)
        st = ShardedTable(ctx, comm=comm)
        lo, hi = shard_tracks(n_tracks, r, world)
        for tr in range(lo, hi):
            st.insert(*blocks[tr])
        st.finalize()
        rows = st.rows()[0]
        res = st.match(qk, qo, qoff, 4)
        st.close()
        comm.close()
        ctx.close()
        return rows, res

    outs = _run_ranks(world, rank_fn)
    assert sum(o[0] for o in outs) == ref.rows()[0]
    for rows, res in outs:
        for name in ("sid", "delta", "aligned", "dedup", "nres"):
            assert np.array_equal(res[name], want[name]), name
    # every hash and every table row belongs to exactly one shard: the per-rank counts add up to the table's
    for name in ("nhash", "npairs"):
        assert np.array_equal(sum(o[1][name].astype(np.uint64) for o in outs), want[name].astype(np.uint64)), name
    ref.close()
    ref_ctx.close()


def test_pipelined_build_over_a_one_rank_rccl_communicator():
    """The exchange rounds over REAL RCCL (one rank: all a one-GPU lease allows -- ncclCommInitRank refuses two ranks on one
    device): ncclAllGather of the round's block on the communicator's exchange stream, the event that orders that stream
    behind the seal, the list exchange's early return, rounds counted; the table equals the plain build's."""
    import shazam_amd as S
    from shazam_amd import _ffi
    rng = np.random.default_rng(12)
    blocks = [_rows(rng, 900, tr + 1, tr + 2) for tr in range(60)]
    want = np.unique(np.concatenate([np.stack(b, 1) for b in blocks]).astype(np.uint64), axis=0)
    ctx = _ffi.Context(0)
    try:
        comm = _ffi.Comm(ctx, _ffi.comm_unique_id(), 0, 1)
    except _ffi.ShzError as e:
        pytest.skip(f"librccl not usable here: {e}")
    comm.warmup()
    tbl = S.Table(ctx)
    tbl.set_segment_rows(20000)
    tbl.reserve(0, 0, gather=True)
    for i, b in enumerate(blocks):
        tbl.insert(*b)
        if i % 15 == 14:
            tbl.exchange_run(comm)
    assert tbl.exchange_stats()["rounds"] == 4 and tbl.exchange_stats()["runs_held"] == 4
    recv = tbl.allgather(comm)
    k, s, o = tbl.export()
    rows = np.stack([k, s, o], 1).astype(np.uint64)
    assert recv == 0 and np.array_equal(rows, want) and tbl.segments() >= 2
    tbl.close()
    comm.close()
    ctx.close()


def test_mixed_ingest_and_query_stream_over_thread_ranks():
    """BASELINE configs[4] at N > 1, on one GPU: a gathered base table (runs sent on the way), then batches of NEW songs --
    every rank stages its share, shz_table_allgather takes the column path into the live replicas (they hold rows), queries
    run between the batches.  After every batch all ranks hold the same table, equal to a one-rank table that inserted the
    same songs, and answer the queries alike (a query for a song ingested in the last batch finds it)."""
    import shazam_amd as S
    from shazam_amd import _ffi
    from shazam_amd.ingest import shard_tracks
    world, n_base, n_new, batches = 3, 90, 18, 3
    rng = np.random.default_rng(404)
    blocks = [_rows(rng, 700, tr + 1, tr + 2) for tr in range(n_base + n_new * batches)]
    gid = 606060

    def queries(upto):
        tr = list(range(3, upto, 11)) + [upto - 1]           # incl. the newest song
        qk = np.concatenate([blocks[t][0][:50] for t in tr])
        qo = np.concatenate([blocks[t][2][:50] for t in tr])
        return qk, qo, np.arange(0, len(qk) + 1, 50).astype(np.uint64), np.array(tr) + 1

    def rank_fn(r):
        ctx = _ffi.Context(0)
        comm = _ffi.Comm.local(ctx, gid, r, world)
        tbl = S.Table(ctx)
        tbl.set_segment_rows(30000)
        tbl.reserve(0, 0, gather=True)
        lo, hi = shard_tracks(n_base, r, world)
        for i, tr in enumerate(range(lo, hi)):
            tbl.insert(*blocks[tr])
            if i % 10 == 9:
                tbl.exchange_run(comm)
        tbl.allgather(comm)
        out = []
        for b in range(batches):
            first = n_base + b * n_new
            lo, hi = shard_tracks(n_new, r, world)
            for tr in range(first + lo, first + hi):
                tbl.insert(*blocks[tr])
            tbl.allgather(comm)                               # the replicas hold rows: every rank takes the column path
            qk, qo, qoff, want_sid = queries(first + n_new)
            res = tbl.match(qk, qo, qoff, 2)
            k, s, o = tbl.export()
            rows = np.stack([k, s, o], 1).astype(np.uint64)
            out.append((rows[np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0]))], res, want_sid))
        tbl.close(); comm.close(); ctx.close()
        return out

    outs = _run_ranks(world, rank_fn)
    for b in range(batches):
        upto = n_base + (b + 1) * n_new
        want = np.unique(np.concatenate([np.stack(x, 1) for x in blocks[:upto]]).astype(np.uint64), axis=0)
        for r in range(world):
            rows, res, want_sid = outs[r][b]
            assert len(rows) == len(want) and np.array_equal(rows, want), (b, r)
            assert np.array_equal(res["sid"][:, 0], want_sid), (b, r)      # every queried song is found, the newest included
            for name in res:
                assert np.array_equal(res[name], outs[0][b][1][name]), (b, r, name)
