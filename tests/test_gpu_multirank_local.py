"""GPU: the N > 1 logic of the sharded database build on ONE GPU ("configs[2] / [4] untested on hardware",
"the collectives have only ever run with nranks = 1").

Ranks are threads of this process, one context each, over the in-process transport (shz_comm_create_local: the same rank
protocol as over RCCL -- counts, maxima and path flags all-gathered, runs placed behind each other, one k-way merge -- with
rendezvous + device copies instead of ncclSend/ncclRecv).  What RCCL itself moves is not tested here; everything around it is.

* replicated table: every rank fingerprints its block of tracks (ingest.shard_tracks, song_id = track + 1),
  shz_table_allgather: all ranks end with the same table, equal to the one-rank build row for row, and answer queries alike;
  also: a rank without rows, a rank that sealed runs on the way, ids too wide to pack (every rank takes the column path),
  a rank whose table already holds rows (ditto);
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
