"""CPU, world_size 2 over gloo: the N > 1 database-build logic.  Each rank takes its block of tracks
(shazam_amd.ingest.shard_tracks), derives rows (key32, song_id, offset) with the oracle standing in
for the GPU extraction, all-gathers them (gloo standing in for RCCL) and merges; the merged table
must equal the single-rank table row for row -- the invariant shz_table_allgather must keep."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from shazam_amd.ingest import GatherRounds, merge_rows, merge_sorted_runs, pack_rows, shard_tracks, song_id_of_track

N_TRACKS = 7


def _rows_of_tracks(lo, hi):
    from oracle import cpu_ref as O, synth
    ks, ss, os_ = [], [], []
    for i in range(lo, hi):
        x = synth.synth_clip(99, i, 2048 * 40 + 13 * i, 3000, 1500)
        k, t1, _, _ = O.fingerprint_keys(x)
        ks.append(k)
        ss.append(np.full(len(k), song_id_of_track(i), np.uint32))
        os_.append(t1)
    if not ks:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint32)
    return np.concatenate(ks), np.concatenate(ss), np.concatenate(os_)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_tracks(N_TRACKS, rank, world)
    mine = _rows_of_tracks(lo, hi)
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    k, s, o = merge_rows(parts)
    # the path shz_table_allgather takes into an empty table: maxima agreed first, every rank sorts its own rows packed,
    # the sorted runs travel, every rank merges them and cuts segments (small segments here to cross a cut)
    mx = [None] * world
    dist.all_gather_object(mx, (int(mine[1].max(initial=0)), int(mine[2].max(initial=0))))
    sb, ob = max(a for a, _ in mx).bit_length() or 1, max(b for _, b in mx).bit_length() or 1
    run = np.sort(pack_rows(*mine, sb, ob))
    runs = [None] * world
    dist.all_gather_object(runs, run)
    segs = merge_sorted_runs(runs, sb, ob, segment_rows=5000)
    q.put((rank, lo, hi, k, s, o, segs))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_tracks_partition():
    for n in (0, 1, 7, 8, 1000, 100003):
        for w in (1, 2, 3, 8):
            blocks = [shard_tracks(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_tracks(4, 2, 2)


def test_two_rank_build_equals_single_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = merge_rows([_rows_of_tracks(0, N_TRACKS)])
    assert sorted((g[1], g[2]) for g in got) == [(0, 4), (4, 7)]
    for _, _, _, k, s, o, segs in got:   # every rank ends with the same, complete table
        assert np.array_equal(k, want[0]) and np.array_equal(s, want[1]) and np.array_equal(o, want[2])
        # ... and so does the sorted-run merge: its segments, concatenated, are that table (rows unique across the cuts)
        assert len(segs) > 1
        for col in range(3):
            assert np.array_equal(np.concatenate([sg[col] for sg in segs]), want[col])
    assert set(np.unique(want[1]).tolist()) == set(range(1, N_TRACKS + 1))


# ---- the exchange rounds of the gathered build (ingest.GatherRounds == csrc/shz_build.hip gx_round) over gloo ----------
def _synthetic_rows(rng, n, sid, noff):
    k = (rng.integers(0, 300, n).astype(np.uint32) << np.uint32(20)) | (rng.integers(0, 300, n).astype(np.uint32) << np.uint32(8)) | \
        rng.integers(0, 5, n).astype(np.uint32)
    return k, np.full(n, sid, np.uint32), np.sort(rng.integers(0, noff, n).astype(np.uint32))


def _rounds_tracks(case):
    rng = np.random.default_rng(5)
    wide = case == "wide_ids_and_runs"
    return [_synthetic_rows(rng, 400, t + 1 + ((1 << 24) if wide and t < 10 else 0),
                            9000 if (case == "widening" and t >= 36) or (wide and t < 10) else 500) for t in range(40)]


def _rounds_worker(rank, world, port, case, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def allgather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    tracks = _rounds_tracks(case)
    lo, hi = shard_tracks(len(tracks), rank, world)
    g = GatherRounds(rank, world, allgather, run_rows=700 if case == "many_runs" else (1 << 32) - 4096)
    staged = []
    err = None
    try:
        for i, t in enumerate(range(lo, hi)):
            staged.append(tracks[t])
            every = (3, 7)[rank]                  # the ranks call exchange() different numbers of times
            if case in ("pipelined", "widening") and i % every == every - 1:
                g.seal(*(np.concatenate([s_[c] for s_ in staged]) for c in range(3)))
                staged = []
                g.exchange()
            if case == "wide_ids_and_runs" and rank == 1 and i % 5 == 4:   # rank 1 seals runs, rank 0's ids do not pack
                g.seal(*(np.concatenate([s_[c] for s_ in staged]) for c in range(3)))
                staged = []
        rest = tuple(np.concatenate([s_[c] for s_ in staged]) for c in range(3)) if staged else None
        table = g.finish(rest)
    except RuntimeError as e:
        err, table = str(e), None
    q.put((rank, table, err, g.rounds))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["plain", "pipelined", "widening", "many_runs", "wide_ids_and_runs"])
def test_exchange_rounds_over_gloo(case):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rounds_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in procs), key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if case == "wide_ids_and_runs":                 # the tables would differ: EVERY rank refuses
        assert all(g[2] is not None and "column path" in g[2] for g in got), got
        return
    want = merge_rows(_rounds_tracks(case))
    for _, table, err, rounds in got:
        assert err is None
        for c in range(3):
            assert np.array_equal(table[c], want[c])
    assert got[0][3] == got[1][3]                  # both ranks ran the same number of rounds
    if case == "many_runs":                        # 8,000 rows a rank in runs of 700: 12 runs, one round; the peer's all arrive
        assert got[0][3] >= 2
    if case in ("pipelined", "widening"):
        assert got[0][3] >= 5
