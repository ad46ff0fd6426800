"""CPU, world_size 2 over gloo: the N > 1 database-build logic.  Each rank takes its block of tracks
(shazam_amd.ingest.shard_tracks), derives rows (key32, song_id, offset) with the oracle standing in
for the GPU extraction, all-gathers them (gloo standing in for RCCL) and merges; the merged table
must equal the single-rank table row for row -- the invariant shz_table_allgather must keep."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from shazam_amd.ingest import merge_rows, merge_sorted_runs, pack_rows, shard_tracks, song_id_of_track

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
