"""CPU, world_size 2 over gloo: the N > 1 logic of the key-sharded table.

Each rank derives the rows of its own tracks (the oracle stands in for the GPU extraction), routes every row to the
rank that owns its key with the library's shard function (gloo stands in for the RCCL all-to-all of
shz_table_shard_exchange), answers the same queries from its shard only -- looking up just the hashes it owns, like
shz_match_pairs -- and all-gathers the (song, offset difference) votes (shz_pairs_allgather).  The ranked result on
every rank must equal the reference vote over the unsharded table (shz_pairs_vote == align_matches)."""
import multiprocessing as mp
import os
import socket

import numpy as np

from shazam_amd.ingest import shard_tracks, song_id_of_track
from shazam_amd.shard import shard_of_keys

N_TRACKS = 6


def _rows_of_tracks(lo, hi):
    from oracle import cpu_ref as O, synth
    rows = []
    for i in range(lo, hi):
        x = synth.synth_clip(123, i, 2048 * 36 + 11 * i, 3000, 1500)
        k, t1, _, _ = O.fingerprint_keys(x)
        rows += [(int(a), song_id_of_track(i), int(b)) for a, b in zip(k.tolist(), t1.tolist())]
    return rows


def _queries():
    from oracle import cpu_ref as O, synth
    qs = []
    for i, start in ((1, 5), (4, 12)):
        x = synth.synth_clip(123, i, 2048 * 36 + 11 * i, 3000, 1500)[start * 2048:start * 2048 + 2048 * 14]
        k, t1, _, _ = O.fingerprint_keys(x)
        qs.append(set(zip(k.tolist(), t1.tolist())))
    return qs


def _db(rows):
    from oracle import cpu_ref as O
    db = O.DictDB()
    for s in range(1, N_TRACKS + 1):
        db.insert_song(str(s), "00", 1)
    for k, s, o in rows:
        db.insert_hashes(s, [(k, o)])
    return db


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import cpu_ref as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_tracks(N_TRACKS, rank, world)
    mine = _rows_of_tracks(lo, hi)
    dest = shard_of_keys(np.array([r[0] for r in mine], np.uint32), world) if mine else np.zeros(0, np.uint32)
    send = [[r for r, d in zip(mine, dest.tolist()) if d == p] for p in range(world)]
    allsend = [None] * world          # all-to-all as an all-gather of the send lists + this rank's column
    dist.all_gather_object(allsend, send)
    recv = [allsend[p][rank] for p in range(world)]
    shard_rows = [r for part in recv for r in part]
    db = _db(shard_rows)
    votes, dedup, nhash = [], [], []
    for hs in _queries():
        keys = np.array([h[0] for h in hs], np.uint32)
        owned = {h for h, d in zip(hs, shard_of_keys(keys, world).tolist()) if d == rank}
        m, dd = O.return_matches(owned, db) if owned else ([], {})
        votes.append(m)
        dedup.append(dd)
        nhash.append(len(owned))
    gathered = [None] * world
    dist.all_gather_object(gathered, (votes, dedup, nhash))
    ranked = []
    for qi in range(len(votes)):
        allm = [v for g in gathered for v in g[0][qi]]
        alld = {}
        for g in gathered:
            for s, c in g[1][qi].items():
                alld[s] = alld.get(s, 0) + c
        ranked.append((O.vote(allm, 3), alld, sum(g[2][qi] for g in gathered), len(allm)))
    q.put((rank, len(shard_rows), sorted(set(shard_rows)) == sorted(shard_rows), ranked))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_key_sharded_match_equals_unsharded():
    from oracle import cpu_ref as O
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
    all_rows = _rows_of_tracks(0, N_TRACKS)
    full = _db(all_rows)
    # every distinct row lives on exactly one rank (rows repeated inside a track collapse there, like the table's dedup)
    assert sum(g[1] for g in got) >= len(set(all_rows)) and all(g[1] > 0 for g in got)
    want = []
    for hs in _queries():
        m, dd = O.return_matches(hs, full)
        want.append((O.vote(m, 3), dd, len(hs), len(m)))
    for _, _, _, ranked in got:          # every rank computes the same, complete answer
        assert len(ranked) == len(want)
        for (v, dd, nh, npairs), (wv, wdd, wnh, wnp) in zip(ranked, want):
            assert v == wv and dd == wdd and nh == wnh and npairs == wnp
    assert want[0][0][0][0] == song_id_of_track(1) and want[0][0][0][1] == 5
    assert want[1][0][0][0] == song_id_of_track(4) and want[1][0][0][1] == 12
