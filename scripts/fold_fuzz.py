"""Randomised campaign for the vote fold (not part of the suite): tables of 1e3..1e6 songs, queries of 1e4..3e6 votes, planted
ties, random topn and batch sizes -- the tile path against the exact full sort (SHZ_MATCH_FULL_SORT), array for array.
python scripts/fold_fuzz.py [seconds] [seed0]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import shazam_amd as S  # noqa: E402

ctx = S.get_context(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget
n_cases = n_q = 0
votes = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    n_songs = int(10 ** rng.uniform(3, 6))
    n_keys = int(rng.choice([50, 400, 3000]))
    rows_per_key = int(10 ** rng.uniform(2, 4))
    n_off = int(rng.choice([1, 3, 40, 1000]))
    keys = np.unique(((rng.integers(0, 2049, n_keys) << 20) | (rng.integers(0, 2049, n_keys) << 8) | rng.integers(0, 201, n_keys)).astype(np.uint32))
    n_keys = len(keys)
    tk = np.repeat(keys, rows_per_key)
    hot = rng.random() < 0.5
    ts = (rng.integers(1, n_songs + 1, len(tk)) if not hot else 1 + (rng.zipf(1.3, len(tk)) % n_songs)).astype(np.uint32)
    to = (50 + rng.integers(0, n_off, len(tk))).astype(np.uint32)
    nq = int(rng.choice([1, 2, 5, 20]))
    per_q = int(min(n_keys, rng.choice([10, 60, 250])))
    qsel = [rng.choice(n_keys, per_q, replace=False) for _ in range(nq)]
    # planted ties: a few songs with the same count at one delta per query
    ek, es, eo = [], [], []
    for q in range(nq):
        cnt = int(rng.integers(2, 9))
        for s_ in rng.integers(1, n_songs + 1, int(rng.integers(0, 30))):
            ks = keys[qsel[q][: min(cnt, per_q)]]
            ek.append(ks); es.append(np.full(len(ks), s_, np.uint32)); eo.append(np.full(len(ks), 5000 + q, np.uint32))
    if ek:
        tk, ts, to = np.concatenate([tk] + ek), np.concatenate([ts] + es), np.concatenate([to] + eo)
    t = S.Table(ctx)
    if rng.random() < 0.3:
        t.set_segment_rows(max(1000, len(tk) // 4))
    t.insert(tk, ts, to)
    t.finalize()
    qk = np.concatenate([keys[s] for s in qsel])
    qo = np.concatenate([np.full(per_q, 7 + q, np.uint32) for q in range(nq)])
    qoff = np.arange(nq + 1, dtype=np.uint64) * per_q
    topn = int(rng.integers(1, 9))
    fast = t.match(qk, qo, qoff, topn)
    full = t.match(qk, qo, qoff, topn, full_sort=True)
    for k in ("sid", "delta", "aligned", "dedup", "nres", "nhash", "npairs"):
        if not np.array_equal(fast[k], full[k]):
            print("MISMATCH seed", seed, k, dict(n_songs=n_songs, n_keys=n_keys, rows_per_key=rows_per_key, n_off=n_off, nq=nq, per_q=per_q, topn=topn, hot=hot), flush=True)
            sys.exit(1)
    votes += int(fast["npairs"].sum())
    n_cases += 1
    n_q += nq
    t.close()
    seed += 1
print("fold fuzz:", n_cases, "cases,", n_q, "queries,", votes, "votes, next seed", seed, "-- all equal to the full sort")
