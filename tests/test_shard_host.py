"""CPU: the host side of the key-sharded table.

* the shard function exported by libshz.so equals its numpy twin and spreads keys evenly;
* the additivity the design rests on, stated with the oracle alone: partition the DB rows by key, run the
  reference's return_matches on every part, concatenate the matches and add up dedup_hashes -- the result is the
  unsharded one (recognizer.py:222-271), so align_matches sees identical input."""
import numpy as np

from oracle import cpu_ref as O


def test_shard_function_matches_numpy_twin_and_balances():
    from shazam_amd.shard import shard_of_keys, shard_of_keys_numpy
    rng = np.random.default_rng(3)
    f1, f2, dt = rng.integers(0, 2049, 200000), rng.integers(0, 2049, 200000), rng.integers(0, 201, 200000)
    keys = ((f1 << 20) | (f2 << 8) | dt).astype(np.uint32)
    for n in (1, 2, 3, 7, 8, 64):
        a, b = shard_of_keys(keys, n), shard_of_keys_numpy(keys, n)
        assert np.array_equal(a, b) and a.max() < n
        counts = np.bincount(a, minlength=n)
        assert counts.min() > 0.9 * len(keys) / n and counts.max() < 1.1 * len(keys) / n
    # deterministic per key: duplicates of a DB row always meet on one shard (the table dedups per shard)
    assert np.array_equal(shard_of_keys(keys[:100], 8), shard_of_keys(keys[:100].copy(), 8))


def test_votes_are_additive_over_key_shards():
    from shazam_amd.shard import shard_of_keys_numpy
    rng = np.random.default_rng(11)
    n = 6000
    key = ((rng.integers(0, 30, n) << 20) | (rng.integers(0, 30, n) << 8) | rng.integers(0, 4, n)).astype(np.uint32)
    sid = rng.integers(1, 25, n)
    off = rng.integers(0, 300, n)
    full = O.DictDB()
    for s_ in range(1, 30):
        full.insert_song(str(s_), "00", 1)
    nsh = 3
    parts = [O.DictDB() for _ in range(nsh)]
    for p in parts:
        for s_ in range(1, 30):
            p.insert_song(str(s_), "00", 1)
    sh = shard_of_keys_numpy(key, nsh)
    for k, s, o, h in zip(key.tolist(), sid.tolist(), off.tolist(), sh.tolist()):
        full.insert_hashes(s, [(k, o)])
        parts[h].insert_hashes(s, [(k, o)])
    for q in range(10):
        m = int(rng.integers(1, 120))
        hs = set(zip(key[rng.integers(0, n, m)].tolist(), rng.integers(0, 50, m).tolist()))
        want_m, want_dd = O.return_matches(hs, full)
        got_m, got_dd = [], {}
        for p in parts:
            mm, dd = O.return_matches(hs, p)
            got_m += mm
            for k_, v_ in dd.items():
                got_dd[k_] = got_dd.get(k_, 0) + v_
        assert sorted(want_m) == sorted(got_m) and want_dd == got_dd
        assert O.vote(want_m, 5) == O.vote(got_m, 5)
