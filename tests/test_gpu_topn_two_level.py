"""GPU: queries with tens of thousands of (query, song) groups take the two-level top-n (slices per query + final
ranking, shz_table.hip: m_topn_partial_kernel / m_topn_final_kernel).  Same ranking as align_matches
(recognizer.py:289-338): count descending, ties -> smaller song id, and inside a song the smallest offset difference
among the best."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_many_groups_per_query_equal_oracle_vote():
    import shazam_amd as S
    from oracle import cpu_ref as O
    ctx = S.get_context(0)
    rng = np.random.default_rng(17)
    hot = np.array([(7 << 20) | (9 << 8) | 1, (100 << 20) | (3 << 8) | 2], np.uint32)     # two very popular hashes
    n_hot = 60000
    sid = np.concatenate([rng.permutation(np.arange(1, n_hot + 1)).astype(np.uint32),     # every song once: all ties
                          rng.integers(1, n_hot + 1, n_hot).astype(np.uint32)])
    key = np.concatenate([np.full(n_hot, hot[0]), np.full(n_hot, hot[1])]).astype(np.uint32)
    off = rng.integers(0, 50, 2 * n_hot).astype(np.uint32)
    # a few songs get real aligned evidence on other keys
    ek = ((rng.integers(200, 260, 400) << 20) | (rng.integers(0, 50, 400) << 8) | 3).astype(np.uint32)
    es = rng.choice(np.array([5, 77, 40000, 59999], np.uint32), 400)
    eo = rng.integers(0, 30, 400).astype(np.uint32)
    key, sid, off = np.concatenate([key, ek]), np.concatenate([sid, es]), np.concatenate([off, eo])
    t = S.Table(ctx)
    t.insert(key, sid, off)
    t.finalize()
    # query 0: both hot hashes + the evidence keys; query 1: one hot hash at two offsets; query 2: nothing hot
    q0k = np.concatenate([hot, ek[:150]])
    q0o = np.concatenate([np.array([3, 4], np.uint32), eo[:150]])
    q1k, q1o = np.array([hot[0], hot[0]], np.uint32), np.array([0, 7], np.uint32)
    q2k, q2o = ek[150:170], eo[150:170]
    qk, qo = np.concatenate([q0k, q1k, q2k]), np.concatenate([q0o, q1o, q2o])
    qoff = np.array([0, len(q0k), len(q0k) + 2, len(qk)], np.uint64)
    res = t.match(qk, qo, qoff, 5)
    assert int(res["npairs"][0]) > 100000            # large enough for the sliced path (>= 2 slices per query)
    odb = O.DictDB()
    for s_ in range(1, n_hot + 1):
        odb.insert_song(str(s_), "00", 1)
    for k_, s_, o_ in zip(key.tolist(), sid.tolist(), off.tolist()):
        odb.insert_hashes(s_, [(k_, o_)])
    for q in range(3):
        a, b = int(qoff[q]), int(qoff[q + 1])
        hs = set(zip(qk[a:b].tolist(), qo[a:b].tolist()))
        m, dd = O.return_matches(hs, odb)
        want = O.vote(m, 5)
        got = [(int(res["sid"][q, i]), int(res["delta"][q, i]), int(res["aligned"][q, i])) for i in range(int(res["nres"][q]))]
        assert got == [tuple(w) for w in want], q
        assert [int(res["dedup"][q, i]) for i in range(len(got))] == [dd[w[0]] for w in want]
        assert int(res["npairs"][q]) == len(m) and int(res["nhash"][q]) == len(hs)
    t.close()


def test_long_groups_take_the_workgroup_fold():
    """(query, song) groups of thousands of votes leave m_reduce_kernel for m_reduce_long_kernel (one workgroup per
    group): one run of a single delta crossing every piece, and ~10,000 runs with the best count tied from delta 0
    upwards (the smallest delta must win)."""
    import shazam_amd as S
    ctx = S.get_context(0)
    K = lambda i: np.uint32(((i % 2049) << 20) | (((i // 2049) % 2049) << 8) | 7)  # noqa: E731
    n_line = 6000
    keys_a = np.array([K(i) for i in range(n_line)], np.uint32)                 # song 1: distinct hashes, offset = i + 7
    key_hot = np.uint32((2000 << 20) | (11 << 8) | 1)
    tk = np.concatenate([keys_a, np.full(10000, key_hot)])
    ts = np.concatenate([np.full(n_line, 1), np.full(10000, 2)]).astype(np.uint32)
    to = np.concatenate([np.arange(n_line) + 7, np.arange(10000)]).astype(np.uint32)
    t = S.Table(ctx)
    t.insert(tk, ts, to)
    t.finalize()
    # query 0: every hash of song 1 at offset i -> 6,000 votes, all delta 7; query 1: the hot hash at offsets 0 and 5
    qk = np.concatenate([keys_a, np.array([key_hot, key_hot], np.uint32)])
    qo = np.concatenate([np.arange(n_line), np.array([0, 5])]).astype(np.uint32)
    qoff = np.array([0, n_line, n_line + 2], np.uint64)
    res = t.match(qk, qo, qoff, 3)
    assert int(res["nres"][0]) == 1 and (int(res["sid"][0, 0]), int(res["delta"][0, 0]), int(res["aligned"][0, 0]),
                                         int(res["dedup"][0, 0])) == (1, 7, n_line, n_line)
    assert int(res["npairs"][1]) == 20000 and int(res["nres"][1]) == 1
    # deltas r and r - 5 for r = 0..9999: count 2 for delta 0..9994, count 1 below and above -> (count 2, delta 0)
    assert (int(res["sid"][1, 0]), int(res["delta"][1, 0]), int(res["aligned"][1, 0]), int(res["dedup"][1, 0])) == (2, 0, 2, 10000)
    t.close()
