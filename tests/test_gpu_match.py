"""GPU parity: the HBM fingerprint table and the match/align path (through the C ABI) against the
golden results of the reference's return_matches/align_matches and against the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import shazam_amd
    return shazam_amd


@pytest.fixture(scope="module")
def ctx(S):
    return S.get_context(0)


@pytest.fixture(scope="module")
def O():
    from oracle import cpu_ref
    return cpu_ref


@pytest.fixture(scope="module")
def match_golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "match_cases.json")))


def _song_pcm(s, p):
    from oracle import synth
    if s == 7:
        return synth.synth_clip(p["seed"], 3, p["n"], p["tone_amp"], p["noise_amp"])
    if s == 11:
        half = synth.synth_clip(p["seed"], 11, 2048 * 100, p["tone_amp"], p["noise_amp"])
        return np.concatenate([half, half])
    return synth.synth_clip(p["seed"], s, p["n"], p["tone_amp"], p["noise_amp"])


def _norm(res):
    out = []
    for r in res:
        r = dict(r)
        for k, v in r.items():
            if isinstance(v, bytes):
                r[k] = v.decode()
            elif isinstance(v, np.integer):
                r[k] = int(v)
        out.append(r)
    return out


def test_table_build_dedup_and_lookup(S, ctx):
    rng = np.random.default_rng(1)
    n = 200000
    key = (rng.integers(0, 2049, n).astype(np.uint32) << 20) | (rng.integers(0, 2049, n).astype(np.uint32) << 8) | \
        rng.integers(0, 6, n).astype(np.uint32)
    key[:5000] = key[5000:10000]  # popular keys
    sid = rng.integers(1, 300, n).astype(np.uint32)
    off = rng.integers(0, 4000, n).astype(np.uint32)
    t = S.Table(ctx)
    t.insert(key[:n // 2], sid[:n // 2], off[:n // 2])
    t.insert(key[n // 2:], sid[n // 2:], off[n // 2:])
    t.insert(key[:1000], sid[:1000], off[:1000])  # INSERT IGNORE duplicates
    assert t.rows() == (0, n + 1000)
    t.finalize()
    rows = np.unique(np.stack([key, sid, off], 1).astype(np.uint64), axis=0)  # lexicographic (key, sid, off)
    k, s, o = t.export()
    assert len(k) == len(rows) == t.rows()[0]
    assert np.array_equal(k, rows[:, 0]) and np.array_equal(s, rows[:, 1]) and np.array_equal(o, rows[:, 2])
    # incremental insert after finalize re-sorts everything
    t.insert(np.array([5, 5, 1 << 31], np.uint32), np.array([9, 9, 2], np.uint32), np.array([1, 1, 3], np.uint32))
    t.finalize()
    rows2 = np.unique(np.concatenate([rows, np.array([[5, 9, 1], [1 << 31, 2, 3]], np.uint64)]), axis=0)
    k, s, o = t.export()
    assert np.array_equal(np.stack([k, s, o], 1).astype(np.uint64), rows2)
    # lookup: rows of listed keys, in key-list order
    want_keys = np.array([int(rows[100, 0]), 0xFFFFFFFF, 5, int(rows[-1, 0]), int(rows[100, 0])], np.uint32)
    lk, ls, lo = t.lookup(want_keys)
    exp = np.concatenate([rows2[rows2[:, 0] == kk] for kk in want_keys.astype(np.uint64)])
    assert np.array_equal(np.stack([lk, ls, lo], 1).astype(np.uint64), exp)
    assert t.song_rows(9) == int((rows2[:, 1] == 9).sum())
    t.close()
    e = S.Table(ctx)
    e.finalize()
    assert e.rows() == (0, 0) and len(e.lookup(np.array([1, 2], np.uint32))[0]) == 0
    r = e.match(np.array([1, 2], np.uint32), np.array([0, 1], np.uint32), np.array([0, 2], np.uint64), 3)
    assert r["nres"][0] == 0 and r["nhash"][0] == 2 and r["npairs"][0] == 0
    e.close()


@pytest.fixture(scope="module")
def mini_db(S, ctx, match_golden):
    p = match_golden["song_params"]
    db = S.get_database("hip")(ctx=ctx)
    pcm = {s: _song_pcm(s, p) for s in range(20)}
    for s in range(20):
        fp = set(S.fingerprint(pcm[s]))
        sid = db.insert_song(f"{s:06d}", hashlib.sha1(pcm[s].tobytes()).hexdigest().upper(), len(fp))
        assert sid == match_golden["songs"][s]["sid"] and len(fp) == match_golden["songs"][s]["total_hashes"]
        db.insert_hashes(sid, fp)
        db.set_song_fingerprinted(sid)
    return db, pcm


def _query_pcm(q, pcm):
    from oracle import synth
    sig = pcm[q["song"]][q["start"]:q["start"] + 220500]
    if q["snr"] is not None:
        sig = synth.mix_query(sig, synth.synth_clip(777, q["q"], 220500, 0, 8000), q["snr"])
    return sig


def test_recognize_against_reference_goldens(S, match_golden, mini_db):
    db, pcm = mini_db
    assert db.num_fingerprints() == sum(s["total_hashes"] for s in match_golden["songs"])
    assert db.table.song_rows(4) == match_golden["songs"][3]["total_hashes"]
    for q in match_golden["queries"]:
        res, ft, qt, at = S.recognize(_query_pcm(q, pcm), db=db, topn=3)
        assert _norm(res) == q["results"], q["q"]
    # batched form, stereo queries (same channel twice -> set union leaves the hashes unchanged)
    qs = [[_query_pcm(q, pcm), _query_pcm(q, pcm)] for q in match_golden["queries"][:12]]
    results, tm = S.recognize_batch(qs, db, topn=3)
    for q, r in zip(match_golden["queries"][:12], results):
        assert _norm(r) == q["results"]
    assert [int(v) for v in tm["n_hashes"]] == [q["n_hashes"] for q in match_golden["queries"][:12]]
    assert [int(v) for v in tm["n_matches"]] == [q["n_matches"] for q in match_golden["queries"][:12]]


def test_reference_style_return_matches_through_cursor(S, O, match_golden, mini_db):
    """The reference's own match code shape (recognizer.py:237-269) run against HipFingerprintDB.cursor()."""
    db, pcm = mini_db
    for q in match_golden["queries"][:8]:
        hashes = set(S.fingerprint(_query_pcm(q, pcm)))
        mapper = {}
        for hsh, offset in hashes:
            mapper.setdefault(hsh.upper(), []).append(offset)
        values = list(mapper.keys())
        dedup, results = {}, []
        with db.cursor() as cur:
            for index in range(0, len(values), 1000):
                query = db.SELECT_MULTIPLE % ", ".join([db.IN_MATCH] * len(values[index: index + 1000]))
                cur.execute(query, values[index: index + 1000])
                for hsh, sid, offset in cur:
                    dedup[sid] = dedup.get(sid, 0) + 1
                    for so in mapper[hsh]:
                        results.append((sid, offset - so))
        assert len(results) == q["n_matches"]
        assert {str(k): v for k, v in sorted(dedup.items())} == q["dedup"]
        assert _norm(O.align_matches(results, dedup, len(hashes), db, topn=3)) == q["results"]


def test_crafted_ties(S, ctx, match_golden):
    c = match_golden["crafted"]
    keys = c["keys"]
    db = S.get_database("hip")(ctx=ctx)
    for sid_s, rr in c["rows"].items():
        sid = db.insert_song(f"c{sid_s}", "AB" * 20, len(set(map(tuple, rr))))
        assert sid == int(sid_s)
        db.insert_keys(sid, np.array([keys[h] for h, _ in rr], np.uint32), np.array([o for _, o in rr], np.uint32))
        db.set_song_fingerprinted(sid)
    qk = np.array([keys[h] for h, _ in c["query"]] * 2, np.uint32)      # duplicated: set semantics
    qo = np.array([o for _, o in c["query"]] * 2, np.uint32)
    for topn in (1, 2, 3, 10):
        res = db.match(qk, qo, np.array([0, len(qk)], np.uint64), topn)
        assert int(res["nhash"][0]) == len(c["query"]) and int(res["npairs"][0]) == c["n_matches"]
        got = _norm(S._result_dicts(db, res, 0, int(res["nhash"][0])))
        assert got == c[f"results_top{topn}"], topn
    # many queries in one call, some empty
    qoff = np.array([0, 0, len(c["query"]), len(c["query"]), 2 * len(c["query"])], np.uint64)
    res = db.match(qk, qo, qoff, 3)
    assert list(res["nres"]) == [0, len(c["results_top3"]), 0, len(c["results_top3"])]
    for q in (1, 3):
        assert _norm(S._result_dicts(db, res, q, int(res["nhash"][q]))) == c["results_top3"]


def test_vote_against_oracle_random(S, ctx, O):
    """Random tables/queries with heavy key collisions: (sid, delta, count, dedup) of the top-n
    must equal the oracle's ranking exactly, including tie-breaks."""
    rng = np.random.default_rng(11)
    for trial in range(3):
        n = 60000
        key = (rng.integers(0, 40, n).astype(np.uint32) << 20) | (rng.integers(0, 40, n).astype(np.uint32) << 8) | \
            rng.integers(0, 3, n).astype(np.uint32)
        sid = rng.integers(1, 50, n).astype(np.uint32)
        off = rng.integers(0, 60, n).astype(np.uint32)
        t = S.Table(ctx)
        t.insert(key, sid, off)
        t.finalize()
        odb = O.DictDB()
        for s in range(1, 50):
            odb.insert_song(str(s), "00", 1)
        for k_, s_, o_ in zip(key.tolist(), sid.tolist(), off.tolist()):
            odb.insert_hashes(s_, [(k_, o_)])
        nq = 9
        qk, qo, qoff = [], [], [0]
        for q in range(nq):
            m = int(rng.integers(1, 120))
            sel = rng.integers(0, n, m)
            qk.append(key[sel])
            qo.append(rng.integers(0, 30, m).astype(np.uint32))
            qoff.append(qoff[-1] + m)
        qk, qo = np.concatenate(qk), np.concatenate(qo)
        res = t.match(qk, qo, np.array(qoff, np.uint64), 5)
        for q in range(nq):
            hs = set(zip(qk[qoff[q]:qoff[q + 1]].tolist(), qo[qoff[q]:qoff[q + 1]].tolist()))
            matches, dedup = O.return_matches(hs, odb)
            want = O.vote(matches, 5)
            assert int(res["nhash"][q]) == len(hs) and int(res["npairs"][q]) == len(matches)
            got = [(int(res["sid"][q, i]), int(res["delta"][q, i]), int(res["aligned"][q, i])) for i in range(int(res["nres"][q]))]
            assert got == [tuple(w) for w in want], (trial, q)
            assert [int(res["dedup"][q, i]) for i in range(len(got))] == [dedup[w[0]] for w in want]
        t.close()


def test_single_rank_rccl_allgather(S, ctx):
    """The RCCL path with a 1-rank communicator: gathered table == locally finalized table."""
    from shazam_amd import _ffi
    rng = np.random.default_rng(2)
    n = 30000
    key = rng.integers(0, 2 ** 31, n).astype(np.uint32)
    sid = rng.integers(1, 100, n).astype(np.uint32)
    off = rng.integers(0, 3000, n).astype(np.uint32)
    a, b = S.Table(ctx), S.Table(ctx)
    a.insert(key, sid, off)
    a.finalize()
    b.insert(key, sid, off)
    comm = _ffi.Comm(ctx, _ffi.comm_unique_id(), 0, 1)
    assert b.allgather(comm) == 0
    comm.barrier()
    comm.close()
    for x, y in zip(a.export(), b.export()):
        assert np.array_equal(x, y)
    a.close()
    b.close()
