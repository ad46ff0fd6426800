"""CPU: the oracle's restatement of the early-exit match (recognizer_apriori.py:237-310, 602-609) against goldens made
by running the reference's own definitions on ordered hash lists (tests/golden/make_golden_apriori.py)."""
import json
import os

import numpy as np
import pytest

from oracle import cpu_ref as O, synth

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "apriori_cases.json")))


def apriori_query(case):
    """the ordered hash list of a golden case, rebuilt from its seeds (oracle fingerprint == reference fingerprint)"""
    sp = G["meta"]["song_params"]
    src = 2 if case["song"] == 5 else case["song"]
    x = synth.synth_clip(sp["seed"], src, sp["n"], sp["tone_amp"], sp["noise_amp"])
    x = x[case["start_frame"] * 2048:(case["start_frame"] + case["frames"]) * 2048]
    if case["snr"] is not None:
        x = synth.mix_query(x, synth.synth_clip(99, case["noise_clip"], len(x), 0, 8000), case["snr"])
    hs = list(dict.fromkeys(O.fingerprint(x)))
    if case["order"] == "reversed":
        hs = hs[::-1]
    elif case["order"] == "shuffled":
        hs = [hs[i] for i in np.random.default_rng(case["case"]).permutation(len(hs))]
    elif case["order"] == "by_offset_desc":
        hs = sorted(hs, key=lambda t: (-t[1], t[0]))
    return hs


@pytest.fixture(scope="module")
def apriori_db():
    sp = G["meta"]["song_params"]
    db = O.DictDB()
    for s in G["songs"]:
        x = synth.synth_clip(sp["seed"], s["source_clip"], sp["n"], sp["tone_amp"], sp["noise_amp"])
        fp = list(dict.fromkeys(O.fingerprint(x)))
        import hashlib
        sid = db.insert_song(f"a{s['song']:02d}", hashlib.sha1(x.tobytes()).hexdigest().upper(), len(set(fp)))
        assert sid == s["sid"]
        db.insert_hashes(sid, fp)
    return db


def _clean(res):
    return [{k: (v.decode() if isinstance(v, bytes) else v) for k, v in r.items()} for r in res]


def test_oracle_apriori_equals_reference(apriori_db):
    exits = []
    for case in G["cases"]:
        hs = apriori_query(case)
        assert len(hs) == case["n_hashes"]
        results, dedup, songs_arr, batches = O.return_matches_apriori(hs, apriori_db, case["batch_size"])
        assert len(results) == case["n_matches"], case["case"]
        assert {str(k): v for k, v in sorted(dedup.items())} == case["dedup"]
        assert (len(songs_arr) > 0) == case["early_exit"]
        final, b2, early = O.recognize_apriori(hs, apriori_db, case["batch_size"])
        assert _clean(final) == case["final_results"], case["case"]
        exits.append((early, batches, -(-case["n_distinct_hashes"] // case["batch_size"])))
    assert any(e and b > 1 for e, b, _ in exits), exits       # a stop after more than one batch is covered
    assert any((not e) and b == t and t > 1 for e, b, t in exits), exits   # and a query that never stops


def test_oracle_apriori_failures_of_the_reference():
    sp = G["meta"]["song_params"]
    db = O.DictDB()
    x = synth.synth_clip(sp["seed"], 0, sp["n"], sp["tone_amp"], sp["noise_amp"])
    fp = list(dict.fromkeys(O.fingerprint(x)))
    db.insert_hashes(db.insert_song("only", "AB" * 20, len(fp)), fp)
    assert G["single_song_table"] == {"first_1500_hashes": "IndexError", "empty_query": "UnboundLocalError"}
    with pytest.raises(IndexError):
        O.return_matches_apriori(fp[:1500], db)
    with pytest.raises(UnboundLocalError):
        O.return_matches_apriori([], db)
