"""GPU: early-exit recognition (shazam_amd/apriori.py) against the goldens made by running the reference's
recognizer_apriori.py definitions (tests/golden/make_golden_apriori.py): same ordered hash list in, same result dicts,
same stop / no-stop decision, same failures."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "apriori_cases.json")))


def _query(case):
    from oracle import synth
    import shazam_amd as S
    sp = G["meta"]["song_params"]
    src = 2 if case["song"] == 5 else case["song"]
    x = synth.synth_clip(sp["seed"], src, sp["n"], sp["tone_amp"], sp["noise_amp"])
    x = x[case["start_frame"] * 2048:(case["start_frame"] + case["frames"]) * 2048]
    if case["snr"] is not None:
        x = synth.mix_query(x, synth.synth_clip(99, case["noise_clip"], len(x), 0, 8000), case["snr"])
    hs = list(dict.fromkeys(S.fingerprint(x)))
    if case["order"] == "reversed":
        hs = hs[::-1]
    elif case["order"] == "shuffled":
        hs = [hs[i] for i in np.random.default_rng(case["case"]).permutation(len(hs))]
    elif case["order"] == "by_offset_desc":
        hs = sorted(hs, key=lambda t: (-t[1], t[0]))
    return x, hs


def _clean(res):
    return [{k: (v.decode() if isinstance(v, bytes) else v) for k, v in r.items()} for r in res]


def test_apriori_equals_reference():
    import shazam_amd as S
    from shazam_amd.apriori import find_matches_apriori, recognize_apriori
    from oracle import synth
    ctx = S.get_context(0)
    sp = G["meta"]["song_params"]
    db = S.get_database("hip")(ctx=ctx)
    for s in G["songs"]:
        x = synth.synth_clip(sp["seed"], s["source_clip"], sp["n"], sp["tone_amp"], sp["noise_amp"])
        fp = list(dict.fromkeys(S.fingerprint(x)))
        sid = db.insert_song(f"a{s['song']:02d}", hashlib.sha1(x.tobytes()).hexdigest().upper(), len(set(fp)))
        assert sid == s["sid"]
        db.insert_hashes(sid, fp)
        db.set_song_fingerprinted(sid)
    for case in G["cases"]:
        x, hs = _query(case)
        assert len(hs) == case["n_hashes"]
        final, early, batches, n_matches, _ = find_matches_apriori(hs, db, case["batch_size"])
        assert early == case["early_exit"], case["case"]
        assert n_matches == case["n_matches"], case["case"]
        assert _clean(final) == case["final_results"], case["case"]
        # packed keys instead of hex strings: the same answer
        k = S.keys_of_hexes([h for h, _ in hs], ctx)
        o = np.array([t for _, t in hs], np.uint32)
        f2, e2, b2, n2, _ = find_matches_apriori((k, o), db, case["batch_size"])
        assert (_clean(f2), e2, b2, n2) == (_clean(final), early, batches, n_matches)
    # the flow entry point: generation order = the order of cases with order "generation"
    case = G["cases"][0]
    x, hs = _query(case)
    res, t_fp, t_q, t_al = recognize_apriori(x, db, batch_size=case["batch_size"])
    assert _clean(res) == case["final_results"] and t_al == 0
    with pytest.raises(ValueError):
        find_matches_apriori(hs + hs[:1], db)
    db.close()


def test_apriori_failures_of_the_reference():
    import shazam_amd as S
    from shazam_amd.apriori import find_matches_apriori
    from oracle import synth
    ctx = S.get_context(0)
    sp = G["meta"]["song_params"]
    db = S.get_database("hip")(ctx=ctx)
    x = synth.synth_clip(sp["seed"], 0, sp["n"], sp["tone_amp"], sp["noise_amp"])
    fp = list(dict.fromkeys(S.fingerprint(x)))
    db.insert_hashes(db.insert_song("only", "AB" * 20, len(fp)), fp)
    with pytest.raises(IndexError):                       # recognizer_apriori.py:303 with one song in the table
        find_matches_apriori(fp[:1500], db)
    with pytest.raises(UnboundLocalError):                # :310 with an empty query
        find_matches_apriori([], db)
    final, early, batches, n, _ = find_matches_apriori(fp[:1500], db, strict=False)
    assert early and batches == 1 and final[0]["song_id"] == 1 and final[0]["hashes_matched_in_input"] == n > 0
    assert find_matches_apriori([], db, strict=False)[0] == []
    db.close()
