"""GPU: foreign hex hashes (not produced by this process) are resolved by inverting SHA-1 over the
preimage space; the reference's hex-keyed DB calls then work with hashes from anywhere."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sha1_invert_and_foreign_hex_insert():
    import shazam_amd as S
    ctx = S.get_context(0)
    rng = np.random.default_rng(4)
    f1, f2, dt = rng.integers(0, 2049, 500), rng.integers(0, 2049, 500), rng.integers(0, 201, 500)
    f1[:4], f2[:4], dt[:4] = [0, 2048, 7, 999], [0, 2048, 10, 1000], [0, 200, 9, 100]   # digit-count corners
    keys = (f1.astype(np.uint32) << 20) | (f2.astype(np.uint32) << 8) | dt.astype(np.uint32)
    hexes = [hashlib.sha1(b"%d|%d|%d" % (a, b, c)).hexdigest()[:20] for a, b, c in zip(f1, f2, dt)]
    dig = np.frombuffer(bytes.fromhex("".join(hexes)), np.uint8).reshape(-1, 10)
    got = ctx.sha1_invert(np.concatenate([dig, np.zeros((1, 10), np.uint8), dig[:3]]))
    assert np.array_equal(got[:500], keys) and got[500] == 0xFFFFFFFF and np.array_equal(got[501:], keys[:3])
    # hex-keyed reference API with hashes this process never produced (upper-case like MySQL HEX())
    S._HEX2KEY.clear()
    db = S.get_database("hip")(ctx=ctx)
    sid = db.insert_song("ext", "AB" * 20, 500)
    db.insert_hashes(sid, [(h.upper(), i) for i, h in enumerate(hexes)])
    db.finalize()
    k, s, o = db.table.export()
    assert sorted(zip(k.tolist(), o.tolist())) == sorted(zip(keys.tolist(), range(500)))
    with db.cursor() as cur:
        cur.execute(db.SELECT_MULTIPLE % ", ".join([db.IN_MATCH] * 3), [hexes[5].upper(), hexes[6].upper(), "00" * 10])
        rows = list(cur)
    assert sorted(rows) == sorted([(hexes[5].upper(), sid, 5), (hexes[6].upper(), sid, 6)])
    with pytest.raises(KeyError):
        db.insert_hashes(sid, [("zz" * 10, 1)])
