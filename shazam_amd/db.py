"""HipFingerprintDB -- the HBM-resident table behind the reference's duck-typed database
interface (MySQLDatabase, mysql_database.py:28-255), registrable as ``DATABASES['hip']``.

What lives where: the ``fingerprints`` rows (hash, song_id, offset) are device columns inside
``libshz.so`` (sorted by key, UNIQUE(song_id, offset, hash) enforced at finalize like the
MySQL constraint + INSERT IGNORE); the tiny ``songs`` table is a host dict.  ``cursor()``
emulates just enough of the SQL surface for the reference's own ``return_matches``
(recognizer.py:251-259) to run unchanged against it; ``match()`` is the fast batched path.
"""
from __future__ import annotations

from contextlib import contextmanager
from datetime import datetime

import numpy as np

from . import _ffi
from . import get_context, hex_of_keys, keys_of_hexes


class _Cursor:
    def __init__(self, db):
        self.db, self.rows, self.lastrowid = db, [], None

    def execute(self, query, values=()):
        q = " ".join(str(query).split()).upper()
        self.rows = []
        if q.startswith("SELECT HEX(`HASH`)") and " IN (" in q:          # SELECT_MULTIPLE
            self.rows = self.db._select_multiple(list(values))
        elif q.startswith("CREATE TABLE"):
            pass
        elif q.startswith("DROP TABLE"):       # empty() of the reference drops both tables (mysql_database.py:143-153)
            self.db.drop_table("songs" if "SONGS" in q else "fingerprints")
        elif q.startswith("DELETE FROM `SONGS` WHERE `FINGERPRINTED` = 0"):  # DELETE_UNFINGERPRINTED
            self.db.delete_unfingerprinted()
        else:
            raise NotImplementedError(f"HipFingerprintDB cursor does not implement: {query!r}")

    def executemany(self, query, seq):
        q = " ".join(str(query).split()).upper()
        if q.startswith("INSERT IGNORE INTO `FINGERPRINTS`"):                # INSERT_FINGERPRINT
            rows = list(seq)
            if rows:
                self.db._insert_rows([r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows])
        else:
            raise NotImplementedError(f"HipFingerprintDB cursor does not implement: {query!r}")

    def __iter__(self):
        return iter(self.rows)

    def fetchone(self):
        return self.rows[0] if self.rows else None


class HipFingerprintDB:
    type = "hip"

    # SQL text the reference passes to cursor.execute (mysql_database.py:32-141); kept so callers
    # that reference db.CREATE_* / db.SELECT_MULTIPLE keep working -- the cursor recognises them.
    CREATE_SONGS_TABLE = "CREATE TABLE IF NOT EXISTS `songs` (...)"
    CREATE_FINGERPRINTS_TABLE = "CREATE TABLE IF NOT EXISTS `fingerprints` (...)"
    DELETE_UNFINGERPRINTED = "DELETE FROM `songs` WHERE `fingerprinted` = 0;"
    INSERT_FINGERPRINT = "INSERT IGNORE INTO `fingerprints` (`song_id`, `hash`, `offset`) VALUES (%s, UNHEX(%s), %s);"
    SELECT_MULTIPLE = "SELECT HEX(`hash`), `song_id`, `offset` FROM `fingerprints` WHERE `hash` IN (%s);"
    IN_MATCH = "UNHEX(%s)"

    def __init__(self, device: int | None = None, ctx: _ffi.Context = None, shards: int = 1, **options):
        """shards > 1: the rows are partitioned by key into that many tables on this GPU (shazam_amd/shard.py);
        results are identical, each shard's sort scratch is 1/shards of the whole."""
        self.ctx = ctx or get_context(device)
        if int(shards) > 1:
            from .shard import ShardedTable
            self.table = ShardedTable(self.ctx, nshards=int(shards))
        else:
            self.table = _ffi.Table(self.ctx)
        self.songs = {}          # sid -> dict(song_name, file_sha1, total_hashes, fingerprinted, date_created)
        self._next_sid = 1       # AUTO_INCREMENT (mysql_database.py:34)
        self._dirty = False
        self._options = options

    # ---- lifecycle ---------------------------------------------------------------------------
    def setup(self) -> None:
        self.delete_unfingerprinted()

    def after_fork(self) -> None:
        raise RuntimeError("a GPU context cannot be shared across fork(); create the DB in the child")

    def close(self):
        self.table.close()

    @contextmanager
    def cursor(self, **options):
        yield _Cursor(self)

    # ---- songs table (host) ---------------------------------------------------------------------
    def insert_song(self, song_name: str, file_hash: str, total_hashes: int) -> int:
        sid = self._next_sid
        self._next_sid += 1
        self.songs[sid] = {"song_name": song_name, "file_sha1": str(file_hash).upper(), "total_hashes": int(total_hashes),
                           "fingerprinted": 0, "date_created": datetime.now()}
        return sid

    def set_song_fingerprinted(self, song_id):
        self.songs[song_id]["fingerprinted"] = 1

    def delete_unfingerprinted(self):
        """DELETE_UNFINGERPRINTED (mysql_database.py:132-134) with the schema's ON DELETE CASCADE (:57-58): songs whose
        ingest never reached set_song_fingerprinted disappear together with their fingerprint rows, so a later match
        can neither return them nor spend a top-n slot on them."""
        gone = [s for s, v in self.songs.items() if not v["fingerprinted"]]
        if gone:
            self.table.delete_songs(gone)
            for sid in gone:
                del self.songs[sid]

    def drop_table(self, which: str):
        """DROP TABLE IF EXISTS `songs` / `fingerprints` (DROP_SONGS / DROP_FINGERPRINTS, mysql_database.py:143-153)."""
        if which == "songs":
            self.songs.clear()
            self._next_sid = 1
        self.table.clear()       # fingerprints reference songs ON DELETE CASCADE: either drop empties them
        self._dirty = False

    def empty(self):
        """MySQLDatabase.empty(): drop both tables and set them up again."""
        self.drop_table("fingerprints")
        self.drop_table("songs")
        self.setup()

    def get_songs(self):
        """SELECT_SONGS rows: (song_id, song_name, HEX(file_sha1), total_hashes, date_created)."""
        return [(sid, s["song_name"], s["file_sha1"], s["total_hashes"], s["date_created"])
                for sid, s in sorted(self.songs.items()) if s["fingerprinted"]]

    def get_song_by_id(self, song_id: int):
        s = self.songs[int(song_id)]
        return {"song_name": s["song_name"], "total_hashes": s["total_hashes"], "file_sha1": s["file_sha1"]}

    def get_metadata(self, song_id: int):
        raise NotImplementedError("the FMA METADATA table is external to the fingerprint path")

    # ---- fingerprints table (device) -------------------------------------------------------------
    def _insert_rows(self, sids, hexes, offsets):
        keys = keys_of_hexes(hexes, self.ctx)
        self.table.insert(keys, np.asarray(sids, np.uint32), np.asarray([int(o) for o in offsets], np.uint32))
        self._dirty = True

    def insert_hashes(self, song_id: int, hashes, batch_size: int = 1000):
        """mysql_database.py:167-181: (hex, offset) pairs of one song; duplicates ignored."""
        hashes = list(hashes)
        if hashes:
            self._insert_rows(np.full(len(hashes), song_id, np.uint32), [h for h, _ in hashes], [o for _, o in hashes])

    def insert_keys(self, song_id, key32, offsets):
        """Packed-key form of insert_hashes (no hex round trip); song_id scalar or array."""
        self.table.insert(key32, song_id, offsets)
        self._dirty = True

    def insert_clips(self, key32, t1, hash_off, sid0, device=False):
        """Rows of many songs at once: clip c of the CSR gets song id sid0 + c."""
        self.table.insert_clips(key32, t1, hash_off, sid0, device=device)
        self._dirty = True

    def finalize(self):
        if self._dirty or self.table.rows()[1] or self.table.rows()[0] == 0:
            self.table.finalize()
            self._dirty = False

    def _select_multiple(self, hex_values):
        self.finalize()
        if not hex_values:
            return []
        k, s, o = self.table.lookup(keys_of_hexes(hex_values, self.ctx, strict=False))
        hexes = hex_of_keys(self.ctx, k)
        return [(h.upper(), int(a), int(b)) for h, a, b in zip(hexes, s.tolist(), o.tolist())]  # HEX() upper-cases

    def match(self, key32, q_off, query_off, topn=2):
        self.finalize()
        return self.table.match(key32, q_off, query_off, topn)

    # ---- dump / load / export (checkpoint-resume of a long build) ----------------------
    def save(self, path: str):
        """Write the table (sorted unique rows) and the songs dict to one .npz file."""
        self.finalize()
        k, s, o = self.table.export()
        sids = sorted(self.songs)
        np.savez(path, key32=k, song_id=s, offset=o, next_sid=np.int64(self._next_sid),
                 songs_sid=np.array(sids, np.int64),
                 songs_name=np.array([self.songs[i]["song_name"] for i in sids], dtype=object).astype("U"),
                 songs_sha1=np.array([self.songs[i]["file_sha1"] for i in sids], dtype=object).astype("U"),
                 songs_total=np.array([self.songs[i]["total_hashes"] for i in sids], np.int64),
                 songs_fp=np.array([self.songs[i]["fingerprinted"] for i in sids], np.int8))

    @classmethod
    def load(cls, path: str, **options):
        z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz", allow_pickle=False)
        db = cls(**options)
        if len(z["key32"]):
            db.table.insert(z["key32"], z["song_id"], z["offset"])
        db.table.finalize()
        for sid, name, sha, tot, fp in zip(z["songs_sid"].tolist(), z["songs_name"].tolist(), z["songs_sha1"].tolist(),
                                           z["songs_total"].tolist(), z["songs_fp"].tolist()):
            db.songs[sid] = {"song_name": name, "file_sha1": sha, "total_hashes": int(tot), "fingerprinted": int(fp),
                             "date_created": datetime.now()}
        db._next_sid = int(z["next_sid"])
        return db

    def export_mysql_rows(self, chunk: int = 1 << 20):
        """Yield (song_id, hash10 bytes, offset) in table order: the VALUES of the reference's
        INSERT_FINGERPRINT after UNHEX (mysql_database.py:62-68), BINARY(10) digests from the GPU."""
        self.finalize()
        k, s, o = self.table.export()
        for i in range(0, len(k), chunk):
            dig = self.ctx.sha1_prefix(k[i:i + chunk])
            for j in range(len(dig)):
                yield int(s[i + j]), dig[j].tobytes(), int(o[i + j])

    def num_fingerprints(self) -> int:
        self.finalize()
        return self.table.rows()[0]
