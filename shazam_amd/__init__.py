"""shazam_amd -- the fingerprint / recognise hot path of CarlosArturoMe/shazam on MI355X.

Host side is Python + numpy + ctypes over the C ABI of ``libshz.so`` (``include/shz.h``);
every stage runs as hand-written HIP on the GPU.  There is NO CPU fallback: importing works
anywhere, but the first call needs the built library and a ROCm GPU and raises otherwise.

The functions below keep the reference's names, arguments and return types
(``__init__.py:116-245`` = ``recognizer.py:86-212`` for extraction, ``recognizer.py:222-338``
for match/align, ``__init__.py:24-27,54-67`` for the database registry) so this package can
be dropped in behind them; ``fingerprint_batch`` / ``recognize_batch`` are the batched forms
the GPU wants.
"""
from __future__ import annotations

import importlib
from operator import itemgetter
from time import time

import numpy as np

from . import _ffi
from ._ffi import HOP, NFFT, Context, ShzError, Table  # noqa: F401

# reference constants (__init__.py:41-51, recognizer.py:21-38,40-58,68)
RATE = 44100
DEFAULT_FS = 44100
DEFAULT_WINDOW_SIZE = 4096
DEFAULT_OVERLAP_RATIO = 0.5
DEFAULT_FAN_VALUE = 5
DEFAULT_AMP_MIN = 10
CONNECTIVITY_MASK = 2
PEAK_NEIGHBORHOOD_SIZE = 10
PEAK_SORT = True
MIN_HASH_TIME_DELTA = 0
MAX_HASH_TIME_DELTA = 200
FINGERPRINT_REDUCTION = 20
TOPN = 2
FIELD_FILE_SHA1 = "file_sha1"
SONG_ID = "song_id"
SONG_NAME = "song_name"
FIELD_TOTAL_HASHES = "total_hashes"
INPUT_HASHES = "input_total_hashes"
INPUT_CONFIDENCE = "input_confidence"
FINGERPRINTED_HASHES = "fingerprinted_hashes_in_db"
HASHES_MATCHED = "hashes_matched_in_input"
FINGERPRINTED_CONFIDENCE = "fingerprinted_confidence"
OFFSET = "offset"
OFFSET_SECS = "offset_seconds"

# the reference's own plugin registry (__init__.py:24-27) with the HIP table added
DATABASES = {
    "hip": ("shazam_amd.db", "HipFingerprintDB"),
    "mysql": ("mysql_database", "MySQLDatabase"),
    "postgres": ("dejavu.database_handler.postgres_database", "PostgreSQLDatabase"),
}


def get_database(database_type: str = "hip"):
    """__init__.py:54-67: class for a registry key; unknown/unimportable -> TypeError."""
    try:
        path, db_class_name = DATABASES[database_type]
        return getattr(importlib.import_module(path), db_class_name)
    except (ImportError, KeyError):
        raise TypeError("Unsupported database type supplied.")


_ctx = None


def get_context(device_id: int | None = None) -> Context:
    """Process-wide default context (device from SHZ_DEVICE / LOCAL_RANK, else 0)."""
    global _ctx
    if _ctx is None:
        import os
        if device_id is None:
            device_id = int(os.environ.get("SHZ_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _ctx = Context(device_id)
    return _ctx


# hex20 -> key32 of every hash this process has produced, so the hex-keyed reference API
# (insert_hashes / return_matches) can address the key-indexed device table.
_HEX2KEY: dict = {}


def hex_of_keys(ctx: Context, key32: np.ndarray) -> list:
    """sha1(f"{f1}|{f2}|{dt}")[:20] (lower-case hex) of packed keys, computed on the GPU."""
    key32 = np.ascontiguousarray(key32, np.uint32)
    uniq, inv = np.unique(key32, return_inverse=True)
    h = ctx.sha1_prefix(uniq).tobytes().hex()
    hexes = [h[i:i + 20] for i in range(0, len(h), 20)]
    _HEX2KEY.update(zip(hexes, uniq.tolist()))
    return [hexes[i] for i in inv.tolist()]


def keys_of_hexes(hexes, ctx: Context = None, strict: bool = True) -> np.ndarray:
    """key32 of hex hashes.  Hashes this process produced come from the cache; any other (a MySQL
    dump, another process) is inverted on the GPU by hashing the whole 8.4e8-string preimage space
    (shz_sha1_invert).  Raises KeyError for strings that are not a hash of any (f1, f2, dt)."""
    hexes = [h.lower() for h in hexes]
    out = np.empty(len(hexes), np.uint32)
    missing = []
    for i, h in enumerate(hexes):
        k = _HEX2KEY.get(h)
        if k is None:
            missing.append(i)
        else:
            out[i] = k
    if missing:
        uniq = sorted({hexes[i] for i in missing})
        try:
            dig = np.frombuffer(bytes.fromhex("".join(uniq)), np.uint8).reshape(-1, 10)
        except ValueError:
            raise KeyError("hashes must be 20 hex characters (FINGERPRINT_REDUCTION, __init__.py:51)")
        keys = (ctx or get_context()).sha1_invert(dig)
        bad = [u for u, k in zip(uniq, keys.tolist()) if k == 0xFFFFFFFF]
        if bad and strict:
            raise KeyError(f"{len(bad)} hash(es) are not sha1(f1|f2|dt)[:20] of any peak pair, e.g. {bad[0]!r}")
        found = dict(zip(uniq, keys.tolist()))
        _HEX2KEY.update((u, k) for u, k in found.items() if k != 0xFFFFFFFF)
        for i in missing:
            out[i] = found[hexes[i]]   # 0xFFFFFFFF (matches no table row) for non-hashes when not strict
    return out


def key_of_hex(hexstr: str) -> int:
    return int(keys_of_hexes([hexstr])[0])


def _as_pcm(channel_samples) -> np.ndarray:
    x = np.asarray(channel_samples)
    if x.ndim != 1:
        raise ValueError("channel_samples must be 1-D")
    if x.dtype == np.int16:
        return np.ascontiguousarray(x)
    if x.dtype.kind in "iu" or x.size == 0:
        if x.size and (x.min() < -32768 or x.max() > 32767):
            raise NotImplementedError("the HIP path takes 16-bit PCM; samples outside int16 range")
        return np.ascontiguousarray(x.astype(np.int16))
    raise NotImplementedError(f"the HIP path takes 16-bit integer PCM, got dtype {x.dtype}")


def _check_window(wsize, wratio) -> int:
    """noverlap = int(wsize * wratio) as fingerprint() hands it to mlab.specgram (__init__.py:232-237)."""
    noverlap = int(int(wsize) * float(wratio))
    if noverlap >= int(wsize):   # what mlab.specgram raises for this call (mlab:242)
        raise ValueError("noverlap must be less than NFFT")
    if noverlap < 0:
        raise ValueError("wratio must not be negative")
    w = int(wsize)
    if w != DEFAULT_WINDOW_SIZE and (w < 64 or w > 2048 or (w & (w - 1))):
        raise NotImplementedError("window sizes: 4096 (the reference's own, the fast kernel) and the powers of two 64 .. 2048 (a "
                                  "generic GPU spectrogram); 8192 has no packed key (key32 gives a frequency 12 bits), other sizes "
                                  "are not implemented, and there is no CPU fallback")
    return noverlap


def fingerprint_batch(clips, Fs: int = RATE, fan_value: int = DEFAULT_FAN_VALUE, amp_min=DEFAULT_AMP_MIN, ctx: Context = None,
                      wratio: float = DEFAULT_OVERLAP_RATIO):
    """Batched fingerprint(): clips = list of 1-D int16 arrays (or a 2-D array).
    Returns (key32, t1, hash_off): hashes of clip c are [hash_off[c], hash_off[c+1])."""
    ctx = ctx or get_context()
    noverlap = _check_window(DEFAULT_WINDOW_SIZE, wratio)
    if NFFT - noverlap != getattr(ctx, "hop", HOP):
        ctx.set_overlap(noverlap)
        try:
            return fingerprint_batch(clips, Fs, fan_value, amp_min, ctx, wratio)
        finally:
            ctx.set_overlap(NFFT - HOP)
    arrs = [_as_pcm(c) for c in clips]
    off = np.zeros(len(arrs) + 1, np.uint64)
    if arrs:
        off[1:] = np.cumsum([len(a) for a in arrs])
    pcm = np.concatenate(arrs) if arrs else np.zeros(0, np.int16)
    if pcm.size == 0:
        pcm = np.zeros(1, np.int16)
    k, t1, ho, _ = ctx.fingerprint_batch(pcm, off, fs=int(Fs), amp_min=float(amp_min), fan_value=int(fan_value))
    return k, t1, ho


def fingerprint(channel_samples, Fs: int = RATE, wsize: int = DEFAULT_WINDOW_SIZE, wratio: float = DEFAULT_OVERLAP_RATIO,
                fan_value: int = DEFAULT_FAN_VALUE, amp_min: int = DEFAULT_AMP_MIN):
    """__init__.py:212-245: FFT the channel, log transform, local maxima, pair hashes.
    Returns list[(hex20, t1)] in the reference's generation order."""
    noverlap = _check_window(wsize, wratio)
    ctx = get_context()
    if int(wsize) != DEFAULT_WINDOW_SIZE:
        # another window size: the reference's own composition (__init__.py:232-245) of GPU stages -- generic spectrogram,
        # get_2D_peaks on the dB array, generate_hashes on the peaks
        arr2D = ctx.stft_db_any(_as_pcm(channel_samples), int(Fs), int(wsize), noverlap)
        return generate_hashes(get_2D_peaks(arr2D, plot=False, amp_min=amp_min), fan_value=fan_value)
    k, t1, _ = fingerprint_batch([channel_samples], Fs, fan_value, amp_min, ctx, wratio)
    return list(zip(hex_of_keys(ctx, k), t1.tolist()))


def get_2D_peaks(arr2D, plot: bool = False, amp_min: int = DEFAULT_AMP_MIN):
    """__init__.py:116-177: list[(freq, time)] of the 21x21 local maxima above amp_min,
    in np.where order (freq asc, time asc)."""
    if plot:
        raise NotImplementedError("plotting is out of scope of the HIP path")
    f, t = get_context().peaks_from_db(np.asarray(arr2D, np.float64), float(amp_min))
    return list(zip(f.astype(np.int64), t.astype(np.int64)))


def generate_hashes(peaks, fan_value: int = DEFAULT_FAN_VALUE):
    """__init__.py:179-210: sorts ``peaks`` in place by time (stable) like the reference, then
    pairs every peak with its next fan_value-1 successors (0 <= dt <= 200)."""
    if PEAK_SORT:
        peaks.sort(key=itemgetter(1))
    ctx = get_context()
    f = np.array([p[0] for p in peaks], np.int64)
    t = np.array([p[1] for p in peaks], np.int64)
    if len(f) and (f.min() < 0 or f.max() > 4095 or t.min() < 0 or t.max() >= 2 ** 32):
        raise NotImplementedError("peak coordinates outside the packed-key range (freq < 4096)")
    k, t1, _ = ctx.pair_hash(f.astype(np.uint16), t.astype(np.uint32), np.array([0, len(f)], np.uint64), int(fan_value))
    return list(zip(hex_of_keys(ctx, k), t1.tolist()))


# ---- match / align (recognizer.py:222-338) -------------------------------------------------
def _result_dicts(db, res, q, queried_hashes):
    out = []
    for n in range(int(res["nres"][q])):
        song_id = int(res["sid"][q, n])
        song = db.get_song_by_id(song_id)
        offset = int(res["delta"][q, n])
        hashes_matched = int(res["dedup"][q, n])
        song_hashes = song.get(FIELD_TOTAL_HASHES, None)
        out.append({
            SONG_ID: song_id,
            SONG_NAME: song.get(SONG_NAME, None).encode("utf8"),
            INPUT_HASHES: queried_hashes,
            FINGERPRINTED_HASHES: song_hashes,
            HASHES_MATCHED: hashes_matched,
            INPUT_CONFIDENCE: round(hashes_matched / queried_hashes, 2),
            FINGERPRINTED_CONFIDENCE: round(hashes_matched / song_hashes, 2),
            OFFSET: offset,
            OFFSET_SECS: round(float(offset) / DEFAULT_FS * DEFAULT_WINDOW_SIZE * DEFAULT_OVERLAP_RATIO, 5),
            FIELD_FILE_SHA1: song.get(FIELD_FILE_SHA1, None).encode("utf8"),
        })
    return out


def recognize_batch(queries, db, Fs: int = RATE, topn: int = TOPN):
    """Recognise flow (recognizer.py:377-392) for many queries at once.  Each query is a list of
    channels (1-D int16 arrays) or a single 1-D array.  Returns (results_per_query, timings)."""
    ctx = db.ctx
    chans, owner = [], []
    for qi, q in enumerate(queries):
        cs = [q] if (isinstance(q, np.ndarray) and q.ndim == 1) else list(q)
        chans.extend(cs)
        owner.extend([qi] * len(cs))
    t0 = time()
    k, t1, ho = fingerprint_batch(chans, Fs, ctx=ctx)
    fingerprint_time = time() - t0
    # channels of one query are adjacent, so the per-query CSR is a sub-sampling of hash_off
    owner = np.asarray(owner, np.int64)
    nq = len(queries)
    first = np.searchsorted(owner, np.arange(nq + 1))
    qoff = ho[first]
    t0 = time()
    res = db.match(k, t1, qoff, topn)
    query_time = time() - t0
    t0 = time()
    results = [_result_dicts(db, res, q, int(res["nhash"][q])) for q in range(nq)]
    align_time = time() - t0
    return results, {"fingerprint_time": fingerprint_time, "query_time": query_time, "align_time": align_time,
                     "n_matches": res["npairs"], "n_hashes": res["nhash"]}


def recognize(channels_or_samples, db=None, Fs: int = RATE, topn: int = TOPN):
    """recognizer.py:377-396: (final_results, fingerprint_time, query_time, align_time)."""
    if db is None:
        raise ValueError("recognize() needs the HipFingerprintDB holding the fingerprints")
    x = channels_or_samples
    q = [np.asarray(x)] if (not isinstance(x, (list, tuple)) or (len(x) and np.isscalar(x[0]))) else list(x)
    results, tm = recognize_batch([q], db, Fs, topn)
    return results[0], tm["fingerprint_time"], tm["query_time"], tm["align_time"]
