"""numpy restatement of the reference's fingerprint + match hot path (CHECKER ONLY).

TEST INFRASTRUCTURE: see ``oracle/__init__.py``.  Every function cites the
reference lines it restates (paths relative to the reference repo root; ``mlab:``
= matplotlib/mlab.py 3.10.8, the third-party routine the reference calls).

Pinned by ``tests/golden/*.npz|json`` which hold outputs of the reference's own
``fingerprint`` / ``get_2D_peaks`` / ``generate_hashes`` / ``return_matches`` /
``align_matches`` run in the build container (``tests/golden/make_golden.py``).
"""
from __future__ import annotations

import hashlib
from itertools import groupby

import numpy as np

# reference constants, __init__.py:41-51 == recognizer.py:21-38
RATE = 44100
DEFAULT_WINDOW_SIZE = 4096
DEFAULT_OVERLAP_RATIO = 0.5
DEFAULT_FAN_VALUE = 5
DEFAULT_AMP_MIN = 10
PEAK_NEIGHBORHOOD_SIZE = 10
MIN_HASH_TIME_DELTA = 0
MAX_HASH_TIME_DELTA = 200
FINGERPRINT_REDUCTION = 20
TOPN = 2  # recognizer.py:68


# --------------------------------------------------------------------------- STFT
def frame_count(n_samples: int, nfft: int = DEFAULT_WINDOW_SIZE, noverlap: int | None = None) -> int:
    """Frames mlab produces: sliding windows of nfft with hop nfft-noverlap
    (mlab:307-308); inputs shorter than nfft are zero-padded to one frame
    (mlab:268-271)."""
    if noverlap is None:
        noverlap = nfft // 2
    n = max(int(n_samples), nfft)
    return (n - nfft) // (nfft - noverlap) + 1


def hann(m: int) -> np.ndarray:
    """np.hanning(m) restated: symmetric Hann, 0.5 - 0.5 cos(2 pi n / (m-1))
    (mlab.window_hanning = np.hanning(len(x)) * x, mlab:58-66)."""
    n = np.arange(1 - m, m, 2, dtype=np.float64)
    return 0.5 + 0.5 * np.cos(np.pi * n / (m - 1))


def stft_psd(x, Fs: int = RATE, nfft: int = DEFAULT_WINDOW_SIZE, noverlap: int | None = None) -> np.ndarray:
    """One-sided PSD exactly as mlab.specgram(..., mode='psd') computes it
    (call site __init__.py:232-237; mlab:213-373): frames x[k*hop : k*hop+nfft],
    no detrend, Hann window, |FFT|^2 on bins 0..nfft/2, bins 1..nfft/2-1 doubled,
    / Fs, / sum(w^2).  Returns float64 [nfft/2+1, F] (freq-major)."""
    if noverlap is None:
        noverlap = nfft // 2
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("1-D samples expected")
    if len(x) < nfft:  # mlab:268-271
        xp = np.zeros(nfft, dtype=x.dtype if x.size else np.int16)
        xp[: len(x)] = x
        x = xp
    hop = nfft - noverlap
    w = hann(nfft)
    frames = np.lib.stride_tricks.sliding_window_view(x, nfft)[::hop]  # [F, nfft]
    out = np.empty((nfft // 2 + 1, frames.shape[0]), dtype=np.float64)
    scale = 1.0 / (float(Fs) * float((w * w).sum()))
    # chunked to bound memory on 3-minute tracks
    for s in range(0, frames.shape[0], 512):
        X = np.fft.rfft(frames[s : s + 512].astype(np.float64) * w, axis=1)
        P = X.real * X.real + X.imag * X.imag
        P[:, 1:-1] *= 2.0  # mlab:339-345 (DC and Nyquist unscaled)
        P *= scale  # mlab:350-354
        out[:, s : s + 512] = P.T
    return out


def log_db(P: np.ndarray) -> np.ndarray:
    """10*log10 where P != 0, else 0.0 (__init__.py:241)."""
    A = np.zeros_like(P)
    nz = P != 0
    A[nz] = 10.0 * np.log10(P[nz])
    return A


def spectrogram_db(x, Fs: int = RATE, nfft: int = DEFAULT_WINDOW_SIZE, wratio: float = DEFAULT_OVERLAP_RATIO):
    return log_db(stft_psd(x, Fs, nfft, int(nfft * wratio)))


# --------------------------------------------------------------------------- peaks
def _running_max(a: np.ndarray, r: int, axis: int) -> np.ndarray:
    """max over the window [i-r, i+r] clipped to the array (scipy maximum_filter
    with mode='reflect' equals edge truncation for a max filter)."""
    a = np.moveaxis(a, axis, -1)
    n = a.shape[-1]
    pad = np.full(a.shape[:-1] + (r,), -np.inf, dtype=a.dtype)
    p = np.concatenate([pad, a, pad], axis=-1)
    w = 2 * r + 1
    # out[i] = max(p[i : i+w]); tables[k][i] = max(p[i : i+k]) for k a power of two
    tables = {1: p}
    k = 1
    while k * 2 <= w:
        tables[k * 2] = np.maximum(tables[k][..., : tables[k].shape[-1] - k], tables[k][..., k:])
        k *= 2
    out, off, rem = None, 0, w
    for k in sorted(tables, reverse=True):
        while rem >= k:
            seg = tables[k][..., off : off + n]
            out = seg.copy() if out is None else np.maximum(out, seg)
            off += k
            rem -= k
    return np.moveaxis(out, -1, axis)


def peaks_2d(A: np.ndarray, amp_min: float = DEFAULT_AMP_MIN, r: int = PEAK_NEIGHBORHOOD_SIZE):
    """get_2D_peaks (__init__.py:116-177):
    peak <=> A[f,t] == max(A[f-r..f+r, t-r..t+r] within the array) and A[f,t] > amp_min,
    except zero-valued cells whose whole window is zero (`local_max != binary_erosion(A == 0, ..., border_value=1)`,
    __init__.py:147-151: the XOR removes local maxima that are eroded background; outside the array counts as zero).
    For amp_min >= 0 the exception is void -- such cells fail `> amp_min` anyway.
    Returns (freqs, times) int64 arrays in np.where row-major order (freq asc, time asc)."""
    m = _running_max(_running_max(A, r, 0), r, 1)
    det = (m == A) & (A > amp_min)
    if amp_min < 0:
        nz = (A != 0).astype(np.float64)                       # any non-zero cell in the (clipped) window?
        any_nz = _running_max(_running_max(nz, r, 0), r, 1) > 0
        det &= ~((A == 0) & ~any_nz)
    f, t = np.where(det)
    return f.astype(np.int64), t.astype(np.int64)


def sort_peaks(f: np.ndarray, t: np.ndarray):
    """peaks.sort(key=itemgetter(1)) -- stable by time (__init__.py:194-195);
    input is (freq asc, time asc) so the result is (time asc, freq asc)."""
    order = np.argsort(t, kind="stable")
    return f[order], t[order]


# --------------------------------------------------------------------------- hashes
def pack_key(f1, f2, dt):
    """Injective 32-bit packing of the SHA-1 preimage (SURVEY 7 hard part 3)."""
    return (np.asarray(f1, np.uint32) << np.uint32(20)) | (np.asarray(f2, np.uint32) << np.uint32(8)) | np.asarray(dt, np.uint32)


def unpack_key(key):
    key = np.asarray(key, np.uint32)
    return (key >> np.uint32(20)).astype(np.int64), ((key >> np.uint32(8)) & np.uint32(0xFFF)).astype(np.int64), (key & np.uint32(0xFF)).astype(np.int64)


def pair_keys(f: np.ndarray, t: np.ndarray, fan_value: int = DEFAULT_FAN_VALUE):
    """generate_hashes (__init__.py:179-210) on time-major-sorted peaks, packed:
    for i, for j in 1..fan_value-1, i+j<n, 0<=dt<=200 -> (key32, t1) in (i, j) order."""
    n = len(f)
    if n == 0:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32)
    nj = max(fan_value - 1, 0)
    i = np.repeat(np.arange(n), nj)
    j = np.tile(np.arange(1, nj + 1), n)
    k = i + j
    ok = k < n
    i, k = i[ok], k[ok]
    dt = t[k] - t[i]
    ok = (dt >= MIN_HASH_TIME_DELTA) & (dt <= MAX_HASH_TIME_DELTA)
    i, k, dt = i[ok], k[ok], dt[ok]
    return pack_key(f[i], f[k], dt), t[i].astype(np.uint32)


def sha1_hex20(key32) -> list:
    """sha1(f"{f1}|{f2}|{dt}")[:20] lowercase hex (__init__.py:207-208)."""
    f1, f2, dt = unpack_key(key32)
    return [hashlib.sha1(b"%d|%d|%d" % (a, b, c)).hexdigest()[:FINGERPRINT_REDUCTION]
            for a, b, c in zip(f1.tolist(), f2.tolist(), dt.tolist())]


def sha1_prefix10(key32) -> np.ndarray:
    """First 10 digest bytes per key, uint8 [n, 10] (BINARY(10), mysql_database.py:48)."""
    f1, f2, dt = unpack_key(key32)
    out = np.empty((len(f1), 10), np.uint8)
    for n, (a, b, c) in enumerate(zip(f1.tolist(), f2.tolist(), dt.tolist())):
        out[n] = np.frombuffer(hashlib.sha1(b"%d|%d|%d" % (a, b, c)).digest()[:10], np.uint8)
    return out


def fingerprint_keys(x, Fs: int = RATE, wsize: int = DEFAULT_WINDOW_SIZE, wratio: float = DEFAULT_OVERLAP_RATIO,
                     fan_value: int = DEFAULT_FAN_VALUE, amp_min: float = DEFAULT_AMP_MIN):
    """fingerprint() (__init__.py:212-245) in packed form: (key32, t1, peak_f, peak_t)."""
    A = spectrogram_db(x, Fs, wsize, wratio)
    f, t = sort_peaks(*peaks_2d(A, amp_min))
    k, t1 = pair_keys(f, t, fan_value)
    return k, t1, f, t


def fingerprint(x, Fs: int = RATE, wsize: int = DEFAULT_WINDOW_SIZE, wratio: float = DEFAULT_OVERLAP_RATIO,
                fan_value: int = DEFAULT_FAN_VALUE, amp_min: float = DEFAULT_AMP_MIN):
    """fingerprint() with the reference's return type: list[(hex20, t1)]."""
    k, t1, _, _ = fingerprint_keys(x, Fs, wsize, wratio, fan_value, amp_min)
    return list(zip(sha1_hex20(k), [int(v) for v in t1]))


# --------------------------------------------------------------------------- store + match
class DictDB:
    """In-memory stand-in for the MySQL schema: fingerprints(hash, song_id, offset)
    UNIQUE(song_id, offset, hash) with INSERT IGNORE (mysql_database.py:46-68);
    songs(song_id auto-increment from 1, song_name, fingerprinted, file_sha1,
    total_hashes) (mysql_database.py:32-44, 188-200)."""

    def __init__(self):
        self.rows = {}      # HASH (upper hex or int) -> list[(sid, off)]
        self.seen = set()   # (sid, off, hash)
        self.songs = {}     # sid -> dict

    def insert_song(self, song_name, file_hash, total_hashes):
        sid = len(self.songs) + 1
        self.songs[sid] = {"song_name": song_name, "file_sha1": file_hash,
                           "total_hashes": total_hashes, "fingerprinted": 0}
        return sid

    def insert_hashes(self, sid, hashes, batch_size=1000):
        for h, off in hashes:
            h = h.upper() if isinstance(h, str) else int(h)
            k = (sid, int(off), h)
            if k in self.seen:
                continue
            self.seen.add(k)
            self.rows.setdefault(h, []).append((sid, int(off)))

    def set_song_fingerprinted(self, sid):
        self.songs[sid]["fingerprinted"] = 1

    def get_song_by_id(self, sid):
        s = self.songs[sid]
        return {"song_name": s["song_name"], "total_hashes": s["total_hashes"], "file_sha1": s["file_sha1"]}

    def select_multiple(self, values):
        """SELECT HEX(hash), song_id, offset WHERE hash IN (values) (recognizer.py:60-65)."""
        for h in values:
            for sid, off in self.rows.get(h, ()):
                yield h, sid, off


def return_matches(hashes, db: DictDB):
    """recognizer.py:222-271: mapper hash->[q_off]; for every DB row whose hash is
    queried: dedup_hashes[sid] += 1 (once per row), and one (sid, db_off - q_off)
    per query offset of that hash."""
    mapper = {}
    for h, off in hashes:
        h = h.upper() if isinstance(h, str) else int(h)
        mapper.setdefault(h, []).append(int(off))
    dedup = {}
    results = []
    for h, sid, off in db.select_multiple(list(mapper.keys())):
        dedup[sid] = dedup.get(sid, 0) + 1
        for q in mapper[h]:
            results.append((sid, off - q))
    return results, dedup


def align_matches(matches, dedup_hashes, queried_hashes, db: DictDB, topn: int = TOPN):
    """recognizer.py:289-338: count per (sid, delta); per sid the first max in
    delta-ascending order; songs by count desc (stable: ties -> smaller sid); topn dicts."""
    sorted_matches = sorted(matches, key=lambda m: (m[0], m[1]))
    counts = [(*key, len(list(group))) for key, group in groupby(sorted_matches, key=lambda m: (m[0], m[1]))]
    songs_matches = sorted(
        [max(list(group), key=lambda g: g[2]) for key, group in groupby(counts, key=lambda c: c[0])],
        key=lambda c: c[2], reverse=True)
    out = []
    for song_id, offset, aligned in songs_matches[0:topn]:
        song = db.get_song_by_id(song_id)
        hashes_matched = dedup_hashes[song_id]
        name = song["song_name"]
        sha = song["file_sha1"]
        out.append({
            "song_id": song_id,
            "song_name": name.encode("utf8") if isinstance(name, str) else name,
            "input_total_hashes": queried_hashes,
            "fingerprinted_hashes_in_db": song["total_hashes"],
            "hashes_matched_in_input": hashes_matched,
            "input_confidence": round(hashes_matched / queried_hashes, 2),
            "fingerprinted_confidence": round(hashes_matched / song["total_hashes"], 2),
            "offset": offset,
            "offset_seconds": round(float(offset) / RATE * DEFAULT_WINDOW_SIZE * DEFAULT_OVERLAP_RATIO, 5),
            "file_sha1": sha.encode("utf8") if isinstance(sha, str) else sha,
        })
    return out


def vote(matches, topn: int = TOPN):
    """The (sid, delta, aligned_count) triples align_matches ranks, before the dict
    build -- what the device vote kernel must reproduce."""
    sorted_matches = sorted(matches, key=lambda m: (m[0], m[1]))
    counts = [(*key, len(list(group))) for key, group in groupby(sorted_matches, key=lambda m: (m[0], m[1]))]
    songs = sorted([max(list(g), key=lambda c: c[2]) for _, g in groupby(counts, key=lambda c: c[0])],
                   key=lambda c: c[2], reverse=True)
    return songs[:topn]


def recognize(channels, db: DictDB, Fs: int = RATE, topn: int = TOPN):
    """Recognise flow recognizer.py:377-392: union of per-channel fingerprints,
    find_matches, align_matches(len(hashes))."""
    hashes = set()
    for ch in channels:
        hashes |= set(fingerprint(ch, Fs=Fs))
    matches, dedup = return_matches(hashes, db)
    return align_matches(matches, dedup, len(hashes), db, topn)


def return_matches_apriori(hashes, db: DictDB, batch_size: int = 1000):
    """recognizer_apriori.py:237-310 (the early-exit variant of return_matches): distinct hashes in first-occurrence
    order are looked up in batches of `batch_size`; after every batch the matches so far are aligned (topn = TOPN,
    queried_hashes = len(hashes)) and the loop stops as soon as the leader has more than twice the runner-up's
    `hashes_matched_in_input` (:302-305); if it never does, songs_arr ends as [] (:307).  Like the reference this
    raises IndexError while fewer than two songs have matched (`songs_arr[1]`, :303) and UnboundLocalError for an empty
    query (:310).  Returns (results, dedup_hashes, songs_arr, batches_looked_up)."""
    hashes = list(hashes)
    mapper = {}
    for h, off in hashes:
        h = h.upper() if isinstance(h, str) else int(h)
        mapper.setdefault(h, []).append(int(off))
    values = list(mapper.keys())
    dedup, results, batches = {}, [], 0
    for index in range(0, len(values), batch_size):
        batches += 1
        for h, sid, off in db.select_multiple(values[index:index + batch_size]):
            dedup[sid] = dedup.get(sid, 0) + 1
            for q in mapper[h]:
                results.append((sid, off - q))
        songs_arr = align_matches(results, dedup, len(hashes), db)
        if songs_arr[0]["hashes_matched_in_input"] / 2 > songs_arr[1]["hashes_matched_in_input"]:
            break
        songs_arr = []
    return results, dedup, songs_arr, batches


def recognize_apriori(hashes, db: DictDB, batch_size: int = 1000):
    """recognizer_apriori.py:602-609: the early result if the loop stopped, else align_matches over everything."""
    hashes = list(hashes)
    results, dedup, songs_arr, batches = return_matches_apriori(hashes, db, batch_size)
    if len(songs_arr) > 0:
        return songs_arr, batches, True
    return align_matches(results, dedup, len(hashes), db), batches, False
