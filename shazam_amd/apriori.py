"""Early-exit ("apriori") recognition: the variant of the match in the reference's recognizer_apriori.py.

Reference (recognizer_apriori.py:237-310): the distinct hashes of the query are looked up in batches of 1000, in the
iteration order of the query; after every batch everything matched so far is aligned, and the loop stops as soon as
the leader's `hashes_matched_in_input` is more than twice the runner-up's.  :602-609: the early result stands, else
`align_matches` over all matches.

Here the table lives in HBM and a lookup costs microseconds, so nothing is saved by stopping; what this module keeps
is the RESULT of the rule.  Every prefix of the batch sequence becomes one query of ONE batched device match
(shz_match_batch: prefix j = the hashes whose distinct hash is among the first (j+1) x batch_size), and the host walks
the prefixes' top-2 in order.  Same sequence of (hash, offset) tuples in, same dicts out as the reference functions.
The reference iterates a Python set, whose order changes from process to process; callers that want its exact
behaviour pass the sequence explicitly (`find_matches_apriori`), `recognize_apriori` uses generation order."""
from time import time

import numpy as np

from . import RATE, TOPN, _result_dicts, fingerprint_batch, keys_of_hexes

BATCH_SIZE = 1000   # recognizer_apriori.py:237


def _as_key_offset(hashes, ctx):
    if isinstance(hashes, tuple) and len(hashes) == 2 and isinstance(hashes[0], np.ndarray):
        k, o = hashes
        return np.ascontiguousarray(k, np.uint32), np.ascontiguousarray(o, np.uint32)
    hashes = list(hashes)
    if not hashes:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32)
    first = hashes[0][0]
    if isinstance(first, (str, bytes)):
        k = keys_of_hexes([h for h, _ in hashes], ctx)
    else:
        k = np.array([int(h) for h, _ in hashes], np.uint32)
    return k, np.array([int(o) for _, o in hashes], np.uint32)


def find_matches_apriori(hashes, db, batch_size: int = BATCH_SIZE, strict: bool = True):
    """recognizer_apriori.py:237-325.  hashes: sequence of DISTINCT (hash, offset) tuples (hash = 20-hex string or
    packed key32) or a pair of arrays (key32, offset); the order of the sequence is the order of the batches.
    Returns (final_results, early_exit, batches_looked_up, n_matches, query_time): final_results are the reference's
    result dicts (topn = TOPN = 2, `input_total_hashes` = len(hashes)) of the prefix the loop stopped at, or of the
    whole query if it never stopped.
    strict=True keeps the reference's failures: IndexError while fewer than two songs have matched when the rule is
    evaluated (`songs_arr[1]`, :303), UnboundLocalError for an empty query (:310).  strict=False treats a missing
    runner-up as zero matches and an empty query as no result."""
    t0 = time()
    k, o = _as_key_offset(hashes, db.ctx)
    n = len(k)
    if n == 0:
        if strict:
            raise UnboundLocalError("local variable 'songs_arr' referenced before assignment")
        return [], False, 0, 0, time() - t0
    tup = (k.astype(np.uint64) << np.uint64(32)) | o.astype(np.uint64)
    if len(np.unique(tup)) != n:
        raise ValueError("find_matches_apriori: the (hash, offset) tuples must be distinct (the reference passes a set)")
    # distinct hashes in first-occurrence order (the keys of the reference's `mapper`, :237-243) -> batch of each tuple
    uniq, first_idx, inv = np.unique(k, return_index=True, return_inverse=True)
    rank = np.empty(len(uniq), np.int64)
    rank[np.argsort(first_idx, kind="stable")] = np.arange(len(uniq))
    batch = rank[inv] // int(batch_size)
    nb = int(batch.max()) + 1
    # prefix j = tuples of batches 0..j, one device query each
    sel = [np.nonzero(batch <= j)[0] for j in range(nb)]
    qoff = np.zeros(nb + 1, np.uint64)
    qoff[1:] = np.cumsum([len(s) for s in sel])
    cat = np.concatenate(sel)
    res = db.match(k[cat], o[cat], qoff, TOPN)
    stop, early = nb - 1, False
    for j in range(nb):
        nres = int(res["nres"][j])
        if strict and nres < 2:
            raise IndexError("list index out of range")   # songs_arr[0] / songs_arr[1] of the reference
        lead = int(res["dedup"][j, 0]) if nres > 0 else 0
        second = int(res["dedup"][j, 1]) if nres > 1 else 0
        if nres > 0 and lead / 2 > second:
            stop, early = j, True
            break
    final = _result_dicts(db, res, stop, n)
    return final, early, stop + 1, int(res["npairs"][stop]), time() - t0


def recognize_apriori(channels_or_samples, db, Fs: int = RATE, batch_size: int = BATCH_SIZE, strict: bool = True):
    """recognizer_apriori.py:586-611: fingerprint every channel, union of the hashes (here: generation order, channel
    after channel, repeats dropped), early-exit match.  Returns (final_results, fingerprint_time, query_time,
    align_time) -- align_time is 0 like the reference's when the loop stopped early, and part of query_time otherwise
    (one device call does both)."""
    x = channels_or_samples
    chans = [np.asarray(x)] if (not isinstance(x, (list, tuple)) or (len(x) and np.isscalar(x[0]))) else list(x)
    t0 = time()
    k, t1, _ = fingerprint_batch(chans, Fs, ctx=db.ctx)
    tup = (k.astype(np.uint64) << np.uint64(32)) | t1.astype(np.uint64)
    _, keep = np.unique(tup, return_index=True)
    keep.sort()
    fingerprint_time = time() - t0
    final, _, _, _, query_time = find_matches_apriori((k[keep], t1[keep]), db, batch_size, strict)
    return final, fingerprint_time, query_time, 0
