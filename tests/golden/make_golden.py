#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Build-container only (needs /root/reference; never runs on the GPU box).  The
reference's extraction functions are imported unmodified from
``/root/reference/__init__.py`` (two empty placeholder modules satisfy its
unrelated ``pydub`` imports); its match functions
(``return_matches``/``find_matches``/``align_matches``) are compiled out of
``recognizer.py`` by AST (the module itself records from a microphone at import
time) against an in-memory table standing in for MySQL.  Only inputs and
outputs are written here -- no reference source text.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz|json
"""
from __future__ import annotations

import ast
import hashlib
import importlib.util
import json
import os
import sys
import types
import wave
from contextlib import contextmanager
from itertools import groupby
from time import time

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import synth  # noqa: E402
from oracle.cpu_ref import DictDB  # noqa: E402


def load_reference_extraction():
    for name, attrs in (("pydub", ("AudioSegment",)), ("pydub.utils", ("audioop",))):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, None)
        sys.modules.setdefault(name, m)
    import warnings
    warnings.simplefilter("ignore")
    spec = importlib.util.spec_from_file_location("shazam_reference_init", os.path.join(REF, "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Cur:
    def __init__(self, store):
        self.store, self.rows = store, []

    def execute(self, query, values=()):
        self.rows = list(self.store.select_multiple(list(values)))

    def __iter__(self):
        return iter(self.rows)


class StandInDB(DictDB):
    """MySQL stand-in with the duck-typed surface the reference's match code uses
    (recognizer.py:251-259, 314): cursor() ctx-mgr, execute(query, values), row
    iteration of (HEX-upper hash, sid, offset), get_song_by_id."""

    @contextmanager
    def cursor(self, **kw):
        yield _Cur(self)


def load_reference_match(db):
    tree = ast.parse(open(os.path.join(REF, "recognizer.py")).read())
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("return_matches", "find_matches", "align_matches"):
            keep.append(node)
        elif isinstance(node, ast.Assign) and all(isinstance(t, ast.Name) and t.id.isupper() for t in node.targets):
            src = ast.unparse(node.value)
            if "pyaudio" in src or "audio." in src:
                continue
            keep.append(node)
    ns = {"groupby": groupby, "time": time, "db": db}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "recognizer_defs", "exec"), ns)
    return ns


def sha256(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def probes(A, n=200, seed=7):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, A.shape[0], n)
    t = rng.integers(0, A.shape[1], n)
    # always include DC / Nyquist rows and the corners
    f[:4] = [0, A.shape[0] - 1, 0, A.shape[0] - 1]
    t[:4] = [0, 0, A.shape[1] - 1, A.shape[1] - 1]
    return f.astype(np.int32), t.astype(np.int32), A[f, t].astype(np.float64)


def run_case(ref, x, Fs):
    """Stage outputs of the reference for one channel."""
    import matplotlib.mlab as mlab
    P = mlab.specgram(x, NFFT=ref.DEFAULT_WINDOW_SIZE, Fs=Fs, window=mlab.window_hanning,
                      noverlap=int(ref.DEFAULT_WINDOW_SIZE * ref.DEFAULT_OVERLAP_RATIO))[0]
    A = 10 * np.log10(P, out=np.zeros_like(P), where=(P != 0))
    peaks = ref.get_2D_peaks(A, plot=False, amp_min=ref.DEFAULT_AMP_MIN)
    pf = np.array([p[0] for p in peaks], np.int32)
    pt = np.array([p[1] for p in peaks], np.int32)
    hashes = ref.fingerprint(x, Fs=Fs)  # the real entry point, end to end
    hh = np.array([h for h, _ in hashes], dtype="S20")
    ht = np.array([int(o) for _, o in hashes], np.int64)
    pr_f, pr_t, pr_v = probes(A)
    return dict(n_frames=np.int64(A.shape[1]), peaks_f=pf, peaks_t=pt, hash_hex=hh, hash_t1=ht,
                probe_f=pr_f, probe_t=pr_t, probe_db=pr_v, sum_db=np.float64(A.sum()),
                col0_db=A[:, 0].copy(), Fs=np.int64(Fs))


def edge_inputs():
    rng = np.random.default_rng(20261004)
    cases = {}
    cases["short_3000"] = rng.integers(-8000, 8000, 3000).astype(np.int16)
    cases["exact_4096"] = rng.integers(-8000, 8000, 4096).astype(np.int16)
    cases["ragged_6143"] = rng.integers(-8000, 8000, 4096 + 2047).astype(np.int16)
    cases["two_frames_6144"] = rng.integers(-8000, 8000, 6144).astype(np.int16)
    cases["silence_20000"] = np.zeros(20000, np.int16)
    n = 2048 * 40
    sq = np.where((np.arange(n) // 32) % 2 == 0, 32767, -32768).astype(np.int16)
    cases["square_p64"] = sq
    burst = synth.synth_clip(99, 0, 2048 * 31, tone_amp=6000, noise_amp=500)
    gap = np.zeros(2048 * 250, np.int16)
    cases["gap_250_frames"] = np.concatenate([burst, gap, synth.synth_clip(99, 1, 2048 * 31, 6000, 500)])
    cases["loud_fullscale"] = rng.integers(-32768, 32767, 2048 * 30, endpoint=True).astype(np.int16)
    cases["dc_offset"] = (rng.integers(-50, 50, 2048 * 24) + 12000).astype(np.int16)
    return cases


def tie_cases(ref):
    """(v) near-tie inputs: the reference's peak test is an equality on dB values, and 10*log10 maps several adjacent
    doubles of power to one dB value -- these inputs make that visible (stationary tones, clicks, DC)."""
    tie = {}
    for name, x in synth.tie_inputs().items():
        r = run_case(ref, x, 44100)
        r.pop("col0_db")
        r["pcm_sha256"] = np.array(sha256(x.tobytes()), dtype="S64")
        for k, v in r.items():
            tie[f"{name}_{k}"] = v
        print(name, int(r["n_frames"]), "frames", len(r["peaks_f"]), "peaks", len(r["hash_hex"]), "hashes")
    np.savez_compressed(os.path.join(HERE, "tie_cases.npz"), **tie)


VARIANTS = {"amp0": dict(amp_min=0), "amp25": dict(amp_min=25), "ampneg5": dict(amp_min=-5), "amp33p3": dict(amp_min=33.3),
            "fan2": dict(fan_value=2), "fan10": dict(fan_value=10), "fan1": dict(fan_value=1),
            "fs8000": dict(Fs=8000), "fs48000": dict(Fs=48000),
            # wratio: noverlap = int(4096 * wratio), hop = 4096 - noverlap (round 4): 1024, 3072, 4096 and the odd 411
            "wr075": dict(wratio=0.75), "wr025": dict(wratio=0.25), "wr0": dict(wratio=0.0), "wr08999": dict(wratio=0.8999),
            # wsize: the generic spectrogram (powers of two 64 .. 2048), alone and with wratio
            "ws2048": dict(wsize=2048), "ws1024": dict(wsize=1024), "ws512_wr075": dict(wsize=512, wratio=0.75),
            "ws256_wr0": dict(wsize=256, wratio=0.0), "ws64": dict(wsize=64), "ws2048_fs8000": dict(wsize=2048, Fs=8000)}


def param_variants(ref):
    """(iii-b) non-default parameters of fingerprint(): amp_min, fan_value, Fs, wratio"""
    var = {}
    xv = synth.synth_clip(1234, 7, 2048 * 90 + 5, 4000, 1500)
    var["pcm_params"] = np.array([1234, 7, 2048 * 90 + 5, 4000, 1500], np.int64)
    for tag, kw in VARIANTS.items():
        hs = ref.fingerprint(xv, **kw)
        var[f"{tag}_hash_hex"] = np.array([h for h, _ in hs], dtype="S20")
        var[f"{tag}_hash_t1"] = np.array([int(o) for _, o in hs], np.int64)
        print("variant", tag, len(hs), "hashes")
    # the reference's own spectrogram lines (__init__.py:232-241) for two of the window sizes, on a short piece: the dB array
    from matplotlib import mlab
    for tag, nfft, nov, n in (("ws1024", 1024, 512, 1024 * 9 + 77), ("ws256", 256, 0, 256 * 12 + 3), ("ws2048short", 2048, 1024, 1500)):
        a = mlab.specgram(xv[:n], NFFT=nfft, Fs=44100, window=mlab.window_hanning, noverlap=nov)[0]
        with np.errstate(divide="ignore"):
            a = 10 * np.log10(a, out=np.zeros_like(a), where=(a != 0))
        var[f"{tag}_db"] = a
        var[f"{tag}_db_args"] = np.array([nfft, nov, n], np.int64)
    np.savez_compressed(os.path.join(HERE, "param_variants.npz"), **var)


def psd_inputs():
    """name -> (pcm, Fs, wratio) of the spectrogram digests: the tie inputs, a white and a tonal 30 s clip, a clip shorter than a
    window, the variant clip at other overlaps and rates, three edge cases.  Regenerable from integers alone (oracle/synth,
    seeded numpy integers), so the GPU box needs no audio."""
    cases = {name: (x, 44100, 0.5) for name, x in synth.tie_inputs().items()}
    cases["white_30s"] = (synth.synth_clip(1234, 1, 1323000, 0, 8000), 44100, 0.5)
    cases["tonal_30s"] = (synth.synth_clip(1234, 2, 1323000, 4000, 1500), 44100, 0.5)
    xv = synth.synth_clip(1234, 7, 2048 * 90 + 5, 4000, 1500)
    cases["variant_wr075"] = (xv, 44100, 0.75)
    cases["variant_wr08999"] = (xv, 44100, 0.8999)
    cases["variant_wr0"] = (xv, 44100, 0.0)
    cases["variant_fs8000"] = (xv, 8000, 0.5)
    cases["variant_fs48000"] = (xv, 48000, 0.5)
    cases["short_1500"] = (xv[:1500].copy(), 44100, 0.5)
    for k in ("exact_4096", "silence_20000", "loud_fullscale"):
        cases["edge_" + k] = (edge_inputs()[k], 44100, 0.5)
    # other window sizes (a 4th element: NFFT): numpy's plans 8.8.8.4, 2.8.8.8, 8.8.8, 8.8.4, 2.8.8, 8.8
    for nfft, wr in ((2048, 0.5), (1024, 0.5), (512, 0.75), (256, 0.0), (128, 0.5), (64, 0.5)):
        cases[f"ws{nfft}"] = (xv[: 40 * nfft + 13].copy(), 44100, wr, nfft)
    cases["ws2048_short"] = (xv[:1500].copy(), 44100, 0.5, 2048)
    return cases


def psd_digests(ref):
    """(vi) the reference's own spectrogram, every bit of it: sha256 over mlab.specgram(...)[0] of the call in
    __init__.py:232-237 (float64 [2049][F], C order, exact zeros written as 1.0 -- what the device stages), plus a few
    values in hex.  The device's fp64 path follows numpy's arithmetic operation by operation and must hit these digests."""
    from matplotlib import mlab
    out = {"_note": "sha256 of np.where(P == 0, 1.0, P).tobytes(), P = mlab.specgram(x, NFFT=4096, Fs, window_hanning, "
                    "noverlap=int(4096 * wratio))[0] as computed on the host that made the other fixtures",
           "_numpy": np.__version__}
    for name, case in psd_inputs().items():
        x, fs, wr = case[:3]
        nfft = case[3] if len(case) > 3 else ref.DEFAULT_WINDOW_SIZE
        P = mlab.specgram(x, NFFT=nfft, Fs=fs, window=mlab.window_hanning, noverlap=int(nfft * wr))[0]
        P = np.ascontiguousarray(np.where(P == 0, 1.0, P), np.float64)
        rng = np.random.default_rng(7)
        pf, pt = rng.integers(0, P.shape[0], 6), rng.integers(0, P.shape[1], 6)
        out[name] = {"Fs": fs, "wratio": wr, "nfft": int(nfft), "samples": int(len(x)), "pcm_sha256": sha256(x.tobytes()), "shape": list(P.shape),
                     "sha256": sha256(P.tobytes()), "zeros": int((P == 1.0).sum()),
                     "probe": [[int(a), int(b), float(P[a, b]).hex()] for a, b in zip(pf, pt)]}
        print("psd", name, P.shape, out[name]["sha256"][:16])
    with open(os.path.join(HERE, "psd_digests.json"), "w") as f:
        json.dump(out, f, indent=1)


def main():
    ref = load_reference_extraction()
    if "--only-psd" in sys.argv:
        return psd_digests(ref)
    if "--only-ties" in sys.argv:
        return tie_cases(ref)
    if "--only-variants" in sys.argv:
        return param_variants(ref)
    meta = {"numpy": np.__version__}
    import matplotlib
    import scipy
    meta.update(scipy=scipy.__version__, matplotlib=matplotlib.__version__,
                reference_constants={k: getattr(ref, k) for k in (
                    "RATE", "DEFAULT_WINDOW_SIZE", "DEFAULT_OVERLAP_RATIO", "DEFAULT_FAN_VALUE", "DEFAULT_AMP_MIN",
                    "CONNECTIVITY_MASK", "PEAK_NEIGHBORHOOD_SIZE", "PEAK_SORT", "MIN_HASH_TIME_DELTA",
                    "MAX_HASH_TIME_DELTA", "FINGERPRINT_REDUCTION")})

    # (i) bundled WAV known-answer test -------------------------------------------------
    wpath = os.path.join(REF, "signal_with_noise.wav")
    with wave.open(wpath, "rb") as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2
        fs_wav = w.getframerate()
        pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16).copy()
    meta["wav"] = {"sha256_file": sha256(open(wpath, "rb").read()), "frames": int(len(pcm)), "Fs": fs_wav}
    out = {"pcm": pcm}
    for fs in (fs_wav, 44100):
        for k, v in run_case(ref, pcm, fs).items():
            out[f"fs{fs}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "wav_kat.npz"), **out)
    print("wav:", len(out[f"fs{fs_wav}_peaks_f"]), "peaks", len(out[f"fs{fs_wav}_hash_hex"]), "hashes")

    # (ii) seeded synthetic clips (PCM regenerated from oracle/synth.py; digest pinned) -----
    syn = {}
    specs = {"white_5s": (1234, 0, 220500, 0, 8000), "white_30s": (1234, 1, 1323000, 0, 8000),
             "tonal_5s": (1234, 2, 220500, 4000, 1500), "tonal_30s": (1234, 3, 1323000, 4000, 1500),
             "tonal_list_input_2s": (1234, 4, 88200, 4000, 1500)}
    for name, (seed, clip, n, ta, na) in specs.items():
        x = synth.synth_clip(seed, clip, n, ta, na)
        xin = [v for v in x] if "list_input" in name else x  # reference also accepts lists (recognizer.py:368)
        r = run_case(ref, xin, 44100)
        r.pop("col0_db")
        r["params"] = np.array([seed, clip, n, ta, na], np.int64)
        r["pcm_sha256"] = np.array(sha256(x.tobytes()), dtype="S64")
        for k, v in r.items():
            syn[f"{name}_{k}"] = v
        print(name, len(r["peaks_f"]), "peaks", len(r["hash_hex"]), "hashes")
    np.savez_compressed(os.path.join(HERE, "synth_clips.npz"), **syn)

    # (iii) edge cases -----------------------------------------------------------------------
    edge = {}
    for name, x in edge_inputs().items():
        r = run_case(ref, x, 44100)
        r.pop("col0_db")
        r["pcm"] = x
        for k, v in r.items():
            edge[f"{name}_{k}"] = v
        print(name, int(r["n_frames"]), "frames", len(r["peaks_f"]), "peaks", len(r["hash_hex"]), "hashes")
    np.savez_compressed(os.path.join(HERE, "edge_cases.npz"), **edge)

    param_variants(ref)

    tie_cases(ref)
    psd_digests(ref)

    # (iv) match / align goldens ---------------------------------------------------------------
    db = StandInDB()
    m = load_reference_match(db)
    songs = []
    n_song = 441000  # 10 s
    for s in range(20):
        if s == 7:   # exact duplicate of song 3 -> equal counts, tie -> smaller sid first
            x = synth.synth_clip(4321, 3, n_song, 4000, 1500)
        elif s == 11:  # self-repeating song -> two deltas with equal counts candidates
            half = synth.synth_clip(4321, 11, 2048 * 100, 4000, 1500)
            x = np.concatenate([half, half])
        else:
            x = synth.synth_clip(4321, s, n_song, 4000, 1500)
        fp = set(ref.fingerprint(x, Fs=44100))
        sid = db.insert_song(f"{s:06d}", hashlib.sha1(x.tobytes()).hexdigest().upper(), len(fp))
        db.insert_hashes(sid, fp)
        db.set_song_fingerprinted(sid)
        songs.append({"song": s, "sid": sid, "total_hashes": len(fp)})
    rng = np.random.default_rng(5)
    queries = []
    for q in range(50):
        s = int(rng.integers(0, 20))
        aligned = q % 3 == 0
        qlen = 220500
        if s == 11:
            start = int(rng.integers(0, 40)) * 2048
        else:
            start = int(rng.integers(0, (n_song - qlen) // 2048)) * 2048 if aligned else int(rng.integers(0, n_song - qlen))
        src = 3 if s == 7 else s
        if s == 11:
            half = synth.synth_clip(4321, 11, 2048 * 100, 4000, 1500)
            full = np.concatenate([half, half])
        else:
            full = synth.synth_clip(4321, src, n_song, 4000, 1500)
        sig = full[start:start + qlen]
        snr = [None, 10.0, 0.0][q % 3 if q % 2 else 0]
        if snr is not None:
            noise = synth.synth_clip(777, q, qlen, 0, 8000)
            sig = synth.mix_query(sig, noise, snr)
        hashes = set(ref.fingerprint(sig, Fs=44100))
        matches, dedup, _ = m["find_matches"](hashes)
        res = m["align_matches"](matches, dedup, len(hashes), topn=3)
        for r in res:
            for k, v in list(r.items()):
                if isinstance(v, bytes):
                    r[k] = v.decode()
                elif isinstance(v, (np.integer,)):
                    r[k] = int(v)
        queries.append({"q": q, "song": s, "start": start, "snr": snr, "n_hashes": len(hashes),
                        "n_matches": len(matches), "dedup": {str(k): int(v) for k, v in sorted(dedup.items())},
                        "results": res})
    # crafted hash lists: exact ties by construction
    db2 = StandInDB()
    m2 = load_reference_match(db2)
    H = [hashlib.sha1(b"%d|%d|%d" % (i, i + 3, i % 5)).hexdigest()[:20] for i in range(40)]
    K = [(i << 20) | ((i + 3) << 8) | (i % 5) for i in range(40)]
    rows = {1: [(H[i], 10 + i) for i in range(10)] + [(H[i], 50 + i) for i in range(10)],      # two deltas tie (10 and 50)
            2: [(H[i], 10 + i) for i in range(10)] + [(H[i], 10 + i) for i in range(5)],        # duplicates ignored
            3: [(H[i], 7 + i) for i in range(10)],                                               # ties with song 2 on count
            4: [(H[20 + i], 100) for i in range(6)] + [(H[0], 3), (H[0], 4), (H[0], 5)]}
    for sid_want, rr in rows.items():
        sid = db2.insert_song(f"c{sid_want}", "AB" * 20, len(set(rr)))
        db2.insert_hashes(sid, rr)
    qh = [(H[i], i) for i in range(10)] + [(H[0], 2), (H[1], 9)] + [(H[20 + i], 40 + i) for i in range(6)] + [(H[39], 1)]
    matches, dedup, _ = m2["find_matches"](set(qh))
    crafted = {"rows": {str(k): [[h, o] for h, o in v] for k, v in rows.items()},
               "keys": {H[i]: K[i] for i in range(40)},
               "query": [[h, o] for h, o in qh],
               "n_matches": len(matches), "matches_sorted": sorted([list(map(int, x)) for x in matches]),
               "dedup": {str(k): int(v) for k, v in sorted(dedup.items())}}
    for topn in (1, 2, 3, 10):
        res = m2["align_matches"](matches, dedup, len(set(qh)), topn=topn)
        for r in res:
            for k, v in list(r.items()):
                if isinstance(v, bytes):
                    r[k] = v.decode()
        crafted[f"results_top{topn}"] = res
    json.dump({"meta": meta, "songs": songs, "song_params": {"seed": 4321, "n": n_song, "tone_amp": 4000, "noise_amp": 1500},
               "queries": queries, "crafted": crafted}, open(os.path.join(HERE, "match_cases.json"), "w"), indent=1)
    json.dump(meta, open(os.path.join(HERE, "META.json"), "w"), indent=1)
    print("match cases:", len(queries), "correct top-1:",
          sum(1 for q in queries if q["results"] and q["results"][0]["song_id"] in ((q["song"] + 1,) if q["song"] != 7 else (4, 8))))


if __name__ == "__main__":
    main()
