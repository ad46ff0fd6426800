"""Digital restatement of the reference's accuracy/timing harness (recognizer_test.py:516-614,
SURVEY 8f #3): random crop of every song, optional noise at a given SNR, recognise, and a CSV with
the reference's columns.  The speaker -> microphone loop of the reference is hardware and out of
scope; the crop + noise mix are done digitally on the host, recognition runs batched on the GPU.
"""
from __future__ import annotations

import csv
import math
from random import Random

import numpy as np

CSV_COLUMNS = ["file_name_played", "file_name_result", "song_start_time", "correct", "fingerprint_times", "query_time",
               "align_time", "total_time", "final_results"]   # recognizer_test.py:476-477


def get_noise_from_sound(signal: np.ndarray, noise: np.ndarray, SNR: float) -> np.ndarray:
    """Noise scaled so that RMS_n = sqrt(RMS_s^2 / 10^(SNR/10)) (recognizer_test.py:426-435)."""
    RMS_s = math.sqrt(np.mean(signal ** 2))
    RMS_n = math.sqrt(RMS_s ** 2 / (pow(10, SNR / 10)))
    RMS_n_current = math.sqrt(np.mean(noise ** 2))
    return noise * (RMS_n / RMS_n_current)


def mix(signal_i16: np.ndarray, noise_i16: np.ndarray, SNR: float) -> np.ndarray:
    """signal + scaled noise, back to int16 (round half to even, clip) -- the sf.write step at :557."""
    s = signal_i16.astype(np.float64)
    n = get_noise_from_sound(s, noise_i16.astype(np.float64), SNR)
    return np.clip(np.rint(s + n), -32768, 32767).astype(np.int16)


def run(db, songs, record_seconds: int = 5, add_noise: bool = False, snr: float = 0, noise: np.ndarray = None,
        fs: int = 44100, topn: int = 3, seed: int = 0, batch: int = 512):
    """songs: list of (name, int16 mono array).  Returns the list of row dicts (CSV_COLUMNS)."""
    import shazam_amd as S
    rnd = Random(seed)
    qlen = record_seconds * fs
    rows, queries, meta = [], [], []
    for name, pcm in songs:
        duration = len(pcm) / fs
        start_s = rnd.randrange(0, max(1, int(duration) - record_seconds))   # whole seconds, :538
        sig = pcm[start_s * fs: start_s * fs + qlen]
        if add_noise:
            n0 = rnd.randrange(0, len(noise) - len(sig))                      # random start of noise, :553
            sig = mix(sig, noise[n0:n0 + len(sig)], snr)
        queries.append(sig)
        meta.append((name, start_s))
    for b0 in range(0, len(queries), batch):
        res, tm = S.recognize_batch(queries[b0:b0 + batch], db, Fs=fs, topn=topn)
        nb = len(res)
        for i, final_results in enumerate(res):
            name, start_s = meta[b0 + i]
            got = final_results[0]["song_name"].decode() if final_results else "No results"
            ft, qt, at = tm["fingerprint_time"] / nb, tm["query_time"] / nb, tm["align_time"] / nb
            rows.append({"file_name_played": name, "file_name_result": got, "song_start_time": start_s,
                         "correct": 1 if got == name else 0, "fingerprint_times": ft, "query_time": qt, "align_time": at,
                         "total_time": ft + qt + at, "final_results": str(final_results) if final_results else "No results"})
    return rows


def accuracy(rows) -> float:
    return sum(r["correct"] for r in rows) / max(1, len(rows))


def write_csv(rows, path: str):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=CSV_COLUMNS)
        w.writeheader()
        for r in rows:
            w.writerow(r)


def write_reports(rows, csv_name: str):
    """CM_/CMSK_/CRSK_/ASSK_ companions of the results CSV (recognizer_test.py:489-513)."""
    import pandas as pd
    from sklearn.metrics import accuracy_score, classification_report, confusion_matrix
    y_true = [r["file_name_played"] for r in rows]
    y_pred = [r["file_name_result"] for r in rows]
    # CM_: the reference's own table (recognizer_test.py:491-499): played names on both axes with the play counts on the
    # diagonal; every miss sets its diagonal cell to 0 and puts a 1 in the column of the name that came back (a column
    # that appears on demand when that name was never played)
    names = sorted(set(y_true))
    cm = pd.DataFrame(0, index=pd.Index(names, name="Actual"), columns=pd.Index(names, name="Actual"), dtype=object)
    for n in y_true:
        cm.at[n, n] += 1
    for t, p_ in zip(y_true, y_pred):
        if t != p_:
            cm.at[t, t] = 0
            cm.at[t, p_] = 1
    cm.to_csv("CM_" + csv_name)
    pd.DataFrame(confusion_matrix(y_true, y_pred)).to_csv("CMSK_" + csv_name)
    pd.DataFrame(classification_report(y_true, y_pred, output_dict=True, zero_division=0)).transpose().to_csv("CRSK_" + csv_name)
    pd.DataFrame([accuracy_score(y_true, y_pred)]).to_csv("ASSK_" + csv_name)
