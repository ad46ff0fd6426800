"""numpy twin of the on-device synthetic PCM generator (``shz_synth_pcm``) and of
the query-mixing rule -- CHECKER ONLY (see ``oracle/__init__.py``).

The generator is integer-only so the HIP kernel and this twin agree bit for bit:

    key(c)     = splitmix64(seed * 0xD6E8FEB86659FD93 + c)
    noise(c,n) = ((splitmix64(key + n) >> 32) * (2*noise_amp) >> 32) - noise_amp
    tone(c,n)  = (sum_k lut[phase_k >> 20] * tone_amp) >> 17          (6 partials)
       seg = n >> 14, m = n & 16383, r_k = splitmix64(~key + seg*8 + k)
       omega_k = OM_MIN + ((r_k >> 32) * (OM_MAX - OM_MIN) >> 32), phase_k = (r_k + m*omega_k) mod 2^32
    x(c,n)     = clamp(tone + noise, -32768, 32767)

``lut[i] = round(32767 sin(2 pi i / 4096))`` is built once on the host by
``sine_lut()`` and uploaded, so both sides use the same table bits.
tone_amp = 0 gives white noise uniform in [-noise_amp, noise_amp).

Query mixing restates recognizer_test.py:426-435 (get_noise_from_sound): the
noise is scaled so RMS_n = sqrt(RMS_s^2 / 10^(SNR/10)).
"""
from __future__ import annotations

import math

import numpy as np

U64 = np.uint64
GOLD = U64(0x9E3779B97F4A7C15)
MIX1 = U64(0xBF58476D1CE4E5B9)
MIX2 = U64(0x94D049BB133111EB)
SEEDMUL = U64(0xD6E8FEB86659FD93)
NOTE_SHIFT = 14
NPART = 6
OM_MIN = 10713046     # 2^32 * 110 Hz / 44100
OM_MAX = 428521855    # 2^32 * 4400 Hz / 44100


def splitmix64(x):
    with np.errstate(over="ignore"):
        x = np.asarray(x, U64) + GOLD
        z = (x ^ (x >> U64(30))) * MIX1
        z = (z ^ (z >> U64(27))) * MIX2
        return z ^ (z >> U64(31))


def sine_lut() -> np.ndarray:
    i = np.arange(4096, dtype=np.float64)
    return np.round(32767.0 * np.sin(2.0 * np.pi * i / 4096.0)).astype(np.int16)


def clip_key(seed: int, clip) -> np.ndarray:
    with np.errstate(over="ignore"):
        return splitmix64(U64(seed) * SEEDMUL + np.asarray(clip, U64))


def synth_clip(seed: int, clip: int, n_samples: int, tone_amp: int = 0, noise_amp: int = 8000,
               start: int = 0) -> np.ndarray:
    """Samples [start, start+n_samples) of synthetic clip ``clip``."""
    key = clip_key(seed, clip)
    n = np.arange(start, start + n_samples, dtype=U64)
    with np.errstate(over="ignore"):
        acc = np.zeros(n_samples, np.int64)
        if noise_amp > 0:
            u = splitmix64(key + n) >> U64(32)
            acc += ((u * U64(2 * noise_amp)) >> U64(32)).astype(np.int64) - noise_amp
        if tone_amp > 0:
            lut = sine_lut().astype(np.int64)
            seg = n >> U64(NOTE_SHIFT)
            m = n & U64((1 << NOTE_SHIFT) - 1)
            s = np.zeros(n_samples, np.int64)
            for k in range(NPART):
                r = splitmix64(~key + seg * U64(8) + U64(k))
                om = U64(OM_MIN) + (((r >> U64(32)) * U64(OM_MAX - OM_MIN)) >> U64(32))
                ph = (r + m * om) & U64(0xFFFFFFFF)
                s += lut[(ph >> U64(20)).astype(np.int64)]
            acc += (s * tone_amp) >> 17
    return np.clip(acc, -32768, 32767).astype(np.int16)


def mix_query(signal: np.ndarray, noise: np.ndarray, snr_db: float) -> np.ndarray:
    """signal + noise scaled to the requested SNR (recognizer_test.py:426-435,557),
    re-quantised to int16 with rounding-to-nearest-even and clipping."""
    s = signal.astype(np.float64)
    nz = noise.astype(np.float64)
    rms_s = math.sqrt(np.mean(s ** 2))
    rms_n = math.sqrt(rms_s ** 2 / (pow(10, snr_db / 10)))
    rms_cur = math.sqrt(np.mean(nz ** 2))
    nz = nz * (rms_n / rms_cur) if rms_cur > 0 else nz
    return np.clip(np.rint(s + nz), -32768, 32767).astype(np.int16)


def lut_tone(n_samples: int, hz_start: float, hz_end: float | None = None, amp: int = 8000) -> np.ndarray:
    """Integer-only sinusoid / linear chirp from the sine table: phase accumulates omega(n) in 2^-32 turns,
    omega runs linearly from hz_start to hz_end.  x = (lut[phase >> 20] * amp) >> 15."""
    lut = sine_lut().astype(np.int64)
    om0 = int(round(2 ** 32 * hz_start / 44100.0))
    om1 = om0 if hz_end is None else int(round(2 ** 32 * hz_end / 44100.0))
    n = np.arange(n_samples, dtype=np.int64)
    om = om0 + ((om1 - om0) * n) // max(n_samples, 1)
    ph = np.cumsum(om) & 0xFFFFFFFF
    return ((lut[ph >> 20] * amp) >> 15).astype(np.int16)


def tie_inputs() -> dict:
    """Near-tie material for the peak predicate (`maximum_filter(A) == A` on dB values, __init__.py:143 after :241):
    stationary or impulsive signals whose window maxima are shared by cells that differ in the last bits only.
    Integer-only, so the PCM is the same wherever it is regenerated (digests pinned in tests/golden/tie_cases.npz)."""
    out = {}
    x = np.zeros(1323000, np.int16)
    x[::6161] = 20000
    out["click_train_30s"] = x
    out["sine_1k_10s"] = lut_tone(441000, 1000.0)
    out["two_tone_10s"] = (lut_tone(441000, 440.0, amp=6000).astype(np.int32)
                           + lut_tone(441000, 1320.0, amp=5000)).astype(np.int16)
    out["dc_12000_5s"] = np.full(220500, 12000, np.int16)
    out["chirp_200_4000_10s"] = lut_tone(441000, 200.0, 4000.0)
    out["tonal_noiseless_10s"] = synth_clip(1234, 21, 441000, 4000, 0)
    out["sparse_clicks_5s"] = np.zeros(220500, np.int16)
    out["sparse_clicks_5s"][[30000, 30001, 90000, 150017]] = [32767, -32768, 15000, 9000]
    return out
