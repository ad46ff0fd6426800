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


# ---------------------------------------------------------------------------------------------------------------------
# Music-like corpus.  The tonal+noise clips above lose their peaks to a query's noise (the tracks'
# own peaks are mostly noise-born); the reference's 0.82 at SNR 0 was measured on real music under low-passed traffic
# noise (recognizer_test.py:39-40, 542-558; tests_csv/shazam_results_100records_5sec_0SNR.csv).  Stand-ins, integer-only
# and random-access like synth_clip (bit-identical twin on the device: shz_synth_corpus, kinds 1 and 2):
#
#   music(c, n)   = four voices of notes + percussive onsets + a quiet white bed.  Voice v changes note every
#                   L_v = 2^MUSIC_LEN_SHIFT[v] samples; note k of voice v: r = splitmix64(~key + (v << 40) + k); silent if
#                   r >> 60 == 0; fundamental = degree (r >> 8) % 48 of an equal-tempered scale from 110 Hz (MUSIC_OMEGA,
#                   integers) times MUSIC_OCTAVE[v], detuned by a factor (63488 + (r >> 24 & 4095)) / 65536; eight harmonics h with amplitudes amp * MUSIC_HARM[h-1] >> 8, start
#                   phases h * (r >> 16); linear decay over the note (L - m).  Voice 0's notes open with a decaying
#                   pseudo-random burst of 2,048 samples (amplitude `burst`): broadband, but with a texture of its own.
#                   ~9,000 hashes per 30 s (the reference's real music: 11-12 k per song).
#   traffic(c, n) = box-car sum of 16 consecutive white samples (a low-pass at ~1.3 kHz, -13 dB side lobes): most of its
#                   power sits below the bins music peaks live in, like street noise.
MUSIC_LEN_SHIFT = (14, 15, 13, 12)      # note lengths 0.37 s, 0.74 s, 0.19 s, 0.09 s
MUSIC_OCTAVE = (1, 1, 1, 2)
MUSIC_VOICES = len(MUSIC_LEN_SHIFT)
MUSIC_NHARM = 8
MUSIC_HARM = (256, 200, 150, 120, 100, 80, 64, 50)
MUSIC_OMEGA = tuple(int(round(2 ** 32 * 110.0 * 2 ** (s / 12.0) / 44100.0)) for s in range(48))
MUSIC_BURST_LEN = 2048
TRAFFIC_TAPS = 16


def music_clip(seed: int, clip: int, n_samples: int, amp: int = 3000, bed: int = 100, start: int = 0,
               burst: int = 1500) -> np.ndarray:
    """Samples [start, start + n_samples) of music-like clip ``clip``."""
    key = clip_key(seed, clip)
    n = np.arange(start, start + n_samples, dtype=U64)
    lut = sine_lut().astype(np.int64)
    om_tab = np.array(MUSIC_OMEGA, dtype=U64)
    with np.errstate(over="ignore"):
        acc = np.zeros(n_samples, np.int64)
        if bed > 0:
            u = splitmix64(key + n) >> U64(32)
            acc += ((u * U64(2 * bed)) >> U64(32)).astype(np.int64) - bed
        for v in range(MUSIC_VOICES):
            sh = MUSIC_LEN_SHIFT[v]
            k = n >> U64(sh)
            m = n & U64((1 << sh) - 1)
            r = splitmix64(~key + (U64(v) << U64(40)) + k)
            on = (r >> U64(60)) != U64(0)
            # scale degree, octave of the voice, and a detune of up to +-3 % drawn per note (singers and strings are not
            # keyboards: without it a corpus of 100,000 tracks shares a few hundred peak frequencies, and every hash of a
            # query matches tens of thousands of rows)
            om = (om_tab[((r >> U64(8)) % U64(48)).astype(np.int64)] * U64(MUSIC_OCTAVE[v]) *
                  (U64(63488) + ((r >> U64(24)) & U64(0xFFF)))) >> U64(16)
            env = (U64(1 << sh) - m).astype(np.int64)                       # L .. 1
            s = np.zeros(n_samples, np.int64)
            for h in range(1, MUSIC_NHARM + 1):
                ph = ((r >> U64(16)) * U64(h) + m * om * U64(h)) & U64(0xFFFFFFFF)
                s += lut[(ph >> U64(20)).astype(np.int64)] * MUSIC_HARM[h - 1]
            # lut (15 bits) x harm (8 bits) x env (sh bits) x amp: scaled back to ~amp at full envelope
            acc += np.where(on, (((s * env) >> (sh + 8)) * amp) >> 15, 0)
            if v == 0 and burst > 0:
                ub = splitmix64(r + m) >> U64(32)
                nb = ((ub * U64(2 * burst)) >> U64(32)).astype(np.int64) - burst
                e2 = np.maximum(MUSIC_BURST_LEN - m.astype(np.int64), 0)
                acc += np.where(on, (nb * e2) >> 11, 0)
    return np.clip(acc, -32768, 32767).astype(np.int16)


def traffic_noise(seed: int, clip: int, n_samples: int, amp: int = 2000, start: int = 0) -> np.ndarray:
    """Low-passed noise: sum of TRAFFIC_TAPS consecutive white samples, each uniform in [-amp, amp), >> 2."""
    key = clip_key(seed, clip)
    n = np.arange(start, start + n_samples, dtype=U64)
    with np.errstate(over="ignore"):
        acc = np.zeros(n_samples, np.int64)
        for i in range(TRAFFIC_TAPS):
            u = splitmix64(key + n + U64(i)) >> U64(32)
            acc += ((u * U64(2 * amp)) >> U64(32)).astype(np.int64) - amp
    return np.clip(acc >> 2, -32768, 32767).astype(np.int16)
