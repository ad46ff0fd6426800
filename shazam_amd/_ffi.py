"""ctypes binding of libshz.so (include/shz.h).  numpy + ctypes only -- no torch, no CPU fallback:
if the HIP library is missing or a call fails this module raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SHZ_LIB") or os.path.join(_HERE, "libshz.so")  # SHZ_LIB: A/B builds of the same ABI

OK, E_INVALID, E_HIP, E_CAPACITY, E_NOMEM, E_UNSUPPORTED, E_RCCL, E_STATE = 0, -1, -2, -3, -4, -5, -6, -7
PCM_DEVICE, OUT_DEVICE, IN_DEVICE, STFT_POWER, MATCH_FULL_SORT, RESERVE_GATHER, RESERVE_WAIT = 1, 2, 4, 8, 16, 32, 64
NFFT, HOP, NBINS = 4096, 2048, 2049

u8p, u16p, u32p, i32p, u64p, i16p, f64p = (C.POINTER(t) for t in (
    C.c_uint8, C.c_uint16, C.c_uint32, C.c_int32, C.c_uint64, C.c_int16, C.c_double))
vp = C.c_void_p

# name -> (restype, argtypes); the loader test checks every symbol in include/shz.h is exported
SIGNATURES = {
    "shz_ctx_create": (C.c_int32, [C.c_int32, C.POINTER(vp)]),
    "shz_ctx_destroy": (C.c_int32, [vp]),
    "shz_last_error": (C.c_char_p, [vp]),
    "shz_version": (C.c_char_p, []),
    "shz_device_info": (C.c_int32, [vp, C.c_char_p, C.c_uint64, u64p, i32p, i32p]),
    "shz_mem_info": (C.c_int32, [vp, u64p, u64p]),
    "shz_dev_alloc": (C.c_int32, [vp, C.c_uint64, C.POINTER(vp)]),
    "shz_dev_free": (C.c_int32, [vp, vp]),
    "shz_copy_h2d": (C.c_int32, [vp, vp, vp, C.c_uint64]),
    "shz_copy_d2h": (C.c_int32, [vp, vp, vp, C.c_uint64]),
    "shz_sync": (C.c_int32, [vp]),
    "shz_host_alloc": (C.c_int32, [vp, C.c_uint64, C.POINTER(vp)]),
    "shz_host_free": (C.c_int32, [vp, vp]),
    "shz_set_workspace_limit": (C.c_int32, [vp, C.c_uint64]),
    "shz_release_workspace": (C.c_int32, [vp, u64p]),
    "shz_timer_start": (C.c_int32, [vp, C.c_int32]),
    "shz_timer_stop": (C.c_int32, [vp, C.c_int32, C.POINTER(C.c_float)]),
    "shz_set_profiling": (C.c_int32, [vp, C.c_int32]),
    "shz_get_kernel_ms": (C.c_int32, [vp, C.c_int32, C.POINTER(C.c_float), u32p]),
    "shz_synth_pcm": (C.c_int32, [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int32, C.c_int32, C.c_uint64, vp]),
    "shz_synth_corpus": (C.c_int32, [vp, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_uint64, vp]),
    "shz_sumsq_i16": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint64, u64p]),
    "shz_mix_i16": (C.c_int32, [vp, vp, vp, C.c_uint32, C.c_uint64, f64p, vp]),
    "shz_membw": (C.c_int32, [vp, C.c_int32, C.c_uint64, C.c_uint32, C.POINTER(C.c_float)]),
    "shz_sort_pairs": (C.c_int32, [vp, vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32]),
    "shz_sort_keys32": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, vp]),
    "shz_sort_keys32_seg": (C.c_int32, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "shz_frame_count": (C.c_uint32, [C.c_uint64]),
    "shz_frame_count_hop": (C.c_uint32, [C.c_uint64, C.c_uint32]),
    "shz_set_overlap": (C.c_int32, [vp, C.c_uint32]),
    "shz_numpy_tables": (C.c_int32, [C.c_uint32, vp, vp, vp]),
    "shz_set_numpy_window": (C.c_int32, [vp, vp, C.c_double]),
    "shz_set_numpy_product": (C.c_int32, [vp, C.c_int32]),
    "shz_stft_db_any": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint64, u64p]),
    "shz_stft_db": (C.c_int32, [vp, vp, u64p, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint64, u64p]),
    "shz_db_values": (C.c_int32, [vp, C.c_uint64, vp]),
    "shz_peaks": (C.c_int32, [vp, vp, u64p, C.c_uint32, C.c_uint32, C.c_double, C.c_uint32, vp, vp, u64p, C.c_uint64, u64p]),
    "shz_peaks_from_db": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_double, vp, vp, C.c_uint64, u64p]),
    "shz_pair_hash": (C.c_int32, [vp, vp, vp, u64p, C.c_uint32, C.c_uint32, vp, vp, u64p, C.c_uint64, u64p]),
    "shz_fingerprint_batch": (C.c_int32, [vp, vp, u64p, C.c_uint32, C.c_uint32, C.c_double, C.c_uint32, C.c_uint32,
                                          vp, vp, u64p, C.c_uint64, u64p]),
    "shz_set_stage_f64": (C.c_int32, [vp, C.c_int32]),
    "shz_upload_stats": (C.c_int32, [vp, u64p, u64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "shz_extract_stats": (C.c_int32, [vp, u64p, u64p, u64p, u64p, u64p, u64p]),
    "shz_sha1_prefix": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, vp]),
    "shz_sha1_invert": (C.c_int32, [vp, vp, C.c_uint64, vp]),
    "shz_table_create": (C.c_int32, [vp, C.POINTER(vp)]),
    "shz_table_destroy": (C.c_int32, [vp]),
    "shz_table_insert": (C.c_int32, [vp, vp, vp, vp, C.c_uint64, C.c_uint32]),
    "shz_table_insert_clips": (C.c_int32, [vp, vp, vp, u64p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "shz_table_finalize": (C.c_int32, [vp]),
    "shz_table_set_segment_rows": (C.c_int32, [vp, C.c_uint64]),
    "shz_table_reserve": (C.c_int32, [vp, C.c_uint64, C.c_uint64, C.c_uint32]),
    "shz_table_seal_run": (C.c_int32, [vp]),
    "shz_table_rows": (C.c_int32, [vp, u64p, u64p]),
    "shz_table_segments": (C.c_int32, [vp, vp]),
    "shz_table_delete_songs": (C.c_int32, [vp, vp, C.c_uint64, u64p]),
    "shz_table_clear": (C.c_int32, [vp]),
    "shz_table_export": (C.c_int32, [vp, vp, vp, vp, C.c_uint64, u64p]),
    "shz_table_lookup": (C.c_int32, [vp, vp, C.c_uint64, vp, vp, vp, C.c_uint64, u64p]),
    "shz_table_song_rows": (C.c_int32, [vp, C.c_uint32, u64p]),
    "shz_match_batch": (C.c_int32, [vp, vp, vp, vp, u64p, C.c_uint32, C.c_uint32, C.c_uint32,
                                    vp, vp, vp, vp, vp, vp, vp]),
    "shz_match_stats": (C.c_int32, [vp, u64p, u64p, u64p]),
    "shz_set_debug": (C.c_int32, [vp, C.c_uint32]),
    "shz_match_vt_redo": (C.c_int32, [vp, u64p]),
    "shz_match_spec_stats": (C.c_int32, [vp, u64p, u64p]),
    "shz_comm_unique_id": (C.c_int32, [vp]),
    "shz_comm_create": (C.c_int32, [vp, vp, C.c_int32, C.c_int32, C.POINTER(vp)]),
    "shz_comm_create_local": (C.c_int32, [vp, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(vp)]),
    "shz_comm_destroy": (C.c_int32, [vp]),
    "shz_table_allgather": (C.c_int32, [vp, vp, u64p]),
    "shz_table_exchange_run": (C.c_int32, [vp, vp]),
    "shz_table_exchange_stats": (C.c_int32, [vp, u64p, u64p, C.POINTER(C.c_double), u32p]),
    "shz_table_set_run_rows": (C.c_int32, [vp, C.c_uint64]),
    "shz_table_finalize_runs": (C.c_int32, [vp, u64p, C.c_uint32]),
    "shz_table_build_stats": (C.c_int32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double)]),
    "shz_table_phase_stats": (C.c_int32, [vp, C.POINTER(C.c_double), C.c_uint32, u32p, C.c_int32]),
    "shz_table_phase_name": (C.c_char_p, [C.c_uint32]),
    "shz_comm_barrier": (C.c_int32, [vp]),
    "shz_comm_warmup": (C.c_int32, [vp]),
    "shz_shard_of_keys": (C.c_int32, [vp, C.c_uint64, C.c_uint32, vp]),
    "shz_table_keep_shard": (C.c_int32, [vp, C.c_uint32, C.c_uint32]),
    "shz_table_shard_exchange": (C.c_int32, [vp, vp, u64p]),
    "shz_table_stage_from": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32]),
    "shz_table_clear_staged": (C.c_int32, [vp]),
    "shz_table_maxima": (C.c_int32, [vp, u32p, u32p]),
    "shz_match_pairs": (C.c_int32, [vp, vp, vp, vp, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_uint32, vp, C.c_uint64, u64p, vp, vp]),
    "shz_pairs_allgather": (C.c_int32, [vp, C.c_uint64, vp, vp, C.c_uint64, u64p]),
    "shz_pairs_vote": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                   vp, vp, vp, vp, vp]),
}


_NP_FUSED = None


def numpy_product_is_fused() -> bool:
    """Does this host's numpy form the real part of conj(z) * z as fma(re, re, im * im)?  (mlab's `np.conj(result) * result`;
    numpy's SIMD complex product uses FMA3 where the CPU has it.)  Probed once on values for which the two roundings differ;
    an inconclusive probe means fused, the form of the hosts that made the fixtures."""
    global _NP_FUSED
    if _NP_FUSED is None:
        import fractions
        rng = np.random.default_rng(12345)
        z = (rng.standard_normal(64) + 1j * rng.standard_normal(64)) * 1e3
        got = (np.conj(z) * z).real
        votes = []
        for v, g in zip(z, got):
            re, im = float(v.real), float(v.imag)
            plain = re * re + im * im
            exact = fractions.Fraction(re) * fractions.Fraction(re) + fractions.Fraction(im * im)   # what an FMA rounds
            fused = float(exact)                                                                   # (Fraction -> float rounds to nearest even)
            if plain != fused:
                votes.append(g == fused)
        _NP_FUSED = bool(sum(votes) * 2 >= len(votes)) if votes else True
    return _NP_FUSED


class ShzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libshz error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libshz.so (built in-tree by __graft_entry__.build() / make -C shazam_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `make -C shazam_amd/csrc` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def ptr(a):
    """void* of a numpy array (or pass through ints / None)."""
    if a is None:
        return None
    if isinstance(a, (int,)):
        return C.c_void_p(a)
    if isinstance(a, DevBuf):
        return C.c_void_p(a.ptr)
    return C.c_void_p(a.ctypes.data)


class DevBuf:
    """Caller-owned device allocation (shz_dev_alloc)."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = vp()
        ctx.check(lib().shz_dev_alloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def free(self):
        if self.ptr and self.ctx.h:
            self.ctx.check(lib().shz_dev_free(self.ctx.h, vp(self.ptr)))
        self.ptr = None

    def upload(self, arr: np.ndarray, offset_bytes: int = 0):
        arr = np.ascontiguousarray(arr)
        assert offset_bytes + arr.nbytes <= self.nbytes
        self.ctx.check(lib().shz_copy_h2d(self.ctx.h, vp(self.ptr + offset_bytes), ptr(arr), arr.nbytes))

    def download(self, dtype, count: int, offset_bytes: int = 0) -> np.ndarray:
        out = np.empty(count, dtype)
        assert offset_bytes + out.nbytes <= self.nbytes
        self.ctx.check(lib().shz_copy_d2h(self.ctx.h, ptr(out), vp(self.ptr + offset_bytes), out.nbytes))
        return out

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One (device, stream).  Not thread-safe; use one per thread/process."""

    def __init__(self, device_id: int = 0):
        self.h = None
        h = vp()
        rc = lib().shz_ctx_create(device_id, C.byref(h))
        if rc != OK:
            raise ShzError(rc, f"shz_ctx_create(device {device_id}) failed -- is a ROCm GPU visible?")
        self.h = h
        self.device_id = device_id
        # the fp64 path's window as THIS host's numpy forms it (mlab.window_hanning = np.hanning; the scaling by
        # (window ** 2).sum()): numpy's by construction, not by the agreement of two cosine routines
        w = np.hanning(4096)
        self.check(lib().shz_set_numpy_window(self.h, ptr(w), float((w ** 2).sum())))
        self.check(lib().shz_set_numpy_product(self.h, 1 if numpy_product_is_fused() else 0))

    def check(self, rc):
        if rc != OK:
            raise ShzError(rc, (lib().shz_last_error(self.h) or b"").decode(errors="replace"))

    def close(self):
        if self.h:
            lib().shz_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- info / memory -------------------------------------------------------------------
    def device_info(self):
        name = C.create_string_buffer(256)
        hbm, cus, clk = C.c_uint64(), C.c_int32(), C.c_int32()
        self.check(lib().shz_device_info(self.h, name, 256, C.byref(hbm), C.byref(cus), C.byref(clk)))
        return {"name": name.value.decode(), "hbm_bytes": hbm.value, "compute_units": cus.value, "clock_khz": clk.value}

    def set_debug(self, flags: int):
        """SHZ_DEBUG_* test switches (1: tiny hand-over list, 2: LDS probes give up after one round)."""
        self.check(lib().shz_set_debug(self.h, int(flags)))

    def vt_redo_count(self) -> int:
        n = C.c_uint64()
        self.check(lib().shz_match_vt_redo(self.h, C.byref(n)))
        return n.value

    def spec_stats(self):
        """(queued, used): single small queries whose vote kernels were queued ahead of the vote count / that took
        their results from them."""
        a, b = C.c_uint64(), C.c_uint64()
        self.check(lib().shz_match_spec_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def mem_info(self):
        """(free, total) bytes of device memory right now."""
        a, b = C.c_uint64(), C.c_uint64()
        self.check(lib().shz_mem_info(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def alloc(self, nbytes) -> DevBuf:
        return DevBuf(self, nbytes)

    def host_array(self, shape, dtype=np.int16) -> np.ndarray:
        """numpy array in PINNED host memory (shz_host_alloc): PCM decoded into it reaches the GPU by DMA at the link rate.
        The memory lives as long as the array (and its views' base) does."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape))
        p = vp()
        self.check(lib().shz_host_alloc(self.h, max(1, n * dt.itemsize), C.byref(p)))
        buf = (C.c_char * max(1, n * dt.itemsize)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
        addr = p.value
        import weakref
        weakref.finalize(buf, lambda: lib().shz_host_free(None, vp(addr)))   # (the array may outlive the context)
        return arr

    def sync(self):
        self.check(lib().shz_sync(self.h))

    def membw(self, mode: int = 0, nbytes: int = 4 << 30, iters: int = 5) -> float:
        """Measured HBM GB/s of a 16-B/lane stream: mode 0 copy (read + write), 1 read, 2 write."""
        g = C.c_float()
        self.check(lib().shz_membw(self.h, int(mode), int(nbytes), int(iters), C.byref(g)))
        return float(g.value)

    def sort_keys32(self, keys: np.ndarray, bit_lo: int = 0, bit_hi: int = 32, add: int = 0) -> np.ndarray:
        """Stable device radix sort of uint32 keys on bits [bit_lo, bit_hi); returns uint64 keys `key + add`."""
        k = np.ascontiguousarray(keys, np.uint32)
        out = np.empty(len(k), np.uint64)
        self.check(lib().shz_sort_keys32(self.h, ptr(k), len(k), int(bit_lo), int(bit_hi), int(add), ptr(out)))
        return out

    def sort_keys32_seg(self, keys: np.ndarray, seg_off, bit_lo: int = 0, bit_hi: int = 32) -> np.ndarray:
        """Segmented stable device radix sort of uint32 keys on bits [bit_lo, bit_hi): segment i = keys[seg_off[i] : seg_off[i + 1]]
        is ordered among itself (the vote passes: one segment per query)."""
        k = np.ascontiguousarray(keys, np.uint32)
        so = np.ascontiguousarray(seg_off, np.uint64)
        out = np.empty(len(k), np.uint32)
        self.check(lib().shz_sort_keys32_seg(self.h, ptr(k), so.ctypes.data_as(u64p), len(so) - 1, int(bit_lo), int(bit_hi), ptr(out)))
        return out

    def sort_pairs(self, keys: np.ndarray, vals=None, bit_lo: int = 0, bit_hi: int = 64):
        """Stable device radix sort of uint64 keys on bits [bit_lo, bit_hi), with an optional
        uint32 / uint64 payload; returns sorted copies."""
        k = np.ascontiguousarray(keys, np.uint64).copy()
        v, vb = None, 0
        if vals is not None:
            v = np.ascontiguousarray(vals).copy()
            vb = v.dtype.itemsize
            if vb not in (4, 8) or v.shape != k.shape:
                raise ValueError("payload must be 4 or 8 bytes per key")
        self.check(lib().shz_sort_pairs(self.h, k.ctypes.data, v.ctypes.data if v is not None else None, vb, k.size,
                                        int(bit_lo), int(bit_hi)))
        return (k, v) if v is not None else k

    def set_workspace_limit(self, nbytes):
        self.check(lib().shz_set_workspace_limit(self.h, int(nbytes)))

    def release_workspace(self) -> int:
        n = C.c_uint64()
        self.check(lib().shz_release_workspace(self.h, C.byref(n)))
        return n.value

    def timer_start(self, slot=0):
        self.check(lib().shz_timer_start(self.h, slot))

    def timer_stop(self, slot=0) -> float:
        ms = C.c_float()
        self.check(lib().shz_timer_stop(self.h, slot, C.byref(ms)))
        return ms.value

    def set_profiling(self, on: bool):
        self.check(lib().shz_set_profiling(self.h, 1 if on else 0))

    def kernel_ms(self):
        names = ["stft_psd", "peak_pick", "peak_expand", "pair_hash", "peak_verify"]
        out = {}
        for i, n in enumerate(names):
            ms, k = C.c_float(), C.c_uint32()
            self.check(lib().shz_get_kernel_ms(self.h, i, C.byref(ms), C.byref(k)))
            out[n] = (ms.value, k.value)
        return out

    # ---- synthetic PCM ---------------------------------------------------------------------
    def synth_pcm(self, seed, clip0, n_clips, n_samples, tone_amp=0, noise_amp=8000, start=0, out: DevBuf = None) -> DevBuf:
        if out is None:
            out = self.alloc(int(n_clips) * int(n_samples) * 2)
        done = 0
        while done < n_clips:  # the kernel takes at most 65535 clips per launch
            k = min(65535, n_clips - done)
            self.check(lib().shz_synth_pcm(self.h, seed, clip0 + done, k, n_samples, tone_amp, noise_amp, start,
                                           vp(out.ptr + done * n_samples * 2)))
            done += k
        return out

    def synth_corpus(self, kind, seed, clip0, n_clips, n_samples, amp=3000, bed=100, burst=1500, start=0, out: DevBuf = None) -> DevBuf:
        """Music-like tracks (kind 1) or traffic-like noise (kind 2) on the device (twins: oracle/synth.music_clip / traffic_noise)."""
        if out is None:
            out = self.alloc(int(n_clips) * int(n_samples) * 2)
        done = 0
        while done < n_clips:
            k = min(65535, n_clips - done)
            self.check(lib().shz_synth_corpus(self.h, int(kind), seed, clip0 + done, k, n_samples, amp, bed, burst, start,
                                              vp(out.ptr + done * n_samples * 2)))
            done += k
        return out

    def mix_snr(self, sig: "DevBuf", noise: "DevBuf", n_clips: int, n_samples: int, snr_db: float, out: "DevBuf" = None) -> "DevBuf":
        """Per clip: noise scaled to the requested SNR (recognizer_test.py:426-435) and added; int16 out."""
        import math
        ss, sn = np.zeros(n_clips, np.uint64), np.zeros(n_clips, np.uint64)
        self.check(lib().shz_sumsq_i16(self.h, ptr(sig), n_clips, n_samples, ss.ctypes.data_as(u64p)))
        self.check(lib().shz_sumsq_i16(self.h, ptr(noise), n_clips, n_samples, sn.ctypes.data_as(u64p)))
        scale = np.empty(n_clips, np.float64)
        for c in range(n_clips):   # the reference's formula, evaluated exactly like oracle/synth.mix_query
            rms_s = math.sqrt(float(ss[c]) / n_samples)
            rms_n = math.sqrt(rms_s ** 2 / (pow(10, snr_db / 10)))
            rms_cur = math.sqrt(float(sn[c]) / n_samples)
            scale[c] = (rms_n / rms_cur) if rms_cur > 0 else 1.0
        if out is None:
            out = self.alloc(n_clips * n_samples * 2)
        self.check(lib().shz_mix_i16(self.h, ptr(sig), ptr(noise), n_clips, n_samples, scale.ctypes.data_as(f64p), ptr(out)))
        return out

    # ---- extraction ---------------------------------------------------------------------------
    @staticmethod
    def _clip_off(clip_off, n_clips=None):
        co = np.ascontiguousarray(clip_off, np.uint64)
        assert co.ndim == 1 and len(co) >= 1
        return co, len(co) - 1

    def upload_stats(self) -> dict:
        """Host PCM that went through the chunked upload pipeline since the context was created."""
        c, b, cs, ws = C.c_uint64(), C.c_uint64(), C.c_double(), C.c_double()
        self.check(lib().shz_upload_stats(self.h, C.byref(c), C.byref(b), C.byref(cs), C.byref(ws)))
        return {"chunks": c.value, "bytes": b.value, "copy_s": cs.value, "wait_s": ws.value}

    def set_overlap(self, noverlap: int):
        """noverlap of mlab.specgram for every later extraction call of this context (default 2048 = int(4096 * 0.5))."""
        self.check(lib().shz_set_overlap(self.h, int(noverlap)))
        self.hop = NFFT - int(noverlap)

    def frames_of(self, n_samples: int) -> int:
        return int(lib().shz_frame_count_hop(int(n_samples), int(getattr(self, "hop", HOP))))

    def set_stage_f64(self, enabled: bool):
        """fp64 staging of the power spectrogram (exact ties decided in the peak kernel) instead of fp32 + verify."""
        self.check(lib().shz_set_stage_f64(self.h, 1 if enabled else 0))

    def extract_stats(self) -> dict:
        v = [C.c_uint64() for _ in range(6)]
        self.check(lib().shz_extract_stats(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("undecided", "decided_f64", "frames_recomputed", "f64_passes", "f64_clips", "f64_clip_frames"),
                        (int(x.value) for x in v)))

    def stft_db(self, pcm, clip_off, fs=44100, pcm_device=False, power=False):
        co, nc = self._clip_off(clip_off)
        frames = [self.frames_of(int(co[i + 1] - co[i])) for i in range(nc)]
        out = np.empty(sum(frames) * NBINS, np.float64)
        cnt = C.c_uint64()
        self.check(lib().shz_stft_db(self.h, ptr(pcm), co.ctypes.data_as(u64p), nc, fs,
                                     (PCM_DEVICE if pcm_device else 0) | (STFT_POWER if power else 0),
                                     ptr(out), out.size, C.byref(cnt)))
        res, pos = [], 0
        for f in frames:
            res.append(out[pos:pos + f * NBINS].reshape(NBINS, f))
            pos += f * NBINS
        return res

    def set_numpy_product(self, fused: bool):
        """How the host's numpy forms conj(z) * z (see numpy_product_is_fused); set by __init__ from a probe."""
        self.check(lib().shz_set_numpy_product(self.h, 1 if fused else 0))

    def stft_db_any(self, x, fs=44100, nfft=2048, noverlap=1024, power=False) -> np.ndarray:
        """dB spectrogram [nfft/2 + 1, frames] of one channel for a window size other than 4096 (generic kernel)."""
        x = np.ascontiguousarray(x, np.int16)
        n = len(x)
        hop = int(nfft) - int(noverlap)
        frames = 1 if n < nfft or hop <= 0 else (n - int(nfft)) // hop + 1
        out = np.empty((int(nfft) // 2 + 1) * max(frames, 1), np.float64)
        nf = C.c_uint64()
        self.check(lib().shz_stft_db_any(self.h, ptr(x), n, int(fs), int(nfft), int(noverlap), STFT_POWER if power else 0, ptr(out),
                                         out.size, C.byref(nf)))
        return out[:(int(nfft) // 2 + 1) * nf.value].reshape(int(nfft) // 2 + 1, nf.value)

    def peaks(self, pcm, clip_off, fs=44100, amp_min=10.0, pcm_device=False):
        co, nc = self._clip_off(clip_off)
        total_frames = sum(self.frames_of(int(co[i + 1] - co[i])) for i in range(nc))
        cap = max(1024, total_frames * 16)
        flags = PCM_DEVICE if pcm_device else 0
        while True:
            pf, pt = np.empty(cap, np.uint16), np.empty(cap, np.uint32)
            po, cnt = np.zeros(nc + 1, np.uint64), C.c_uint64()
            rc = lib().shz_peaks(self.h, ptr(pcm), co.ctypes.data_as(u64p), nc, fs, float(amp_min), flags, ptr(pf), ptr(pt),
                                 po.ctypes.data_as(u64p), cap, C.byref(cnt))
            if rc == E_CAPACITY:
                cap = int(cnt.value)
                continue
            self.check(rc)
            n = int(cnt.value)
            return pf[:n], pt[:n], po

    def peaks_from_db(self, arr2d, amp_min=10.0):
        a = np.ascontiguousarray(arr2d, np.float64)
        assert a.ndim == 2
        cap = max(1024, a.shape[1] * 16)
        while True:
            of, ot, cnt = np.empty(cap, np.uint32), np.empty(cap, np.uint32), C.c_uint64()
            rc = lib().shz_peaks_from_db(self.h, ptr(a), a.shape[0], a.shape[1], float(amp_min), ptr(of), ptr(ot), cap, C.byref(cnt))
            if rc == E_CAPACITY:
                cap = int(cnt.value)
                continue
            self.check(rc)
            n = int(cnt.value)
            return of[:n], ot[:n]

    def pair_hash(self, peak_f, peak_t, peak_off, fan_value=5):
        pf = np.ascontiguousarray(peak_f, np.uint16)
        pt = np.ascontiguousarray(peak_t, np.uint32)
        po = np.ascontiguousarray(peak_off, np.uint64)
        nc = len(po) - 1
        cap = max(16, len(pf) * max(fan_value - 1, 0))
        k, t1 = np.empty(cap, np.uint32), np.empty(cap, np.uint32)
        ho, cnt = np.zeros(nc + 1, np.uint64), C.c_uint64()
        self.check(lib().shz_pair_hash(self.h, ptr(pf), ptr(pt), po.ctypes.data_as(u64p), nc, fan_value, ptr(k), ptr(t1),
                                       ho.ctypes.data_as(u64p), cap, C.byref(cnt)))
        n = int(cnt.value)
        return k[:n], t1[:n], ho

    def fingerprint_batch(self, pcm, clip_off, fs=44100, amp_min=10.0, fan_value=5, pcm_device=False,
                          out_key: DevBuf = None, out_t1: DevBuf = None, cap=None):
        """(key32, t1, hash_off).  With out_key/out_t1 DevBufs the hashes stay on the device and the
        returned arrays are None; hash_off (host) and the count are always returned."""
        co, nc = self._clip_off(clip_off)
        flags = PCM_DEVICE if pcm_device else 0
        ho, cnt = np.zeros(nc + 1, np.uint64), C.c_uint64()
        if out_key is not None:
            cap = int(cap if cap is not None else out_key.nbytes // 4)
            self.check(lib().shz_fingerprint_batch(self.h, ptr(pcm), co.ctypes.data_as(u64p), nc, fs, float(amp_min), fan_value,
                                                   flags | OUT_DEVICE, ptr(out_key), ptr(out_t1), ho.ctypes.data_as(u64p), cap,
                                                   C.byref(cnt)))
            return None, None, ho, int(cnt.value)
        total_frames = sum(self.frames_of(int(co[i + 1] - co[i])) for i in range(nc))
        cap = max(1024, total_frames * 40)
        while True:
            k, t1 = np.empty(cap, np.uint32), np.empty(cap, np.uint32)
            rc = lib().shz_fingerprint_batch(self.h, ptr(pcm), co.ctypes.data_as(u64p), nc, fs, float(amp_min), fan_value, flags,
                                             ptr(k), ptr(t1), ho.ctypes.data_as(u64p), cap, C.byref(cnt))
            if rc == E_CAPACITY:
                cap = int(cnt.value)
                continue
            self.check(rc)
            n = int(cnt.value)
            return k[:n], t1[:n], ho, n

    def sha1_prefix(self, key32, device=False, n=None) -> np.ndarray:
        if not device:
            key32 = np.ascontiguousarray(key32, np.uint32)
            n = len(key32)
        out = np.empty((int(n), 10), np.uint8)
        self.check(lib().shz_sha1_prefix(self.h, ptr(key32), int(n), IN_DEVICE if device else 0, ptr(out)))
        return out


def _sha1_invert(self, digests10: np.ndarray) -> np.ndarray:
    """key32 of each 10-byte digest (0xFFFFFFFF where none): brute force over the preimage space on the GPU."""
    d = np.ascontiguousarray(digests10, np.uint8).reshape(-1, 10)
    out = np.empty(len(d), np.uint32)
    self.check(lib().shz_sha1_invert(self.h, ptr(d), len(d), ptr(out)))
    return out


Context.sha1_invert = _sha1_invert


class Table:
    """HBM-resident fingerprints table: rows (key32, song_id, offset)."""

    def __init__(self, ctx: Context):
        self.ctx, self.h = ctx, None
        h = vp()
        ctx.check(lib().shz_table_create(ctx.h, C.byref(h)))
        self.h = h

    def close(self):
        if self.h and self.ctx.h:
            lib().shz_table_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def insert(self, key32, sid, off):
        k = np.ascontiguousarray(key32, np.uint32)
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(sid, np.uint32), k.shape))
        o = np.ascontiguousarray(off, np.uint32)
        assert k.shape == s.shape == o.shape
        self.ctx.check(lib().shz_table_insert(self.h, ptr(k), ptr(s), ptr(o), len(k), 0))

    def insert_clips(self, key32, t1, hash_off, sid0, device=False):
        ho = np.ascontiguousarray(hash_off, np.uint64)
        if not device:
            key32 = np.ascontiguousarray(key32, np.uint32)
            t1 = np.ascontiguousarray(t1, np.uint32)
        self.ctx.check(lib().shz_table_insert_clips(self.h, ptr(key32), ptr(t1), ho.ctypes.data_as(u64p), len(ho) - 1, sid0,
                                                    IN_DEVICE if device else 0))

    def set_segment_rows(self, rows: int):
        self.ctx.check(lib().shz_table_set_segment_rows(self.h, int(rows)))

    def finalize(self):
        self.ctx.check(lib().shz_table_finalize(self.h))

    def reserve(self, rows_hint: int, batch_rows_hint: int = 0, gather: bool = False, wait: bool = False):
        """Announce the size of a bulk build: the table's arenas are allocated once, beside the first batches (wait: before
        this call returns).  gather: the table holds its sealed runs until finalize() / allgather() merges them all at once
        (what a gathered build needs -- every row can still travel -- and what cuts segments by key range)."""
        self.ctx.check(lib().shz_table_reserve(self.h, int(rows_hint), int(batch_rows_hint),
                                               (RESERVE_GATHER if gather else 0) | (RESERVE_WAIT if wait else 0)))

    def seal_run(self):
        """Staged rows -> one sorted run (not yet visible to queries); finalize() merges the runs."""
        self.ctx.check(lib().shz_table_seal_run(self.h))

    def delete_songs(self, sids) -> int:
        """Remove every row of the listed song ids (ON DELETE CASCADE, mysql_database.py:57-58); returns rows removed."""
        a = np.ascontiguousarray(sids, np.uint32)
        n = C.c_uint64()
        self.ctx.check(lib().shz_table_delete_songs(self.h, ptr(a), len(a), C.byref(n)))
        return int(n.value)

    def clear(self):
        self.ctx.check(lib().shz_table_clear(self.h))

    def rows(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.ctx.check(lib().shz_table_rows(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def segments(self) -> int:
        n = C.c_uint32()
        self.ctx.check(lib().shz_table_segments(self.h, C.byref(n)))
        return n.value

    def export(self):
        n, _ = self.rows()
        k, s, o, cnt = np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.uint32), C.c_uint64()
        self.ctx.check(lib().shz_table_export(self.h, ptr(k), ptr(s), ptr(o), n, C.byref(cnt)))
        return k, s, o

    def lookup(self, keys):
        """Rows of the listed keys: (key32, sid, off) arrays, grouped in key-list order."""
        kk = np.ascontiguousarray(keys, np.uint32)
        cap = max(1024, 4 * len(kk))
        while True:
            k, s, o, cnt = np.empty(cap, np.uint32), np.empty(cap, np.uint32), np.empty(cap, np.uint32), C.c_uint64()
            rc = lib().shz_table_lookup(self.h, ptr(kk), len(kk), ptr(k), ptr(s), ptr(o), cap, C.byref(cnt))
            if rc == E_CAPACITY:
                cap = int(cnt.value)
                continue
            self.ctx.check(rc)
            n = int(cnt.value)
            return k[:n], s[:n], o[:n]

    def song_rows(self, sid) -> int:
        n = C.c_uint64()
        self.ctx.check(lib().shz_table_song_rows(self.h, int(sid), C.byref(n)))
        return n.value

    def finalize_runs(self, run_rows):
        """finalize() for staged rows that are consecutive blocks of run_rows[r] rows: every block is sorted on its own
        and the sorted runs are merged -- what allgather() does with the ranks' rows, without a communicator."""
        rr = np.ascontiguousarray(run_rows, np.uint64)
        self.ctx.check(lib().shz_table_finalize_runs(self.h, rr.ctypes.data_as(u64p), len(rr)))

    def build_stats(self) -> dict:
        v = [C.c_double() for _ in range(4)]
        self.ctx.check(lib().shz_table_build_stats(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("sort_s", "exchange_s", "merge_s", "segments_s"), (float(x.value) for x in v)))

    def phase_stats(self, reset=False) -> dict:
        """Host seconds per build phase since the last reset (shz_table_phase_stats)."""
        n = C.c_uint32()
        self.ctx.check(lib().shz_table_phase_stats(self.h, None, 0, C.byref(n), 0))
        v = (C.c_double * n.value)()
        self.ctx.check(lib().shz_table_phase_stats(self.h, v, n.value, C.byref(n), 1 if reset else 0))
        return {lib().shz_table_phase_name(i).decode(): float(v[i]) for i in range(n.value)}

    def set_run_rows(self, rows: int):
        """Rows one sealed run may hold (0: 2^32 - 4096); small values force many runs (tests)."""
        self.ctx.check(lib().shz_table_set_run_rows(self.h, int(rows)))

    def exchange_run(self, comm: "Comm"):
        """Collective: seal the staged rows and start this round's runs travelling to every peer on the communicator's own
        stream; returns with the transfers in flight (the next batch is fingerprinted beside them)."""
        self.ctx.check(lib().shz_table_exchange_run(self.h, comm.h))

    def exchange_stats(self) -> dict:
        r, b, w, k = C.c_uint64(), C.c_uint64(), C.c_double(), C.c_uint32()
        self.ctx.check(lib().shz_table_exchange_stats(self.h, C.byref(r), C.byref(b), C.byref(w), C.byref(k)))
        return {"rounds": r.value, "bytes_received": b.value, "wait_s": w.value, "runs_held": k.value}

    def allgather(self, comm: "Comm") -> int:
        """Collective: seal what is staged, exchange every run not yet sent, merge all runs into the node-global table;
        returns the payload bytes this rank received."""
        b = C.c_uint64()
        self.ctx.check(lib().shz_table_allgather(self.h, comm.h, C.byref(b)))
        return b.value

    def match(self, key32, q_off, query_off, topn=2, full_sort=False):
        """Batched return_matches + align_matches.  Returns dict of arrays (see shz.h).  full_sort: the vote as one
        sort of 8-byte votes + record chain (SHZ_MATCH_FULL_SORT), the form the faster vote paths are tested against."""
        k = np.ascontiguousarray(key32, np.uint32)
        o = np.ascontiguousarray(q_off, np.uint32)
        qo = np.ascontiguousarray(query_off, np.uint64)
        nq = len(qo) - 1
        res = {
            "sid": np.zeros((nq, topn), np.uint32), "delta": np.zeros((nq, topn), np.int32),
            "aligned": np.zeros((nq, topn), np.uint32), "dedup": np.zeros((nq, topn), np.uint32),
            "nres": np.zeros(nq, np.uint32), "nhash": np.zeros(nq, np.uint32), "npairs": np.zeros(nq, np.uint64)}
        self.ctx.check(lib().shz_match_batch(self.ctx.h, self.h, ptr(k), ptr(o), qo.ctypes.data_as(u64p), nq, topn,
                                             MATCH_FULL_SORT if full_sort else 0,
                                             ptr(res["sid"]), ptr(res["delta"]), ptr(res["aligned"]), ptr(res["dedup"]),
                                             ptr(res["nres"]), ptr(res["nhash"]), ptr(res["npairs"])))
        return res

    def match_stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.ctx.check(lib().shz_match_stats(self.ctx.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"rows_scanned": a.value, "pairs": b.value, "distinct_keys": c.value}


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    rc = lib().shz_comm_unique_id(buf)
    if rc != OK:
        raise ShzError(rc, "shz_comm_unique_id failed (librccl.so not loadable?)")
    return bytes(buf)


class Comm:
    """RCCL communicator, one rank per GPU."""

    def __init__(self, ctx: Context, unique_id: bytes, rank: int, nranks: int):
        self.ctx, self.h = ctx, None
        assert len(unique_id) == 128
        idb = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        h = vp()
        ctx.check(lib().shz_comm_create(ctx.h, idb, rank, nranks, C.byref(h)))
        self.h, self.rank, self.nranks = h, rank, nranks

    @classmethod
    def local(cls, ctx: Context, group_id: int, rank: int, nranks: int) -> "Comm":
        """Ranks = threads of this process (one Context each): rendezvous + device copies instead of RCCL."""
        self = cls.__new__(cls)
        self.ctx, self.h = ctx, None
        h = vp()
        ctx.check(lib().shz_comm_create_local(ctx.h, int(group_id), rank, nranks, C.byref(h)))
        self.h, self.rank, self.nranks = h, rank, nranks
        return self

    def warmup(self):
        """Collective: the connections between every pair of ranks exist afterwards (RCCL makes them on first use)."""
        self.ctx.check(lib().shz_comm_warmup(self.h))

    def barrier(self):
        self.ctx.check(lib().shz_comm_barrier(self.h))

    def close(self):
        if self.h and self.ctx.h:
            lib().shz_comm_destroy(self.h)
        self.h = None
