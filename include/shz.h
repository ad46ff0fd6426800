/*
 * shz.h -- C ABI of libshz.so, the MI355X (gfx950) fingerprint / match hot path.
 *
 * The reference (CarlosArturoMe/shazam) is pure Python and has no FFI of its own
 * (SURVEY.md 8b); this header is the boundary a maintainer binds with ctypes to put
 * the HIP path behind the reference's own functions.  Each entry point names the
 * reference interface it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - every function returns an int32 status: SHZ_OK or a negative SHZ_E_* code;
 *     shz_last_error(ctx) gives a message owned by the ctx (valid until the next call).
 *   - shz_ctx is one (device, stream); NOT thread-safe per ctx, independent ctxs are.
 *   - the caller allocates every output buffer and passes its capacity; on overflow the
 *     call returns SHZ_E_CAPACITY with the required count in *count (two-call idiom).
 *   - pointers are host pointers unless the matching SHZ_*_DEVICE flag is set, in which
 *     case they are device pointers obtained from shz_dev_alloc on the same ctx.
 *   - the library never keeps a caller pointer past return; no exception crosses the ABI.
 */
#ifndef SHZ_H
#define SHZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHZ_OK 0
#define SHZ_E_INVALID (-1)   /* bad argument                                   */
#define SHZ_E_HIP (-2)       /* a HIP runtime call failed                      */
#define SHZ_E_CAPACITY (-3)  /* output buffer too small; *count = required     */
#define SHZ_E_NOMEM (-4)     /* device or host allocation failed               */
#define SHZ_E_UNSUPPORTED (-5)/* parameter combination the HIP path does not implement */
#define SHZ_E_RCCL (-6)      /* RCCL missing or a collective failed            */
#define SHZ_E_STATE (-7)     /* object used in the wrong state                 */

#define SHZ_PCM_DEVICE 1u    /* pcm pointer is device memory                   */
#define SHZ_OUT_DEVICE 2u    /* output pointers are device memory              */
#define SHZ_IN_DEVICE 4u     /* generic: input arrays are device memory        */
#define SHZ_STFT_POWER 8u    /* shz_stft_db: write the PSD itself, not 10*log10 */
#define SHZ_RESERVE_GATHER 32u /* shz_table_reserve: size the run arena for an all-gathered build (every rank's rows) */
#define SHZ_RESERVE_WAIT 64u   /* shz_table_reserve: return when the allocations exist (setup outside a timed region) */
#define SHZ_MATCH_FULL_SORT 16u /* shz_match_batch: 8-byte votes, full radix sort and record chain (the reference form of
                                  the vote: what the 4-byte votes and the vote tiles must reproduce) */

/* constants of the algorithm: __init__.py:41-51 == recognizer.py:21-38 */
#define SHZ_NFFT 4096
#define SHZ_HOP 2048
#define SHZ_NBINS 2049
#define SHZ_PEAK_RADIUS 10
#define SHZ_MAX_DT 200
#define SHZ_DEFAULT_FS 44100

typedef struct shz_ctx shz_ctx;
typedef struct shz_table shz_table;
typedef struct shz_comm shz_comm;

/* ---- context, memory, timing ------------------------------------------------------ */
int32_t shz_ctx_create(int32_t device_id, shz_ctx** out);
int32_t shz_ctx_destroy(shz_ctx* ctx);
const char* shz_last_error(shz_ctx* ctx);
const char* shz_version(void);
/* name: >=128 bytes; any out pointer may be NULL */
int32_t shz_device_info(shz_ctx* ctx, char* name, uint64_t name_cap, uint64_t* hbm_bytes,
                        int32_t* compute_units, int32_t* clock_khz);
/* device memory free / total right now (hipMemGetInfo): sizes reservations, shows leaks in tests */
int32_t shz_mem_info(shz_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);
int32_t shz_dev_alloc(shz_ctx* ctx, uint64_t bytes, void** dptr);
int32_t shz_dev_free(shz_ctx* ctx, void* dptr);
int32_t shz_copy_h2d(shz_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);
int32_t shz_copy_d2h(shz_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);
int32_t shz_sync(shz_ctx* ctx);
/* cap on the internal scratch arena (dB spectrogram, masks, sort buffers); batches are
 * split into sub-batches that fit.  0 = default (1/4 of HBM). */
int32_t shz_set_workspace_limit(shz_ctx* ctx, uint64_t bytes);
/* free the scratch arena (it regrows on demand); *freed_bytes may be NULL */
int32_t shz_release_workspace(shz_ctx* ctx, uint64_t* freed_bytes);
/* hipEvent timers on the ctx stream (replaces the time() deltas at recognizer.py:214-220,
 * 282-284, 388-390).  slot in [0,16). */
int32_t shz_timer_start(shz_ctx* ctx, int32_t slot);
int32_t shz_timer_stop(shz_ctx* ctx, int32_t slot, float* elapsed_ms);
/* per-kernel accumulated device time of the last profiled call (see shz_set_profiling).
 * which: 0 stft_psd, 1 peak_pick, 2 peak_expand(+scan), 3 pair_hash(+scan) */
int32_t shz_set_profiling(shz_ctx* ctx, int32_t enabled);
int32_t shz_get_kernel_ms(shz_ctx* ctx, int32_t which, float* total_ms, uint32_t* launches);

/* ---- synthetic PCM (bench / tests input; numpy twin: oracle/synth.py) ------------------ */
/* clips [clip0, clip0+n_clips) x n_samples int16, clip-major, written to DEVICE memory.
 * tone_amp = 0 -> white noise uniform in [-noise_amp, noise_amp). */
int32_t shz_synth_pcm(shz_ctx* ctx, uint64_t seed, uint64_t clip0, uint32_t n_clips, uint64_t n_samples,
                      int32_t tone_amp, int32_t noise_amp, uint64_t start_sample, int16_t* dev_out);
/* Music-like tracks (kind SHZ_CORPUS_MUSIC: four voices of decaying harmonic notes, percussive onsets, a quiet noise bed:
 * amp, bed, burst) and traffic-like low-passed noise (SHZ_CORPUS_TRAFFIC: amp) -- stand-ins for what the reference's
 * accuracy figures were measured on (real music under street noise, recognizer_test.py:39-40, 542-558), integer-only with
 * numpy twins (oracle/synth.py: music_clip, traffic_noise).  Same addressing as shz_synth_pcm. */
#define SHZ_CORPUS_MUSIC 1u
#define SHZ_CORPUS_TRAFFIC 2u
int32_t shz_synth_corpus(shz_ctx* ctx, uint32_t kind, uint64_t seed, uint64_t clip0, uint32_t n_clips, uint64_t n_samples,
                         int32_t amp, int32_t bed, int32_t burst, uint64_t start_sample, int16_t* dev_out);

/* HBM bandwidth probe (the measured ceiling beside the vendor peak): mode 0 copy (bytes read +
 * bytes written are counted), 1 read only, 2 write only; two scratch buffers of `bytes` each are allocated and
 * freed inside; gb_per_s = bytes moved / hipEvent time over `iters` launches (16 B per lane, grid-stride).
 * Host link probe (what a host-fed fingerprint call can reach at best): mode 3 pinned host -> device, 4 pageable host ->
 * device, 5 device -> pinned host; `iters` copies of `bytes`, host clock around them. */
int32_t shz_membw(shz_ctx* ctx, int32_t mode, uint64_t bytes, uint32_t iters, float* gb_per_s);

/* The device radix sort the table build and the vote use (tests / tools): stable sort of n 64-bit keys on bits
 * [bit_lo, bit_hi) (other bits are not compared), carrying a payload of val_bytes = 0, 4 or 8
 * bytes per key.  keys / vals are HOST arrays, sorted in place.  n < 2^32. */
int32_t shz_sort_pairs(shz_ctx* ctx, uint64_t* keys, void* vals, uint32_t val_bytes, uint64_t n, uint32_t bit_lo,
                       uint32_t bit_hi);
/* The 4-byte form the vote uses when a pass's query index, song id and offset delta fit 31 bits (tests / tools): stable
 * sort of n 32-bit keys on bits [bit_lo, bit_hi), written as 64-bit keys `key + add` (the last pass widens).
 * keys / out64 are HOST arrays.  n < 2^32. */
int32_t shz_sort_keys32(shz_ctx* ctx, const uint32_t* keys, uint64_t n, uint32_t bit_lo, uint32_t bit_hi, uint64_t add,
                        uint64_t* out64);

/* The SEGMENTED form behind the vote passes (tests / tools): segment i = keys [seg_off[i], seg_off[i+1]), n_segs <= 128;
 * every segment is ordered (stably) among its own keys on bits [bit_lo, bit_hi), no key leaves its segment.
 * keys / seg_off / out are HOST arrays; seg_off[0] = 0, seg_off[n_segs] = n < 2^32. */
int32_t shz_sort_keys32_seg(shz_ctx* ctx, const uint32_t* keys, const uint64_t* seg_off, uint32_t n_segs, uint32_t bit_lo,
                            uint32_t bit_hi, uint32_t* out);

/* Query preparation (bench / tests): exact sum of squares of each clip (device PCM, clip-major, equal
 * lengths) to HOST, and out = clip(rint(sig + scale[c] * noise)) on the device: the digital form of
 * get_noise_from_sound + sf.write (recognizer_test.py:426-435, 557); twin: oracle/synth.mix_query. */
int32_t shz_sumsq_i16(shz_ctx* ctx, const int16_t* dev_pcm, uint32_t n_clips, uint64_t n_samples, uint64_t* out_host);
int32_t shz_mix_i16(shz_ctx* ctx, const int16_t* dev_sig, const int16_t* dev_noise, uint32_t n_clips,
                    uint64_t n_samples, const double* scale_host, int16_t* dev_out);

/* Pinned host memory for PCM (and any other array handed to the library): the reference's callers hold their samples in
 * host arrays (read(), __init__.py:70-113; recognizer.py:357-382); a decoder that writes them into a buffer from here lets
 * shz_fingerprint_batch feed the GPU by DMA at the link rate, chunk by chunk beside the kernels of the chunk before. */
int32_t shz_host_alloc(shz_ctx* ctx, uint64_t bytes, void** out);
int32_t shz_host_free(shz_ctx* ctx, void* p);   /* ctx may be NULL */

/* ---- extraction --------------------------------------------------------------------- */
/* Frames mlab produces for n_samples (mlab.specgram via __init__.py:232-237). */
uint32_t shz_frame_count(uint64_t n_samples);

/* Stage parity/debug: dB spectrogram of a batch, replaces
 *   mlab.specgram(x, NFFT=4096, Fs, window_hanning, noverlap=2048)[0] + 10*log10
 *   (__init__.py:232-241).  clip_off: n_clips+1 sample offsets into pcm (host memory).
 * out_db (HOST): per clip a float64 [2049, F_c] freq-major block (the reference's layout),
 * blocks concatenated in clip order; cap_doubles = capacity of out_db in doubles.
 * The logarithm is the correctly rounded one (csrc/shz_log10.h).  With SHZ_STFT_POWER the block holds what
 * specgram returns (the PSD before the log, __init__.py:232-237) in the staged form peak picking reads:
 * exact zeros read as 1.0, the power whose dB value is the 0.0 the reference assigns them (:241).
 * The power is computed with the reference's arithmetic operation by operation (numpy 2.x: pocketfft's radix-8 passes on the
 * complex frame, its complex product on a host with FMA3, mlab's scaling: csrc/shz_extract.hip np_fft4096): every value is
 * the one specgram returns, bit for bit (tests/golden/psd_digests.json).  The same arithmetic decides every tie of the
 * batch path (shz_peaks / shz_fingerprint_batch). */
int32_t shz_stft_db(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t n_clips,
                    uint32_t fs, uint32_t flags, double* out_db, uint64_t cap_doubles, uint64_t* count);

/* The log transform alone, on the HOST (no device, no ctx): out_db[i] = 10*log10(power[i]) where power != 0, else 0.0
 * (__init__.py:241), with the same correctly rounded logarithm the kernels use (csrc/shz_log10.h) -- the function
 * that decides ties in the peak test, exposed so that tests can pin it without a GPU. */
int32_t shz_db_values(const double* power, uint64_t n, double* out_db);

/* Constellation peaks of a batch: replaces get_2D_peaks(10*log10(specgram)) +
 * the stable time sort at generate_hashes (__init__.py:116-177, 194-195).
 * Outputs (host unless SHZ_OUT_DEVICE): peak_f/peak_t in (clip, time asc, freq asc)
 * order, peak_off[n_clips+1] CSR offsets. */
int32_t shz_peaks(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t n_clips,
                  uint32_t fs, double amp_min, uint32_t flags,
                  uint16_t* peak_f, uint32_t* peak_t, uint64_t* peak_off, uint64_t cap, uint64_t* count);

/* get_2D_peaks(arr2D, amp_min) on a caller-supplied 2-D float64 array (__init__.py:116-177):
 * arr2d is HOST, C-contiguous [n_rows(freq), n_cols(time)]; outputs in np.where row-major
 * order (freq asc, time asc) like the reference's return value. */
int32_t shz_peaks_from_db(shz_ctx* ctx, const double* arr2d, uint32_t n_rows, uint32_t n_cols,
                          double amp_min, uint32_t* out_f, uint32_t* out_t, uint64_t cap, uint64_t* count);

/* generate_hashes(peaks, fan_value) in packed form (__init__.py:179-210): peaks must be in
 * (time asc, freq asc) order per clip (peak_off CSR, host).  key32 = f1<<20 | f2<<8 | dt. */
int32_t shz_pair_hash(shz_ctx* ctx, const uint16_t* peak_f, const uint32_t* peak_t, const uint64_t* peak_off,
                      uint32_t n_clips, uint32_t fan_value,
                      uint32_t* key32, uint32_t* t1, uint64_t* hash_off, uint64_t cap, uint64_t* count);

/* fingerprint() for a batch of channels (__init__.py:212-245): PCM -> (key32, t1) in the
 * reference's generation order, hash_off[n_clips+1] CSR offsets (always HOST).
 * key32/t1 are host unless SHZ_OUT_DEVICE. */
int32_t shz_fingerprint_batch(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t n_clips,
                              uint32_t fs, double amp_min, uint32_t fan_value, uint32_t flags,
                              uint32_t* key32, uint32_t* t1, uint64_t* hash_off, uint64_t cap, uint64_t* count);

/* Host PCM of 192 MB or more (no SHZ_PCM_DEVICE) is fed in chunks of whole clips (16 MB, then ~64 MB): a helper thread uploads chunk
 * i + 1 on its own stream while the extraction pass of chunk i runs -- the call's rate is the link's.  Results are those of
 * one pass.  SHZ_UPLOAD_PIPELINE=0 in the environment turns it off.  Counters since the context was created: chunks and
 * bytes that went that way, seconds the upload thread spent copying, seconds the passes waited for a chunk. */
int32_t shz_upload_stats(shz_ctx* ctx, uint64_t* chunks, uint64_t* bytes, double* copy_s, double* wait_s);

/* The window of mlab.specgram as fingerprint() calls it (__init__.py:232-237): NFFT = wsize = 4096 is what the STFT kernel is
 * built for (its radix plan, LDS layout and the 2049-bin peak stage); `noverlap = int(wsize * wratio)` is free: every
 * extraction call of the context then cuts frames x[k hop : k hop + 4096], hop = 4096 - noverlap (mlab:307-308), and a clip
 * of n >= 4096 samples has (n - 4096) / hop + 1 frames (shz_frame_count_hop; shz_frame_count is the default hop 2048).
 * noverlap >= 4096: SHZ_E_INVALID (mlab raises ValueError, mlab:242).  Other wsize: not implemented -- the Python layer raises
 * NotImplementedError, there is no CPU fallback. */
int32_t shz_set_overlap(shz_ctx* ctx, uint32_t noverlap);
/* numpy's tables for a window of nfft samples, as the fp64 path uses them (no GPU needed; any argument may be NULL):
 * window[nfft] = np.hanning(nfft) (mlab.window_hanning), twiddles[2 nfft] = (re, im) of the values pocketfft's
 * sincos_2pibyn(nfft) holds (cos, +sin of 2 pi i / nfft: the forward passes conjugate), *sumsq = (window ** 2).sum() in
 * numpy's pairwise order.  For tests: tests/test_numpy_tables.py compares them with numpy's own bits on the host. */
int32_t shz_numpy_tables(uint32_t nfft, double* window, double* twiddles, double* sumsq);
/* The window of the fp64 path as the host's numpy forms it: window[4096] = np.hanning(4096), sumsq = (window ** 2).sum()
 * (mlab.window_hanning and the scaling of mlab._spectral_helper behind __init__.py:232-237).  The Python layer calls this for
 * every context it creates, so the window is numpy's by construction; without the call the libm form of shz_numpy_tables is
 * used (equal to numpy's on the hosts seen).  Affects the fp64 path only (ties, shz_stft_db), not the fp32 staging kernel. */
int32_t shz_set_numpy_window(shz_ctx* ctx, const double* window, double sumsq);
/* How the host's numpy multiplies complex numbers (`np.conj(result) * result` in mlab._spectral_helper): fused = 1 (default):
 * real part fma(re, re, im * im) -- numpy's SIMD product on x86-64 with FMA3, the hosts of the fixtures; fused = 0:
 * re * re + im * im.  The Python layer probes its numpy when it creates a context and says which. */
int32_t shz_set_numpy_product(shz_ctx* ctx, int32_t fused);
/* mlab.specgram(x, NFFT=nfft, Fs, window_hanning, noverlap)[0] -> 10*log10 where != 0 (__init__.py:232-241) for window sizes
 * OTHER than 4096: nfft a power of two in [64, 2048].  A generic kernel (one workgroup per frame, radix-2 in fp64) -- correct,
 * not fast; the reference and every caller of it use 4096.  pcm: host, one channel; out_db: host [nfft/2 + 1][n_frames]
 * (the reference's layout), cap_doubles its capacity; SHZ_STFT_POWER: the PSD instead of dB.  With shz_peaks_from_db and
 * shz_pair_hash this is fingerprint(wsize=nfft) as the reference composes it.  Other sizes: SHZ_E_UNSUPPORTED (8192 has no
 * packed key: key32 gives a frequency 12 bits; non-powers of two are not implemented). */
int32_t shz_stft_db_any(shz_ctx* ctx, const int16_t* pcm, uint64_t n_samples, uint32_t fs, uint32_t nfft, uint32_t noverlap,
                        uint32_t flags, double* out_db, uint64_t cap_doubles, uint64_t* n_frames);
uint32_t shz_frame_count_hop(uint64_t n_samples, uint32_t hop);

/* Staging precision of shz_peaks / shz_fingerprint_batch.  Default (0): the power spectrogram is staged in fp32 and
 * the cells fp32 cannot decide (shared window maxima, threshold within 1e-7) are re-derived in fp64; results are
 * those of the fp64 path bit for bit.  1: stage fp64 and decide everything in the peak kernel (twice the HBM traffic;
 * what a clip falls back to on stationary / plateau material, and always used for amp_min < 0).
 * Env SHZ_STAGE_F64=1 forces it process-wide. */
int32_t shz_set_stage_f64(shz_ctx* ctx, int32_t enabled);
/* Counters since ctx creation: cells left undecided by the fp32 pass, those that needed fp64 values, FFT frames
 * recomputed for them, whole passes repeated with fp64 staging, single clips re-run with fp64 staging (windows with more
 * tied cells than the verification kernel takes: only those clips are redone and spliced into the batch) and the frames
 * of those clips.  Any pointer may be NULL. */
int32_t shz_extract_stats(shz_ctx* ctx, uint64_t* undecided, uint64_t* decided_f64, uint64_t* frames_recomputed,
                          uint64_t* f64_passes, uint64_t* f64_clips, uint64_t* f64_clip_frames);

/* sha1(f"{f1}|{f2}|{dt}")[:10 bytes] per key (__init__.py:207-208; BINARY(10) at
 * mysql_database.py:48).  key32: host or device (SHZ_IN_DEVICE); out10: host [n][10]. */
int32_t shz_sha1_prefix(shz_ctx* ctx, const uint32_t* key32, uint64_t n, uint32_t flags, uint8_t* out10);

/* Inverse of the above for hashes that arrive as hex/BINARY(10) from outside (a MySQL dump, another
 * process): the preimage space is only 2049 x 2049 x 201 strings, so the GPU hashes all of it and
 * looks the digests up.  digests10: host [n][10]; key32_out: host [n], 0xFFFFFFFF where no preimage
 * exists.  Lets insert_hashes / SELECT_MULTIPLE (mysql_database.py:62-68, 82-86) take foreign hashes. */
int32_t shz_sha1_invert(shz_ctx* ctx, const uint8_t* digests10, uint64_t n, uint32_t* key32_out);

/* ---- fingerprint table (replaces the MySQL fingerprints table, mysql_database.py:46-68) - */
int32_t shz_table_create(shz_ctx* ctx, shz_table** out);
int32_t shz_table_destroy(shz_table* t);
/* INSERT IGNORE of rows (hash, song_id, offset) (mysql_database.py:62-68, 167-181);
 * rows are staged; duplicates on (song_id, offset, hash) are dropped at finalize. */
int32_t shz_table_insert(shz_table* t, const uint32_t* key32, const uint32_t* sid, const uint32_t* off,
                         uint64_t n, uint32_t flags);
/* same, one song per clip: song id of clip c = sid0 + c, rows from a CSR (key32, t1, hash_off) */
int32_t shz_table_insert_clips(shz_table* t, const uint32_t* key32, const uint32_t* t1, const uint64_t* hash_off,
                               uint32_t n_clips, uint32_t sid0, uint32_t flags);
/* sort staged+existing rows by (key, sid, off), drop duplicates, build the bucket index */
int32_t shz_table_finalize(shz_table* t);
/* Bulk build (the insert loop of fingerprint_directory, __init__.py:378-386, at database scale):
 * shz_table_reserve announces how many rows the table will hold and how many arrive between two seals; ONE slab for
 * the segments' columns, the run arena, the staging columns and the sort scratch are then allocated once, on a helper
 * thread beside the first fingerprint batches, and the build performs no further device allocation.  Without it
 * everything still works, allocating as it goes.  flags: SHZ_RESERVE_WAIT (return when the memory is there),
 * SHZ_RESERVE_GATHER: the table HOLDS its sealed runs in the arena (sized for all rows_hint rows, this rank's and its
 * peers') until shz_table_finalize / shz_table_allgather merges all of them at once -- seal_run never cuts a segment on
 * the way, so every row can still travel, and the one merge cuts segments by KEY RANGE (a query hash is then looked up
 * in one segment, not in all).  The flag takes effect with rows_hint = 0 too (nothing is allocated ahead then).
 * shz_table_seal_run turns the staged rows into a sorted run (bounded scratch: one batch; more than 2^32 - 4096 staged
 * rows become several runs) WITHOUT making them visible to queries.  Without SHZ_RESERVE_GATHER full segments are cut as
 * soon as enough rows wait (bounded arena).  shz_table_finalize merges what is left (k-way merge of the runs, 8 bytes
 * read + 12 written per row) and makes everything visible.  On a table whose active segment holds rows, or whose song
 * ids + offsets need more than 32 bits, seal_run is finalize -- on a table that holds its runs it is SHZ_E_UNSUPPORTED
 * instead and the rows stay staged (rows put into segments could not travel any more). */
int32_t shz_table_reserve(shz_table* t, uint64_t rows_hint, uint64_t batch_rows_hint, uint32_t flags);
int32_t shz_table_seal_run(shz_table* t);
/* rows one sealed run may hold (0 = the limit of a radix sort, 2^32 - 4096); small values make many runs of few rows (tests) */
int32_t shz_table_set_run_rows(shz_table* t, uint64_t rows);
/* A table is a list of sorted segments (each one radix sort, < 2^32 rows) that every probe visits; rows
 * beyond `rows` per segment open a new one at finalize.  Default 2^31; smaller values only for tests.
 * UNIQUE(song_id, offset, hash) + INSERT IGNORE (mysql_database.py:54-55, 62-68) hold across segments: staged rows
 * that already sit in a frozen segment are dropped at finalize, duplicates inside the batch by the sort. */
int32_t shz_table_set_segment_rows(shz_table* t, uint64_t rows);
/* ON DELETE CASCADE of fingerprints when songs are deleted (mysql_database.py:57-58; DELETE_UNFINGERPRINTED :132-134,
 * the reference's crash recovery at __init__.py:424): every row of the listed song ids leaves the table (all segments
 * and the staged rows), the order of the rest is kept.  sids: host array. */
int32_t shz_table_delete_songs(shz_table* t, const uint32_t* sids, uint64_t n_sids, uint64_t* rows_deleted);
/* DROP TABLE fingerprints: no rows, no segments; allocations of the active segment are kept for the rows to come. */
int32_t shz_table_clear(shz_table* t);
int32_t shz_table_rows(shz_table* t, uint64_t* n_rows, uint64_t* n_staged);
/* number of sorted segments the finalized rows live in (every query hash is looked up in each of them) */
int32_t shz_table_segments(shz_table* t, uint32_t* n_segments);
/* sorted rows to host (dump / parity): arrays of cap rows */
int32_t shz_table_export(shz_table* t, uint32_t* key32, uint32_t* sid, uint32_t* off, uint64_t cap, uint64_t* count);
/* SELECT hash, song_id, offset WHERE hash IN (keys) (SELECT_MULTIPLE, mysql_database.py:82-86;
 * recognizer.py:252-259): rows of every listed key, grouped in the order the keys are given,
 * inside a key ordered by (song_id, offset).  keys: host; outputs: host arrays of cap rows. */
int32_t shz_table_lookup(shz_table* t, const uint32_t* keys, uint64_t n_keys,
                         uint32_t* key32, uint32_t* sid, uint32_t* off, uint64_t cap, uint64_t* count);
/* distinct (hash, offset) rows of one song = songs.total_hashes candidates (__init__.py:381) */
int32_t shz_table_song_rows(shz_table* t, uint32_t sid, uint64_t* n_rows);

/* ---- match + align (replaces return_matches/align_matches, recognizer.py:222-338) ------- */
/* Queries are CSR: query q owns (key32, q_off) pairs [query_off[q], query_off[q+1]); duplicate
 * (key, q_off) pairs inside a query are collapsed (set semantics, recognizer.py:378-382).
 * Outputs (host), per query up to topn results ranked like align_matches:
 *   out_sid/out_delta/out_aligned [n_queries*topn]  (sid, db_off - q_off of the winning bin, its count)
 *   out_dedup  [n_queries*topn]  dedup_hashes[sid] (DB rows matched, once per row, recognizer.py:261-264)
 *   out_nres   [n_queries]       results valid for q
 *   out_nhash  [n_queries]       len(set(hashes)) = queried_hashes (recognizer.py:389)
 *   out_npairs [n_queries]       len(matches)
 */
int32_t shz_match_batch(shz_ctx* ctx, shz_table* t, const uint32_t* key32, const uint32_t* q_off,
                        const uint64_t* query_off, uint32_t n_queries, uint32_t topn, uint32_t flags,
                        uint32_t* out_sid, int32_t* out_delta, uint32_t* out_aligned, uint32_t* out_dedup,
                        uint32_t* out_nres, uint32_t* out_nhash, uint64_t* out_npairs);
/* Test switches that force the vote tiles' rare paths (never set in production): SHZ_DEBUG_VT_TINY_HEAVY gives the list
 * of ranges handed from vt_stream to vt_fold room for ONE range; SHZ_DEBUG_VT_PROBE1 lets an LDS hash probe give up
 * after one round (what a full table would cause).  Either way the pass's flag word is set, and the sub-batch is voted
 * again through the full sort (align_matches' result is a function of the votes, recognizer.py:289-338: same arrays).
 * shz_match_vt_redo: how many sub-batches went that way since the context was created. */
#define SHZ_DEBUG_VT_TINY_HEAVY 1u
#define SHZ_DEBUG_VT_PROBE1 2u
int32_t shz_set_debug(shz_ctx* ctx, uint32_t flags);
int32_t shz_match_vt_redo(shz_ctx* ctx, uint64_t* count);
/* A single query of at most 8,192 hashes handed over in host memory has its vote kernels queued before the number of its
 * votes is known (they do nothing when it exceeds 32,768; the call then continues as for any other query): how many
 * calls queued them / took their results from them, since the context was created.  Same results either way;
 * SHZ_MATCH_NO_SPEC=1 in the environment turns the queueing off. */
int32_t shz_match_spec_stats(shz_ctx* ctx, uint64_t* queued, uint64_t* used);
/* rows streamed / pairs voted by the last shz_match_batch (for HBM accounting) */
int32_t shz_match_stats(shz_ctx* ctx, uint64_t* rows_scanned, uint64_t* pairs, uint64_t* distinct_keys);

/* ---- multi-GPU database build (new) ---------------------------------------- */
/* RCCL communicator, one rank per GPU.  id: 128-byte ncclUniqueId made by rank 0 and shipped
 * to the other ranks by the host (a file, a socket, any key-value store the launcher offers). */
int32_t shz_comm_unique_id(uint8_t id_out[128]);
int32_t shz_comm_create(shz_ctx* ctx, const uint8_t id[128], int32_t rank, int32_t nranks, shz_comm** out);
/* The same communicator interface with the ranks as THREADS of one process (one context each, on one device or on several):
 * exchanges are rendezvous + device copies out of the peers' buffers.  Lets the N > 1 logic of the sharded build run on a
 * one-GPU box (tests), and a single process drive several GPUs without RCCL.  group_id: any number the ranks agree on. */
int32_t shz_comm_create_local(shz_ctx* ctx, uint64_t group_id, int32_t rank, int32_t nranks, shz_comm** out);
int32_t shz_comm_destroy(shz_comm* c);
/* The gathered build (the reference's analogue is the pool + insert loop of fingerprint_directory,
 * __init__.py:341, 357-386, which overlaps fingerprinting of the next song with the insert of the last).
 * Collective calls: every rank of the communicator makes them, on tables reserved with SHZ_RESERVE_GATHER.
 *
 * shz_table_exchange_run: the staged rows become a sorted run (as shz_table_seal_run) and ONE exchange round runs: all
 *   ranks all-gather a 320-byte block (largest song id / offset so far, flags, row counts and song-id ranges of the runs
 *   they have sealed and not yet sent -- up to 16 a round, each < 2^32 rows), agree on one packing layout from the global
 *   maxima, and every rank's announced runs start travelling to every peer, 8 bytes a row in pieces of <= 1 GB, each
 *   pair of GPUs on its own xGMI link (grouped ncclSend / ncclRecv), on the communicator's own stream: the call returns
 *   with the transfers in flight, and the next batch is fingerprinted beside them.  Ranks need not call it equally often.
 * shz_table_allgather: seals what is staged, runs rounds until every rank has arrived here and sent all its runs, waits
 *   for the transfers, and merges ALL runs -- its own and its peers' -- in one k-way merge into segments cut by key range
 *   (more than 32 runs: the smallest are merged first).  No rank sorts another rank's rows.  Afterwards every rank
 *   holds the same table.  Which rows travel: everything inserted since the table was last finalized / gathered --
 *   staged rows and sealed runs.
 * The column path: when any rank's table already holds rows, or its song ids + offsets need more than 32 bits, or
 *   SHZ_ALLGATHER=columns is set, the ranks' STAGED rows travel as unsorted columns and finalize sorts them into the
 *   table each rank holds.  Every rank takes it if any rank needs it.  Sealed runs do not travel on it: if any rank
 *   holds one, or its seal_run has already moved rows into segments (a table not reserved with SHZ_RESERVE_GATHER that
 *   sealed past a segment's worth), EVERY rank returns SHZ_E_STATE -- never a table that differs between ranks.
 * bytes_recv: payload bytes this rank received (all rounds of this build). */
int32_t shz_table_exchange_run(shz_table* t, shz_comm* c);
int32_t shz_table_allgather(shz_table* t, shz_comm* c, uint64_t* bytes_recv);
/* exchange rounds / payload bytes received / host seconds spent waiting for peers and transfers since the last
 * shz_table_allgather, and the runs the arena holds now.  Any pointer may be NULL. */
int32_t shz_table_exchange_stats(shz_table* t, uint64_t* rounds, uint64_t* bytes_recv, double* wait_s, uint32_t* runs_held);
/* The sort + merge + segments half of the above without a communicator: the staged rows are n_runs consecutive
 * blocks of run_rows[r] rows (what n_runs ranks would have staged); the result equals shz_table_finalize's. */
int32_t shz_table_finalize_runs(shz_table* t, const uint64_t* run_rows, uint32_t n_runs);
/* seconds the last shz_table_allgather / shz_table_finalize_runs spent sorting its own rows, inside exchange rounds
 * (host time: waiting for peers and for transfers, queueing them -- with pipelined rounds most of a transfer runs beside
 * fingerprinting and shows up nowhere), merging the runs and cutting segments.  Any pointer may be NULL. */
int32_t shz_table_build_stats(shz_table* t, double* sort_s, double* exchange_s, double* merge_s, double* segments_s);
/* Host seconds the table spent per phase of the build since the last reset (stream drained at each phase border):
 * staging allocation, insert, INSERT-IGNORE anti-join against frozen segments, segment top-up, maxima, sort, merge,
 * unique + scan, column allocation, compaction into columns, bucket index, slicing, release of the staging columns.
 * The reference's analogue is the per-file wall time of fingerprint_directory's insert loop (__init__.py:378-386).
 * seconds: host array of cap entries (may be NULL); *n = number of phases; shz_table_phase_name(i) names phase i. */
int32_t shz_table_phase_stats(shz_table* t, double* seconds, uint32_t cap, uint32_t* n, int32_t reset);
const char* shz_table_phase_name(uint32_t i);
int32_t shz_comm_barrier(shz_comm* c);
/* Collective: one tiny exchange of each kind the gathered build uses, on the communicator's exchange stream.  RCCL connects
 * two ranks when they first talk to each other; a build whose time matters calls this before its clock starts. */
int32_t shz_comm_warmup(shz_comm* c);

/* ---- key-sharded table (new; SURVEY.md 8f row 4: the table no longer fits one GPU) -------
 * Rows are partitioned by a hash of key32, so a DB row lives on exactly one shard and both quantities
 * align_matches needs are sums over shards: dedup_hashes[sid] (recognizer.py:261-264) and the
 * (sid, offset difference) histogram (recognizer.py:305).  Build: every rank stages its own tracks' rows,
 * shz_table_shard_exchange routes each row to the rank that owns its key (all-to-all over RCCL) and
 * finalizes.  Query: every rank runs shz_match_pairs on the same queries, shz_pairs_allgather collects the
 * packed votes, shz_pairs_vote ranks them: the same kernels as shz_match_batch on the unsharded table. */
/* shard (0..nshards-1) of each key; host arrays */
int32_t shz_shard_of_keys(const uint32_t* key32, uint64_t n, uint32_t nshards, uint32_t* shard_out);
/* drop the STAGED rows that do not belong to `shard` (several shards on one GPU, tests) */
int32_t shz_table_keep_shard(shz_table* t, uint32_t shard, uint32_t nshards);
/* append the STAGED rows of src that belong to `shard` to dst's staged rows (src unchanged), and forget a
 * table's staged rows: one staging table feeding several shard tables on one GPU */
int32_t shz_table_stage_from(shz_table* dst, shz_table* src, uint32_t shard, uint32_t nshards);
int32_t shz_table_clear_staged(shz_table* t);
/* route every rank's STAGED rows to the owner of their key, then finalize; bytes_recv: payload received */
int32_t shz_table_shard_exchange(shz_table* t, shz_comm* c, uint64_t* bytes_recv);
/* Votes travel packed, 8 bytes each, in ONE layout all shards agree on:
 *   ((query << sid_bits | song_id) << delta_bits | (db_off - q_off) + bias) << 1 | counts-a-DB-row-once flag
 * sid_bits >= bits of the largest song id of the WHOLE table, bias >= the largest q_off of the batch,
 * delta_bits >= bits of (largest offset of the whole table + bias); bits(n_queries-1) + sid_bits + delta_bits + 1
 * must fit 64.  shz_table_maxima gives a table's largest song id / offset (after shz_table_shard_exchange:
 * of the whole sharded table). */
int32_t shz_table_maxima(shz_table* t, uint32_t* max_sid, uint32_t* max_off);
/* probe + expand only (the head of shz_match_batch): the votes of this table's rows for the queries, appended
 * to a DEVICE buffer of cap entries.  More than cap: SHZ_E_CAPACITY with *count = the number needed.
 * The table holds shard `shard` of `nshards` (1 shard: everything): only the query hashes that shard owns are
 * looked up, so S shards together do the work of one table.
 * out_nhash / out_npairs (host, may be NULL): per query the distinct hashes / matches found HERE -- both add up
 * over shards, because every hash and every DB row belongs to exactly one of them. */
int32_t shz_match_pairs(shz_ctx* ctx, shz_table* t, const uint32_t* key32, const uint32_t* q_off,
                        const uint64_t* query_off, uint32_t n_queries, uint32_t flags,
                        uint32_t shard, uint32_t nshards, uint32_t sid_bits, uint32_t delta_bits, uint32_t bias,
                        uint64_t* d_pairs, uint64_t cap, uint64_t* count, uint32_t* out_nhash, uint64_t* out_npairs);
/* all-gather the ranks' votes (device in, device buffer of cap entries out) */
int32_t shz_pairs_allgather(shz_comm* c, uint64_t n_local, const uint64_t* d_pairs, uint64_t* d_all, uint64_t cap,
                            uint64_t* n_total);
/* the tail of shz_match_batch over any collection of votes in that layout: sort, per (query, song) fold, top-n
 * ranked like align_matches (recognizer.py:289-338).  d_pairs (device, 16-byte aligned) is overwritten.  n < 2^32.
 * Outputs (host) as in shz_match_batch. */
int32_t shz_pairs_vote(shz_ctx* ctx, uint64_t* d_pairs, uint64_t n, uint32_t n_queries, uint32_t sid_bits,
                       uint32_t delta_bits, uint32_t bias, uint32_t topn, uint32_t* out_sid, int32_t* out_delta,
                       uint32_t* out_aligned, uint32_t* out_dedup, uint32_t* out_nres);

#ifdef __cplusplus
}
#endif
#endif /* SHZ_H */
