// Internal declarations shared by the libshz.so translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/shz.h"

#define SHZ_FAIL(ctx, code, ...)                                   \
  do {                                                             \
    char _b[512];                                                  \
    snprintf(_b, sizeof(_b), __VA_ARGS__);                         \
    (ctx)->err = _b;                                               \
    return (code);                                                 \
  } while (0)

#define SHZ_HIP(ctx, call)                                                                       \
  do {                                                                                           \
    hipError_t _e = (call);                                                                      \
    if (_e != hipSuccess) SHZ_FAIL(ctx, SHZ_E_HIP, "%s failed: %s (%s:%d)", #call,                \
                                   hipGetErrorString(_e), __FILE__, __LINE__);                   \
  } while (0)

#define SHZ_TRY(expr)                  \
  do {                                 \
    int32_t _s = (expr);               \
    if (_s != SHZ_OK) return _s;       \
  } while (0)

static inline double now_seconds() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct shz_prof_rec;
// a growable device buffer owned by the ctx (scratch arena slot)
struct shz_buf {
  void* p = nullptr;
  uint64_t cap = 0;
};

enum {
  SHZ_WS_DB = 0,     // dB spectrogram [frames][row_stride] f64
  SHZ_WS_MASK,       // peak bit masks
  SHZ_WS_SCAN,       // scan outputs
  SHZ_WS_SCAN_TMP,   // scan block sums
  SHZ_WS_PEAK_F,
  SHZ_WS_PEAK_T,
  SHZ_WS_PEAK_CLIP,
  SHZ_WS_HCNT,
  SHZ_WS_HOFF,
  SHZ_WS_META,       // per-clip metadata (frame offsets, sample offsets)
  SHZ_WS_META2,
  SHZ_WS_PCM,        // staged host PCM
  SHZ_WS_KEY,
  SHZ_WS_T1,
  SHZ_WS_MISC0,
  SHZ_WS_MISC1,
  SHZ_WS_MISC2,
  SHZ_WS_MISC3,
  SHZ_WS_SORT_A,
  SHZ_WS_SORT_B,
  SHZ_WS_SORT_C,
  SHZ_WS_SORT_D,
  SHZ_WS_SORT_H,
  SHZ_WS_DB2,        // second staged spectrogram: stft of sub-batch i+1 runs beside peak picking of sub-batch i
  SHZ_WS_META_B,
  SHZ_WS_PCM_B,
  SHZ_WS_CTL,        // device control block of the extraction pass (xctl)
  SHZ_WS_OFFS,       // per-clip output offsets (u64)
  SHZ_WS_UND,        // undecided cells of fp32 peak picking
  SHZ_WS_M0, SHZ_WS_M1, SHZ_WS_M2, SHZ_WS_M3, SHZ_WS_M4, SHZ_WS_M5, SHZ_WS_M6, SHZ_WS_M7,
  SHZ_WS_M8, SHZ_WS_M9,      // top-n candidates of the vote fold (M3 / M4 hold the probe's group tables until the last vote pass)
  SHZ_WS_VT0, SHZ_WS_VT1, SHZ_WS_VT2, SHZ_WS_VT3,   // vote tiles: tile starts, candidate records
  SHZ_WS_VT4,        // table of the vote passes
  SHZ_WS_VT5,        // sub-group of every expand chunk's first vote (expand by sort blocks)
  SHZ_WS_VT6,        // the bar of every query of a vote pass (vt_stream2_kernel)
  SHZ_WS_COUNT
};

struct shz_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  hipDeviceProp_t prop;
  uint64_t ws_limit = 0;
  shz_buf ws[SHZ_WS_COUNT];
  // constant tables (device)
  double* d_window = nullptr;   // hann(4096)
  double2* d_twiddle = nullptr; // W4096^k, k in [0,1024]
  int16_t* d_sine_lut = nullptr;
  double m_votes_per_hash = 0.0;   // votes per query hash of the last match sub-batch: sizes the next one before it is tried
  double win_sumsq = 0.0;
  // numpy's own tables (shz_numpy_tables): np.hanning(4096), pocketfft's twiddles exp(-2 pi i k / 4096) as its
  // sincos_2pibyn computes them, sum(window ** 2) in numpy's pairwise order -- for the fp64 path that follows numpy's
  // arithmetic operation by operation (stft_np_kernel / peak_verify_kernel)
  double* d_np_window = nullptr;
  double2* d_np_comp = nullptr;
  double np_sumsq = 0.0;
  bool np_unfused = false;       // numpy's complex product on this host has no FMA (shz_set_numpy_product)
  uint32_t hop = SHZ_HOP;          // new samples per frame: NFFT - noverlap (shz_set_overlap; the reference's wratio)
  // timers / profiling
  hipEvent_t tev[16][2];
  bool tev_init = false;
  bool profiling = false;
  float kernel_ms[8] = {0};
  uint32_t kernel_launches[8] = {0};
  std::vector<struct shz_prof_rec> prof_pending;  // recorded, not yet resolved
  std::vector<struct shz_prof_rec> prof_free;
  // match stats
  uint64_t st_rows = 0, st_pairs = 0, st_keys = 0;
  uint64_t st_spec_queued = 0, st_spec_used = 0;   // single small queries whose votes were queued ahead of the count / whose results that gave
  uint64_t st_vt_redo = 0;      // sub-batches whose vote tiles flagged an overflow and were voted again by the full sort
  uint32_t debug = 0;           // SHZ_DEBUG_* (shz_set_debug): switches that force rare paths, for tests
  // extraction stats: cells fp32 peak picking left undecided, of those decided on fp64 values, frames recomputed for
  // that, passes repeated with fp64 staging
  uint64_t st_und = 0, st_und_f64 = 0, st_und_ffts = 0, st_fallbacks = 0;
  uint64_t st_f64_clips = 0, st_f64_frames = 0;   // clips re-run one by one with fp64 staging (per-clip fallback), their frames
  bool stage_f64 = false;       // shz_set_stage_f64: stage fp64 power and decide ties in peak_pick (no fp32 pass)
  // pinned bounce buffers of shz_memcpy (two halves, an event each)
  void* pin[2] = {nullptr, nullptr};
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  bool pin_busy[2] = {false, false};
  hipStream_t stream2 = nullptr;   // second stream of the extraction pipeline (created on first use)
  hipStream_t stream_up = nullptr; // uploads of host PCM, chunk by chunk beside the kernels of the chunk before (created on first use)
  uint64_t st_up_chunks = 0, st_up_bytes = 0;   // chunks / bytes that went through that pipeline
  double st_up_copy_s = 0.0, st_up_wait_s = 0.0;   // seconds the upload thread spent copying / the main thread waited for a chunk
  hipEvent_t ev_stft[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
  shz_ctx* twin = nullptr;         // second pipeline of a dual extraction pass: own stream, own workspace (created on first use)
  hipEvent_t ev_twin = nullptr;
  void* mail = nullptr;         // shz_mailbox
  uint64_t mail_cap = 0;
  // large device blocks (table slabs, run arenas, staging columns) that a destroyed table handed back: kept for the next
  // table instead of going through hipFree + hipMalloc (shz_block_alloc / shz_block_free; shz_release_workspace drops them)
  std::vector<shz_buf> blocks;
  std::mutex blocks_mu;
};

// Copy on ctx->stream.  Host <-> device copies of 16 KB .. 64 MB go through pinned bounce buffers owned by the context
// instead of handing the caller's pages to the driver: HIP registers pageable memory of that size with the kernel
// driver for the DMA, and when the application later frees it the driver evicts and restores every queue of the
// process -- 20-28 ms in the next device call (seen as every other query batch of a stream being 6x slower).  Smaller
// copies use the runtime's own staging, larger ones amortise one such event.  H2D: the source may be reused on return;
// D2H (bounced sizes): the data is there on return.
hipError_t shz_memcpy(shz_ctx* ctx, void* dst, const void* src, uint64_t bytes, hipMemcpyKind kind);

// pinned host memory owned by the ctx for small latency-critical transfers (counts, result blocks, query uploads):
// grows to the largest request; contents are the caller's between two calls
void shz_numpy_tables_host(uint32_t n, double* window, double2* comp, double* sumsq);
int32_t shz_mailbox(shz_ctx* ctx, uint64_t bytes, void** out);

// Big blocks with reuse.  A hipFree'd gigabyte comes back from the driver scrubbed at ~40 GB/s on its next use (measured:
// the same 207 GB reservation took 0 s on untouched memory and 5 s after 70 GB had been freed in the process), so blocks
// a table gives up stay with the context.  alloc: the smallest cached block of bytes <= size <= 1.5 x bytes, else
// hipMalloc (cached blocks are dropped first if the device is short); *got = the block's real size.  Thread-safe.
hipError_t shz_block_alloc(shz_ctx* ctx, uint64_t bytes, void** out, uint64_t* got);
void shz_block_free(shz_ctx* ctx, void* p, uint64_t bytes);

// ensure ws slot has >= bytes (grow-only; contents not preserved)
int32_t shz_ws_reserve(shz_ctx* ctx, int slot, uint64_t bytes, void** out);

// profiling helpers: bracket a kernel group with events when ctx->profiling.  Events are only
// recorded here (no host sync inside a timed region); shz_get_kernel_ms drains them.
struct shz_prof_rec { hipEvent_t a, b; int which; };
void shz_prof_begin(shz_ctx* ctx, int which);
void shz_prof_end(shz_ctx* ctx);
struct shz_prof_scope {
  shz_ctx* c;
  shz_prof_scope(shz_ctx* ctx, int w) : c(ctx) { if (c->profiling) shz_prof_begin(c, w); }
  ~shz_prof_scope() { if (c->profiling) shz_prof_end(c); }
};

// ---- device primitives (shz_prims.hip) ------------------------------------------------
// exclusive scan of n u32 values -> u32 (total written to d_total if non-null, as u64)
int32_t shz_scan_u32(shz_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, uint64_t n, uint64_t* d_total);
// exclusive scan of popcount(mask[i]) for u64 words
int32_t shz_scan_popc64(shz_ctx* ctx, const uint64_t* d_in, uint32_t* d_out, uint64_t n, uint64_t* d_total);
// exclusive scan of n u64 values -> u64
int32_t shz_scan_u64(shz_ctx* ctx, const uint64_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* d_total);
// stable LSD radix sort of u64 keys (bits [bit_lo, bit_hi)) with an optional 4- or 8-byte payload.
// keys/vals are ping-ponged between (k0,v0) and (k1,v1); *out_sel tells which pair holds the result.
int32_t shz_sort_u64(shz_ctx* ctx, uint64_t* k0, uint64_t* k1, void* v0, void* v1, int vbytes, uint64_t n,
                     int bit_lo, int bit_hi, int* out_sel);

// stable LSD radix sort of 4-byte keys on bits [bit_lo, bit_hi) (k0 <-> k1 ping-pong); the result is written to out64 as
// 8-byte keys, key + add
// segments of a segmented 4-byte sort (shz_sort_u32_seg): segment i holds the keys [qv[i], qv[i + 1]) and is cut into
// the blocks [bq[i], bq[i + 1]) of <= 4,096 or 8,192 keys (bq is filled in by the sort); no block crosses a segment border
#define SHZ_SEG_MAX 128
struct shz_seg_plan {
  uint32_t nq;
  uint32_t qv[SHZ_SEG_MAX + 1];
  uint32_t bq[SHZ_SEG_MAX + 1];
};
int32_t shz_sort_u32_seg(shz_ctx* ctx, uint32_t* k0, uint32_t* k1, uint64_t n, int bit_lo, int bit_hi, const shz_seg_plan& sp,
                         int* sel, bool hist0 = false);
uint32_t shz_seg_tile(uint64_t n);                          // keys per block the sort of n keys uses
void shz_seg_blocks(shz_seg_plan* sp, uint32_t tile);       // bq from nq, qv
int shz_seg_first_pass(int bit_lo, int bit_hi, uint32_t* dmask);   // digit width (8 or 9) and mask of the first pass
int32_t shz_sort_u32_widen(shz_ctx* ctx, uint32_t* k0, uint32_t* k1, uint64_t* out64, uint64_t n, int bit_lo, int bit_hi,
                           uint64_t add, int* sel);

// ---- RCCL helpers (shz_comm.hip) ----------------------------------------------------------
int32_t shz_comm_info(shz_comm* c, int* rank, int* nranks);
shz_ctx* shz_comm_ctx(shz_comm* c);
int32_t shz_comm_allgather_bytes(shz_comm* c, const void* d_send, void* d_recv, uint64_t bytes);
int32_t shz_comm_alltoallv_bytes(shz_comm* c, const void* d_send, const uint64_t* scount, const uint64_t* sdispl,
                                 void* d_recv, const uint64_t* rcount, const uint64_t* rdispl);
int32_t shz_comm_allgatherv_bytes(shz_comm* c, const void* d_send, void* d_recv, const uint64_t* counts,
                                  const uint64_t* displ);
// the gathered build's transport: a second stream owned by the communicator, a blocking all-gather of small host
// blocks on it, and the all-gather of lists of device buffers (see shz_comm.hip)
struct shz_xfer { void* p; uint64_t bytes; };
int32_t shz_comm_exchange_stream(shz_comm* c, hipStream_t* out);
int32_t shz_comm_allgather_bytes_on(shz_comm* c, hipStream_t s, const void* d_send, void* d_recv, uint64_t bytes);
int32_t shz_comm_allgather_host(shz_comm* c, const void* h_mine, void* h_all, uint64_t bytes);
int32_t shz_comm_allgather_lists_on(shz_comm* c, hipStream_t s, const std::vector<shz_xfer>& send,
                                    const std::vector<std::vector<shz_xfer>>& recv);
