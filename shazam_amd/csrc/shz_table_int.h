// Internal to libshz.so: the fingerprint table as the build side (shz_build.hip) and the match side (shz_table.hip)
// both see it.  Not part of the ABI.
#pragma once
#include <time.h>

#include <algorithm>

#include "shz_internal.h"

// A table is a list of immutable sorted SEGMENTS (each < 2^32 rows, its own bucket index) plus the
// active segment below that new rows are merged into; a probe visits every segment.  Segments lift
// the 2^32-row limit of one radix sort (BASELINE config 3: 7.1e9 rows = 85 GB fits one GPU's HBM).
struct shz_seg {
  uint32_t *key, *sid, *off, *bucket;
  uint64_t n, nbuckets;
  uint32_t sid_lo = 0, sid_hi = 0xFFFFFFFFu;   // song ids the segment may hold (conservative, inclusive)
  bool slab = false;                           // columns carved from the table's slab: never freed on their own
  uint32_t key_lo = 0;                         // first key of the segment (its last is (nbuckets << 8) - 1 at most): the segments
                                               //   one k-way merge cuts hold disjoint key ranges, and a probe skips the foreign ones
};
struct shz_seg_dev {  // what the match kernels see
  const uint32_t *key, *sid, *off, *bucket;
  uint32_t n;
  uint32_t key_lo;      // first key (keys below it have no rows here)
  uint64_t nbuckets;    // ((last key) >> 8) + 1
};
static inline shz_seg_dev seg_dev_of(const shz_seg& g) { return shz_seg_dev{g.key, g.sid, g.off, g.bucket, (uint32_t)g.n, g.key_lo, g.nbuckets}; }
#define SHZ_MAX_SEGS 32

#define SHZ_TABLE_PHASES 24

// a sorted run of packed rows (key << (sb + ob) | sid << ob | off) waiting in the run arena for the k-way merge
enum { RUN_LOCAL = 0 /* made here, no peer has it yet */, RUN_SENT = 1 /* made here, every peer has it */, RUN_RECV = 2 /* a peer's */ };
struct shz_run {
  uint64_t off, n;             // position / rows in shz_table::rbuf
  uint32_t sid_lo, sid_hi;     // song ids of the run: runs with disjoint ranges cannot hold the same row
  uint8_t where = RUN_LOCAL;   // gathered build: has the run travelled?
};
struct shz_reserve_job;        // background allocation of the arenas (shz_table_reserve)

struct shz_table {
  shz_ctx* ctx = nullptr;
  std::vector<shz_seg> done;           // frozen segments
  uint64_t seg_limit = 1ull << 31;     // rows per segment (bounds the sort scratch: 16 B/row)
  uint32_t *key = nullptr, *sid = nullptr, *off = nullptr;   // active segment
  uint64_t n = 0;
  uint64_t cap = 0, bcap = 0;  // rows the active columns / entries the bucket array can hold (reused across finalize calls)
  uint32_t *skey = nullptr, *ssid = nullptr, *soff = nullptr;
  uint64_t ns = 0, scap = 0;
  uint32_t* bucket = nullptr;
  uint64_t nbuckets = 0;  // bucket has nbuckets+1 entries
  uint32_t max_sid = 0, max_off = 0;
  double bs_sort = 0, bs_exchange = 0, bs_merge = 0, bs_segments = 0;   // seconds of the last run-merge build
  bool broken = false;  // a finalize ran out of memory after giving up the old columns: rows were lost, refuse further use
  double ph[SHZ_TABLE_PHASES] = {0};   // host seconds per build phase since the last reset (shz_table_phase_stats)
  // ---- bulk build: staged rows become sorted runs, runs become segments in one k-way merge (shz_build.hip)
  uint32_t act_sid_lo = 0, act_sid_hi = 0xFFFFFFFFu;   // song ids of the active segment
  uint32_t act_key_lo = 0;                              // its first key
  bool act_slab = false;                                // the active columns are carved from the slab
  char* slab = nullptr;                                 // ONE allocation the segments' columns are carved from
  uint64_t slab_bytes = 0, slab_used = 0;
  uint64_t* rbuf = nullptr;                             // run arena
  uint64_t rcap = 0, rbuf_bytes = 0;                    // ... rows it holds, size of the block
  uint64_t st_bytes[3] = {0, 0, 0};                     // sizes of the reserved staging blocks
  std::vector<shz_run> runs;
  int run_sb = 0, run_ob = 0;                           // packing of the runs (0: none yet)
  double votes_per_hash = 0.0;                          // of the last match against this table (0: none yet): whether a single query's votes are worth queueing ahead of their count
  bool stage_reserved = false;                          // the staging columns were sized by shz_table_reserve: kept
  shz_reserve_job* job = nullptr;
  // ---- gathered build (shz_table_exchange_run / shz_table_allgather)
  bool hold_runs = false;                               // sealed runs wait in the arena until finalize / allgather, seal_run never cuts a segment (until the final merge)
  bool hold_reserved = false;                           // ... asked for by shz_table_reserve(SHZ_RESERVE_GATHER): every bulk build of this table holds its runs
  uint64_t run_limit = 0;                               // rows a sealed run may hold (0: 2^32 - 4096); small values force many runs (tests)
  uint64_t rows_cut = 0;                                // rows seal_run moved into segments since the last allgather / clear: they cannot travel any more
  uint64_t gx_recv_bytes = 0, gx_rounds = 0;            // payload received / exchange rounds since the last allgather
  double gx_wait_s = 0.0, gx_xfer_s = 0.0;              // host seconds waiting for peers / transfers inside exchange rounds; queueing (in-process transport: making) the transfers
  hipEvent_t gx_ev = nullptr;                           // "the runs this round ships are complete" (context's stream -> exchange stream)
  hipStream_t gx_stream = nullptr;                      // a transfer into / out of the arena may be in flight on this stream
  uint32_t* d_kw_err = nullptr;                         // error word of the k-way merge's tile kernels
};

// phases of the single-GPU build, in the order shz_table_phase_stats reports them
enum { PH_STAGE_ALLOC = 0, PH_INSERT, PH_DEDUP_FROZEN, PH_TOPUP, PH_MAXES, PH_SORT, PH_MERGE, PH_UNIQ, PH_COL_ALLOC, PH_COMPACT,
       PH_BUCKET, PH_SLICE, PH_STAGE_FREE, PH_RUN_PACK, PH_RUN_SORT, PH_RUN_UNIQ, PH_KW_PLAN, PH_KW_MERGE, PH_RESERVE_WAIT,
       PH_RESERVE_ALLOC /* helper thread's hipMalloc seconds: beside the build, not part of it */, PH_COUNT };
static_assert(PH_COUNT <= SHZ_TABLE_PHASES, "phase table too small");
static inline double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
// a phase ends here: the stream is drained so that the host clock sees the kernels of the phase
struct ph_clock {
  shz_table* t;
  double t0;
  explicit ph_clock(shz_table* t_) : t(t_), t0(now_s()) {}
  void lap(int which) {
    (void)hipStreamSynchronize(t->ctx->stream);
    const double t1 = now_s();
    t->ph[which] += t1 - t0;
    t0 = t1;
  }
  void skip() { t0 = now_s(); }
};

static inline std::vector<shz_seg> all_segs(const shz_table* t) {
  std::vector<shz_seg> v = t->done;
  if (t->n) v.push_back(shz_seg{t->key, t->sid, t->off, t->bucket, t->n, t->nbuckets, t->act_sid_lo, t->act_sid_hi, t->act_slab, t->act_key_lo});
  return v;
}
static inline uint64_t total_rows(const shz_table* t) {
  uint64_t n = t->n;
  for (const shz_seg& g : t->done) n += g.n;
  return n;
}

// rows inserted but not yet visible to queries: staged columns + sealed runs
static inline uint64_t pending_rows(const shz_table* t) {
  uint64_t n = t->ns;
  for (const shz_run& r : t->runs) n += r.n;
  return n;
}

static inline int bits_for(uint64_t v) {
  int b = 0;
  while (v) { ++b; v >>= 1; }
  return b ? b : 1;
}

// device allocations of one call: freed when the call leaves early (SHZ_TRY / SHZ_FAIL return), handed over with take()
struct dev_cols {
  uint32_t* p[3] = {nullptr, nullptr, nullptr};
  ~dev_cols() {
    for (uint32_t* q : p)
      if (q) (void)hipFree(q);
  }
  bool alloc(uint64_t rows) {
    for (auto& q : p)
      if (hipMalloc(&q, std::max<uint64_t>(rows, 1) * 4) != hipSuccess) return false;
    return true;
  }
  uint32_t* take(int i) { uint32_t* q = p[i]; p[i] = nullptr; return q; }
};

static inline bool ranges_overlap(uint32_t a_lo, uint32_t a_hi, uint32_t b_lo, uint32_t b_hi) { return a_lo <= b_hi && b_lo <= a_hi; }
static inline unsigned nblk(uint64_t n) { return (unsigned)((n + 255) / 256); }

// shard of a key: a different multiplier and bit field than slice_of (segments inside a shard stay balanced)
__host__ __device__ __forceinline__ uint32_t shard_of(uint32_t key, uint32_t nshards) {
  return (((key ^ (key >> 15)) * 0x85EBCA6Bu) >> 10) % nshards;
}
