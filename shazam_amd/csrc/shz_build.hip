// The fingerprint table resident in HBM: storage, INSERT IGNORE / ON DELETE CASCADE / DROP TABLE semantics, and the
// database build (single GPU and the RCCL all-gather of sorted runs).
//   table rows (key32, song_id, offset) replace the MySQL `fingerprints` table
//     (mysql_database.py:46-68: INDEX on hash, UNIQUE(song_id, offset, hash), INSERT IGNORE)
// The match / align path on the table lives in shz_table.hip.
//
// Layout: three u32 column arrays sorted by (key, sid, off), duplicates removed, plus a bucket
// index over key >> 8 (first row of every (f1, f2) prefix) so a probe is one index read and a
// short binary search over dt.
#include "shz_table_int.h"

// ---------------------------------------------------------------------------------------- kernels
__global__ void tbl_expand_clips_kernel(const uint32_t* __restrict__ key32, const uint32_t* __restrict__ t1,
                                        const uint64_t* __restrict__ hash_off, uint32_t n_clips, uint32_t sid0,
                                        uint64_t n, uint32_t* __restrict__ okey, uint32_t* __restrict__ osid,
                                        uint32_t* __restrict__ ooff) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t h = hash_off[0] + i;
  uint32_t lo = 0, hi = n_clips;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (hash_off[mid] <= h) lo = mid; else hi = mid;
  }
  okey[i] = key32[h];
  ooff[i] = t1[h];
  osid[i] = sid0 + lo;
}

__global__ void tbl_compose_kernel(const uint32_t* __restrict__ key, const uint32_t* __restrict__ sid,
                                   const uint32_t* __restrict__ off, uint64_t n, uint64_t dst0, uint64_t* __restrict__ k,
                                   uint32_t* __restrict__ v, uint32_t* __restrict__ maxes) {
  // grid-stride: a bounded number of workgroups so the two atomicMax words see few, not millions of, updates
  uint32_t s = 0, o = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t si = sid[i], oi = off[i];
    k[dst0 + i] = ((uint64_t)si << 32) | oi;
    v[dst0 + i] = key[i];
    s = max(s, si);
    o = max(o, oi);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    s = max(s, (uint32_t)__shfl_xor((int)s, d, 64));
    o = max(o, (uint32_t)__shfl_xor((int)o, d, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&maxes[0], s);
    atomicMax(&maxes[1], o);
  }
}

__global__ void tbl_swap_kernel(const uint64_t* __restrict__ k, const uint32_t* __restrict__ v, uint64_t n,
                                uint64_t* __restrict__ k2, uint64_t* __restrict__ v2) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  k2[i] = v[i];
  v2[i] = k[i];
}

__global__ void tbl_uniq_flag_kernel(const uint64_t* __restrict__ k, const uint64_t* __restrict__ v, uint64_t n,
                                     uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = (i == 0 || k[i] != k[i - 1] || v[i] != v[i - 1]) ? 1u : 0u;
}

__global__ void tbl_compact_kernel(const uint64_t* __restrict__ k, const uint64_t* __restrict__ v,
                                   const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, uint64_t n,
                                   uint32_t* __restrict__ okey, uint32_t* __restrict__ osid, uint32_t* __restrict__ ooff) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const uint32_t p = pos[i];
  okey[p] = (uint32_t)k[i];
  osid[p] = (uint32_t)(v[i] >> 32);
  ooff[p] = (uint32_t)v[i];
}

__global__ void tbl_bucket_kernel(const uint32_t* __restrict__ key, uint32_t n, uint64_t nbuckets,
                                  uint32_t* __restrict__ bucket) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nbuckets) return;
  const uint64_t target = b << 8;  // first key of the bucket (may exceed 32 bits for b = nbuckets)
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = lo + ((hi - lo) >> 1);
    if ((uint64_t)key[mid] < target) lo = mid + 1; else hi = mid;
  }
  bucket[b] = lo;
}

__global__ void tbl_count_sid_kernel(const uint32_t* __restrict__ sid, uint64_t n, uint32_t want,
                                     unsigned long long* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool hit = i < n && sid[i] == want;
  const unsigned long long b = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, (unsigned long long)__popcll(b));
}

// fast path of finalize: when song-id and offset bits fit beside the 32 key bits, a row is ONE u64
// (key | sid | off) and the whole order is a single payload-free radix sort
__global__ void tbl_max_kernel(const uint32_t* __restrict__ sid, const uint32_t* __restrict__ off, uint64_t n,
                               uint32_t* __restrict__ maxes) {
  uint32_t s = 0, o = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    s = max(s, sid[i]);
    o = max(o, off[i]);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    s = max(s, (uint32_t)__shfl_xor((int)s, d, 64));
    o = max(o, (uint32_t)__shfl_xor((int)o, d, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&maxes[0], s);
    atomicMax(&maxes[1], o);
  }
}

__global__ void tbl_compose1_kernel(const uint32_t* __restrict__ key, const uint32_t* __restrict__ sid,
                                    const uint32_t* __restrict__ off, uint64_t n, uint64_t dst0, int sb, int ob,
                                    uint64_t* __restrict__ c) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    c[dst0 + i] = ((uint64_t)key[i] << (sb + ob)) | ((uint64_t)sid[i] << ob) | off[i];
}

__global__ void tbl_uniq1_flag_kernel(const uint64_t* __restrict__ c, uint64_t n, uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || c[i] != c[i - 1]) ? 1u : 0u;
}

__global__ void tbl_compact1_kernel(const uint64_t* __restrict__ c, const uint32_t* __restrict__ flag,
                                    const uint32_t* __restrict__ pos, uint64_t n, int sb, int ob,
                                    uint32_t* __restrict__ okey, uint32_t* __restrict__ osid, uint32_t* __restrict__ ooff) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const uint32_t p = pos[i];
  const uint64_t v = c[i];
  okey[p] = (uint32_t)(v >> (sb + ob));
  osid[p] = (uint32_t)((v >> ob) & ((1ull << sb) - 1));
  ooff[p] = (uint32_t)(v & ((1ull << ob) - 1));
}

__device__ __forceinline__ uint32_t slice_of(uint32_t key, uint32_t nsl) { return ((key * 2654435761u) >> 12) % nsl; }
__global__ void tbl_slice_flag_kernel(const uint32_t* __restrict__ key, uint64_t n, uint32_t nsl, uint32_t want,
                                      uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = slice_of(key[i], nsl) == want ? 1u : 0u;
}
// flag = 1 for the rows of the slices [0, m) of nsl (all copies of a key share a slice)
__global__ void tbl_slice_below_flag_kernel(const uint32_t* __restrict__ key, uint64_t n, uint32_t nsl, uint32_t m,
                                            uint32_t* __restrict__ flag, uint32_t* __restrict__ nflag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t f = slice_of(key[i], nsl) < m ? 1u : 0u;
  flag[i] = f;
  nflag[i] = 1u - f;
}
__global__ void tbl_slice_scatter_kernel(const uint32_t* __restrict__ key, const uint32_t* __restrict__ sid,
                                         const uint32_t* __restrict__ off, const uint32_t* __restrict__ flag,
                                         const uint32_t* __restrict__ pos, uint64_t n, uint32_t* __restrict__ ok,
                                         uint32_t* __restrict__ os, uint32_t* __restrict__ oo) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const uint32_t p = pos[i];
  ok[p] = key[i];
  os[p] = sid[i];
  oo[p] = off[i];
}

// ---------------------------------------------------------------------------------------- table API
// the bulk build (sorted runs + k-way merge, further down)
static void reserve_wait(shz_table* t, int which);
static int32_t seal_staged(shz_table* t, const uint64_t* block_rows, uint32_t n_blocks, bool* packed);
static int32_t flush_runs(shz_table* t, bool final);
static uint64_t runs_end(const shz_table* t);
static int32_t staged_minmax(shz_table* t, uint64_t lo, uint64_t n, uint32_t out[3]);
static void reserve_cancel(shz_table* t);

// columns that were not carved from the slab go back to the device
static void free_cols(bool slab, uint32_t* key, uint32_t* sid, uint32_t* off) {
  if (slab) return;
  void* ps[] = {key, sid, off};
  for (void* p : ps)
    if (p) (void)hipFree(p);
}

extern "C" int32_t shz_table_create(shz_ctx* ctx, shz_table** out) {
  if (!ctx || !out) return SHZ_E_INVALID;
  shz_table* t = new shz_table();
  t->ctx = ctx;
  *out = t;
  return SHZ_OK;
}

extern "C" int32_t shz_table_destroy(shz_table* t) {
  if (!t) return SHZ_E_INVALID;
  (void)hipSetDevice(t->ctx->device);
  (void)hipStreamSynchronize(t->ctx->stream);
  if (t->gx_stream) (void)hipStreamSynchronize(t->gx_stream);
  if (t->d_kw_err) (void)hipFree(t->d_kw_err);
  if (t->gx_ev) (void)hipEventDestroy(t->gx_ev);
  reserve_cancel(t);
  free_cols(t->act_slab, t->key, t->sid, t->off);
  if (t->stage_reserved) {
    uint32_t* st[3] = {t->skey, t->ssid, t->soff};
    for (int i = 0; i < 3; ++i) shz_block_free(t->ctx, st[i], t->st_bytes[i]);
  } else {
    void* st[] = {t->skey, t->ssid, t->soff};
    for (void* p : st)
      if (p) (void)hipFree(p);
  }
  if (t->bucket) (void)hipFree(t->bucket);
  shz_block_free(t->ctx, t->rbuf, t->rbuf_bytes);
  shz_block_free(t->ctx, t->slab, t->slab_bytes);
  for (shz_seg& g : t->done) {
    free_cols(g.slab, g.key, g.sid, g.off);
    if (g.bucket) (void)hipFree(g.bucket);
  }
  delete t;
  return SHZ_OK;
}

static int32_t stage_reserve(shz_table* t, uint64_t extra) {
  shz_ctx* ctx = t->ctx;
  const uint64_t need = t->ns + extra;
  if (t->job && !t->skey) reserve_wait(t, 1 /* RJ_STAGE */);
  if (need <= t->scap) return SHZ_OK;
  uint64_t cap = std::max<uint64_t>(need, t->scap * 2);
  cap = std::max<uint64_t>(cap, 1024);
  ph_clock pc(t);
  struct lap_on_exit { ph_clock& c; ~lap_on_exit() { c.lap(PH_STAGE_ALLOC); } } loe{pc};
  uint32_t* np[3];
  for (int i = 0; i < 3; ++i) {
    hipError_t e = hipMalloc(&np[i], cap * 4);
    if (e != hipSuccess) SHZ_FAIL(ctx, SHZ_E_NOMEM, "table staging: hipMalloc(%llu) failed", (unsigned long long)(cap * 4));
  }
  uint32_t** old[3] = {&t->skey, &t->ssid, &t->soff};
  for (int i = 0; i < 3; ++i) {
    if (t->ns) SHZ_HIP(ctx, shz_memcpy(ctx, np[i], *old[i], t->ns * 4, hipMemcpyDeviceToDevice));
  }
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 3; ++i) {
    if (*old[i]) SHZ_HIP(ctx, hipFree(*old[i]));
    *old[i] = np[i];
  }
  t->scap = cap;
  t->stage_reserved = false;   // plain allocations from here on
  return SHZ_OK;
}

extern "C" int32_t shz_table_insert(shz_table* t, const uint32_t* key32, const uint32_t* sid, const uint32_t* off,
                                    uint64_t n, uint32_t flags) {
  if (!t) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  if (n == 0) return SHZ_OK;
  if (!key32 || !sid || !off) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_table_insert: NULL column");
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  SHZ_TRY(stage_reserve(t, n));
  const hipMemcpyKind kd = (flags & SHZ_IN_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  SHZ_HIP(ctx, shz_memcpy(ctx, t->skey + t->ns, key32, n * 4, kd));
  SHZ_HIP(ctx, shz_memcpy(ctx, t->ssid + t->ns, sid, n * 4, kd));
  SHZ_HIP(ctx, shz_memcpy(ctx, t->soff + t->ns, off, n * 4, kd));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  t->ns += n;
  return SHZ_OK;
}

extern "C" int32_t shz_table_insert_clips(shz_table* t, const uint32_t* key32, const uint32_t* t1,
                                          const uint64_t* hash_off, uint32_t n_clips, uint32_t sid0, uint32_t flags) {
  if (!t) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  if (n_clips == 0) return SHZ_OK;
  if (!hash_off) SHZ_FAIL(ctx, SHZ_E_INVALID, "hash_off is NULL");
  const uint64_t n = hash_off[n_clips] - hash_off[0];
  if (n == 0) return SHZ_OK;
  if (!key32 || !t1) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_table_insert_clips: NULL column");
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  SHZ_TRY(stage_reserve(t, n));
  ph_clock pc(t);
  void* d_ho;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, (uint64_t)(n_clips + 1) * 8, &d_ho));
  SHZ_HIP(ctx, shz_memcpy(ctx, d_ho, hash_off, (uint64_t)(n_clips + 1) * 8, hipMemcpyHostToDevice));
  const uint32_t *dk = key32, *dt = t1;
  if (!(flags & SHZ_IN_DEVICE)) {
    void *a, *b;
    const uint64_t hi = hash_off[n_clips];
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, hi * 4, &a));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M2, hi * 4, &b));
    SHZ_HIP(ctx, shz_memcpy(ctx, a, key32, hi * 4, hipMemcpyHostToDevice));
    SHZ_HIP(ctx, shz_memcpy(ctx, b, t1, hi * 4, hipMemcpyHostToDevice));
    dk = (const uint32_t*)a;
    dt = (const uint32_t*)b;
  }
  hipLaunchKernelGGL(tbl_expand_clips_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dk, dt,
                     (const uint64_t*)d_ho, n_clips, sid0, n, t->skey + t->ns, t->ssid + t->ns, t->soff + t->ns);
  SHZ_HIP(ctx, hipGetLastError());
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  pc.lap(PH_INSERT);
  t->ns += n;
  return SHZ_OK;
}

// merge of two sorted u64 runs (merge path).  A workgroup produces a tile of MERGE_TILE outputs: the tile's share of
// `a` and `b` (found by one binary search per tile edge) is staged in LDS with coalesced loads, every thread merges
// MERGE_PT outputs there, and the tile leaves with coalesced stores.  Ties take the element of `a` first.  Used by
// finalize when a sorted active segment absorbs a (much smaller) sorted batch of new rows: one pass over the data
// instead of a radix sort of everything.
#define MERGE_PT 8
#define MERGE_TILE (256 * MERGE_PT)
__device__ __forceinline__ uint64_t merge_split(const uint64_t* __restrict__ a, uint64_t na, const uint64_t* __restrict__ b,
                                                uint64_t nb, uint64_t diag) {  // elements of a among the first diag outputs
  uint64_t lo = diag > nb ? diag - nb : 0, hi = diag < na ? diag : na;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (a[mid] <= b[diag - 1 - mid]) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__global__ __launch_bounds__(256) void tbl_merge_kernel(const uint64_t* __restrict__ a, uint64_t na,
                                                        const uint64_t* __restrict__ b, uint64_t nb,
                                                        uint64_t* __restrict__ out) {
  __shared__ uint64_t sin[MERGE_TILE];   // the tile's elements of a, then those of b
  __shared__ uint64_t sout[MERGE_TILE];
  __shared__ uint64_t edge[2];
  const uint64_t total = na + nb;
  const uint64_t d0 = (uint64_t)blockIdx.x * MERGE_TILE;
  if (d0 >= total) return;
  const uint64_t d1 = d0 + MERGE_TILE < total ? d0 + MERGE_TILE : total;
  if (threadIdx.x < 2) edge[threadIdx.x] = merge_split(a, na, b, nb, threadIdx.x ? d1 : d0);
  __syncthreads();
  const uint64_t a0 = edge[0], a1 = edge[1], b0 = d0 - a0, b1 = d1 - a1;
  const uint32_t ca = (uint32_t)(a1 - a0), cb = (uint32_t)(b1 - b0), n = ca + cb;
  for (uint32_t i = threadIdx.x; i < n; i += 256) sin[i] = i < ca ? a[a0 + i] : b[b0 + (i - ca)];
  __syncthreads();
  const uint64_t* la = sin;
  const uint64_t* lb = sin + ca;
  const uint32_t diag = threadIdx.x * MERGE_PT;
  if (diag < n) {
    uint32_t i = (uint32_t)merge_split(la, ca, lb, cb, diag), j = diag - i;
    const uint32_t end = diag + MERGE_PT < n ? diag + MERGE_PT : n;
    for (uint32_t o = diag; o < end; ++o) {
      const bool take_a = j >= cb || (i < ca && la[i] <= lb[j]);
      sout[o] = take_a ? la[i++] : lb[j++];
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += 256) out[d0 + i] = sout[i];
}

// merge `ns` staged rows (columns skey/ssid/soff, not freed here) into the active segment
static int32_t finalize_active(shz_table* t, const uint32_t* skey, const uint32_t* ssid, const uint32_t* soff, uint64_t ns,
                               uint32_t sid_lo, uint32_t sid_hi) {
  shz_ctx* ctx = t->ctx;
  const uint64_t total = t->n + ns;
  if (total >= (1ull << 32)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "segment limited to < 2^32 rows (have %llu)", (unsigned long long)total);
  if (t->n == 0) { t->act_sid_lo = 0xFFFFFFFFu; t->act_sid_hi = 0; }
  void *k0, *k1, *v0 = nullptr, *v1 = nullptr, *mx, *fl, *ps, *tot;
  ph_clock pc(t);
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC0, 64, &mx));
  SHZ_HIP(ctx, hipMemsetAsync(mx, 0, 64, ctx->stream));
  const unsigned gmax = 2048;
  if (t->n)
    hipLaunchKernelGGL(tbl_max_kernel, dim3((unsigned)std::min<uint64_t>((t->n + 255) / 256, gmax)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)t->sid, (const uint32_t*)t->off, t->n, (uint32_t*)mx);
  if (ns)
    hipLaunchKernelGGL(tbl_max_kernel, dim3((unsigned)std::min<uint64_t>((ns + 255) / 256, gmax)), dim3(256), 0, ctx->stream,
                       ssid, soff, ns, (uint32_t*)mx);
  SHZ_HIP(ctx, hipGetLastError());
  uint32_t maxes[2];
  SHZ_HIP(ctx, shz_memcpy(ctx, maxes, mx, 8, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  t->max_sid = std::max(t->max_sid, maxes[0]);   // table-wide maxima size the packed vote key
  t->max_off = std::max(t->max_off, maxes[1]);
  const int sb = bits_for(maxes[0]), ob = bits_for(maxes[1]);
  const bool one_key = sb + ob <= 32;
  pc.lap(PH_MAXES);
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_A, total * 8, &k0));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_B, total * 8, &k1));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, total * 4, &fl));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, total * 4, &ps));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, 64, &tot));
  pc.lap(PH_COL_ALLOC);   // growth of the sort workspace counts as allocation
  int sel = 0;
  uint64_t *ka = (uint64_t*)k0, *kb = (uint64_t*)k1;
  void *va = nullptr, *vb = nullptr;
  if (one_key) {
    if (t->n)
      hipLaunchKernelGGL(tbl_compose1_kernel, dim3((unsigned)std::min<uint64_t>((t->n + 255) / 256, 8192)), dim3(256), 0,
                         ctx->stream, (const uint32_t*)t->key, (const uint32_t*)t->sid, (const uint32_t*)t->off, t->n,
                         (uint64_t)0, sb, ob, ka);
    if (ns)
      hipLaunchKernelGGL(tbl_compose1_kernel, dim3((unsigned)std::min<uint64_t>((ns + 255) / 256, 8192)), dim3(256), 0,
                         ctx->stream, skey, ssid, soff, ns, t->n, sb, ob, ka);
    SHZ_HIP(ctx, hipGetLastError());
    if (t->n && ns) {
      // the active rows are already in order (the packing is monotone in (key, sid, off)): sort only the new rows and
      // merge the two runs -- one pass over the segment instead of a radix sort of all of it
      SHZ_TRY(shz_sort_u64(ctx, ka + t->n, kb + t->n, nullptr, nullptr, 0, ns, 0, 32 + sb + ob, &sel));
      if (sel) SHZ_HIP(ctx, shz_memcpy(ctx, ka + t->n, kb + t->n, ns * 8, hipMemcpyDeviceToDevice));
      pc.lap(PH_SORT);
      hipLaunchKernelGGL(tbl_merge_kernel, dim3((unsigned)((total + MERGE_TILE - 1) / MERGE_TILE)), dim3(256), 0, ctx->stream,
                         (const uint64_t*)ka, t->n, (const uint64_t*)(ka + t->n), ns, kb);
      SHZ_HIP(ctx, hipGetLastError());
      std::swap(ka, kb);
      pc.lap(PH_MERGE);
    } else {
      SHZ_TRY(shz_sort_u64(ctx, ka, kb, nullptr, nullptr, 0, total, 0, 32 + sb + ob, &sel));
      if (sel) std::swap(ka, kb);
      pc.lap(PH_SORT);
    }
    hipLaunchKernelGGL(tbl_uniq1_flag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint64_t*)ka, total, (uint32_t*)fl);
  } else {
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_C, total * 8, &v0));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_D, total * 8, &v1));
    va = v0;
    vb = v1;
    if (t->n)
      hipLaunchKernelGGL(tbl_compose_kernel, dim3((unsigned)std::min<uint64_t>((t->n + 255) / 256, 4096)), dim3(256), 0,
                         ctx->stream, t->key, t->sid, t->off, t->n, (uint64_t)0, ka, (uint32_t*)va, (uint32_t*)mx);
    if (ns)
      hipLaunchKernelGGL(tbl_compose_kernel, dim3((unsigned)std::min<uint64_t>((ns + 255) / 256, 4096)), dim3(256), 0,
                         ctx->stream, skey, ssid, soff, ns, t->n, ka, (uint32_t*)va, (uint32_t*)mx);
    SHZ_HIP(ctx, hipGetLastError());
    // 1) stable sort by (sid, off) carrying the key, 2) stable sort by key carrying (sid, off)
    SHZ_TRY(shz_sort_u64(ctx, ka, kb, va, vb, 4, total, 0, ob, &sel));
    if (sel) { std::swap(ka, kb); std::swap(va, vb); }
    SHZ_TRY(shz_sort_u64(ctx, ka, kb, va, vb, 4, total, 32, 32 + sb, &sel));
    if (sel) { std::swap(ka, kb); std::swap(va, vb); }
    hipLaunchKernelGGL(tbl_swap_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, (const uint64_t*)ka,
                       (const uint32_t*)va, total, kb, (uint64_t*)vb);
    SHZ_HIP(ctx, hipGetLastError());
    std::swap(ka, kb);
    std::swap(va, vb);
    SHZ_TRY(shz_sort_u64(ctx, ka, kb, va, vb, 8, total, 0, 32, &sel));
    if (sel) { std::swap(ka, kb); std::swap(va, vb); }
    pc.lap(PH_SORT);
    hipLaunchKernelGGL(tbl_uniq_flag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint64_t*)ka, (const uint64_t*)va, total, (uint32_t*)fl);
  }
  SHZ_HIP(ctx, hipGetLastError());
  SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)fl, (uint32_t*)ps, total, (uint64_t*)tot));
  uint64_t nu = 0;
  SHZ_HIP(ctx, shz_memcpy(ctx, &nu, tot, 8, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  pc.lap(PH_UNIQ);
  // The old columns are dead once composed.  If they can hold the merged rows they are written in place (no
  // hipFree / hipMalloc of gigabytes per finalize: that, not the kernels, dominated incremental ingest); otherwise they
  // are freed before the new ones are allocated (peak memory), with 1/8 headroom for the next batches.
  if (nu > t->cap) {
    // Grow: new columns are allocated while the old ones still exist, so that running out of memory leaves the table
    // as it was (the merged rows are in the sort workspace, nothing of the table has been touched yet).  Only if that
    // fails are the old columns given up first (lower peak); a failure after that has lost rows and marks the table.
    const uint64_t want = std::min<uint64_t>(nu + nu / 8 + 1024, std::max<uint64_t>(nu, t->seg_limit) + 1024);
    dev_cols fresh;
    if (!fresh.alloc(want)) {
      for (int i = 0; i < 3; ++i)
        if (uint32_t* q = fresh.take(i)) (void)hipFree(q);
      (void)hipGetLastError();
      free_cols(t->act_slab, t->key, t->sid, t->off);
      t->key = t->sid = t->off = nullptr;
      t->cap = 0;
      t->n = 0;
      if (!fresh.alloc(want)) {
        t->broken = true;
        SHZ_FAIL(ctx, SHZ_E_NOMEM, "table: hipMalloc of %llu rows failed after the active segment was released; "
                                   "the table lost rows and refuses further use", (unsigned long long)want);
      }
    } else {
      free_cols(t->act_slab, t->key, t->sid, t->off);
    }
    t->key = fresh.take(0);
    t->sid = fresh.take(1);
    t->off = fresh.take(2);
    t->cap = want;
    t->act_slab = false;
  }
  pc.lap(PH_COL_ALLOC);
  t->n = 0;
  uint32_t *nk = t->key, *nsid = t->sid, *noff = t->off;
  if (one_key)
    hipLaunchKernelGGL(tbl_compact1_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint64_t*)ka, (const uint32_t*)fl, (const uint32_t*)ps, total, sb, ob, nk, nsid, noff);
  else
    hipLaunchKernelGGL(tbl_compact_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint64_t*)ka, (const uint64_t*)va, (const uint32_t*)fl, (const uint32_t*)ps, total, nk, nsid,
                       noff);
  SHZ_HIP(ctx, hipGetLastError());
  uint32_t last_key = 0;
  SHZ_HIP(ctx, shz_memcpy(ctx, &last_key, nk + (nu - 1), 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, shz_memcpy(ctx, &t->act_key_lo, nk, 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  pc.lap(PH_COMPACT);
  t->n = nu;
  t->act_sid_lo = std::min(t->act_sid_lo, sid_lo);
  t->act_sid_hi = std::max(t->act_sid_hi, sid_hi);
  t->nbuckets = (uint64_t)(last_key >> 8) + 1;
  if (t->nbuckets + 1 > t->bcap) {
    if (t->bucket) SHZ_HIP(ctx, hipFree(t->bucket));
    t->bucket = nullptr;
    t->bcap = 0;
    SHZ_HIP(ctx, hipMalloc(&t->bucket, (t->nbuckets + 1) * 4));
    t->bcap = t->nbuckets + 1;
  }
  hipLaunchKernelGGL(tbl_bucket_kernel, dim3((unsigned)((t->nbuckets + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const uint32_t*)t->key, (uint32_t)t->n, t->nbuckets, t->bucket);
  SHZ_HIP(ctx, hipGetLastError());
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  pc.lap(PH_BUCKET);
  return SHZ_OK;
}


// ---- rows leaving the table: ON DELETE CASCADE of a song's fingerprints (mysql_database.py:57-58), and INSERT IGNORE
// against rows that already sit in a frozen segment (UNIQUE(song_id, offset, hash), :54-55, 62-68) ----
__global__ void tbl_sid_keep_kernel(const uint32_t* __restrict__ sid, uint64_t n, const uint32_t* __restrict__ bitmap,
                                    uint32_t nbits, uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = sid[i];
  flag[i] = (s < nbits && ((bitmap[s >> 5] >> (s & 31)) & 1u)) ? 0u : 1u;
}
__global__ void tbl_gather_u32_kernel(const uint32_t* __restrict__ in, const uint32_t* __restrict__ flag,
                                      const uint32_t* __restrict__ pos, uint64_t n, uint32_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) out[pos[i]] = in[i];
}
__global__ void tbl_ones_kernel(uint32_t* __restrict__ flag, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = 1u;
}
// flag[i] = 0 where staged row i is a row of the (sorted, unique) segment g
__global__ void tbl_exists_kernel(const uint32_t* __restrict__ skey, const uint32_t* __restrict__ ssid,
                                  const uint32_t* __restrict__ soff, uint64_t ns, shz_seg_dev g,
                                  uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns || !flag[i]) return;
  const uint32_t k = skey[i], sd = ssid[i], of = soff[i];
  const uint64_t b = k >> 8;
  if (b >= g.nbuckets) return;
  uint32_t lo = g.bucket[b], hi = g.bucket[b + 1];
  while (lo < hi) {  // first row >= (k, sd, of)
    const uint32_t mid = lo + ((hi - lo) >> 1);
    const uint32_t mk = g.key[mid];
    bool less = mk < k;
    if (mk == k) {
      const uint32_t ms = g.sid[mid];
      less = ms < sd || (ms == sd && g.off[mid] < of);
    }
    if (less) lo = mid + 1; else hi = mid;
  }
  if (lo < g.n && g.key[lo] == k && g.sid[lo] == sd && g.off[lo] == of) flag[i] = 0u;
}

// keep the flagged rows of three columns, in order, in place (through a scratch column); returns the kept count
static int32_t compact_cols(shz_ctx* ctx, uint32_t* key, uint32_t* sid, uint32_t* off, uint64_t n, const uint32_t* fl,
                            uint64_t* kept_out) {
  void *ps, *tot, *tmp;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, n * 4, &ps));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, 64, &tot));
  SHZ_TRY(shz_scan_u32(ctx, fl, (uint32_t*)ps, n, (uint64_t*)tot));
  uint64_t kept = 0;
  SHZ_HIP(ctx, shz_memcpy(ctx, &kept, tot, 8, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *kept_out = kept;
  if (kept == n || n == 0) return SHZ_OK;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_A, std::max<uint64_t>(kept, 1) * 4, &tmp));
  uint32_t* cols[3] = {key, sid, off};
  for (uint32_t* c : cols) {
    hipLaunchKernelGGL(tbl_gather_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)c, fl, (const uint32_t*)ps, n, (uint32_t*)tmp);
    SHZ_HIP(ctx, hipGetLastError());
    if (kept) SHZ_HIP(ctx, hipMemcpyAsync(c, tmp, kept * 4, hipMemcpyDeviceToDevice, ctx->stream));
  }
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}

// bucket index of a sorted segment whose rows changed
static int32_t rebuild_buckets(shz_ctx* ctx, uint32_t* key, uint64_t n, uint32_t** bucket, uint64_t* nbuckets, uint64_t* bcap,
                               uint32_t* key_lo) {
  if (n == 0) { *nbuckets = 0; *key_lo = 0; return SHZ_OK; }
  uint32_t last_key = 0;
  SHZ_HIP(ctx, shz_memcpy(ctx, &last_key, key + (n - 1), 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, shz_memcpy(ctx, key_lo, key, 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint64_t nb = (uint64_t)(last_key >> 8) + 1;
  if (nb + 1 > *bcap) {
    if (*bucket) SHZ_HIP(ctx, hipFree(*bucket));
    *bucket = nullptr;
    *bcap = 0;
    SHZ_HIP(ctx, hipMalloc(bucket, (nb + 1) * 4));
    *bcap = nb + 1;
  }
  *nbuckets = nb;
  hipLaunchKernelGGL(tbl_bucket_kernel, dim3((unsigned)((nb + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const uint32_t*)key, (uint32_t)n, nb, *bucket);
  SHZ_HIP(ctx, hipGetLastError());
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}

extern "C" int32_t shz_table_delete_songs(shz_table* t, const uint32_t* sids, uint64_t n_sids, uint64_t* rows_deleted) {
  if (!t) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  if (rows_deleted) *rows_deleted = 0;
  if (t->broken) SHZ_FAIL(ctx, SHZ_E_STATE, "table lost rows in a failed finalize");
  if (n_sids == 0) return SHZ_OK;
  if (!sids) SHZ_FAIL(ctx, SHZ_E_INVALID, "sids is NULL");
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  if (!t->runs.empty()) SHZ_TRY(flush_runs(t, true));   // sealed runs become segments first: rows leave segments and staged rows
  uint32_t mx = 0;
  for (uint64_t i = 0; i < n_sids; ++i) mx = std::max(mx, sids[i]);
  const uint32_t nbits = mx + 1;
  std::vector<uint32_t> bm(((uint64_t)nbits + 31) / 32, 0u);
  for (uint64_t i = 0; i < n_sids; ++i) bm[sids[i] >> 5] |= 1u << (sids[i] & 31);
  void* d_bm;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M2, bm.size() * 4, &d_bm));
  SHZ_HIP(ctx, shz_memcpy(ctx, d_bm, bm.data(), bm.size() * 4, hipMemcpyHostToDevice));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t gone = 0;
  auto purge = [&](uint32_t* key, uint32_t* sid, uint32_t* off, uint64_t n, uint64_t* kept) -> int32_t {
    void* fl;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, std::max<uint64_t>(n, 1) * 4, &fl));
    hipLaunchKernelGGL(tbl_sid_keep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)sid, n, (const uint32_t*)d_bm, nbits, (uint32_t*)fl);
    SHZ_HIP(ctx, hipGetLastError());
    return compact_cols(ctx, key, sid, off, n, (const uint32_t*)fl, kept);
  };
  for (shz_seg& g : t->done) {
    uint64_t kept = g.n, bcap = g.nbuckets + 1;
    SHZ_TRY(purge(g.key, g.sid, g.off, g.n, &kept));
    if (kept != g.n) {
      gone += g.n - kept;
      g.n = kept;
      SHZ_TRY(rebuild_buckets(ctx, g.key, g.n, &g.bucket, &g.nbuckets, &bcap, &g.key_lo));
    }
  }
  // frozen segments that became empty disappear
  for (size_t i = t->done.size(); i-- > 0;)
    if (t->done[i].n == 0) {
      free_cols(t->done[i].slab, t->done[i].key, t->done[i].sid, t->done[i].off);
      if (t->done[i].bucket) (void)hipFree(t->done[i].bucket);
      t->done.erase(t->done.begin() + (long)i);
    }
  if (t->n) {
    uint64_t kept = t->n;
    SHZ_TRY(purge(t->key, t->sid, t->off, t->n, &kept));
    if (kept != t->n) {
      gone += t->n - kept;
      t->n = kept;
      SHZ_TRY(rebuild_buckets(ctx, t->key, t->n, &t->bucket, &t->nbuckets, &t->bcap, &t->act_key_lo));
    }
  }
  if (t->ns) {
    uint64_t kept = t->ns;
    SHZ_TRY(purge(t->skey, t->ssid, t->soff, t->ns, &kept));
    gone += t->ns - kept;
    t->ns = kept;
  }
  if (rows_deleted) *rows_deleted = gone;
  return SHZ_OK;
}

extern "C" int32_t shz_table_clear(shz_table* t) {
  if (!t) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (t->gx_stream) SHZ_HIP(ctx, hipStreamSynchronize(t->gx_stream));
  for (shz_seg& g : t->done) {
    free_cols(g.slab, g.key, g.sid, g.off);
    if (g.bucket) (void)hipFree(g.bucket);
  }
  t->done.clear();
  t->runs.clear();
  t->run_sb = t->run_ob = 0;
  t->rows_cut = t->gx_recv_bytes = t->gx_rounds = 0;
  t->gx_wait_s = t->gx_xfer_s = 0.0;
  t->hold_runs = t->hold_reserved;
  if (t->act_slab) { t->key = t->sid = t->off = nullptr; t->cap = 0; t->act_slab = false; }   // the slab starts over: nothing is carved
  t->slab_used = 0;
  t->act_sid_lo = 0xFFFFFFFFu;
  t->act_sid_hi = 0;
  t->act_key_lo = 0;
  t->n = 0;        // the active and staging columns keep their allocations for the rows to come
  t->nbuckets = 0;
  t->ns = 0;
  t->max_sid = t->max_off = 0;
  t->broken = false;
  return SHZ_OK;
}

// INSERT IGNORE across segments: staged rows that already sit in a frozen segment are dropped before they are merged
// (a frozen segment whose song ids do not meet [sid_lo, sid_hi] of the staged rows cannot hold one of them: song_id is
// part of the UNIQUE key, and ingest hands out fresh ids -- the search over every segment per finalize was what made the
// 1M-song build's finalize 4x slower in round 2)
static int32_t drop_staged_duplicates_of_frozen(shz_table* t, uint32_t sid_lo, uint32_t sid_hi) {
  shz_ctx* ctx = t->ctx;
  if (t->done.empty() || t->ns == 0) return SHZ_OK;
  bool any = false;
  for (const shz_seg& g : t->done) any |= g.n && ranges_overlap(g.sid_lo, g.sid_hi, sid_lo, sid_hi);
  if (!any) return SHZ_OK;
  void* fl;
  ph_clock pc(t);
  struct lap_on_exit { ph_clock& c; ~lap_on_exit() { c.lap(PH_DEDUP_FROZEN); } } loe{pc};
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, t->ns * 4, &fl));
  hipLaunchKernelGGL(tbl_ones_kernel, dim3((unsigned)((t->ns + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)fl, t->ns);
  for (const shz_seg& g : t->done) {
    if (!g.n || !ranges_overlap(g.sid_lo, g.sid_hi, sid_lo, sid_hi)) continue;
    const shz_seg_dev gd = seg_dev_of(g);
    hipLaunchKernelGGL(tbl_exists_kernel, dim3((unsigned)((t->ns + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)t->skey, (const uint32_t*)t->ssid, (const uint32_t*)t->soff, t->ns, gd, (uint32_t*)fl);
  }
  SHZ_HIP(ctx, hipGetLastError());
  uint64_t kept = t->ns;
  SHZ_TRY(compact_cols(ctx, t->skey, t->ssid, t->soff, t->ns, (const uint32_t*)fl, &kept));
  t->ns = kept;
  return SHZ_OK;
}

static void freeze_active(shz_table* t) {
  if (!t->n) return;
  t->done.push_back(shz_seg{t->key, t->sid, t->off, t->bucket, t->n, t->nbuckets, t->act_sid_lo, t->act_sid_hi, t->act_slab, t->act_key_lo});
  t->key = t->sid = t->off = t->bucket = nullptr;
  t->n = t->nbuckets = 0;
  t->cap = t->bcap = 0;
  t->act_slab = false;
  t->act_sid_lo = 0xFFFFFFFFu;
  t->act_sid_hi = 0;
  t->act_key_lo = 0;
}

extern "C" int32_t shz_table_finalize(shz_table* t) {
  if (!t) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  if (t->broken) SHZ_FAIL(ctx, SHZ_E_STATE, "table lost rows in a failed finalize");
  static const bool no_runs = [] { const char* e = getenv("SHZ_BUILD_RUNS"); return e && atoi(e) == 0; }();
  if (t->n == 0 && (t->ns || !t->runs.empty()) && (!no_runs || !t->runs.empty())) {
    // the bulk path: an empty active segment takes the rows as sorted runs and one k-way merge
    bool packed = false;
    SHZ_TRY(seal_staged(t, nullptr, 0, &packed));
    if (packed) {
      SHZ_TRY(flush_runs(t, true));
      if (!t->bucket && t->n == 0 && t->done.empty()) {  // every row was a duplicate of nothing: still an empty table
        SHZ_HIP(ctx, hipMalloc(&t->bucket, 2 * 4));
        SHZ_HIP(ctx, hipMemsetAsync(t->bucket, 0, 8, ctx->stream));
        t->nbuckets = 1;
      }
      return SHZ_OK;
    }
    if (!t->runs.empty()) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "song ids and offsets outgrew 32 bits while sealed runs were waiting");
  }
  uint32_t smm[3] = {0u, 0u, 0xFFFFFFFFu};   // largest song id / offset, smallest song id of the staged rows
  if (t->ns) SHZ_TRY(staged_minmax(t, 0, t->ns, smm));
  const uint32_t st_lo = smm[2], st_hi = smm[0];
  if (t->ns && t->n && t->n + t->ns > t->seg_limit) {
    // The staged rows do not fit the active segment.  Before it is frozen it is topped up with the slices (by key) of
    // the staged rows that still fit: segments then hold ~seg_limit rows instead of whatever multiple of the ingest
    // batch fell below it (1.13e9-row batches against 2^31 left every segment half empty: 11 segments for 1.15e10 rows
    // instead of 6 -- and every query hash is looked up in every segment).
    const uint32_t S = 64;
    const uint32_t m = (uint32_t)std::min<uint64_t>(S - 1, (t->seg_limit - t->n) * S / t->ns);
    if (m >= S / 16) {
      SHZ_TRY(drop_staged_duplicates_of_frozen(t, st_lo, st_hi));
      if (t->ns) {
        ph_clock pc(t);
        void *fa, *fb, *pa, *pb, *tot, *ak, *as, *ao, *bk, *bs, *bo;
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, t->ns * 4, &fa));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, t->ns * 4, &pa));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_C, t->ns * 4, &fb));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_D, t->ns * 4, &pb));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, 64, &tot));
        hipLaunchKernelGGL(tbl_slice_below_flag_kernel, dim3((unsigned)((t->ns + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)t->skey, t->ns, S, m, (uint32_t*)fa, (uint32_t*)fb);
        SHZ_HIP(ctx, hipGetLastError());
        SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)fa, (uint32_t*)pa, t->ns, (uint64_t*)tot));
        SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)fb, (uint32_t*)pb, t->ns, nullptr));
        uint64_t na = 0;
        SHZ_HIP(ctx, shz_memcpy(ctx, &na, tot, 8, hipMemcpyDeviceToHost));
        SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const uint64_t nb = t->ns - na;
        if (na) {
          SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M2, na * 4, &ak));
          SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M3, na * 4, &as));
          SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M4, na * 4, &ao));
          SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M5, (nb + 1) * 4, &bk));
          SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M6, (nb + 1) * 4, &bs));
          SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M7, (nb + 1) * 4, &bo));
          const unsigned g = (unsigned)((t->ns + 255) / 256);
          hipLaunchKernelGGL(tbl_slice_scatter_kernel, dim3(g), dim3(256), 0, ctx->stream, (const uint32_t*)t->skey,
                             (const uint32_t*)t->ssid, (const uint32_t*)t->soff, (const uint32_t*)fa, (const uint32_t*)pa, t->ns,
                             (uint32_t*)ak, (uint32_t*)as, (uint32_t*)ao);
          hipLaunchKernelGGL(tbl_slice_scatter_kernel, dim3(g), dim3(256), 0, ctx->stream, (const uint32_t*)t->skey,
                             (const uint32_t*)t->ssid, (const uint32_t*)t->soff, (const uint32_t*)fb, (const uint32_t*)pb, t->ns,
                             (uint32_t*)bk, (uint32_t*)bs, (uint32_t*)bo);
          SHZ_HIP(ctx, hipGetLastError());
          if (nb) {   // the rest stays staged
            SHZ_HIP(ctx, shz_memcpy(ctx, t->skey, bk, nb * 4, hipMemcpyDeviceToDevice));
            SHZ_HIP(ctx, shz_memcpy(ctx, t->ssid, bs, nb * 4, hipMemcpyDeviceToDevice));
            SHZ_HIP(ctx, shz_memcpy(ctx, t->soff, bo, nb * 4, hipMemcpyDeviceToDevice));
          }
          t->ns = nb;
          pc.lap(PH_TOPUP);
          SHZ_TRY(finalize_active(t, (const uint32_t*)ak, (const uint32_t*)as, (const uint32_t*)ao, na, st_lo, st_hi));
        }
      }
    }
    freeze_active(t);   // what is still staged starts new segment(s)
  }
  SHZ_TRY(drop_staged_duplicates_of_frozen(t, st_lo, st_hi));    // UNIQUE(song_id, offset, hash) across segments
  if (t->ns == 0) {
    if (!t->bucket && t->n == 0 && t->done.empty()) {  // empty table: one empty bucket
      SHZ_HIP(ctx, hipMalloc(&t->bucket, 2 * 4));
      SHZ_HIP(ctx, hipMemsetAsync(t->bucket, 0, 8, ctx->stream));
      t->nbuckets = 1;
    }
    return SHZ_OK;
  }
  // staged rows go into the active segment if they fit; otherwise the active segment is frozen and the
  // staged rows are cut into slices BY KEY (all copies of a row land in the same slice, so duplicates
  // inside one batch are still removed), one new segment per slice
  if (t->n + t->ns <= t->seg_limit) {
    SHZ_TRY(finalize_active(t, t->skey, t->ssid, t->soff, t->ns, st_lo, st_hi));
  } else {   // t->n == 0 here: the active segment was frozen above
    const uint32_t nsl = (uint32_t)((t->ns + t->seg_limit - 1) / t->seg_limit) + (t->ns > t->seg_limit ? 1 : 0);
    if (t->done.size() + nsl > SHZ_MAX_SEGS) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "more than %d table segments", SHZ_MAX_SEGS);
    for (uint32_t sl = 0; sl < nsl; ++sl) {
      if (nsl == 1) {
        SHZ_TRY(finalize_active(t, t->skey, t->ssid, t->soff, t->ns, st_lo, st_hi));
      } else {
        void *fl, *ps, *tot, *ck, *cs, *co;
        ph_clock pc(t);
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, t->ns * 4, &fl));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, t->ns * 4, &ps));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, 64, &tot));
        hipLaunchKernelGGL(tbl_slice_flag_kernel, dim3((unsigned)((t->ns + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)t->skey, t->ns, nsl, sl, (uint32_t*)fl);
        SHZ_HIP(ctx, hipGetLastError());
        SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)fl, (uint32_t*)ps, t->ns, (uint64_t*)tot));
        uint64_t cnt = 0;
        SHZ_HIP(ctx, shz_memcpy(ctx, &cnt, tot, 8, hipMemcpyDeviceToHost));
        SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (cnt == 0) continue;
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M2, cnt * 4, &ck));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M3, cnt * 4, &cs));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M4, cnt * 4, &co));
        hipLaunchKernelGGL(tbl_slice_scatter_kernel, dim3((unsigned)((t->ns + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)t->skey, (const uint32_t*)t->ssid, (const uint32_t*)t->soff,
                           (const uint32_t*)fl, (const uint32_t*)ps, t->ns, (uint32_t*)ck, (uint32_t*)cs, (uint32_t*)co);
        SHZ_HIP(ctx, hipGetLastError());
        pc.lap(PH_SLICE);
        SHZ_TRY(finalize_active(t, (const uint32_t*)ck, (const uint32_t*)cs, (const uint32_t*)co, cnt, st_lo, st_hi));
      }
      if (sl + 1 < nsl) freeze_active(t);
    }
  }
  // The staging columns stay allocated for the next batch (a stream of ingest batches otherwise pays three
  // hipMalloc + growth copies per batch) unless they hold more than a quarter of what is free now.
  size_t mem_free = 0, mem_total = 0;
  ph_clock pcf(t);
  SHZ_HIP(ctx, hipMemGetInfo(&mem_free, &mem_total));
  if (!t->stage_reserved && t->scap * 12 > mem_free / 4) {
    void* st[] = {t->skey, t->ssid, t->soff};
    for (void* p : st)
      if (p) SHZ_HIP(ctx, hipFree(p));
    t->skey = t->ssid = t->soff = nullptr;
    t->scap = 0;
  }
  pcf.lap(PH_STAGE_FREE);
  t->ns = 0;
  return SHZ_OK;
}

extern "C" int32_t shz_table_segments(shz_table* t, uint32_t* n_segments) {
  if (!t || !n_segments) return SHZ_E_INVALID;
  *n_segments = (uint32_t)t->done.size() + (t->n ? 1u : 0u);
  return SHZ_OK;
}

extern "C" int32_t shz_table_set_segment_rows(shz_table* t, uint64_t rows) {
  if (!t) return SHZ_E_INVALID;
  if (rows < 16 || rows >= (1ull << 32)) SHZ_FAIL(t->ctx, SHZ_E_INVALID, "segment rows must be in [16, 2^32)");
  t->seg_limit = rows;
  return SHZ_OK;
}

extern "C" int32_t shz_table_rows(shz_table* t, uint64_t* n_rows, uint64_t* n_staged) {
  if (!t) return SHZ_E_INVALID;
  if (n_rows) *n_rows = total_rows(t);
  if (n_staged) *n_staged = pending_rows(t);
  return SHZ_OK;
}

extern "C" int32_t shz_table_export(shz_table* t, uint32_t* key32, uint32_t* sid, uint32_t* off, uint64_t cap,
                                    uint64_t* count) {
  if (!t) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  const uint64_t nrows = total_rows(t);
  if (count) *count = nrows;
  if (pending_rows(t)) SHZ_FAIL(ctx, SHZ_E_STATE, "table has %llu staged rows; call shz_table_finalize first", (unsigned long long)pending_rows(t));
  if (nrows > cap) SHZ_FAIL(ctx, SHZ_E_CAPACITY, "need %llu rows", (unsigned long long)nrows);
  if (nrows == 0) return SHZ_OK;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  uint64_t pos = 0;
  for (const shz_seg& g : all_segs(t)) {  // segment after segment; rows are sorted inside a segment
    SHZ_HIP(ctx, shz_memcpy(ctx, key32 + pos, g.key, g.n * 4, hipMemcpyDeviceToHost));
    SHZ_HIP(ctx, shz_memcpy(ctx, sid + pos, g.sid, g.n * 4, hipMemcpyDeviceToHost));
    SHZ_HIP(ctx, shz_memcpy(ctx, off + pos, g.off, g.n * 4, hipMemcpyDeviceToHost));
    pos += g.n;
  }
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}

extern "C" int32_t shz_table_song_rows(shz_table* t, uint32_t sid, uint64_t* n_rows) {
  if (!t || !n_rows) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  if (pending_rows(t)) SHZ_FAIL(ctx, SHZ_E_STATE, "table has staged rows; call shz_table_finalize first");
  *n_rows = 0;
  if (total_rows(t) == 0) return SHZ_OK;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  void* d;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC0, 64, &d));
  SHZ_HIP(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
  for (const shz_seg& g : all_segs(t))
    hipLaunchKernelGGL(tbl_count_sid_kernel, dim3((unsigned)((g.n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)g.sid, g.n, sid, (unsigned long long*)d);
  SHZ_HIP(ctx, hipGetLastError());
  SHZ_HIP(ctx, shz_memcpy(ctx, n_rows, d, 8, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}

__global__ void tbl_lookup_count_kernel(const uint32_t* __restrict__ keys, uint64_t nk, const uint32_t* __restrict__ tkey,
                                        uint32_t tn, const uint32_t* __restrict__ bucket, uint64_t nbuckets,
                                        uint32_t* __restrict__ lo_out, uint64_t* __restrict__ cnt) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nk) return;
  if (i == nk) { cnt[i] = 0; return; }
  const uint32_t key = keys[i];
  const uint64_t b = key >> 8;
  uint32_t lo = 0, rows = 0;
  if (b < nbuckets && tn) {
    uint32_t l = bucket[b], h = bucket[b + 1];
    const uint32_t h0 = h;
    while (l < h) { uint32_t mid = l + ((h - l) >> 1); if (tkey[mid] < key) l = mid + 1; else h = mid; }
    lo = l;
    h = h0;
    while (l < h) { uint32_t mid = l + ((h - l) >> 1); if (tkey[mid] <= key) l = mid + 1; else h = mid; }
    rows = l - lo;
  }
  lo_out[i] = lo;
  cnt[i] = rows;
}

__global__ void tbl_lookup_gather_kernel(const uint32_t* __restrict__ lo, const uint64_t* __restrict__ po, uint64_t nk,
                                         uint64_t total, const uint32_t* __restrict__ tkey,
                                         const uint32_t* __restrict__ tsid, const uint32_t* __restrict__ toff,
                                         uint32_t* __restrict__ okey, uint32_t* __restrict__ osid,
                                         uint32_t* __restrict__ ooff) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  uint64_t l = 0, h = nk;
  while (h - l > 1) { uint64_t mid = (l + h) >> 1; if (po[mid] <= p) l = mid; else h = mid; }
  const uint32_t row = lo[l] + (uint32_t)(p - po[l]);
  okey[p] = tkey[row];
  osid[p] = tsid[row];
  ooff[p] = toff[row];
}

// rows of the listed keys inside ONE segment: device gather, host arrays + per-key prefix (n_keys+1)
static int32_t lookup_segment(shz_table* t, const shz_seg& g, const uint32_t* keys, uint64_t n_keys,
                              std::vector<uint32_t>& ok_, std::vector<uint32_t>& os_, std::vector<uint32_t>& oo_,
                              std::vector<uint64_t>& po_) {
  shz_ctx* ctx = t->ctx;
  void *dk, *dlo, *dcnt, *dpo, *tot;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, n_keys * 4, &dk));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, n_keys * 4, &dlo));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M2, (n_keys + 1) * 8, &dcnt));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M3, (n_keys + 1) * 8, &dpo));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC0, 64, &tot));
  SHZ_HIP(ctx, shz_memcpy(ctx, dk, keys, n_keys * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(tbl_lookup_count_kernel, dim3((unsigned)((n_keys + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const uint32_t*)dk, n_keys, (const uint32_t*)g.key, (uint32_t)g.n, (const uint32_t*)g.bucket,
                     g.nbuckets, (uint32_t*)dlo, (uint64_t*)dcnt);
  SHZ_HIP(ctx, hipGetLastError());
  SHZ_TRY(shz_scan_u64(ctx, (const uint64_t*)dcnt, (uint64_t*)dpo, n_keys + 1, (uint64_t*)tot));
  po_.resize(n_keys + 1);
  SHZ_HIP(ctx, shz_memcpy(ctx, po_.data(), dpo, (n_keys + 1) * 8, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint64_t total = po_[n_keys];
  ok_.resize(total); os_.resize(total); oo_.resize(total);
  if (total == 0) return SHZ_OK;
  void *ok, *os, *oo;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M4, total * 4, &ok));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M5, total * 4, &os));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M6, total * 4, &oo));
  hipLaunchKernelGGL(tbl_lookup_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const uint32_t*)dlo, (const uint64_t*)dpo, n_keys, total, (const uint32_t*)g.key,
                     (const uint32_t*)g.sid, (const uint32_t*)g.off, (uint32_t*)ok, (uint32_t*)os, (uint32_t*)oo);
  SHZ_HIP(ctx, hipGetLastError());
  SHZ_HIP(ctx, shz_memcpy(ctx, ok_.data(), ok, total * 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, shz_memcpy(ctx, os_.data(), os, total * 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, shz_memcpy(ctx, oo_.data(), oo, total * 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}

extern "C" int32_t shz_table_lookup(shz_table* t, const uint32_t* keys, uint64_t n_keys, uint32_t* key32, uint32_t* sid,
                                    uint32_t* off, uint64_t cap, uint64_t* count) {
  if (!t) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  if (count) *count = 0;
  if (pending_rows(t) || (!t->bucket && t->done.empty())) SHZ_FAIL(ctx, SHZ_E_STATE, "table not finalized");
  if (n_keys == 0) return SHZ_OK;
  if (!keys) SHZ_FAIL(ctx, SHZ_E_INVALID, "keys is NULL");
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  const std::vector<shz_seg> segs = all_segs(t);
  std::vector<std::vector<uint32_t>> K(segs.size()), S(segs.size()), O(segs.size());
  std::vector<std::vector<uint64_t>> PO(segs.size());
  uint64_t total = 0;
  for (size_t i = 0; i < segs.size(); ++i) {
    SHZ_TRY(lookup_segment(t, segs[i], keys, n_keys, K[i], S[i], O[i], PO[i]));
    total += K[i].size();
  }
  if (count) *count = total;
  if (total > cap) SHZ_FAIL(ctx, SHZ_E_CAPACITY, "need %llu rows", (unsigned long long)total);
  if (total == 0) return SHZ_OK;
  if (!key32 || !sid || !off) SHZ_FAIL(ctx, SHZ_E_INVALID, "NULL output column");
  uint64_t pos = 0;  // grouped in key-list order; inside a key: segment order, then (song_id, offset)
  for (uint64_t k = 0; k < n_keys; ++k)
    for (size_t i = 0; i < segs.size(); ++i)
      for (uint64_t r = PO[i][k]; r < PO[i][k + 1]; ++r) {
        key32[pos] = K[i][r];
        sid[pos] = S[i][r];
        off[pos] = O[i][r];
        ++pos;
      }
  return SHZ_OK;
}

// ---------------------------------------------------------------------------------------- bulk build: runs + k-way merge
// The database build (fingerprint_directory's insert loop, __init__.py:378-386, at the scale of BASELINE configs 2-4):
//   staged rows --seal--> SORTED RUNS of packed rows in the run arena --one k-way merge--> segment columns in the slab.
// Rows travel, sort and merge in the packed form key << (sb + ob) | sid << ob | off (8 bytes a row; its numeric order is
// the table's order), which needs song id and offset to fit 32 bits together -- true for every configuration of
// BASELINE (1M songs = 20 bits, 3-minute tracks = 12 bits); other tables take the column path (finalize_active).
//   * a run is sorted once, by the rank (or ingest batch) that made it, and never again;
//   * N runs are merged in ONE pass ("every rank merges 8 sorted runs"): sampled splitters cut the value
//     range into tiles of ~2,048 rows, a workgroup loads its tile's share of every run into LDS, ranks every element by
//     binary searches in the other runs' shares, and writes the rows straight into the segment's three columns --
//     8 bytes read and 12 written per row, no intermediate merged run, no pass per level of a merge tree;
//   * after shz_table_reserve the build performs no device allocation: the slab (columns), the run arena, the staging
//     columns and the sort scratch exist before the first batch arrives (allocated on a helper thread beside it).

#define SHZ_I_GENERAL_PATH 1   // internal: the packed sorted-run path does not apply, take the column path

#define KW_MAXK 32             // runs one merge takes (more: the smallest are merged into one first)
#define KW_THREADS 256
// Nominal rows of a tile = samples per tile x sample stride; a tile holds fewer than (c + 2 k) M <= 3 c M = 3 x nominal rows
// (c samples per tile >= k runs, stride M): the LDS buffer of kw_tile_kernel<MODE, TILE> holds 3 TILE rows.  Two sizes
// (round 4, 1.09e9 rows, plan + merge): 2,048 rows (48 KB of LDS, 3 workgroups a CU) k = 2 / 8 / 16: 9.7 / 18.2 / 25.1 ms;
// 1,024 rows (24 KB, 6 workgroups: twice the waves to hide the merge rounds' dependent LDS reads): 6.8 / 14.5 / 21.5 ms
// (512 rows: 7.0 / 15.3 / 23.4; 768: 7.0 / 15.4 / 22.9; 512 threads at 1,024: 7.5 / 17.0 / 25.4).  Beyond 16 runs the sample
// stride of the small tile (1,024 / 32) doubles the samples to sort -- the plan's cost -- and the larger tile is kept.
#define KW_TILE_SMALL 1024
#define KW_TILE_LARGE 2048
#define KW_TMAX (3 * KW_TILE_LARGE)   // the largest tile any instantiation holds (bounds of the flush's scratch)

struct kw_runs {
  const uint64_t* p[KW_MAXK];
  uint64_t n[KW_MAXK];          // a run may hold more than 2^32 rows (a rank's collapsed runs at BASELINE configs[4])
  uint32_t soff[KW_MAXK + 1];   // first sample of every run in the sample array
  uint32_t k;
};

// reserve job: three allocations in the order the build needs them, made by a helper thread
#include <condition_variable>
#include <mutex>
#include <thread>
struct shz_reserve_job {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  int done = 0;                  // allocations finished so far (1: staging, 2: run arena + sort scratch, 3: slab)
  int device = 0;
  shz_ctx* ctx = nullptr;
  uint64_t stage_rows = 0, run_rows = 0, slab_bytes = 0, sort_bytes = 0;
  uint64_t st_bytes[3] = {0, 0, 0}, rbuf_bytes = 0;   // real sizes of the blocks (a cached block may be larger)
  uint32_t* st[3] = {nullptr, nullptr, nullptr};
  uint64_t* rbuf = nullptr;
  void* sortbuf[2] = {nullptr, nullptr};
  int n_sort = 2;
  char* slab = nullptr;
  bool failed = false;
  double alloc_s = 0.0;          // seconds the helper thread spent inside hipMalloc
};
enum { RJ_STAGE = 1, RJ_RUNS = 2, RJ_SLAB = 3 };

static void reserve_cancel(shz_table* t) {   // the table goes away (or starts over): take what the helper thread made
  if (t->job) reserve_wait(t, RJ_SLAB);
}

static void reserve_worker(shz_reserve_job* j) {
  (void)hipSetDevice(j->device);
  const double t0 = now_s();
  auto step = [&](int n) { std::lock_guard<std::mutex> lk(j->mu); j->done = n; j->alloc_s = now_s() - t0; j->cv.notify_all(); };
  bool ok = true;
  for (int i = 0; i < 3; ++i) ok = ok && j->stage_rows && shz_block_alloc(j->ctx, j->stage_rows * 4, (void**)&j->st[i], &j->st_bytes[i]) == hipSuccess;
  if (!ok) { for (int i = 0; i < 3; ++i) { if (j->st[i]) shz_block_free(j->ctx, j->st[i], j->st_bytes[i]); j->st[i] = nullptr; } (void)hipGetLastError(); }
  step(RJ_STAGE);
  if (j->run_rows && shz_block_alloc(j->ctx, j->run_rows * 8, (void**)&j->rbuf, &j->rbuf_bytes) != hipSuccess) { j->rbuf = nullptr; (void)hipGetLastError(); }
  for (int i = 0; i < j->n_sort; ++i)
    if (j->sort_bytes && hipMalloc(&j->sortbuf[i], j->sort_bytes) != hipSuccess) { j->sortbuf[i] = nullptr; (void)hipGetLastError(); }
  step(RJ_RUNS);
  if (j->slab_bytes) {
    const uint64_t want = j->slab_bytes;
    if (shz_block_alloc(j->ctx, want, (void**)&j->slab, &j->slab_bytes) != hipSuccess) { j->slab = nullptr; j->failed = true; (void)hipGetLastError(); }
  }
  step(RJ_SLAB);
}

// block until the helper thread has made allocation `which`, and take over what it made
static void reserve_wait(shz_table* t, int which) {
  shz_reserve_job* j = t->job;
  if (!j) return;
  const double t0 = now_s();
  {
    std::unique_lock<std::mutex> lk(j->mu);
    j->cv.wait(lk, [&] { return j->done >= which; });
  }
  shz_ctx* ctx = t->ctx;
  if (j->st[0] && t->ns == 0 && !t->skey) {
    t->skey = j->st[0]; t->ssid = j->st[1]; t->soff = j->st[2];
    t->scap = j->stage_rows;
    for (int i = 0; i < 3; ++i) t->st_bytes[i] = j->st_bytes[i];
    t->stage_reserved = true;
    j->st[0] = j->st[1] = j->st[2] = nullptr;
  }
  if (which >= RJ_RUNS) {
    if (j->rbuf && !t->rbuf) { t->rbuf = j->rbuf; t->rcap = j->run_rows; t->rbuf_bytes = j->rbuf_bytes; j->rbuf = nullptr; }
    const int slots[2] = {SHZ_WS_SORT_A, SHZ_WS_SORT_B};
    for (int i = 0; i < 2; ++i)
      if (j->sortbuf[i]) {
        shz_buf& b = ctx->ws[slots[i]];
        if (b.cap < j->sort_bytes) {
          if (b.p) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(b.p); }
          b.p = j->sortbuf[i];
          b.cap = j->sort_bytes;
        } else {
          (void)hipFree(j->sortbuf[i]);
        }
        j->sortbuf[i] = nullptr;
      }
  }
  if (which >= RJ_SLAB) {
    if (j->slab && !t->slab) { t->slab = j->slab; t->slab_bytes = j->slab_bytes; t->slab_used = 0; j->slab = nullptr; }
    j->th.join();
    t->ph[PH_RESERVE_ALLOC] += j->alloc_s;
    for (int i = 0; i < 3; ++i) if (j->st[i]) shz_block_free(ctx, j->st[i], j->st_bytes[i]);
    if (j->rbuf) shz_block_free(ctx, j->rbuf, j->rbuf_bytes);
    delete j;
    t->job = nullptr;
  }
  t->ph[PH_RESERVE_WAIT] += now_s() - t0;
}

extern "C" int32_t shz_table_reserve(shz_table* t, uint64_t rows_hint, uint64_t batch_rows_hint, uint32_t flags) {
  if (!t) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  if (flags & SHZ_RESERVE_GATHER) t->hold_runs = t->hold_reserved = true;   // (also without a row count: the runs wait, the arena grows as it must)
  if (t->job || t->slab) return SHZ_OK;   // one reservation per table
  if (rows_hint == 0) return SHZ_OK;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  if (batch_rows_hint == 0) batch_rows_hint = rows_hint;
  batch_rows_hint = std::min<uint64_t>(batch_rows_hint, (1ull << 32) - 4096);
  shz_reserve_job* j = new shz_reserve_job();
  j->device = ctx->device;
  j->ctx = ctx;
  j->stage_rows = t->skey ? 0 : batch_rows_hint + batch_rows_hint / 16 + 1024;
  // run arena: what is merged at once.  One GPU: a segment's worth of runs + the batch being sealed + the remainder a
  // flush leaves; the gathered build (SHZ_RESERVE_GATHER): this rank's run beside every rank's.
  const uint64_t seg = std::min<uint64_t>(rows_hint, t->seg_limit);
  j->run_rows = (flags & SHZ_RESERVE_GATHER) ? rows_hint + rows_hint / 64 + batch_rows_hint + batch_rows_hint / 8 + 65536
                                             : seg + 2 * (batch_rows_hint + batch_rows_hint / 16) + 65536;
  j->sort_bytes = (batch_rows_hint + batch_rows_hint / 16 + 65536) * 8;
  j->n_sort = (flags & SHZ_RESERVE_GATHER) ? 1 : 2;   // (the second scratch holds what a mid-build flush leaves over: never with held runs)
  // columns: 12 bytes a row, 256-byte aligned per column, + 2 % for duplicates-free estimates that run a little over
  j->slab_bytes = ((rows_hint + rows_hint / 50 + 65536) * 12 + 4095) & ~4095ull;
  size_t mem_free = 0, mem_total = 0;
  SHZ_HIP(ctx, hipMemGetInfo(&mem_free, &mem_total));
  const uint64_t want = j->slab_bytes + j->run_rows * 8 + j->stage_rows * 12 + j->n_sort * j->sort_bytes;
  uint64_t avail = mem_free;   // + what the context keeps from earlier tables (reused where it fits, dropped where it does not)
  {
    std::lock_guard<std::mutex> lk(ctx->blocks_mu);
    for (const shz_buf& b : ctx->blocks) avail += b.cap;
  }
  avail += ctx->ws[SHZ_WS_SORT_A].cap + ctx->ws[SHZ_WS_SORT_B].cap;
  if (want > avail - (avail >> 4)) {
    delete j;
    SHZ_FAIL(ctx, SHZ_E_NOMEM, "shz_table_reserve: %llu rows need %.1f GB of device memory, %.1f GB are free",
             (unsigned long long)rows_hint, want / 1e9, avail / 1e9);
  }
  t->job = j;
  j->th = std::thread(reserve_worker, j);
  static const bool sync_alloc = [] { const char* e = getenv("SHZ_RESERVE_SYNC"); return e && atoi(e) != 0; }();
  if (sync_alloc || (flags & SHZ_RESERVE_WAIT)) reserve_wait(t, RJ_SLAB);   // the allocations on the caller's clock, nothing beside them
  return SHZ_OK;
}

// three columns of `rows` rows: from the slab while it lasts, else three allocations
static int32_t carve_cols(shz_table* t, uint64_t rows, uint32_t** k, uint32_t** s, uint32_t** o, bool* from_slab) {
  shz_ctx* ctx = t->ctx;
  reserve_wait(t, RJ_SLAB);
  const uint64_t col = (std::max<uint64_t>(rows, 1) * 4 + 255) & ~255ull;
  if (t->slab && t->slab_used + 3 * col <= t->slab_bytes) {
    char* b = t->slab + t->slab_used;
    *k = (uint32_t*)b; *s = (uint32_t*)(b + col); *o = (uint32_t*)(b + 2 * col);
    t->slab_used += 3 * col;
    *from_slab = true;
    return SHZ_OK;
  }
  dev_cols c;
  if (!c.alloc(rows)) SHZ_FAIL(ctx, SHZ_E_NOMEM, "table: hipMalloc of %llu rows failed", (unsigned long long)rows);
  *k = c.take(0); *s = c.take(1); *o = c.take(2);
  *from_slab = false;
  return SHZ_OK;
}

// the run arena holds at least `rows` rows (contents kept)
static int32_t rbuf_reserve(shz_table* t, uint64_t rows) {
  shz_ctx* ctx = t->ctx;
  reserve_wait(t, RJ_RUNS);
  if (rows <= t->rcap) return SHZ_OK;
  if (t->gx_stream) SHZ_HIP(ctx, hipStreamSynchronize(t->gx_stream));   // the arena moves: no transfer may be on its way into it
  uint64_t used = 0;
  for (const shz_run& r : t->runs) used = std::max(used, r.off + r.n);
  const uint64_t cap = std::max<uint64_t>(rows + rows / 8 + 1024, 1024);
  uint64_t* nb = nullptr;
  if (hipMalloc(&nb, cap * 8) != hipSuccess) SHZ_FAIL(ctx, SHZ_E_NOMEM, "run arena: hipMalloc(%llu) failed", (unsigned long long)(cap * 8));
  if (used) SHZ_HIP(ctx, hipMemcpyAsync(nb, t->rbuf, used * 8, hipMemcpyDeviceToDevice, ctx->stream));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (t->rbuf) shz_block_free(ctx, t->rbuf, t->rbuf_bytes);
  t->rbuf = nb;
  t->rcap = cap;
  t->rbuf_bytes = cap * 8;
  return SHZ_OK;
}

// ---- kernels of the seal ----
// largest song id / offset and smallest song id of n rows: out[0] max sid, out[1] max off, out[2] min sid (preset to ~0)
__global__ void tbl_minmax_kernel(const uint32_t* __restrict__ sid, const uint32_t* __restrict__ off, uint64_t n,
                                  uint32_t* __restrict__ out) {
  uint32_t s = 0, o = 0, m = 0xFFFFFFFFu;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t x = sid[i];
    s = max(s, x);
    m = min(m, x);
    o = max(o, off[i]);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    s = max(s, (uint32_t)__shfl_xor((int)s, d, 64));
    o = max(o, (uint32_t)__shfl_xor((int)o, d, 64));
    m = min(m, (uint32_t)__shfl_xor((int)m, d, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&out[0], s);
    atomicMax(&out[1], o);
    atomicMin(&out[2], m);
  }
}

// pack rows and note whether they already come ordered by (song id, offset) -- the order ingest produces (clips in id
// order, a clip's hashes in time order, __init__.py:194-210): a stable sort on the key bits alone then yields the full
// order, four radix passes instead of eight.  *unordered is set when some row sorts before its predecessor.
__global__ void run_pack_kernel(const uint32_t* __restrict__ key, const uint32_t* __restrict__ sid,
                                const uint32_t* __restrict__ off, uint64_t n, int sb, int ob, uint64_t* __restrict__ out,
                                uint32_t* __restrict__ unordered) {
  bool bad = false;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t lo = ((uint64_t)sid[i] << ob) | off[i];
    out[i] = ((uint64_t)key[i] << (sb + ob)) | lo;
    if (i) bad |= lo < (((uint64_t)sid[i - 1] << ob) | off[i - 1]);
  }
  if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(unordered, 1u);
}

// rows of a sorted run that equal their predecessor
__global__ void run_dups_kernel(const uint64_t* __restrict__ c, uint64_t n, unsigned long long* __restrict__ out) {
  uint32_t d = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    d += (i && c[i] == c[i - 1]) ? 1u : 0u;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) d += (uint32_t)__shfl_xor((int)d, s, 64);
  if ((threadIdx.x & 63) == 0 && d) atomicAdd(out, (unsigned long long)d);
}
__global__ void run_compact_kernel(const uint64_t* __restrict__ c, const uint32_t* __restrict__ flag,
                                   const uint32_t* __restrict__ pos, uint64_t n, uint64_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) out[pos[i]] = c[i];
}
// packed rows from one layout to another (both monotone in (key, sid, off): a sorted run stays sorted)
__global__ void run_repack_kernel(uint64_t* __restrict__ c, uint64_t n, int sb0, int ob0, int sb1, int ob1) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t v = c[i];
    const uint64_t key = v >> (sb0 + ob0), sid = (v >> ob0) & ((1ull << sb0) - 1), off = v & ((1ull << ob0) - 1);
    c[i] = (key << (sb1 + ob1)) | (sid << ob1) | off;
  }
}

// ---- kernels of the k-way merge ----
__global__ void kw_sample_kernel(kw_runs R, uint32_t M, uint64_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= R.soff[R.k]) return;
  uint32_t r = 0;
  while (j >= R.soff[r + 1]) ++r;
  out[j] = R.p[r][(uint64_t)(j - R.soff[r]) * M];
}

// bounds[t * k + r] = first element of run r that belongs to tile t or a later one: tile t takes the values
// [smp[t * c], smp[(t + 1) * c)), the first tile everything below, the last everything above
__global__ void kw_bounds_kernel(kw_runs R, const uint64_t* __restrict__ smp, uint32_t c, uint32_t ntiles,
                                 uint64_t* __restrict__ bounds) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t k = R.k;
  if (e >= ((uint64_t)ntiles + 1) * k) return;
  const uint32_t t = (uint32_t)(e / k), r = (uint32_t)(e - (uint64_t)t * k);
  uint64_t lo = 0, hi = R.n[r];
  if (t == 0) hi = 0;
  else if (t == ntiles) lo = hi;
  else {
    const uint64_t v = smp[(uint64_t)t * c];
    const uint64_t* __restrict__ p = R.p[r];
    while (lo < hi) {
      const uint64_t mid = lo + ((hi - lo) >> 1);
      if (p[mid] < v) lo = mid + 1; else hi = mid;
    }
  }
  bounds[e] = lo;
}

__global__ void kw_tile_rows_kernel(const uint64_t* __restrict__ bounds, uint32_t k, uint32_t ntiles, uint64_t* __restrict__ rows) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > ntiles) return;
  uint64_t s = 0;
  if (t < ntiles)
    for (uint32_t r = 0; r < k; ++r) s += bounds[(uint64_t)(t + 1) * k + r] - bounds[(uint64_t)t * k + r];
  rows[t] = s;   // rows[ntiles] = 0: the exclusive scan ends in the total
}

// output pieces of at most L rows, taken greedily in whole tiles (a tile alone always counts): tile[j] = first tile of
// piece j, row[j] = its first output row; the last entry is (ntiles, total); n = number of entries (pieces + 1)
struct kw_cutlist {
  uint32_t n, pad;
  uint32_t tile[SHZ_MAX_SEGS + 4];
  uint64_t row[SHZ_MAX_SEGS + 4];
};
__global__ void kw_cuts_kernel(const uint64_t* __restrict__ base, uint32_t ntiles, uint64_t L, kw_cutlist* __restrict__ out) {
  if (blockIdx.x || threadIdx.x) return;
  uint32_t pos = 0, n = 0;
  out->tile[n] = 0; out->row[n] = 0; ++n;
  while (pos < ntiles && n < SHZ_MAX_SEGS + 3) {
    uint32_t lo = pos + 1, hi = ntiles;   // largest e in [pos + 1, ntiles] with base[e] - base[pos] <= L
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo + 1) >> 1);
      if (base[mid] - base[pos] <= L) lo = mid; else hi = mid - 1;
    }
    pos = lo;
    out->tile[n] = pos; out->row[n] = base[pos]; ++n;
  }
  out->n = n;
}

// One tile of the merge.  MODE 0: rows -> columns; 1: count the tile's distinct rows; 2: rows -> columns, duplicates
// dropped; 3: rows -> packed; 4: packed, duplicates dropped.  base[t] = output row of tile t (of the distinct rows in
// the modes that drop duplicates); this launch writes relative to row out0.
template <int MODE, int TILE>
__global__ __launch_bounds__(KW_THREADS) void kw_tile_kernel(kw_runs R, const uint64_t* __restrict__ bounds, uint32_t tile0,
                                                             const uint64_t* __restrict__ base, uint64_t out0, int sb, int ob,
                                                             uint32_t* __restrict__ okey, uint32_t* __restrict__ osid,
                                                             uint32_t* __restrict__ ooff, uint64_t* __restrict__ opacked,
                                                             uint64_t* __restrict__ uniq, uint32_t* __restrict__ err) {
  constexpr int TMAX = 3 * TILE, PER = TMAX / KW_THREADS;
  __shared__ uint64_t buf[TMAX];
  __shared__ uint64_t s_lo[KW_MAXK];
  __shared__ uint32_t s_off[KW_MAXK + 1], s_w[KW_THREADS / 64 + 1];
  const uint32_t t = tile0 + blockIdx.x, k = R.k, tid = threadIdx.x, lane = tid & 63;
  if (tid < 64) {
    uint64_t lo = 0;
    uint32_t cnt = 0;
    if (tid < k) {
      lo = bounds[(uint64_t)t * k + tid];
      const uint64_t c64 = bounds[(uint64_t)(t + 1) * k + tid] - lo;
      cnt = (uint32_t)(c64 < 0x7FFFFFFull ? c64 : 0x7FFFFFFull);   // (a share that large has broken the plan: flagged below)
    }
    uint32_t inc = cnt;
#pragma unroll
    for (int d = 1; d < KW_MAXK; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
      if ((int)lane >= d) inc += o;
    }
    if (tid < k) { s_lo[tid] = lo; s_off[tid + 1] = inc; }
    if (tid == 0) s_off[0] = 0;
  }
  __syncthreads();
  // The plan bounds a tile below KW_TMAX rows.  Should it ever not, the tile says so: the host fails the merge and
  // marks the table (rows beyond the clamp would be lost, the columns would hold holes) -- never a silent drop.
  if (s_off[k] > (uint32_t)TMAX && tid == 0) atomicOr(err, 1u);
  const uint32_t total = min(s_off[k], (uint32_t)TMAX);
  for (uint32_t r = 0; r < k; ++r) {
    const uint64_t* __restrict__ p = R.p[r] + s_lo[r];
    const uint32_t o = s_off[r], c = min(s_off[r + 1], total) - min(o, total);
    for (uint32_t i = tid; i < c; i += KW_THREADS) buf[o + i] = p[i];
  }
  __syncthreads();
  // log2 k rounds of pairwise merges inside LDS (merge path): in round `stride` the neighbouring regions
  // [s_off[q], s_off[q + stride]) and [s_off[q + stride], s_off[q + 2 stride]) -- each sorted -- become one.  A thread
  // produces `per` consecutive outputs: one binary search along its diagonal, then a serial merge of the two heads;
  // ties take the left (lower) run first.  Outputs wait in registers until every thread has read its inputs.
  const uint32_t per = (total + KW_THREADS - 1) / KW_THREADS;          // uniform, <= PER
  const uint32_t g0 = min(tid * per, total), g1 = min(g0 + per, total);
  for (uint32_t stride = 1; stride < k; stride <<= 1) {
    uint64_t out[PER];
    uint32_t g = g0;
    while (g < g1) {                                                    // one piece per region the thread's outputs touch (1, rarely 2)
      uint32_t q = 0;
      while (q + 2 * stride < k && s_off[min(k, q + 2 * stride)] <= g) q += 2 * stride;
      const uint32_t a0 = min(s_off[q], total), mid = min(s_off[min(k, q + stride)], total), e = min(s_off[min(k, q + 2 * stride)], total);
      const uint32_t na = mid - a0, nb = e - mid, d = g - a0, cnt = min(g1, e) - g, c0 = g - g0;
      uint32_t lo = d > nb ? d - nb : 0, hi = min(d, na);
      while (lo < hi) {                                                 // elements of A among the first d outputs
        const uint32_t m = (lo + hi) >> 1;
        if (buf[a0 + m] <= buf[mid + (d - 1 - m)]) lo = m + 1; else hi = m;
      }
      uint32_t i = lo, j = d - lo;
      uint64_t ca = i < na ? buf[a0 + i] : 0, cb = j < nb ? buf[mid + j] : 0;
#pragma unroll
      for (int c = 0; c < PER; ++c) {
        if ((uint32_t)c < per && (uint32_t)c >= c0 && (uint32_t)c < c0 + cnt) {
          const bool take_a = j >= nb || (i < na && ca <= cb);
          out[c] = take_a ? ca : cb;
          if (take_a) { ++i; ca = i < na ? buf[a0 + i] : 0; } else { ++j; cb = j < nb ? buf[mid + j] : 0; }
        }
      }
      g += cnt;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < PER; ++c)
      if ((uint32_t)c < per && g0 + c < g1) buf[g0 + c] = out[c];
    __syncthreads();
  }
  // the tile is sorted in buf[0, total)
  const int shift = sb + ob;
  const uint64_t smask = (1ull << sb) - 1, omask = (1ull << ob) - 1;
  if (MODE == 0 || MODE == 3) {
    const uint64_t row0 = base[t] - out0;
    for (uint32_t i = tid; i < total; i += KW_THREADS) {
      const uint64_t v = buf[i];
      if (MODE == 0) {
        okey[row0 + i] = (uint32_t)(v >> shift);
        osid[row0 + i] = (uint32_t)((v >> ob) & smask);
        ooff[row0 + i] = (uint32_t)(v & omask);
      } else {
        opacked[row0 + i] = v;
      }
    }
    return;
  }
  // duplicates (equal neighbours; all copies of a value sit in one tile) are dropped: position = number of kept rows before
  uint32_t run = 0;   // kept rows of the rounds before this one (uniform)
  const uint64_t row0 = MODE == 1 ? 0 : base[t] - out0;
  for (uint32_t i0 = 0; i0 < total; i0 += KW_THREADS) {
    const uint32_t i = i0 + tid;
    const bool keep = i < total && (i == 0 || buf[i] != buf[i - 1]);
    const unsigned long long b = __ballot(keep);
    if (lane == 0) s_w[tid >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < KW_THREADS / 64; ++w) {
      const uint32_t c = s_w[w];
      if (w < (int)(tid >> 6)) before += c;
      all += c;
    }
    if (MODE != 1 && keep) {
      const uint64_t v = buf[i];
      const uint64_t o = row0 + run + before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
      if (MODE == 2) {
        okey[o] = (uint32_t)(v >> shift);
        osid[o] = (uint32_t)((v >> ob) & smask);
        ooff[o] = (uint32_t)(v & omask);
      } else {
        opacked[o] = v;
      }
    }
    run += all;
    __syncthreads();
  }
  if (MODE == 1 && tid == 0) uniq[t] = run;
}

// launch of one instantiation by the plan's tile size
template <int MODE>
static void kw_launch(hipStream_t st, uint32_t tile_rows, uint32_t grid, const kw_runs& R, const uint64_t* bounds, uint32_t tile0,
                      const uint64_t* base, uint64_t out0, int sb, int ob, uint32_t* okey, uint32_t* osid, uint32_t* ooff,
                      uint64_t* opacked, uint64_t* uniq, uint32_t* err) {
  if (tile_rows == KW_TILE_SMALL)
    hipLaunchKernelGGL((kw_tile_kernel<MODE, KW_TILE_SMALL>), dim3(grid), dim3(KW_THREADS), 0, st, R, bounds, tile0, base, out0, sb, ob, okey,
                       osid, ooff, opacked, uniq, err);
  else
    hipLaunchKernelGGL((kw_tile_kernel<MODE, KW_TILE_LARGE>), dim3(grid), dim3(KW_THREADS), 0, st, R, bounds, tile0, base, out0, sb, ob, okey,
                       osid, ooff, opacked, uniq, err);
}

// ---- host side of the bulk build ----
static uint64_t runs_end(const shz_table* t) {
  uint64_t e = 0;
  for (const shz_run& r : t->runs) e = std::max(e, r.off + r.n);
  return e;
}
static uint64_t runs_rows(const shz_table* t) {
  uint64_t n = 0;
  for (const shz_run& r : t->runs) n += r.n;
  return n;
}

// The layout every run of the table is packed in is a PURE function of the largest offset the table has seen:
// ob = its bits, sb = 32 - ob (all spare bits to the song id, which grows with ingest).  Widening -- a later batch
// brings longer tracks -- repacks the runs the arena holds in place (monotone: sorted runs stay sorted).  Being pure,
// the ranks of a gathered build arrive at the same layout from the same global maxima, whatever each sealed before.
// false: ids and offsets do not fit 32 bits together.
static int32_t run_layout(shz_table* t, uint32_t max_sid, uint32_t max_off, bool* ok) {
  shz_ctx* ctx = t->ctx;
  const int need_sb = bits_for(max_sid), need_ob = bits_for(max_off);
  const int ob = std::max(need_ob, t->run_ob), sb = 32 - ob;   // (run_ob only grows: it covers everything packed so far)
  *ok = need_sb + ob <= 32;
  if (!*ok) return SHZ_OK;
  if (ob == t->run_ob) return SHZ_OK;
  if (t->run_ob && !t->runs.empty()) {
    if (t->gx_stream) SHZ_HIP(ctx, hipStreamSynchronize(t->gx_stream));   // no transfer touches a run while it is rewritten
    for (const shz_run& r : t->runs)
      if (r.n)
        hipLaunchKernelGGL(run_repack_kernel, dim3((unsigned)std::min<uint64_t>((r.n + 255) / 256, 8192)), dim3(256), 0, ctx->stream,
                           t->rbuf + r.off, r.n, t->run_sb, t->run_ob, sb, ob);
    SHZ_HIP(ctx, hipGetLastError());
  }
  t->run_sb = sb;
  t->run_ob = ob;
  return SHZ_OK;
}

// n rows of three columns -> one sorted run without duplicates at the end of the run arena
static int32_t seal_rows(shz_table* t, const uint32_t* key, const uint32_t* sid, const uint32_t* off, uint64_t n, uint32_t sid_lo,
                         uint32_t sid_hi) {
  shz_ctx* ctx = t->ctx;
  if (n == 0) return SHZ_OK;
  if (n >= (1ull << 32) - 4096) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "a run is limited to < 2^32 rows (have %llu)", (unsigned long long)n);
  ph_clock pc(t);
  const uint64_t tail = (runs_end(t) + 31) & ~31ull;   // runs start 256-byte aligned (the sort loads 16 bytes per lane)
  SHZ_TRY(rbuf_reserve(t, tail + n));
  pc.lap(PH_RESERVE_WAIT);
  uint64_t* dst = t->rbuf + tail;
  void *flag, *tmp;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC0, 64, &flag));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_A, n * 8, &tmp));
  SHZ_HIP(ctx, hipMemsetAsync(flag, 0, 64, ctx->stream));
  const int sb = t->run_sb, ob = t->run_ob;
  hipLaunchKernelGGL(run_pack_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 16384)), dim3(256), 0, ctx->stream, key, sid,
                     off, n, sb, ob, dst, (uint32_t*)flag);
  SHZ_HIP(ctx, hipGetLastError());
  uint32_t unordered = 1;
  SHZ_HIP(ctx, shz_memcpy(ctx, &unordered, flag, 4, hipMemcpyDeviceToHost));
  pc.lap(PH_RUN_PACK);
  // rows that arrive ordered by (song id, offset) need a stable sort on the key bits only
  int sel = 0;
  SHZ_TRY(shz_sort_u64(ctx, dst, (uint64_t*)tmp, nullptr, nullptr, 0, n, unordered ? 0 : sb + ob, 32 + sb + ob, &sel));
  if (sel) SHZ_HIP(ctx, hipMemcpyAsync(dst, tmp, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
  pc.lap(PH_RUN_SORT);
  // duplicates inside the run (INSERT IGNORE): counted first, compacted only if there are any
  unsigned long long* d_dups = (unsigned long long*)flag + 1;
  hipLaunchKernelGGL(run_dups_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 16384)), dim3(256), 0, ctx->stream,
                     (const uint64_t*)dst, n, d_dups);
  SHZ_HIP(ctx, hipGetLastError());
  unsigned long long dups = 0;
  SHZ_HIP(ctx, shz_memcpy(ctx, &dups, d_dups, 8, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t kept = n;
  if (dups) {
    void *fl, *ps;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, n * 4, &fl));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, n * 4, &ps));
    hipLaunchKernelGGL(tbl_uniq1_flag_kernel, dim3(nblk(n)), dim3(256), 0, ctx->stream, (const uint64_t*)dst, n, (uint32_t*)fl);
    SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)fl, (uint32_t*)ps, n, nullptr));
    hipLaunchKernelGGL(run_compact_kernel, dim3(nblk(n)), dim3(256), 0, ctx->stream, (const uint64_t*)dst, (const uint32_t*)fl,
                       (const uint32_t*)ps, n, (uint64_t*)tmp);
    SHZ_HIP(ctx, hipGetLastError());
    kept = n - dups;
    SHZ_HIP(ctx, hipMemcpyAsync(dst, tmp, kept * 8, hipMemcpyDeviceToDevice, ctx->stream));
  }
  pc.lap(PH_RUN_UNIQ);
  t->runs.push_back(shz_run{tail, kept, sid_lo, sid_hi});
  return SHZ_OK;
}


// can two of the runs hold the same row?  Only if their song-id ranges meet.
static bool runs_may_share_rows(const std::vector<shz_run>& runs) {
  for (size_t i = 0; i < runs.size(); ++i)
    for (size_t j = i + 1; j < runs.size(); ++j)
      if (runs[i].n && runs[j].n && ranges_overlap(runs[i].sid_lo, runs[i].sid_hi, runs[j].sid_lo, runs[j].sid_hi)) return true;
  return false;
}


// The k-way merge of `runs` (pointers into device memory, each sorted and duplicate-free).  Output pieces of at most
// `piece_rows` rows: the first `n_seg_pieces` of them (all if 0xFFFFFFFF) become table segments -- the last one the
// active segment when `last_active` --, what is left is written packed to `rest` (capacity rest_cap rows) and its
// row count to *rest_rows.  dedup: rows may repeat across runs.  Pieces follow each other in value order: the segments
// of one merge hold disjoint key ranges (but for a key whose rows straddle a cut).
static int32_t kway_merge(shz_table* t, const std::vector<const uint64_t*>& ptr, const std::vector<shz_run>& runs, bool dedup,
                          uint64_t piece_rows, uint32_t n_seg_pieces, bool last_active, uint64_t* rest, uint64_t rest_cap,
                          uint64_t* rest_rows) {
  shz_ctx* ctx = t->ctx;
  if (rest_rows) *rest_rows = 0;
  kw_runs R;
  memset(&R, 0, sizeof(R));
  uint32_t k = 0, sid_lo = 0xFFFFFFFFu, sid_hi = 0;
  for (size_t i = 0; i < runs.size(); ++i) {
    if (!runs[i].n) continue;
    if (k == KW_MAXK) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "k-way merge of more than %d runs", KW_MAXK);
    R.p[k] = ptr[i];
    R.n[k] = runs[i].n;
    sid_lo = std::min(sid_lo, runs[i].sid_lo);
    sid_hi = std::max(sid_hi, runs[i].sid_hi);
    ++k;
  }
  if (k == 0) return SHZ_OK;
  R.k = k;
  ph_clock pc(t);
  if (!t->d_kw_err) SHZ_HIP(ctx, hipMalloc(&t->d_kw_err, 64));
  SHZ_HIP(ctx, hipMemsetAsync(t->d_kw_err, 0, 64, ctx->stream));
  // plan: every M-th element of every run is a sample; every c-th of the sorted samples starts a tile
  uint32_t kp2 = 1;
  while (kp2 < k) kp2 <<= 1;
  const uint32_t tile_rows = kp2 <= 16 ? KW_TILE_SMALL : KW_TILE_LARGE;
  const uint32_t nominal = (uint32_t)std::min<uint64_t>(tile_rows, std::max<uint64_t>(kp2, piece_rows / 8));
  const uint32_t M = std::max<uint32_t>(1, nominal / kp2), c = kp2;
  uint64_t S = 0;
  for (uint32_t r = 0; r < k; ++r) {
    if (S >= (1ull << 32)) break;
    R.soff[r] = (uint32_t)S;
    S += (R.n[r] + M - 1) / M;
  }
  if (S >= (1ull << 32)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "k-way merge: %llu samples", (unsigned long long)S);
  R.soff[k] = (uint32_t)S;
  const uint32_t ntiles = (uint32_t)((S + c - 1) / c);
  void *smp0, *smp1, *bnd, *rows, *base, *cutp;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M2, S * 8, &smp0));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M3, S * 8, &smp1));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M4, ((uint64_t)ntiles + 1) * k * 8, &bnd));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M5, ((uint64_t)ntiles + 1) * 8, &rows));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M6, ((uint64_t)ntiles + 1) * 8, &base));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, sizeof(kw_cutlist), &cutp));
  hipLaunchKernelGGL(kw_sample_kernel, dim3(nblk(S)), dim3(256), 0, ctx->stream, R, M, (uint64_t*)smp0);
  SHZ_HIP(ctx, hipGetLastError());
  int sel = 0;
  if (k > 1) SHZ_TRY(shz_sort_u64(ctx, (uint64_t*)smp0, (uint64_t*)smp1, nullptr, nullptr, 0, S, 0, 32 + t->run_sb + t->run_ob, &sel));
  const uint64_t* smp = sel ? (const uint64_t*)smp1 : (const uint64_t*)smp0;
  hipLaunchKernelGGL(kw_bounds_kernel, dim3(nblk(((uint64_t)ntiles + 1) * k)), dim3(256), 0, ctx->stream, R, smp, c, ntiles,
                     (uint64_t*)bnd);
  const int sb = t->run_sb, ob = t->run_ob;
  const uint64_t* cb = (const uint64_t*)bnd;
  if (!dedup) {
    hipLaunchKernelGGL(kw_tile_rows_kernel, dim3(nblk((uint64_t)ntiles + 1)), dim3(256), 0, ctx->stream, cb, k, ntiles, (uint64_t*)rows);
  } else {   // a first pass over the tiles counts their distinct rows
    SHZ_HIP(ctx, hipMemsetAsync((uint64_t*)rows + ntiles, 0, 8, ctx->stream));
    kw_launch<1>(ctx->stream, tile_rows, ntiles, R, cb, 0u, (const uint64_t*)nullptr, (uint64_t)0, sb, ob, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)rows, t->d_kw_err);
  }
  SHZ_HIP(ctx, hipGetLastError());
  SHZ_TRY(shz_scan_u64(ctx, (const uint64_t*)rows, (uint64_t*)base, (uint64_t)ntiles + 1, nullptr));
  hipLaunchKernelGGL(kw_cuts_kernel, dim3(1), dim3(1), 0, ctx->stream, (const uint64_t*)base, ntiles, piece_rows, (kw_cutlist*)cutp);
  SHZ_HIP(ctx, hipGetLastError());
  kw_cutlist hc;
  SHZ_HIP(ctx, shz_memcpy(ctx, &hc, cutp, sizeof(hc), hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t n_cuts = hc.n;
  const uint32_t* cuts = hc.tile;
  const uint64_t* cut_row = hc.row;
  if (n_cuts < 2 || cuts[n_cuts - 1] != ntiles) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "more than %d table segments", SHZ_MAX_SEGS);
  pc.lap(PH_KW_PLAN);
  // the tile kernels' verdict on the plan (a tile beyond KW_TMAX rows would have dropped rows): read after every launch
  // of them has run, before anything is reported as done
  auto check_tiles = [&]() -> int32_t {
    uint32_t bad = 0;
    SHZ_HIP(ctx, shz_memcpy(ctx, &bad, t->d_kw_err, 4, hipMemcpyDeviceToHost));
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (bad) {
      t->broken = true;
      SHZ_FAIL(ctx, SHZ_E_STATE, "k-way merge: a tile outgrew its bound of %d rows (k = %u, stride %u); the table is incomplete and refuses further use",
               (int)(3 * tile_rows), k, M);
    }
    return SHZ_OK;
  };
  const uint32_t n_pieces = n_cuts - 1;
  const uint32_t n_seg = std::min(n_seg_pieces, n_pieces);
  if (t->done.size() + n_seg > SHZ_MAX_SEGS) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "more than %d table segments", SHZ_MAX_SEGS);
  for (uint32_t i = 0; i < n_seg; ++i) {
    const uint64_t r0 = cut_row[i], nrows = cut_row[i + 1] - r0;
    const uint32_t t0 = cuts[i], nt = cuts[i + 1] - t0;
    if (nrows == 0) continue;
    if (t->n) freeze_active(t);
    uint32_t *ck, *cs, *co;
    bool from_slab = false;
    SHZ_TRY(carve_cols(t, nrows, &ck, &cs, &co, &from_slab));
    pc.lap(PH_COL_ALLOC);
    if (dedup)
      kw_launch<2>(ctx->stream, tile_rows, nt, R, cb, t0, (const uint64_t*)base, r0, sb, ob, ck, cs, co, (uint64_t*)nullptr, (uint64_t*)nullptr, t->d_kw_err);
    else
      kw_launch<0>(ctx->stream, tile_rows, nt, R, cb, t0, (const uint64_t*)base, r0, sb, ob, ck, cs, co, (uint64_t*)nullptr, (uint64_t*)nullptr, t->d_kw_err);
    SHZ_HIP(ctx, hipGetLastError());
    // the new segment is the active one until the next piece (or the caller) freezes it
    if (t->key && !t->act_slab) { void* olds[] = {t->key, t->sid, t->off}; for (void* p : olds) (void)hipFree(p); }
    t->key = ck; t->sid = cs; t->off = co;
    t->n = nrows;
    t->cap = nrows;
    t->act_slab = from_slab;
    t->act_sid_lo = sid_lo;
    t->act_sid_hi = sid_hi;
    pc.lap(PH_KW_MERGE);
    SHZ_TRY(rebuild_buckets(ctx, t->key, t->n, &t->bucket, &t->nbuckets, &t->bcap, &t->act_key_lo));
    pc.lap(PH_BUCKET);
  }
  if (t->n && !(last_active && n_seg == n_pieces)) freeze_active(t);
  if (n_seg < n_pieces) {   // the rest, as one packed run
    const uint64_t r0 = cut_row[n_seg], nrows = cut_row[n_pieces] - r0;
    const uint32_t t0 = cuts[n_seg], nt = ntiles - t0;
    if (nrows > rest_cap || !rest) SHZ_FAIL(ctx, SHZ_E_STATE, "k-way merge: %llu rows left over, room for %llu", (unsigned long long)nrows, (unsigned long long)rest_cap);
    if (nrows) {
      if (dedup)
        kw_launch<4>(ctx->stream, tile_rows, nt, R, cb, t0, (const uint64_t*)base, r0, sb, ob, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, rest, (uint64_t*)nullptr, t->d_kw_err);
      else
        kw_launch<3>(ctx->stream, tile_rows, nt, R, cb, t0, (const uint64_t*)base, r0, sb, ob, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, rest, (uint64_t*)nullptr, t->d_kw_err);
      SHZ_HIP(ctx, hipGetLastError());
    }
    if (rest_rows) *rest_rows = nrows;
    pc.lap(PH_KW_MERGE);
  }
  return check_tiles();
}

// merge the smallest runs of the arena into one, at the arena's tail, until at most `keep` runs are left: a merge takes
// KW_MAXK runs (the places the merged runs leave stay unused until the arena starts over)
static int32_t collapse_runs(shz_table* t, size_t keep) {
  keep = std::max<size_t>(keep, 1);
  while (t->runs.size() > keep) {
    const size_t m = std::min<size_t>(KW_MAXK, t->runs.size() - keep + 1);
    std::vector<size_t> order(t->runs.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return t->runs[a].n < t->runs[b].n; });
    std::vector<size_t> pick(order.begin(), order.begin() + (long)m);
    std::sort(pick.begin(), pick.end());
    std::vector<shz_run> sub;
    uint64_t total = 0;
    uint32_t sid_lo = 0xFFFFFFFFu, sid_hi = 0;
    uint8_t where = RUN_SENT;
    for (size_t i : pick) {
      const shz_run& r = t->runs[i];
      sub.push_back(r);
      total += r.n;
      if (r.n) { sid_lo = std::min(sid_lo, r.sid_lo); sid_hi = std::max(sid_hi, r.sid_hi); }
      if (r.where == RUN_LOCAL) where = RUN_LOCAL;   // (the exchange never collapses a mix of travelled and local runs)
    }
    const uint64_t tail = (runs_end(t) + 31) & ~31ull;
    SHZ_TRY(rbuf_reserve(t, tail + total));
    std::vector<const uint64_t*> ptr;
    for (const shz_run& r : sub) ptr.push_back(t->rbuf + r.off);
    uint64_t rest_rows = 0;
    SHZ_TRY(kway_merge(t, ptr, sub, runs_may_share_rows(sub), ~0ull >> 1, 0, false, t->rbuf + tail, total, &rest_rows));
    for (size_t j = pick.size(); j-- > 0;) t->runs.erase(t->runs.begin() + (long)pick[j]);
    if (rest_rows) {
      shz_run nr{tail, rest_rows, sid_lo, sid_hi};
      nr.where = where;
      t->runs.push_back(nr);
    }
    if (t->runs.empty()) break;
  }
  return SHZ_OK;
}

// Runs of the arena -> table.  final: every row becomes a segment row (the last piece stays the active segment).
// Otherwise only FULL segments are cut (seg_limit rows each) and what is left goes back to the arena as one run.
static int32_t flush_runs(shz_table* t, bool final) {
  shz_ctx* ctx = t->ctx;
  uint64_t total = runs_rows(t);
  if (total == 0) { t->runs.clear(); return SHZ_OK; }
  if (t->gx_stream) {   // every run has arrived; nothing of this table is on the communicator's stream any more (the table may outlive it)
    SHZ_HIP(ctx, hipStreamSynchronize(t->gx_stream));
    t->gx_stream = nullptr;
  }
  if (t->runs.size() > KW_MAXK) SHZ_TRY(collapse_runs(t, KW_MAXK));
  std::vector<const uint64_t*> ptr;
  for (const shz_run& r : t->runs) ptr.push_back(t->rbuf + r.off);
  const bool dedup = runs_may_share_rows(t->runs);
  uint32_t sid_lo = 0xFFFFFFFFu, sid_hi = 0;
  for (const shz_run& r : t->runs)
    if (r.n) { sid_lo = std::min(sid_lo, r.sid_lo); sid_hi = std::max(sid_hi, r.sid_hi); }
  const uint64_t L = std::min<uint64_t>(t->seg_limit, (1ull << 32) - 4096);
  if (final) {
    SHZ_TRY(kway_merge(t, ptr, t->runs, dedup, L, 0xFFFFFFFFu, true, nullptr, 0, nullptr));
    t->runs.clear();
    t->hold_runs = false;   // the bulk build is over: rows that come now join a table that holds rows
    return SHZ_OK;
  }
  const uint32_t full = (uint32_t)(total / L);   // (duplicates across runs can only make the pieces fewer)
  if (full == 0) return SHZ_OK;
  // the leftover is written to scratch and moved to the front of the arena
  const uint64_t rest_cap = total - (uint64_t)full * L + (uint64_t)KW_TMAX * (full + 1);
  void* rest;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_B, rest_cap * 8, &rest));
  uint64_t rest_rows = 0;
  SHZ_TRY(kway_merge(t, ptr, t->runs, dedup, L, full, false, (uint64_t*)rest, rest_cap, &rest_rows));
  t->runs.clear();
  if (rest_rows) {
    SHZ_TRY(rbuf_reserve(t, rest_rows));
    SHZ_HIP(ctx, hipMemcpyAsync(t->rbuf, rest, rest_rows * 8, hipMemcpyDeviceToDevice, ctx->stream));
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    t->runs.push_back(shz_run{0, rest_rows, sid_lo, sid_hi});
  }
  return SHZ_OK;
}


// maxima / minimum song id of the staged rows [lo, lo + n)
static int32_t staged_minmax(shz_table* t, uint64_t lo, uint64_t n, uint32_t out[3]) {
  shz_ctx* ctx = t->ctx;
  void* mx;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC0, 64, &mx));
  const uint32_t init[3] = {0u, 0u, 0xFFFFFFFFu};
  SHZ_HIP(ctx, shz_memcpy(ctx, mx, init, 12, hipMemcpyHostToDevice));
  if (n)
    hipLaunchKernelGGL(tbl_minmax_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 2048)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)t->ssid + lo, (const uint32_t*)t->soff + lo, n, (uint32_t*)mx);
  SHZ_HIP(ctx, hipGetLastError());
  SHZ_HIP(ctx, shz_memcpy(ctx, out, mx, 12, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}

// rows a run may hold when it is made (a debug limit forces several runs out of few rows: tests)
static uint64_t run_rows_limit(const shz_table* t) {
  const uint64_t hard = (1ull << 32) - 4096;
  return t->run_limit ? std::min<uint64_t>(t->run_limit, hard) : hard;
}

// The staged rows become sorted runs: one per block of block_rows[] rows if blocks are given, else one (cut into
// several when they outgrow what a run takes).  *packed = false when the packed form does not apply (ids + offsets wider
// than 32 bits, or the active segment holds rows): nothing was done, the caller takes the column path.
static int32_t seal_staged(shz_table* t, const uint64_t* block_rows, uint32_t n_blocks, bool* packed) {
  shz_ctx* ctx = t->ctx;
  *packed = false;
  if (t->n) return SHZ_OK;
  if (t->ns == 0) { *packed = true; return SHZ_OK; }
  ph_clock pc(t);
  uint32_t mm[3];
  SHZ_TRY(staged_minmax(t, 0, t->ns, mm));
  pc.lap(PH_MAXES);
  bool ok = false;
  SHZ_TRY(run_layout(t, std::max(t->max_sid, mm[0]), std::max(t->max_off, mm[1]), &ok));
  if (!ok) return SHZ_OK;
  // INSERT IGNORE against the rows of frozen segments (only where song-id ranges meet)
  SHZ_TRY(drop_staged_duplicates_of_frozen(t, mm[2], mm[0]));
  t->max_sid = std::max(t->max_sid, mm[0]);
  t->max_off = std::max(t->max_off, mm[1]);
  std::vector<uint64_t> blocks;
  if (n_blocks > 1) {
    uint64_t o = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) {
      const uint64_t n = std::min<uint64_t>(block_rows[b], t->ns - std::min(o, t->ns));   // (the anti-join may have shortened the rows)
      blocks.push_back(n);
      o += n;
    }
    if (o < t->ns) SHZ_FAIL(ctx, SHZ_E_INVALID, "runs cover %llu rows, %llu are staged", (unsigned long long)o, (unsigned long long)t->ns);
  } else {
    blocks.push_back(t->ns);
  }
  const uint64_t lim = run_rows_limit(t);
  uint64_t o = 0;
  for (uint64_t bn : blocks) {
    for (uint64_t done = 0; done < bn;) {
      const uint64_t n = std::min(bn - done, lim);
      uint32_t bm[3] = {mm[0], mm[1], mm[2]};
      if (blocks.size() > 1 || n < t->ns) SHZ_TRY(staged_minmax(t, o, n, bm));
      // a merge takes KW_MAXK runs: beyond that the smallest local ones are merged first (never runs that travelled with
      // runs that did not -- the exchange ships whole local runs)
      if (!t->hold_runs && t->runs.size() + 1 > KW_MAXK) SHZ_TRY(collapse_runs(t, KW_MAXK - 1));
      SHZ_TRY(seal_rows(t, t->skey + o, t->ssid + o, t->soff + o, n, bm[2], bm[0]));
      o += n;
      done += n;
    }
  }
  t->ns = 0;
  *packed = true;
  return SHZ_OK;
}

extern "C" int32_t shz_table_seal_run(shz_table* t) {
  if (!t) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  if (t->broken) SHZ_FAIL(ctx, SHZ_E_STATE, "table lost rows in a failed finalize");
  if (t->ns == 0) return SHZ_OK;
  bool packed = false;
  SHZ_TRY(seal_staged(t, nullptr, 0, &packed));
  if (!packed) {
    // rows in the active segment, or ids too wide to pack.  A table that holds its runs for a gathered build must not
    // put rows where the exchange cannot find them: the rows stay staged and the caller hears about it.
    if (t->hold_runs)
      SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "seal_run on a table reserved with SHZ_RESERVE_GATHER: %s; the rows stay staged (shz_table_allgather "
                                       "takes staged rows through the column path as long as no run has been sealed)",
               t->n ? "the table already holds rows" : "song id + offset need more than 32 bits");
    const uint64_t before = total_rows(t);
    const int32_t rc = shz_table_finalize(t);   // the column path, rows visible at once
    t->rows_cut += total_rows(t) - before;
    return rc;
  }
  if (t->hold_runs) return SHZ_OK;   // runs wait in the arena: one merge at finalize / allgather cuts segments by key range
  while (runs_rows(t) >= std::min<uint64_t>(t->seg_limit, (1ull << 32) - 4096)) {
    const uint64_t before = runs_rows(t);
    SHZ_TRY(flush_runs(t, false));
    t->rows_cut += before - runs_rows(t);
    if (runs_rows(t) >= before) break;
  }
  return SHZ_OK;
}

// ---------------------------------------------------------------------------------------- gathered build
// One table on every GPU from every rank's tracks (BASELINE configs[2], [4]).  The reference's only
// parallelism already pipelines -- the parent inserts a song while the pool fingerprints the next
// (__init__.py:341, 357-386, imap_unordered) -- and so does this: a rank's sealed runs travel on the communicator's
// exchange stream while its own stream fingerprints the next batch.
//
// An exchange ROUND (collective; gx_round):
//   1. every rank all-gathers one fixed block: its largest song id / offset so far, flags, and the row counts and
//      song-id ranges of up to GX_MAXR runs it has sealed and not yet sent (a blocking exchange of 320 bytes a rank on
//      the exchange stream: being in order, it also waits for the transfers of the round before);
//   2. all ranks derive the same verdicts from the same blocks: the packing layout (run_layout, pure in the global
//      maxima; runs packed narrower are repacked in place), whether anybody needs the column path, whether anybody has
//      more to send;
//   3. every rank places its peers' runs behind each other at the tail of its arena and queues the transfers
//      (shz_comm_allgather_lists_on: piece i of every rank's list in one RCCL group, each pair of GPUs on its own xGMI
//      link); its own runs stay where they were sealed.  The call returns with the transfers in flight.
// shz_table_exchange_run = seal + one round.  shz_table_allgather = seal + rounds until every rank is finishing and has
// nothing left to send, then ONE k-way merge over all runs (more than KW_MAXK: the smallest are merged first).  Ranks
// may call exchange_run different numbers of times: a rank that is already finishing keeps taking part in rounds.
// Every run is < 2^32 rows; a rank may hold any number of them (configs[4] at N = 2: 5.7e9 rows a rank).
#define GX_MAXR 16
#define GX_HDR 8
#define GX_BLOCK (GX_HDR + 2 * GX_MAXR)   // u64 words a rank contributes to a round
enum { GXF_GENERAL = 1, GXF_MORE = 2, GXF_FINISHING = 4, GXF_CUT = 8, GXF_HOLDS_RUNS = 16, GXF_STAGED = 32, GXF_BROKEN = 64 };

struct gx_verdict {
  bool any_general = false, any_cut = false, any_runs = false, any_more = false, all_finishing = true, any_broken = false;
  uint64_t rows_sent = 0;   // rows that travelled in this round (all ranks')
};

// pending: this rank has staged rows it will seal after this round (largest song id / offset among them: st_sid, st_off)
static int32_t gx_round(shz_table* t, shz_comm* c, bool finishing, bool local_general, bool local_failed, bool pending, uint32_t st_sid,
                        uint32_t st_off, gx_verdict* v) {
  shz_ctx* ctx = t->ctx;
  int rank = 0, nranks = 1;
  shz_comm_info(c, &rank, &nranks);
  hipStream_t xs;
  SHZ_TRY(shz_comm_exchange_stream(c, &xs));
  // 1) this rank's block
  std::vector<size_t> mine;   // indices of the runs this round ships
  size_t unsent = 0;
  bool holds = false;
  for (size_t i = 0; i < t->runs.size(); ++i) {
    holds |= t->runs[i].n != 0;
    if (t->runs[i].where != RUN_LOCAL) continue;
    ++unsent;
    if (mine.size() < GX_MAXR) mine.push_back(i);
  }
  uint64_t blk[GX_BLOCK];
  memset(blk, 0, sizeof(blk));
  blk[0] = unsent;
  blk[1] = std::max(t->max_sid, st_sid);
  blk[2] = std::max(t->max_off, st_off);
  blk[3] = total_rows(t);
  blk[4] = (local_general ? GXF_GENERAL : 0) | (unsent > mine.size() || pending ? GXF_MORE : 0) | (finishing ? GXF_FINISHING : 0) |
           (t->rows_cut ? GXF_CUT : 0) | (holds ? GXF_HOLDS_RUNS : 0) | (t->ns ? GXF_STAGED : 0) | (t->broken || local_failed ? GXF_BROKEN : 0);
  blk[5] = mine.size();
  for (size_t j = 0; j < mine.size(); ++j) {
    const shz_run& r = t->runs[mine[j]];
    blk[GX_HDR + 2 * j] = r.n;
    blk[GX_HDR + 2 * j + 1] = (uint64_t)r.sid_lo | ((uint64_t)r.sid_hi << 32);
  }
  std::vector<uint64_t> all((size_t)GX_BLOCK * nranks);
  const double w0 = now_s();
  SHZ_TRY(shz_comm_allgather_host(c, blk, all.data(), sizeof(blk)));
  t->gx_wait_s += now_s() - w0;
  ++t->gx_rounds;
  // 2) verdicts every rank derives alike
  *v = gx_verdict();
  uint64_t gmax_sid = 0, gmax_off = 0;
  for (int r = 0; r < nranks; ++r) {
    const uint64_t* b = &all[(size_t)GX_BLOCK * r];
    gmax_sid = std::max(gmax_sid, b[1]);
    gmax_off = std::max(gmax_off, b[2]);
    v->any_general |= (b[4] & GXF_GENERAL) != 0;
    v->any_cut |= (b[4] & GXF_CUT) != 0;
    v->any_runs |= (b[4] & GXF_HOLDS_RUNS) != 0;
    v->any_more |= (b[4] & GXF_MORE) != 0;
    v->any_broken |= (b[4] & GXF_BROKEN) != 0;
    v->all_finishing &= (b[4] & GXF_FINISHING) != 0;
    if (b[5] > GX_MAXR) SHZ_FAIL(ctx, SHZ_E_STATE, "exchange round: rank %d announces %llu runs", r, (unsigned long long)b[5]);
    for (uint64_t j = 0; j < b[5]; ++j) v->rows_sent += b[GX_HDR + 2 * j];
  }
  if (bits_for(gmax_sid) + bits_for(gmax_off) > 32) v->any_general = true;
  if (v->any_general || v->any_cut || v->any_broken) return SHZ_OK;   // nothing travels: the caller decides (alike on every rank)
  // 3) one layout: every run this rank holds is (re)packed in it before anything leaves
  bool ok = false;
  SHZ_TRY(run_layout(t, (uint32_t)gmax_sid, (uint32_t)gmax_off, &ok));
  if (!ok) SHZ_FAIL(ctx, SHZ_E_STATE, "exchange round: the agreed layout does not fit (song id %llu, offset %llu)", (unsigned long long)gmax_sid, (unsigned long long)gmax_off);
  t->max_sid = std::max<uint32_t>(t->max_sid, (uint32_t)gmax_sid);
  t->max_off = std::max<uint32_t>(t->max_off, (uint32_t)gmax_off);
  if (v->rows_sent == 0) return SHZ_OK;
  // 4) places for the peers' runs, then the transfers
  std::vector<shz_xfer> send;
  std::vector<std::vector<shz_xfer>> recv(nranks);
  std::vector<shz_run> arrivals;
  uint64_t at = (runs_end(t) + 31) & ~31ull, recv_rows = 0;
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) continue;
    const uint64_t* b = &all[(size_t)GX_BLOCK * r];
    for (uint64_t j = 0; j < b[5]; ++j) {
      const uint64_t n = b[GX_HDR + 2 * j];
      if (n >= (1ull << 32) - 4096) SHZ_FAIL(ctx, SHZ_E_STATE, "exchange round: rank %d announces a run of %llu rows", r, (unsigned long long)n);
      shz_run nr{at, n, (uint32_t)b[GX_HDR + 2 * j + 1], (uint32_t)(b[GX_HDR + 2 * j + 1] >> 32)};
      nr.where = RUN_RECV;
      arrivals.push_back(nr);
      at = (at + n + 31) & ~31ull;
      recv_rows += n;
    }
  }
  SHZ_TRY(rbuf_reserve(t, at));   // (a growing arena waits for the transfers in flight; a reserved one never grows)
  size_t ai = 0;
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) continue;
    const uint64_t* b = &all[(size_t)GX_BLOCK * r];
    for (uint64_t j = 0; j < b[5]; ++j, ++ai) recv[r].push_back(shz_xfer{t->rbuf + arrivals[ai].off, arrivals[ai].n * 8});
  }
  for (size_t i : mine) send.push_back(shz_xfer{t->rbuf + t->runs[i].off, t->runs[i].n * 8});
  // the exchange stream takes over once the seal (and a repack) on the context's stream is done
  if (!t->gx_ev) SHZ_HIP(ctx, hipEventCreateWithFlags(&t->gx_ev, hipEventDisableTiming));
  SHZ_HIP(ctx, hipEventRecord(t->gx_ev, ctx->stream));
  SHZ_HIP(ctx, hipStreamWaitEvent(xs, t->gx_ev, 0));
  t->gx_stream = xs;
  const double x0 = now_s();
  SHZ_TRY(shz_comm_allgather_lists_on(c, xs, send, recv));
  t->gx_xfer_s += now_s() - x0;
  for (size_t i : mine) t->runs[i].where = RUN_SENT;
  for (const shz_run& r : arrivals)
    if (r.n) t->runs.push_back(r);
  t->gx_recv_bytes += recv_rows * 8;
  return SHZ_OK;
}

// what a round's verdict means for a caller that wanted runs to travel
static int32_t gx_refuse(shz_table* t, const gx_verdict& v, const char* who) {
  shz_ctx* ctx = t->ctx;
  if (v.any_broken) SHZ_FAIL(ctx, SHZ_E_STATE, "%s: a rank failed to seal its rows, or its table lost rows in a failed finalize", who);
  if (v.any_cut)
    SHZ_FAIL(ctx, SHZ_E_STATE, "%s: a rank's seal_run moved rows into table segments before the exchange (they cannot travel any more): "
                               "reserve the tables of a gathered build with SHZ_RESERVE_GATHER, which keeps sealed runs in the arena", who);
  if (v.any_general && v.any_runs)
    SHZ_FAIL(ctx, SHZ_E_STATE, "%s: a rank needs the column path (its table holds rows, or song id + offset need more than 32 bits) "
                               "while sealed runs wait on some rank: the tables would differ between ranks", who);
  return SHZ_OK;
}

extern "C" int32_t shz_table_exchange_run(shz_table* t, shz_comm* c) {
  if (!t || !c) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  if (shz_comm_ctx(c) != ctx) SHZ_FAIL(ctx, SHZ_E_INVALID, "exchange_run: table and communicator belong to different contexts");
  t->hold_runs = true;
  // the round is entered whatever happened locally: a rank that fails alone would leave its peers waiting in theirs
  int32_t rc_local = SHZ_OK;
  bool local_general = t->n || !t->done.empty();
  if (!local_general && t->ns) {
    bool packed = false;
    rc_local = seal_staged(t, nullptr, 0, &packed);
    if (rc_local == SHZ_OK && !packed) local_general = true;
  }
  const std::string err_local = ctx->err;
  gx_verdict v;
  SHZ_TRY(gx_round(t, c, false, local_general, rc_local != SHZ_OK, false, 0, 0, &v));
  if (rc_local != SHZ_OK) { ctx->err = err_local; return rc_local; }
  SHZ_TRY(gx_refuse(t, v, "shz_table_exchange_run"));
  if (v.any_general)
    SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "shz_table_exchange_run: a rank's rows cannot travel as packed runs (its table holds rows, or song id + "
                                     "offset need more than 32 bits); stage everything and call shz_table_allgather");
  return SHZ_OK;
}

// the build without a communicator: the staged rows (n_runs_in blocks, what that many ranks would have staged) and the
// sealed runs -> table, one k-way merge
static int32_t build_from_runs(shz_table* t, const uint64_t* run_rows_in, uint32_t n_runs_in) {
  shz_ctx* ctx = t->ctx;
  t->bs_sort = t->bs_exchange = t->bs_merge = t->bs_segments = 0.0;
  uint32_t mm[3] = {0u, 0u, 0xFFFFFFFFu};
  if (t->ns) SHZ_TRY(staged_minmax(t, 0, t->ns, mm));
  for (const shz_run& r : t->runs)
    if (r.n) mm[0] = std::max(mm[0], r.sid_hi);
  const uint64_t off_max = std::max<uint64_t>(mm[1], t->runs.empty() ? 0 : t->max_off);   // (max_off covers every sealed run)
  if (t->n || !t->done.empty() || bits_for(mm[0]) + bits_for(off_max) > 32) return SHZ_I_GENERAL_PATH;
  if (t->ns + runs_rows(t) == 0) return shz_table_finalize(t);
  double t0 = now_s();
  if (t->ns) {
    bool packed = false;
    if (n_runs_in) {
      uint64_t sum = 0;
      for (uint32_t i = 0; i < n_runs_in; ++i) sum += run_rows_in[i];
      if (sum != t->ns) SHZ_FAIL(ctx, SHZ_E_INVALID, "runs cover %llu rows, %llu are staged", (unsigned long long)sum, (unsigned long long)t->ns);
    }
    SHZ_TRY(seal_staged(t, run_rows_in, n_runs_in, &packed));
    if (!packed) SHZ_FAIL(ctx, SHZ_E_STATE, "build_from_runs: rows could not be packed");
  }
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  t->bs_sort = now_s() - t0;
  t0 = now_s();
  const double m0 = t->ph[PH_KW_PLAN] + t->ph[PH_KW_MERGE], s0 = t->ph[PH_COL_ALLOC] + t->ph[PH_BUCKET];
  SHZ_TRY(flush_runs(t, true));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  t->bs_merge = t->ph[PH_KW_PLAN] + t->ph[PH_KW_MERGE] - m0;
  t->bs_segments = std::max(now_s() - t0 - t->bs_merge, t->ph[PH_COL_ALLOC] + t->ph[PH_BUCKET] - s0);
  return SHZ_OK;
}

// the exchange of unsorted columns followed by one sort of everything: tables that already hold rows, or ids / offsets
// too wide for the packed form.  Only STAGED rows travel (the caller has made sure no rank holds sealed runs).
static int32_t allgather_columns(shz_table* t, shz_comm* c, uint64_t* bytes_recv) {
  shz_ctx* ctx = t->ctx;
  int rank, nranks;
  shz_comm_info(c, &rank, &nranks);
  // 1) exchange staged-row counts
  void* d_cnt;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC2, 8ull * (nranks + 1), &d_cnt));
  uint64_t mine = t->ns;
  SHZ_HIP(ctx, shz_memcpy(ctx, d_cnt, &mine, 8, hipMemcpyHostToDevice));
  SHZ_TRY(shz_comm_allgather_bytes(c, d_cnt, (uint64_t*)d_cnt + 1, 8));
  std::vector<uint64_t> cnt(nranks);
  SHZ_HIP(ctx, shz_memcpy(ctx, cnt.data(), (uint64_t*)d_cnt + 1, 8ull * nranks, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t total = 0;
  std::vector<uint64_t> displ(nranks), bytes(nranks);
  for (int r = 0; r < nranks; ++r) {
    displ[r] = total * 4;
    bytes[r] = cnt[r] * 4;
    total += cnt[r];
  }
  if (bytes_recv) *bytes_recv = (total - mine) * 12;
  if (total == 0) return shz_table_finalize(t);
  // 2) per column: every rank's block lands at its displacement in the gathered column
  dev_cols gc;
  if (!gc.alloc(total)) SHZ_FAIL(ctx, SHZ_E_NOMEM, "allgather: hipMalloc(%llu) failed", (unsigned long long)(total * 4));
  const uint32_t* mine_cols[3] = {t->skey, t->ssid, t->soff};
  const double t0 = now_s();
  for (int i = 0; i < 3; ++i) SHZ_TRY(shz_comm_allgatherv_bytes(c, mine_cols[i], gc.p[i], bytes.data(), displ.data()));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  t->bs_sort = t->bs_merge = 0.0;
  t->bs_exchange = now_s() - t0;
  // 3) the gathered columns become the staged rows; finalize sorts + dedups them with the existing table
  if (t->stage_reserved) {
    uint32_t* st[3] = {t->skey, t->ssid, t->soff};
    for (int i = 0; i < 3; ++i) shz_block_free(ctx, st[i], t->st_bytes[i]);
  } else {
    void* olds[] = {t->skey, t->ssid, t->soff};
    for (void* p : olds)
      if (p) SHZ_HIP(ctx, hipFree(p));
  }
  t->skey = gc.take(0); t->ssid = gc.take(1); t->soff = gc.take(2);
  t->ns = t->scap = total;
  t->stage_reserved = false;
  const double t1 = now_s();
  const bool hold = t->hold_runs;
  t->hold_runs = false;   // (finalize of a table that holds rows: the column path proper)
  const int32_t rc = shz_table_finalize(t);
  t->hold_runs = hold;
  t->bs_segments = now_s() - t1;
  return rc;
}

extern "C" int32_t shz_table_allgather(shz_table* t, shz_comm* c, uint64_t* bytes_recv) {
  if (!t || !c) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  if (shz_comm_ctx(c) != ctx) SHZ_FAIL(ctx, SHZ_E_INVALID, "allgather: table and communicator belong to different contexts");
  if (bytes_recv) *bytes_recv = 0;
  static const bool force_cols = [] { const char* e = getenv("SHZ_ALLGATHER"); return e && !strcmp(e, "columns"); }();
  t->bs_sort = t->bs_exchange = t->bs_merge = t->bs_segments = 0.0;
  // 1) A first round with nothing sealed here yet: does ANY rank need the column path (its table holds rows, ids too
  //    wide, SHZ_ALLGATHER=columns)?  Then the staged rows must stay staged -- on every rank -- to travel as columns.  The
  //    block carries the staged rows' maxima, so the layout agreed in this round already covers them.
  double t0 = now_s();
  uint32_t mm[3] = {0u, 0u, 0xFFFFFFFFu};
  int32_t rc_local = SHZ_OK;
  if (t->ns) rc_local = staged_minmax(t, 0, t->ns, mm);
  const uint32_t lmax_sid = std::max(t->max_sid, mm[0]), lmax_off = std::max(t->max_off, mm[1]);
  const bool local_general = t->n || !t->done.empty() || force_cols || (t->ns && bits_for(lmax_sid) + bits_for(lmax_off) > 32);
  std::string err_local = ctx->err;
  gx_verdict v;
  SHZ_TRY(gx_round(t, c, true, local_general, rc_local != SHZ_OK, t->ns != 0, mm[0], mm[1], &v));
  // 2) the packed path: staged rows become runs (a local failure is carried into the next round: a rank that left
  //    alone would leave its peers waiting), then rounds until every rank is here and has sent everything
  if (!(v.any_general || v.any_cut || v.any_broken)) {
    if (rc_local == SHZ_OK && t->ns) {
      bool packed = false;
      rc_local = seal_staged(t, nullptr, 0, &packed);
      if (rc_local == SHZ_OK && !packed) { rc_local = SHZ_E_STATE; ctx->err = "allgather: staged rows could not be packed in the agreed layout"; }
      err_local = ctx->err;
    }
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    t->bs_sort = now_s() - t0;
    while (!(v.all_finishing && !v.any_more)) {   // (what the last round queued is the last of it)
      SHZ_TRY(gx_round(t, c, true, false, rc_local != SHZ_OK, false, 0, 0, &v));
      if (v.any_general || v.any_cut || v.any_broken) break;
    }
  }
  if (rc_local != SHZ_OK) { ctx->err = err_local; return rc_local; }
  SHZ_TRY(gx_refuse(t, v, "shz_table_allgather"));
  if (v.any_general) {
    t->rows_cut = 0;
    return allgather_columns(t, c, bytes_recv);   // (no rank holds a run: only staged rows exist, and they all travel)
  }
  // 3) every run is here (or on its way: flush_runs waits): one merge
  if (t->gx_stream) {
    const double w0 = now_s();
    SHZ_HIP(ctx, hipStreamSynchronize(t->gx_stream));
    t->gx_wait_s += now_s() - w0;
    t->gx_stream = nullptr;
  }
  if (bytes_recv) *bytes_recv = t->gx_recv_bytes;
  t->bs_exchange = t->gx_xfer_s + t->gx_wait_s;   // host seconds inside exchange rounds (waiting for peers included), not link time
  t->gx_recv_bytes = 0;
  t->gx_wait_s = t->gx_xfer_s = 0.0;
  t->rows_cut = 0;
  if (runs_rows(t) == 0) { t->runs.clear(); return shz_table_finalize(t); }
  t0 = now_s();
  const double m0 = t->ph[PH_KW_PLAN] + t->ph[PH_KW_MERGE];
  SHZ_TRY(flush_runs(t, true));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  t->bs_merge = t->ph[PH_KW_PLAN] + t->ph[PH_KW_MERGE] - m0;
  t->bs_segments = now_s() - t0 - t->bs_merge;
  return SHZ_OK;
}

extern "C" int32_t shz_table_finalize_runs(shz_table* t, const uint64_t* run_rows, uint32_t n_runs) {
  if (!t || (n_runs && !run_rows)) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  if (t->broken) SHZ_FAIL(ctx, SHZ_E_STATE, "table lost rows in a failed finalize");
  const int32_t rc = build_from_runs(t, run_rows, n_runs);
  return rc == SHZ_I_GENERAL_PATH ? shz_table_finalize(t) : rc;   // not an empty table / ids too wide
}

/* rows a sealed run may hold (0: the hard limit, 2^32 - 4096): small values force several runs out of few rows (tests) */
extern "C" int32_t shz_table_set_run_rows(shz_table* t, uint64_t rows) {
  if (!t) return SHZ_E_INVALID;
  if (rows && rows < 16) SHZ_FAIL(t->ctx, SHZ_E_INVALID, "run rows must be 0 or >= 16");
  t->run_limit = rows;
  return SHZ_OK;
}

extern "C" int32_t shz_table_exchange_stats(shz_table* t, uint64_t* rounds, uint64_t* bytes_recv, double* wait_s, uint32_t* runs_held) {
  if (!t) return SHZ_E_INVALID;
  if (rounds) *rounds = t->gx_rounds;
  if (bytes_recv) *bytes_recv = t->gx_recv_bytes;
  if (wait_s) *wait_s = t->gx_wait_s;
  if (runs_held) *runs_held = (uint32_t)t->runs.size();
  return SHZ_OK;
}

extern "C" int32_t shz_table_build_stats(shz_table* t, double* sort_s, double* exchange_s, double* merge_s, double* segments_s) {
  if (!t) return SHZ_E_INVALID;
  if (sort_s) *sort_s = t->bs_sort;
  if (exchange_s) *exchange_s = t->bs_exchange;
  if (merge_s) *merge_s = t->bs_merge;
  if (segments_s) *segments_s = t->bs_segments;
  return SHZ_OK;
}

static const char* const k_phase_names[] = {"stage_alloc", "insert", "dedup_frozen", "topup", "maxes", "sort", "merge", "uniq_scan",
                                            "column_alloc", "compact", "bucket", "slice", "stage_free", "run_pack", "run_sort", "run_uniq",
                                            "kway_plan", "kway_merge", "reserve_wait", "reserve_alloc_thread"};
static_assert(sizeof(k_phase_names) / sizeof(k_phase_names[0]) == PH_COUNT, "one name per phase");
extern "C" int32_t shz_table_phase_stats(shz_table* t, double* seconds, uint32_t cap, uint32_t* n, int32_t reset) {
  if (!t) return SHZ_E_INVALID;
  const uint32_t np = (uint32_t)(sizeof(k_phase_names) / sizeof(k_phase_names[0]));
  if (n) *n = np;
  if (seconds)
    for (uint32_t i = 0; i < np && i < cap; ++i) seconds[i] = t->ph[i];
  if (reset)
    for (double& x : t->ph) x = 0.0;
  return SHZ_OK;
}
extern "C" const char* shz_table_phase_name(uint32_t i) {
  return i < sizeof(k_phase_names) / sizeof(k_phase_names[0]) ? k_phase_names[i] : nullptr;
}

// ======================================================================================== key-sharded table
// when the replicated table no longer fits one GPU's HBM, rows are partitioned by a hash
// of the key.  A DB row lives on exactly one shard, so both quantities align_matches needs are additive over
// shards: dedup_hashes[sid] (rows matched, recognizer.py:261-264) and counts[(sid, delta)] (recognizer.py:305).
// Each shard probes its own rows and emits their votes packed in one agreed layout (shz_match_pairs), the votes
// travel (shz_pairs_allgather, 8 bytes each) and the normal tail of the match -- one sort, run lengths, per-group
// fold, top-n -- runs once over all of them (shz_pairs_vote).

__global__ void tbl_shard_flag_kernel(const uint32_t* __restrict__ key, uint64_t n, uint32_t nsh, uint32_t want,
                                      uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = shard_of(key[i], nsh) == want ? 1u : 0u;
}

extern "C" int32_t shz_shard_of_keys(const uint32_t* key32, uint64_t n, uint32_t nshards, uint32_t* shard_out) {
  if (!key32 || !shard_out || nshards == 0) return SHZ_E_INVALID;
  for (uint64_t i = 0; i < n; ++i) shard_out[i] = shard_of(key32[i], nshards);
  return SHZ_OK;
}

// compact the staged rows of shard `want` to (ok, os, oo); *cnt = how many
static int32_t stage_select_shard(shz_table* t, uint32_t nsh, uint32_t want, uint32_t* ok, uint32_t* os, uint32_t* oo,
                                  uint64_t* cnt) {
  shz_ctx* ctx = t->ctx;
  void *fl, *ps, *tot;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, t->ns * 4, &fl));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, t->ns * 4, &ps));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, 64, &tot));
  hipLaunchKernelGGL(tbl_shard_flag_kernel, dim3(nblk(t->ns)), dim3(256), 0, ctx->stream, (const uint32_t*)t->skey, t->ns,
                     nsh, want, (uint32_t*)fl);
  SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)fl, (uint32_t*)ps, t->ns, (uint64_t*)tot));
  hipLaunchKernelGGL(tbl_slice_scatter_kernel, dim3(nblk(t->ns)), dim3(256), 0, ctx->stream, (const uint32_t*)t->skey,
                     (const uint32_t*)t->ssid, (const uint32_t*)t->soff, (const uint32_t*)fl, (const uint32_t*)ps, t->ns, ok,
                     os, oo);
  SHZ_HIP(ctx, hipGetLastError());
  SHZ_HIP(ctx, shz_memcpy(ctx, cnt, tot, 8, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}

// one-pass partition of the staged rows by destination shard: pack (shard | key) with (sid | off) as payload, ONE
// stable radix pass on the shard bits, unpack -- instead of one compaction per destination
__global__ void tbl_shard_pack_kernel(const uint32_t* __restrict__ key, const uint32_t* __restrict__ sid,
                                      const uint32_t* __restrict__ off, uint64_t n, uint32_t nsh, uint64_t* __restrict__ k,
                                      uint64_t* __restrict__ v) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  k[i] = ((uint64_t)shard_of(key[i], nsh) << 32) | key[i];
  v[i] = ((uint64_t)sid[i] << 32) | off[i];
}
__global__ void tbl_shard_unpack_kernel(const uint64_t* __restrict__ k, const uint64_t* __restrict__ v, uint64_t n,
                                        uint32_t* __restrict__ ok, uint32_t* __restrict__ os, uint32_t* __restrict__ oo,
                                        unsigned long long* __restrict__ start /* [nsh]: first row of every shard */) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t kk = k[i], vv = v[i];
  ok[i] = (uint32_t)kk;
  os[i] = (uint32_t)(vv >> 32);
  oo[i] = (uint32_t)vv;
  const uint32_t sh = (uint32_t)(kk >> 32);
  if (i == 0 || (uint32_t)(k[i - 1] >> 32) != sh) start[sh] = i;
}

// staged rows of `t` -> (ok, os, oo) grouped by destination shard 0, 1, ...; cnt[d] rows go to shard d
static int32_t stage_partition(shz_table* t, uint32_t nsh, uint32_t* ok, uint32_t* os, uint32_t* oo, std::vector<uint64_t>& cnt) {
  shz_ctx* ctx = t->ctx;
  const uint64_t ns = t->ns;
  cnt.assign(nsh, 0);
  if (ns == 0) return SHZ_OK;
  void *k0, *k1, *v0, *v1, *st;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_A, ns * 8, &k0));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_B, ns * 8, &k1));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_C, ns * 8, &v0));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_D, ns * 8, &v1));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, 8ull * nsh, &st));
  SHZ_HIP(ctx, hipMemsetAsync(st, 0xFF, 8ull * nsh, ctx->stream));
  hipLaunchKernelGGL(tbl_shard_pack_kernel, dim3(nblk(ns)), dim3(256), 0, ctx->stream, (const uint32_t*)t->skey,
                     (const uint32_t*)t->ssid, (const uint32_t*)t->soff, ns, nsh, (uint64_t*)k0, (uint64_t*)v0);
  SHZ_HIP(ctx, hipGetLastError());
  int sel = 0;
  if (nsh > 1) SHZ_TRY(shz_sort_u64(ctx, (uint64_t*)k0, (uint64_t*)k1, v0, v1, 8, ns, 32, 32 + bits_for(nsh - 1), &sel));
  hipLaunchKernelGGL(tbl_shard_unpack_kernel, dim3(nblk(ns)), dim3(256), 0, ctx->stream,
                     (const uint64_t*)(sel ? k1 : k0), (const uint64_t*)(sel ? v1 : v0), ns, ok, os, oo, (unsigned long long*)st);
  SHZ_HIP(ctx, hipGetLastError());
  std::vector<uint64_t> start(nsh);
  SHZ_HIP(ctx, shz_memcpy(ctx, start.data(), st, 8ull * nsh, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t next = ns;  // shards without rows start where the next one does
  for (int d = (int)nsh - 1; d >= 0; --d) {
    if (start[d] == ~0ull) start[d] = next;
    cnt[d] = next - start[d];
    next = start[d];
  }
  return SHZ_OK;
}

extern "C" int32_t shz_table_keep_shard(shz_table* t, uint32_t shard, uint32_t nshards) {
  if (!t) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  if (nshards == 0 || shard >= nshards) SHZ_FAIL(ctx, SHZ_E_INVALID, "keep_shard: shard %u of %u", shard, nshards);
  if (t->ns == 0 || nshards == 1) return SHZ_OK;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  dev_cols g;
  if (!g.alloc(t->ns)) SHZ_FAIL(ctx, SHZ_E_NOMEM, "keep_shard: hipMalloc(%llu) failed", (unsigned long long)(t->ns * 4));
  std::vector<uint64_t> cnt;
  SHZ_TRY(stage_partition(t, nshards, g.p[0], g.p[1], g.p[2], cnt));   // the same partition the exchange uses
  uint64_t first = 0;
  for (uint32_t d = 0; d < shard; ++d) first += cnt[d];
  const uint64_t kept = cnt[shard];
  uint32_t* dst[3] = {t->skey, t->ssid, t->soff};                      // the staged columns are big enough
  for (int i = 0; i < 3; ++i)
    if (kept) SHZ_HIP(ctx, shz_memcpy(ctx, dst[i], g.p[i] + first, kept * 4, hipMemcpyDeviceToDevice));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  t->ns = kept;
  return SHZ_OK;
}

extern "C" int32_t shz_table_shard_exchange(shz_table* t, shz_comm* c, uint64_t* bytes_recv) {
  if (!t || !c) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  int rank, nranks;
  shz_comm_info(c, &rank, &nranks);
  // 1) partition the staged rows by destination: the send columns hold the blocks for rank 0, 1, ... back to back
  const uint64_t ns = t->ns;
  dev_cols sndc, rcvc;
  if (!sndc.alloc(ns)) SHZ_FAIL(ctx, SHZ_E_NOMEM, "shard exchange: hipMalloc(%llu) failed", (unsigned long long)(ns * 4));
  uint32_t** snd = sndc.p;
  std::vector<uint64_t> scnt, sdis(nranks, 0);
  SHZ_TRY(stage_partition(t, (uint32_t)nranks, snd[0], snd[1], snd[2], scnt));
  for (int d = 1; d < nranks; ++d) sdis[d] = sdis[d - 1] + scnt[d - 1];
  // 2) everyone learns the whole count matrix: row r = what rank r sends to each destination
  void* d_cnt;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC2, 8ull * nranks * (nranks + 1), &d_cnt));
  SHZ_HIP(ctx, shz_memcpy(ctx, d_cnt, scnt.data(), 8ull * nranks, hipMemcpyHostToDevice));
  SHZ_TRY(shz_comm_allgather_bytes(c, d_cnt, (uint64_t*)d_cnt + nranks, 8ull * nranks));
  std::vector<uint64_t> mat((size_t)nranks * nranks);
  SHZ_HIP(ctx, shz_memcpy(ctx, mat.data(), (uint64_t*)d_cnt + nranks, 8ull * nranks * nranks, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<uint64_t> rcnt(nranks), rdis(nranks);
  uint64_t total = 0;
  for (int r = 0; r < nranks; ++r) {
    rcnt[r] = mat[(size_t)r * nranks + rank];
    rdis[r] = total;
    total += rcnt[r];
  }
  if (bytes_recv) *bytes_recv = (total - rcnt[rank]) * 12;
  // 3) one grouped all-to-all per column
  if (!rcvc.alloc(total)) SHZ_FAIL(ctx, SHZ_E_NOMEM, "shard exchange: hipMalloc(%llu) failed", (unsigned long long)(total * 4));
  uint32_t** rcv = rcvc.p;
  std::vector<uint64_t> sb(nranks), sd(nranks), rb(nranks), rd(nranks);
  for (int r = 0; r < nranks; ++r) { sb[r] = scnt[r] * 4; sd[r] = sdis[r] * 4; rb[r] = rcnt[r] * 4; rd[r] = rdis[r] * 4; }
  for (int i = 0; i < 3; ++i) SHZ_TRY(shz_comm_alltoallv_bytes(c, snd[i], sb.data(), sd.data(), rcv[i], rb.data(), rd.data()));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // 4) the received rows are this rank's shard: they replace the staged rows
  void* olds[] = {t->skey, t->ssid, t->soff};
  for (void* p : olds)
    if (p) SHZ_HIP(ctx, hipFree(p));
  t->skey = rcvc.take(0); t->ssid = rcvc.take(1); t->soff = rcvc.take(2);  // the send columns go with sndc
  t->stage_reserved = false;
  t->ns = total;
  t->scap = std::max<uint64_t>(total, 1);
  SHZ_TRY(shz_table_finalize(t));
  // every shard packs its votes in one layout: agree on the largest song id / offset of the whole table
  uint32_t mx[2] = {t->max_sid, t->max_off};
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC2, 8ull * (nranks + 1), &d_cnt));
  SHZ_HIP(ctx, shz_memcpy(ctx, d_cnt, mx, 8, hipMemcpyHostToDevice));
  SHZ_TRY(shz_comm_allgather_bytes(c, d_cnt, (uint64_t*)d_cnt + 1, 8));
  std::vector<uint32_t> all(2 * (size_t)nranks);
  SHZ_HIP(ctx, shz_memcpy(ctx, all.data(), (uint64_t*)d_cnt + 1, 8ull * nranks, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int r = 0; r < nranks; ++r) {
    t->max_sid = std::max(t->max_sid, all[2 * r]);
    t->max_off = std::max(t->max_off, all[2 * r + 1]);
  }
  return SHZ_OK;
}

// append the STAGED rows of `src` that belong to `shard` to the staged rows of `dst` (src is left untouched):
// one staging table feeds several shard tables on the same GPU
extern "C" int32_t shz_table_stage_from(shz_table* dst, shz_table* src, uint32_t shard, uint32_t nshards) {
  if (!dst || !src) return SHZ_E_INVALID;
  shz_ctx* ctx = dst->ctx;
  if (src->ctx != ctx || src == dst) SHZ_FAIL(ctx, SHZ_E_INVALID, "stage_from: tables must differ and share a context");
  if (nshards == 0 || shard >= nshards) SHZ_FAIL(ctx, SHZ_E_INVALID, "stage_from: shard %u of %u", shard, nshards);
  if (src->ns == 0) return SHZ_OK;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  SHZ_TRY(stage_reserve(dst, src->ns));  // upper bound; the selection writes behind dst's staged rows
  uint64_t k = 0;
  SHZ_TRY(stage_select_shard(src, nshards, shard, dst->skey + dst->ns, dst->ssid + dst->ns, dst->soff + dst->ns, &k));
  dst->ns += k;
  return SHZ_OK;
}

extern "C" int32_t shz_table_clear_staged(shz_table* t) {
  if (!t) return SHZ_E_INVALID;
  t->ns = 0;
  return SHZ_OK;
}

