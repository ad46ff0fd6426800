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
  void* ps[] = {t->key, t->sid, t->off, t->skey, t->ssid, t->soff, t->bucket};
  for (void* p : ps)
    if (p) (void)hipFree(p);
  for (shz_seg& g : t->done) {
    void* qs[] = {g.key, g.sid, g.off, g.bucket};
    for (void* p : qs)
      if (p) (void)hipFree(p);
  }
  delete t;
  return SHZ_OK;
}

static int32_t stage_reserve(shz_table* t, uint64_t extra) {
  shz_ctx* ctx = t->ctx;
  const uint64_t need = t->ns + extra;
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
static int32_t finalize_active(shz_table* t, const uint32_t* skey, const uint32_t* ssid, const uint32_t* soff, uint64_t ns) {
  shz_ctx* ctx = t->ctx;
  const uint64_t total = t->n + ns;
  if (total >= (1ull << 32)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "segment limited to < 2^32 rows (have %llu)", (unsigned long long)total);
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
      void* olds[] = {t->key, t->sid, t->off};
      for (void* p : olds)
        if (p) (void)hipFree(p);
      t->key = t->sid = t->off = nullptr;
      t->cap = 0;
      t->n = 0;
      if (!fresh.alloc(want)) {
        t->broken = true;
        SHZ_FAIL(ctx, SHZ_E_NOMEM, "table: hipMalloc of %llu rows failed after the active segment was released; "
                                   "the table lost rows and refuses further use", (unsigned long long)want);
      }
    } else {
      void* olds[] = {t->key, t->sid, t->off};
      for (void* p : olds)
        if (p) SHZ_HIP(ctx, hipFree(p));
    }
    t->key = fresh.take(0);
    t->sid = fresh.take(1);
    t->off = fresh.take(2);
    t->cap = want;
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
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  pc.lap(PH_COMPACT);
  t->n = nu;
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
static int32_t rebuild_buckets(shz_ctx* ctx, uint32_t* key, uint64_t n, uint32_t** bucket, uint64_t* nbuckets, uint64_t* bcap) {
  if (n == 0) { *nbuckets = 0; return SHZ_OK; }
  uint32_t last_key = 0;
  SHZ_HIP(ctx, shz_memcpy(ctx, &last_key, key + (n - 1), 4, hipMemcpyDeviceToHost));
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
      SHZ_TRY(rebuild_buckets(ctx, g.key, g.n, &g.bucket, &g.nbuckets, &bcap));
    }
  }
  // frozen segments that became empty disappear
  for (size_t i = t->done.size(); i-- > 0;)
    if (t->done[i].n == 0) {
      void* qs[] = {t->done[i].key, t->done[i].sid, t->done[i].off, t->done[i].bucket};
      for (void* p : qs)
        if (p) (void)hipFree(p);
      t->done.erase(t->done.begin() + (long)i);
    }
  if (t->n) {
    uint64_t kept = t->n;
    SHZ_TRY(purge(t->key, t->sid, t->off, t->n, &kept));
    if (kept != t->n) {
      gone += t->n - kept;
      t->n = kept;
      SHZ_TRY(rebuild_buckets(ctx, t->key, t->n, &t->bucket, &t->nbuckets, &t->bcap));
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
  for (shz_seg& g : t->done) {
    void* qs[] = {g.key, g.sid, g.off, g.bucket};
    for (void* p : qs)
      if (p) (void)hipFree(p);
  }
  t->done.clear();
  t->n = 0;        // the active and staging columns keep their allocations for the rows to come
  t->nbuckets = 0;
  t->ns = 0;
  t->max_sid = t->max_off = 0;
  t->broken = false;
  return SHZ_OK;
}

// INSERT IGNORE across segments: staged rows that already sit in a frozen segment are dropped before they are merged
static int32_t drop_staged_duplicates_of_frozen(shz_table* t) {
  shz_ctx* ctx = t->ctx;
  if (t->done.empty() || t->ns == 0) return SHZ_OK;
  void* fl;
  ph_clock pc(t);
  struct lap_on_exit { ph_clock& c; ~lap_on_exit() { c.lap(PH_DEDUP_FROZEN); } } loe{pc};
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, t->ns * 4, &fl));
  hipLaunchKernelGGL(tbl_ones_kernel, dim3((unsigned)((t->ns + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)fl, t->ns);
  for (const shz_seg& g : t->done) {
    if (!g.n) continue;
    shz_seg_dev gd{g.key, g.sid, g.off, g.bucket, (uint32_t)g.n, g.nbuckets};
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
  t->done.push_back(shz_seg{t->key, t->sid, t->off, t->bucket, t->n, t->nbuckets});
  t->key = t->sid = t->off = t->bucket = nullptr;
  t->n = t->nbuckets = 0;
  t->cap = t->bcap = 0;
}

extern "C" int32_t shz_table_finalize(shz_table* t) {
  if (!t) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  if (t->broken) SHZ_FAIL(ctx, SHZ_E_STATE, "table lost rows in a failed finalize");
  if (t->ns && t->n && t->n + t->ns > t->seg_limit) {
    // The staged rows do not fit the active segment.  Before it is frozen it is topped up with the slices (by key) of
    // the staged rows that still fit: segments then hold ~seg_limit rows instead of whatever multiple of the ingest
    // batch fell below it (1.13e9-row batches against 2^31 left every segment half empty: 11 segments for 1.15e10 rows
    // instead of 6 -- and every query hash is looked up in every segment).
    const uint32_t S = 64;
    const uint32_t m = (uint32_t)std::min<uint64_t>(S - 1, (t->seg_limit - t->n) * S / t->ns);
    if (m >= S / 16) {
      SHZ_TRY(drop_staged_duplicates_of_frozen(t));
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
          SHZ_TRY(finalize_active(t, (const uint32_t*)ak, (const uint32_t*)as, (const uint32_t*)ao, na));
        }
      }
    }
    freeze_active(t);   // what is still staged starts new segment(s)
  }
  SHZ_TRY(drop_staged_duplicates_of_frozen(t));                  // UNIQUE(song_id, offset, hash) across segments
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
    SHZ_TRY(finalize_active(t, t->skey, t->ssid, t->soff, t->ns));
  } else {   // t->n == 0 here: the active segment was frozen above
    const uint32_t nsl = (uint32_t)((t->ns + t->seg_limit - 1) / t->seg_limit) + (t->ns > t->seg_limit ? 1 : 0);
    if (t->done.size() + nsl > SHZ_MAX_SEGS) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "more than %d table segments", SHZ_MAX_SEGS);
    for (uint32_t sl = 0; sl < nsl; ++sl) {
      if (nsl == 1) {
        SHZ_TRY(finalize_active(t, t->skey, t->ssid, t->soff, t->ns));
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
        SHZ_TRY(finalize_active(t, (const uint32_t*)ck, (const uint32_t*)cs, (const uint32_t*)co, cnt));
      }
      if (sl + 1 < nsl) freeze_active(t);
    }
  }
  // The staging columns stay allocated for the next batch (a stream of ingest batches otherwise pays three
  // hipMalloc + growth copies per batch) unless they hold more than a quarter of what is free now.
  size_t mem_free = 0, mem_total = 0;
  ph_clock pcf(t);
  SHZ_HIP(ctx, hipMemGetInfo(&mem_free, &mem_total));
  if (t->scap * 12 > mem_free / 4) {
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
  if (n_staged) *n_staged = t->ns;
  return SHZ_OK;
}

extern "C" int32_t shz_table_export(shz_table* t, uint32_t* key32, uint32_t* sid, uint32_t* off, uint64_t cap,
                                    uint64_t* count) {
  if (!t) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  const uint64_t nrows = total_rows(t);
  if (count) *count = nrows;
  if (t->ns) SHZ_FAIL(ctx, SHZ_E_STATE, "table has %llu staged rows; call shz_table_finalize first", (unsigned long long)t->ns);
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
  if (t->ns) SHZ_FAIL(ctx, SHZ_E_STATE, "table has staged rows; call shz_table_finalize first");
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
  if (t->ns || (!t->bucket && t->done.empty())) SHZ_FAIL(ctx, SHZ_E_STATE, "table not finalized");
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

// ---------------------------------------------------------------------------------------- all-gather build
// ---- building a table from SORTED RUNS ("every rank merges 8 sorted runs") ---------------------------------
// Rows travel and merge in the packed form key << (sb + ob) | sid << ob | off (8 bytes a row instead of 12; its order is
// the table's order), which needs sid and offset to fit 32 bits together -- true for every configuration of BASELINE
// (1M songs = 20 bits, 3-minute tracks = 12 bits).

#define SHZ_I_GENERAL_PATH 1   // internal: the packed sorted-run path does not apply, take the column path

// flag[i] = 1 where c[i] differs from its predecessor (prev = the element before c[0], if the chunk has one)
__global__ void tbl_uniq1_chunk_kernel(const uint64_t* __restrict__ c, uint64_t n, bool has_prev, uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = ((i == 0 && !has_prev) || c[i] != c[(int64_t)i - 1]) ? 1u : 0u;
}

// pack + sort rows [0, n) of three columns into `dst` (scratch `tmp`, both n entries)
static int32_t pack_sort_run(shz_ctx* ctx, const uint32_t* key, const uint32_t* sid, const uint32_t* off, uint64_t n, int sb,
                             int ob, uint64_t* dst, uint64_t* tmp) {
  if (n == 0) return SHZ_OK;
  if (n >= (1ull << 32)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "a run is limited to < 2^32 rows (have %llu)", (unsigned long long)n);
  hipLaunchKernelGGL(tbl_compose1_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, ctx->stream,
                     key, sid, off, n, (uint64_t)0, sb, ob, dst);
  SHZ_HIP(ctx, hipGetLastError());
  int sel = 0;
  SHZ_TRY(shz_sort_u64(ctx, dst, tmp, nullptr, nullptr, 0, n, 0, 32 + sb + ob, &sel));
  if (sel) SHZ_HIP(ctx, hipMemcpyAsync(dst, tmp, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
  return SHZ_OK;
}

// merge the sorted runs run_off[r] .. run_off[r+1] of `a` pairwise, ping-ponging with `b`, until one run is left;
// *out = the buffer that holds it
static int32_t merge_runs(shz_ctx* ctx, uint64_t* a, uint64_t* b, std::vector<uint64_t> run_off, uint64_t** out) {
  while (run_off.size() > 2) {
    std::vector<uint64_t> next{0};
    const size_t nr = run_off.size() - 1;
    for (size_t r = 0; r < nr; r += 2) {
      const uint64_t o0 = run_off[r], o1 = run_off[r + 1], o2 = r + 2 <= nr ? run_off[r + 2] : o1;
      if (r + 1 == nr) {  // odd run out: carried over
        if (o1 > o0) SHZ_HIP(ctx, hipMemcpyAsync(b + o0, a + o0, (o1 - o0) * 8, hipMemcpyDeviceToDevice, ctx->stream));
        next.push_back(o1);
      } else {
        const uint64_t tot = o2 - o0;
        if (tot) {
          if ((tot + MERGE_TILE - 1) / MERGE_TILE >= (1ull << 31)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "merge of %llu rows", (unsigned long long)tot);
          hipLaunchKernelGGL(tbl_merge_kernel, dim3((unsigned)((tot + MERGE_TILE - 1) / MERGE_TILE)), dim3(256), 0, ctx->stream,
                             (const uint64_t*)(a + o0), o1 - o0, (const uint64_t*)(a + o1), o2 - o1, b + o0);
          SHZ_HIP(ctx, hipGetLastError());
        }
        next.push_back(o2);
      }
    }
    std::swap(a, b);
    run_off.swap(next);
  }
  *out = a;
  return SHZ_OK;
}

// an EMPTY table takes one sorted packed run of `total` rows: duplicates dropped, cut into segments of <= seg_limit rows
static int32_t segments_from_sorted(shz_table* t, const uint64_t* g, uint64_t total, int sb, int ob) {
  shz_ctx* ctx = t->ctx;
  const uint64_t L = std::min<uint64_t>(t->seg_limit, (1ull << 32) - 4096);
  if ((total + L - 1) / L > SHZ_MAX_SEGS) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "more than %d table segments", SHZ_MAX_SEGS);
  for (uint64_t o = 0; o < total; o += L) {
    const uint64_t n = std::min(L, total - o);
    void *fl, *ps, *tot;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, n * 4, &fl));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, n * 4, &ps));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, 64, &tot));
    hipLaunchKernelGGL(tbl_uniq1_chunk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, g + o, n, o > 0,
                       (uint32_t*)fl);
    SHZ_HIP(ctx, hipGetLastError());
    SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)fl, (uint32_t*)ps, n, (uint64_t*)tot));
    uint64_t nu = 0;
    SHZ_HIP(ctx, shz_memcpy(ctx, &nu, tot, 8, hipMemcpyDeviceToHost));
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (nu == 0) continue;
    if (t->n) freeze_active(t);   // the previous chunk's segment
    dev_cols cols;
    if (!cols.alloc(nu)) SHZ_FAIL(ctx, SHZ_E_NOMEM, "table: hipMalloc of %llu rows failed", (unsigned long long)nu);
    t->key = cols.take(0); t->sid = cols.take(1); t->off = cols.take(2);
    t->cap = nu;
    hipLaunchKernelGGL(tbl_compact1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, g + o,
                       (const uint32_t*)fl, (const uint32_t*)ps, n, sb, ob, t->key, t->sid, t->off);
    SHZ_HIP(ctx, hipGetLastError());
    t->n = nu;
    SHZ_TRY(rebuild_buckets(ctx, t->key, t->n, &t->bucket, &t->nbuckets, &t->bcap));
  }
  return SHZ_OK;
}


// staged rows = n_runs consecutive blocks of run_rows[r] rows: sort every block, merge the runs, build the segments.
// With `c`, the blocks are the ranks' staged rows and travel between the sort and the merge.
static int32_t build_from_runs(shz_table* t, shz_comm* c, const uint64_t* run_rows_in, uint32_t n_runs_in, uint64_t* bytes_recv) {
  shz_ctx* ctx = t->ctx;
  int rank = 0, nranks = 1;
  if (c) shz_comm_info(c, &rank, &nranks);
  t->bs_sort = t->bs_exchange = t->bs_merge = t->bs_segments = 0.0;
  // maxima of this rank's staged rows
  void* mx;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC0, 64 + 24ull * (nranks + 1), &mx));
  SHZ_HIP(ctx, hipMemsetAsync(mx, 0, 64, ctx->stream));
  if (t->ns)
    hipLaunchKernelGGL(tbl_max_kernel, dim3((unsigned)std::min<uint64_t>((t->ns + 255) / 256, 2048)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)t->ssid, (const uint32_t*)t->soff, t->ns, (uint32_t*)mx);
  SHZ_HIP(ctx, hipGetLastError());
  uint32_t maxes[2];
  SHZ_HIP(ctx, shz_memcpy(ctx, maxes, mx, 8, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // counts and maxima of every rank
  std::vector<uint64_t> info(3 * (size_t)nranks, 0);
  info[3 * (size_t)rank] = t->ns; info[3 * (size_t)rank + 1] = maxes[0]; info[3 * (size_t)rank + 2] = maxes[1];
  if (c && nranks > 1) {
    uint64_t* d_info = (uint64_t*)((char*)mx + 64);
    SHZ_HIP(ctx, shz_memcpy(ctx, d_info, &info[3 * (size_t)rank], 24, hipMemcpyHostToDevice));
    SHZ_TRY(shz_comm_allgather_bytes(c, d_info, d_info + 3, 24));
    SHZ_HIP(ctx, shz_memcpy(ctx, info.data(), d_info + 3, 24ull * nranks, hipMemcpyDeviceToHost));
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  uint64_t total = 0, gmax_sid = 0, gmax_off = 0;
  std::vector<uint64_t> cnt(nranks), displ(nranks);
  for (int r = 0; r < nranks; ++r) {
    cnt[r] = info[3 * (size_t)r];
    displ[r] = total;
    total += cnt[r];
    gmax_sid = std::max(gmax_sid, info[3 * (size_t)r + 1]);
    gmax_off = std::max(gmax_off, info[3 * (size_t)r + 2]);
  }
  if (bytes_recv) *bytes_recv = 0;
  const int sb = bits_for(gmax_sid), ob = bits_for(gmax_off);
  static const bool force_cols = [] { const char* e = getenv("SHZ_ALLGATHER"); return e && !strcmp(e, "columns"); }();
  if (total == 0) return shz_table_finalize(t);
  if (sb + ob > 32 || t->n || !t->done.empty() || force_cols) return SHZ_I_GENERAL_PATH;
  // 1) local runs: the staged rows of this rank as sorted packed runs, back to back
  std::vector<uint64_t> my_runs;
  if (c) my_runs.push_back(t->ns);
  else my_runs.assign(run_rows_in, run_rows_in + n_runs_in);
  uint64_t sum = 0;
  for (uint64_t r : my_runs) sum += r;
  if (sum != t->ns) SHZ_FAIL(ctx, SHZ_E_INVALID, "runs cover %llu rows, %llu are staged", (unsigned long long)sum, (unsigned long long)t->ns);
  double t0 = now_s();
  void *pa, *pb;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_C, std::max<uint64_t>(t->ns, 1) * 8, &pa));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_D, std::max<uint64_t>(t->ns, 1) * 8, &pb));
  uint64_t o = 0;
  for (uint64_t r : my_runs) {
    SHZ_TRY(pack_sort_run(ctx, t->skey + o, t->ssid + o, t->soff + o, r, sb, ob, (uint64_t*)pa + o, (uint64_t*)pb + o));
    o += r;
  }
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  t->bs_sort = now_s() - t0;
  // the staged columns have done their work: release them before the gathered buffers are allocated
  void* olds[] = {t->skey, t->ssid, t->soff};
  for (void* p : olds)
    if (p) SHZ_HIP(ctx, hipFree(p));
  t->skey = t->ssid = t->soff = nullptr;
  const uint64_t ns_local = t->ns;
  t->ns = t->scap = 0;
  // 2) exchange: every rank's run(s) into one buffer at the rank's displacement
  uint64_t *ga = nullptr, *gb = nullptr;
  struct guard { uint64_t** p[2]; ~guard() { for (auto q : p) if (*q) (void)hipFree(*q); } } gd{{&ga, &gb}};
  std::vector<uint64_t> run_off{0};
  t0 = now_s();
  if (c && nranks > 1) {
    if (hipMalloc(&ga, total * 8) != hipSuccess) SHZ_FAIL(ctx, SHZ_E_NOMEM, "allgather: hipMalloc(%llu) failed", (unsigned long long)(total * 8));
    std::vector<uint64_t> bcnt(nranks), bdis(nranks);
    for (int r = 0; r < nranks; ++r) { bcnt[r] = cnt[r] * 8; bdis[r] = displ[r] * 8; run_off.push_back(displ[r] + cnt[r]); }
    SHZ_TRY(shz_comm_allgatherv_bytes(c, pa, ga, bcnt.data(), bdis.data()));
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (bytes_recv) *bytes_recv = (total - ns_local) * 8;
  } else {
    if (hipMalloc(&ga, total * 8) != hipSuccess) SHZ_FAIL(ctx, SHZ_E_NOMEM, "hipMalloc(%llu) failed", (unsigned long long)(total * 8));
    SHZ_HIP(ctx, hipMemcpyAsync(ga, pa, total * 8, hipMemcpyDeviceToDevice, ctx->stream));
    uint64_t acc = 0;
    for (uint64_t r : my_runs) { acc += r; run_off.push_back(acc); }
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  t->bs_exchange = now_s() - t0;
  // 3) merge the runs
  t0 = now_s();
  uint64_t* g = ga;
  if (run_off.size() > 2) {
    if (hipMalloc(&gb, total * 8) != hipSuccess) SHZ_FAIL(ctx, SHZ_E_NOMEM, "merge: hipMalloc(%llu) failed", (unsigned long long)(total * 8));
    SHZ_TRY(merge_runs(ctx, ga, gb, run_off, &g));
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t** other = (g == ga) ? &gb : &ga;   // the buffer that does not hold the result goes before the columns come
    (void)hipFree(*other);
    *other = nullptr;
  }
  t->bs_merge = now_s() - t0;
  // 4) segments
  t0 = now_s();
  t->max_sid = std::max<uint32_t>(t->max_sid, (uint32_t)gmax_sid);
  t->max_off = std::max<uint32_t>(t->max_off, (uint32_t)gmax_off);
  SHZ_TRY(segments_from_sorted(t, g, total, sb, ob));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  t->bs_segments = now_s() - t0;
  return SHZ_OK;
}

// the exchange of unsorted columns followed by one sort of everything: tables that already hold rows, or ids / offsets
// too wide for the packed form
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
  void* olds[] = {t->skey, t->ssid, t->soff};
  for (void* p : olds)
    if (p) SHZ_HIP(ctx, hipFree(p));
  t->skey = gc.take(0); t->ssid = gc.take(1); t->soff = gc.take(2);
  t->ns = t->scap = total;
  const double t1 = now_s();
  const int32_t rc = shz_table_finalize(t);
  t->bs_segments = now_s() - t1;
  return rc;
}

extern "C" int32_t shz_table_allgather(shz_table* t, shz_comm* c, uint64_t* bytes_recv) {
  if (!t || !c) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  if (t->broken) SHZ_FAIL(ctx, SHZ_E_STATE, "table lost rows in a failed finalize");
  const int32_t rc = build_from_runs(t, c, nullptr, 0, bytes_recv);
  return rc == SHZ_I_GENERAL_PATH ? allgather_columns(t, c, bytes_recv) : rc;
}

extern "C" int32_t shz_table_finalize_runs(shz_table* t, const uint64_t* run_rows, uint32_t n_runs) {
  if (!t || (n_runs && !run_rows)) return SHZ_E_INVALID;
  shz_ctx* ctx = t->ctx;
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  if (t->broken) SHZ_FAIL(ctx, SHZ_E_STATE, "table lost rows in a failed finalize");
  const int32_t rc = build_from_runs(t, nullptr, run_rows, n_runs, nullptr);
  return rc == SHZ_I_GENERAL_PATH ? shz_table_finalize(t) : rc;   // not an empty table / ids too wide
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
                                            "column_alloc", "compact", "bucket", "slice", "stage_free"};
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

