// The match / align path on the HBM-resident fingerprint table (storage and build: shz_build.hip).
//   shz_match_batch replaces return_matches + align_matches (recognizer.py:222-338)
#include "shz_table_int.h"

// ======================================================================================== match
#define QOFF_BITS 20
#define QKEY_SHIFT 20
#define QIDX_SHIFT 52
#define MAX_Q_SUB 4096

__global__ void m_compose_kernel(const uint32_t* __restrict__ key32, const uint32_t* __restrict__ q_off,
                                 const uint64_t* __restrict__ query_off /*sub-batch CSR, nq+1*/, uint32_t nq, uint64_t m,
                                 uint32_t nshards, uint32_t shard, uint64_t* __restrict__ c, uint32_t* __restrict__ err) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // a shard only looks at the hashes it owns; the others collapse into ONE filler element behind the last query
  // (index nq), which the host drops after the sort.  err[2] != 0: some element is not owned (the filler exists).
  const bool own = i < m && (nshards <= 1 || shard_of(key32[query_off[0] + i], nshards) == shard);
  // (a flag, set by the first wave that sees a foreign element.  A COUNT of the owned elements kept with one atomicAdd per
  // wave -- 41,000 waves on ONE word, and as many atomicMax on the next -- was what this kernel cost: 0.7 ms per 2.6 M
  // elements)
  if (nshards > 1) {
    const bool foreign = i < m && !own;
    if (__ballot(foreign) && (threadIdx.x & 63) == 0 && *(volatile uint32_t*)(err + 2) == 0u) atomicOr(err + 2, 1u);
  }
  // largest query offset (the vote key biases deltas by it) and the "offset too wide" flag: one atomic per wave
  uint32_t o = 0;
  uint64_t h = 0;
  if (own) {
    h = query_off[0] + i;
    o = q_off[h];
  }
  uint32_t omax = o;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) omax = max(omax, (uint32_t)__shfl_xor((int)omax, d, 64));
  // the running maximum only grows: a wave whose own maximum is not above what it reads there has nothing to add (the
  // wave that wrote that value also set the flag if it was too wide)
  if ((threadIdx.x & 63) == 0 && omax > *(volatile uint32_t*)(err + 1)) {
    atomicMax(err + 1, omax);
    if (omax >> QOFF_BITS) atomicOr(err, 1u);
  }
  // the query of an element = the last one whose first hash is not behind it.  A wave's 64 consecutive elements lie in
  // one or two queries: ONE search per wave (scalar loads, for the wave's first element), then each lane walks on from
  // there -- a step or two instead of log2(nq) dependent loads per element (2.9 % of a match at 100k songs before)
  const uint64_t iw = i & ~63ull;   // the wave's first element: the same in every lane, and told so to the compiler
  const uint64_t i0 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)iw) |
                      ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(iw >> 32)) << 32);
  if (i0 >= m) return;   // uniform
  const uint64_t h0 = query_off[0] + i0;
  uint32_t w_lo = 0, w_hi = nq;
  while (w_hi - w_lo > 1) {   // uniform
    const uint32_t mid = (w_lo + w_hi) >> 1;
    if (query_off[mid] <= h0) w_lo = mid; else w_hi = mid;
  }
  if (i >= m) return;
  if (!own) { c[i] = (uint64_t)nq << QIDX_SHIFT; return; }
  uint32_t lo = w_lo;
  while (lo + 1 < nq && query_off[lo + 1] <= h) ++lo;
  c[i] = ((uint64_t)lo << QIDX_SHIFT) | ((uint64_t)key32[h] << QKEY_SHIFT) | (o & ((1u << QOFF_BITS) - 1));
}

// Counts that used to travel to the host between the stages (unique elements, groups) stay on the device: kernels
// are launched over a host-known bound and read the count here.  mctl sits in the first 256 bytes of SHZ_WS_MISC0.
struct mctl {
  unsigned long long mu;      // unique (query, key, offset) elements
  unsigned long long ng;      // groups = distinct (query, key)
  unsigned long long rows;    // table rows under the probed keys
  unsigned long long P;       // votes
  unsigned long long G;       // (query, song) groups of the votes
  unsigned long long pad[3];
  unsigned int err[4];        // [0] offset too wide, [1] largest query offset, [2] some hash belongs to another shard
};
// rows under the probed keys: striped over the 8-byte words [16, 16 + M_ROW_STRIPES) of the 256-byte control block
#define M_ROW_STRIPES 16
static_assert(sizeof(mctl) <= 128 && 128 + M_ROW_STRIPES * 8 <= 256, "the stripes sit in the second half of the control block");
// the filler element that stands for the hashes other shards own sorts last: it is not an element
__global__ void m_fix_mu_kernel(mctl* c, unsigned long long m) {
  if (c->err[2] && c->mu) --c->mu;
}
// flag[i] = 1 where (c[i] >> shift) differs from its predecessor, for i < *n (0 beyond, up to the launch bound)
__global__ void m_head_flag_dn_kernel(const uint64_t* __restrict__ c, const unsigned long long* __restrict__ n,
                                      uint64_t bound, int shift, uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bound) return;
  flag[i] = (i < *n && (i == 0 || (c[i] >> shift) != (c[i - 1] >> shift))) ? 1u : 0u;
}

// ---- head of the match for SMALL query sets (<= MH_MAX hashes, no shard filter) in ONE workgroup: compose, sort,
// unique, group heads.  Replaces m_compose + the radix sort (1 to 21 launches) + 11 flag / scan / compact launches:
// for one 5-10 s query the launches, not the work, were the latency.
#define MH_THREADS 1024
#define MH_ROWS 8
#define MH_MAX (MH_THREADS * MH_ROWS)   // 8,192 elements: two 64 KB key buffers in LDS

// exclusive scan of one value per thread over the 1024-thread workgroup; *total = sum.  tmp: >= 17 entries of LDS.
__device__ __forceinline__ uint32_t mh_block_scan(uint32_t v, uint32_t* total, uint32_t* tmp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();   // tmp may still be read from the previous use
  if (lane == 63) tmp[wave] = inc;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < MH_THREADS / 64; ++w) {
    const uint32_t x = tmp[w];
    if (w < wave) woff += x;
    tot += x;
  }
  *total = tot;
  return woff + inc - v;
}

// Stable LSD radix sort of m <= MH_MAX keys that sit in sk[0], on bits [bit_lo, bit_hi), 8 bits per pass, all in LDS.
// Wave w owns the slice [w * 64 rows, (w + 1) * 64 rows), rows = ceil(m / 1024): every pass ranks a slice's elements among themselves by ballots
// (the scheme of sort_scatter_kernel), prefixes the per-wave digit counts over the waves and the digits, and scatters
// into the other buffer.  Bits on which no two keys differ (`diff` = OR of all keys ^ AND of all keys) are skipped: a
// digit begins at the next differing bit.  Returns the index of the buffer that holds the result.  Ends with a barrier.
__device__ __forceinline__ int mh_lds_sort(uint64_t (*sk)[MH_MAX], uint16_t (*wrun)[256], uint32_t* dbase, uint32_t m,
                                           int bit_lo, int bit_hi, unsigned long long diff) {
  const int rows = (int)((m + MH_THREADS - 1) / MH_THREADS);   // rows of 64 elements per wave: the slices shrink with m
  const int j = threadIdx.x, lane = j & 63, wave = j >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  int cur = 0;
  for (int shift = bit_lo; shift < bit_hi; shift += 8) {
    // a digit starts at the next bit on which any two keys differ (not at a multiple of 8): offsets of 8 bits + 32 key bits
    // are five passes wherever the fields sit, not six
    const unsigned long long rest = shift < 64 ? diff >> shift : 0ull;
    if (rest == 0ull) break;                        // uniform
    shift += __ffsll((long long)rest) - 1;
    if (shift >= bit_hi) break;
    const uint32_t dmask = (1u << min(8, bit_hi - shift)) - 1u;
    for (int i = j; i < (MH_THREADS / 64) * 256; i += MH_THREADS) (&wrun[0][0])[i] = 0;
    __syncthreads();
    uint64_t k[MH_ROWS];
    uint32_t rank[MH_ROWS];
#pragma unroll
    for (int r = 0; r < MH_ROWS; ++r) {
      if (r >= rows) break;   // uniform
      const uint32_t li = (uint32_t)wave * (rows * 64) + (uint32_t)r * 64 + lane;
      const bool valid = li < m;
      k[r] = valid ? sk[cur][li] : 0;
      const uint32_t d = (uint32_t)(k[r] >> shift) & dmask;
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long mm = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? mm : ~mm;
      }
      const uint32_t rk = (uint32_t)__popcll(peers & lt);
      const uint32_t run = wrun[wave][d];
      rank[r] = run + rk;
      if (valid && rk == 0) wrun[wave][d] = (uint16_t)(run + (uint32_t)__popcll(peers));
    }
    __syncthreads();
    if (j < 256) {   // per digit: exclusive prefix over the waves, total of the digit
      uint32_t acc = 0;
#pragma unroll
      for (int w = 0; w < MH_THREADS / 64; ++w) {
        const uint32_t c = wrun[w][j];
        wrun[w][j] = (uint16_t)acc;
        acc += c;
      }
      dbase[j] = acc;
    }
    __syncthreads();
    if (wave == 0) {  // exclusive scan of the 256 digit totals: 4 per lane
      uint32_t a0 = dbase[4 * lane], a1 = dbase[4 * lane + 1], a2 = dbase[4 * lane + 2], a3 = dbase[4 * lane + 3];
      uint32_t sum = a0 + a1 + a2 + a3, inc = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
      }
      uint32_t ex = inc - sum;
      dbase[4 * lane] = ex; ex += a0;
      dbase[4 * lane + 1] = ex; ex += a1;
      dbase[4 * lane + 2] = ex; ex += a2;
      dbase[4 * lane + 3] = ex;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MH_ROWS; ++r) {
      if (r >= rows) break;   // uniform
      const uint32_t li = (uint32_t)wave * (rows * 64) + (uint32_t)r * 64 + lane;
      if (li < m) {
        const uint32_t d = (uint32_t)(k[r] >> shift) & dmask;
        sk[cur ^ 1][dbase[d] + wrun[wave][d] + rank[r]] = k[r];
      }
    }
    cur ^= 1;
    __syncthreads();
  }
  return cur;
}

// the votes of a small match (<= MH_MAX of them) sorted in one workgroup instead of 3-4 passes x 3 launches
__global__ __launch_bounds__(MH_THREADS) void m_sort_small_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                                  uint32_t n, int bit_lo, int bit_hi) {
  __shared__ uint64_t sk[2][MH_MAX];
  __shared__ uint16_t wrun[MH_THREADS / 64][256];
  __shared__ uint32_t dbase[256];
  for (uint32_t i = threadIdx.x; i < n; i += MH_THREADS) sk[0][i] = in[i];
  __syncthreads();
  const int cur = mh_lds_sort(sk, wrun, dbase, n, bit_lo, bit_hi, ~0ull);
  for (uint32_t i = threadIdx.x; i < n; i += MH_THREADS) out[i] = sk[cur][i];
}

__global__ __launch_bounds__(MH_THREADS) void m_head_small_kernel(const uint32_t* __restrict__ key32,
                                                                  const uint32_t* __restrict__ q_off,
                                                                  const uint64_t* __restrict__ query_off, uint32_t nq,
                                                                  uint32_t m, int bit_hi, uint64_t* __restrict__ E,
                                                                  uint32_t* __restrict__ gs, mctl* __restrict__ ctl) {
  __shared__ uint64_t sk[2][MH_MAX];
  __shared__ uint16_t wrun[MH_THREADS / 64][256];
  __shared__ uint32_t dbase[256], tmp[20];
  __shared__ unsigned long long s_or, s_and;
  __shared__ uint32_t s_omax;
  const int j = threadIdx.x, lane = j & 63;
  if (j == 0) { s_or = 0; s_and = ~0ull; s_omax = 0; }
  __syncthreads();
  // compose (query, key, offset) elements; wave w owns the slice [w * 512, w * 512 + 512)
  const uint64_t h0 = query_off[0];
  unsigned long long vor = 0, vand = ~0ull;
  uint32_t omax = 0;
#pragma unroll
  for (int r = 0; r < MH_ROWS; ++r) {
    const uint32_t li = (uint32_t)r * MH_THREADS + j;   // (any order: the sort follows)
    if (li < m) {
      const uint64_t h = h0 + li;
      const uint32_t o = q_off[h];
      uint32_t lo = 0, hi = nq;
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (query_off[mid] <= h) lo = mid; else hi = mid;
      }
      const uint64_t c = ((uint64_t)lo << QIDX_SHIFT) | ((uint64_t)key32[h] << QKEY_SHIFT) | (o & ((1u << QOFF_BITS) - 1));
      sk[0][li] = c;
      vor |= c;
      vand &= c;
      omax = max(omax, o);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    vor |= __shfl_xor((long long)vor, d, 64);
    vand &= __shfl_xor((long long)vand, d, 64);
    omax = max(omax, (uint32_t)__shfl_xor((int)omax, d, 64));
  }
  if (lane == 0) {
    atomicOr(&s_or, vor);
    atomicAnd(&s_and, vand);
    atomicMax(&s_omax, omax);
  }
  __syncthreads();
  const unsigned long long diff = s_or ^ s_and;   // bits on which the elements differ at all: other digits need no pass
  const int cur = mh_lds_sort(sk, wrun, dbase, m, 0, bit_hi, diff);
  // unique elements: thread j looks at the MH_ROWS consecutive sorted elements [j * 8, j * 8 + 8)
  const uint64_t* S = sk[cur];
  uint64_t* U = sk[cur ^ 1];
  uint32_t f = 0, cnt = 0;
#pragma unroll
  for (int r = 0; r < MH_ROWS; ++r) {
    const uint32_t i = (uint32_t)j * MH_ROWS + r;
    if (i < m && (i == 0 || S[i] != S[i - 1])) { f |= 1u << r; ++cnt; }
  }
  uint32_t mu;
  uint32_t pos = mh_block_scan(cnt, &mu, tmp);
  uint64_t mine[MH_ROWS];
#pragma unroll
  for (int r = 0; r < MH_ROWS; ++r) mine[r] = S[(uint32_t)j * MH_ROWS + r];
  __syncthreads();   // everybody has read S before U (the other buffer) is written; U != S, but keep the phases apart
#pragma unroll
  for (int r = 0; r < MH_ROWS; ++r)
    if ((f >> r) & 1u) { U[pos] = mine[r]; E[pos] = mine[r]; ++pos; }
  __syncthreads();
  // group heads: first element of every distinct (query, key)
  f = 0; cnt = 0;
#pragma unroll
  for (int r = 0; r < MH_ROWS; ++r) {
    const uint32_t i = (uint32_t)j * MH_ROWS + r;
    if (i < mu && (i == 0 || (U[i] >> QKEY_SHIFT) != (U[i - 1] >> QKEY_SHIFT))) { f |= 1u << r; ++cnt; }
  }
  uint32_t ng;
  pos = mh_block_scan(cnt, &ng, tmp);
#pragma unroll
  for (int r = 0; r < MH_ROWS; ++r)
    if ((f >> r) & 1u) gs[pos++] = (uint32_t)j * MH_ROWS + r;
  if (j == 0) {
    gs[ng] = mu;
    ctl->mu = mu;
    ctl->ng = ng;
    ctl->err[0] = (s_omax >> QOFF_BITS) ? 1u : 0u;
    ctl->err[1] = s_omax;
    ctl->err[2] = 0;   // (one shard: nothing is foreign)
  }
}

// flag[i] = 1 where (c[i] >> shift) differs from its predecessor
__global__ void m_head_flag_kernel(const uint64_t* __restrict__ c, uint64_t n, int shift, uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = (i == 0 || (c[i] >> shift) != (c[i - 1] >> shift)) ? 1u : 0u;
}

__global__ void m_compact_vals_kernel(const uint64_t* __restrict__ c, const uint32_t* __restrict__ flag,
                                      const uint32_t* __restrict__ pos, uint64_t n, uint64_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) out[pos[i]] = c[i];
}

// starts[pos[i]] = i for heads; starts[n_heads] = n
__global__ void m_compact_idx_kernel(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos,
                                     const unsigned long long* __restrict__ n_dev, uint64_t bound,
                                     const unsigned long long* __restrict__ n_heads, uint32_t* __restrict__ starts) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t n = *n_dev;
  if (i == 0) starts[*n_heads] = (uint32_t)n;
  if (i < bound && i < n && flag[i]) starts[pos[i]] = (uint32_t)i;
}

// probe: rows [lo, lo+rows) of every (query, key) group g in every segment sg, one thread per x = g * nseg + sg.
// g_lo[x] = first row, g_pairs[x] = rows x offsets of the group (the votes it expands to); x = ng * nseg is the
// sentinel that makes the exclusive scan end in the total.
__global__ void m_probe_kernel(const uint64_t* __restrict__ E, const uint32_t* __restrict__ gs,
                               const unsigned long long* __restrict__ ng_dev, uint64_t bound,
                               const shz_seg_dev* __restrict__ segs, uint32_t nseg, uint32_t* __restrict__ g_lo,
                               uint64_t* __restrict__ g_pairs, unsigned long long* __restrict__ rows_total,
                               uint32_t* __restrict__ rb, uint32_t rb_words) {
  const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint64_t i = x; i < rb_words; i += (uint64_t)gridDim.x * blockDim.x) rb[i] = 0u;   // the result block (a fill launch otherwise)
  const uint64_t nx = (uint64_t)*ng_dev * nseg;   // <= bound - 1: slots past nx count no pairs
  unsigned long long s = 0;
  if (x < nx) {
    const uint32_t g = (uint32_t)(x / nseg), sg = (uint32_t)(x - (uint64_t)g * nseg);
    const uint32_t e0 = gs[g];
    const uint32_t key = (uint32_t)(E[e0] >> QKEY_SHIFT);
    const uint64_t b = key >> 8;
    const uint32_t* __restrict__ tkey = segs[sg].key;
    uint32_t lo = 0, rows = 0;
    // a segment is asked only for keys inside [its first key, its last bucket]: the segments of one k-way merge hold
    // disjoint key ranges, so one of them answers and the others cost this compare (not a bucket fetch + searches)
    if (b < segs[sg].nbuckets && key >= segs[sg].key_lo && segs[sg].n) {
      uint32_t l = segs[sg].bucket[b], h = segs[sg].bucket[b + 1];
      const uint32_t h0 = h;
      while (l < h) {  // lower_bound(key)
        uint32_t mid = l + ((h - l) >> 1);
        if (tkey[mid] < key) l = mid + 1; else h = mid;
      }
      lo = l;
      h = h0;
      while (l < h) {  // upper_bound(key)
        uint32_t mid = l + ((h - l) >> 1);
        if (tkey[mid] <= key) l = mid + 1; else h = mid;
      }
      rows = l - lo;
    }
    g_lo[x] = lo;
    g_pairs[x] = (uint64_t)rows * (gs[g + 1] - e0);
    s = rows;
  } else if (x < bound) {
    g_pairs[x] = 0;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor((long long)s, d, 64);
  // (statistics only.  One word for all waves cost more than the searches: 31,000 atomics on one address per call; the
  // count is kept in M_ROW_STRIPES words of the control block instead and summed on the host)
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(rows_total + (blockIdx.x & (M_ROW_STRIPES - 1)), s);
}

__global__ void m_query_stats_kernel(const uint64_t* __restrict__ E, const mctl* __restrict__ ctl,
                                     const uint32_t* __restrict__ gs, const uint64_t* __restrict__ po, uint32_t nseg,
                                     uint32_t nq, uint32_t* __restrict__ nhash, uint64_t* __restrict__ npairs) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const uint32_t mu = (uint32_t)ctl->mu, ng = (uint32_t)ctl->ng;
  // elements of q: E in [q << 52, (q+1) << 52)
  auto lb_e = [&](uint64_t target) {
    uint32_t l = 0, h = mu;
    while (l < h) { uint32_t mid = l + ((h - l) >> 1); if (E[mid] < target) l = mid + 1; else h = mid; }
    return l;
  };
  auto lb_g = [&](uint64_t target) {
    uint32_t l = 0, h = ng;
    while (l < h) { uint32_t mid = l + ((h - l) >> 1); if (E[gs[mid]] < target) l = mid + 1; else h = mid; }
    return l;
  };
  const uint64_t a = (uint64_t)q << QIDX_SHIFT, b = (uint64_t)(q + 1) << QIDX_SHIFT;
  nhash[q] = lb_e(b) - lb_e(a);
  npairs[q] = po[(uint64_t)lb_g(b) * nseg] - po[(uint64_t)lb_g(a) * nseg];
}

struct m_bits { int sb, dbits, qb; uint32_t bias; };  // bias = max query offset of the sub-batch: delta + bias >= 0

// expand: pair p = (query hash element, table row) of sub-group x = (group g, segment sg), packed as a vote.  A
// workgroup owns M_EXP_TILE consecutive pairs.  m_tile_start_kernel finds the sub-group of every tile's first pair; a
// tile spans a handful of sub-groups (the whole po[] has ~1e6 entries), so the workgroup copies their descriptors to
// LDS once and every pair is resolved from there: the only per-pair global traffic is the table row and the store.
#define M_EXP_PER 8
#define M_EXP_TILE (256 * M_EXP_PER)
#define M_EXP_SUB 512   // sub-groups per tile that fit the LDS tables (more: the same search over global memory)

// tile_x[t] = last x in [0, nx) with po[x] <= min(t * M_EXP_TILE, P - 1), t = 0 .. ntiles
// (votes [v_lo, P) of the sub-batch: a vote pass may cover a range of queries only)
__global__ void m_tile_start_kernel(const uint64_t* __restrict__ po, uint32_t nx, uint64_t v_lo, uint64_t P, uint32_t ntiles,
                                    uint32_t* __restrict__ tile_x) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > ntiles) return;
  const uint64_t p = min(v_lo + (uint64_t)t * M_EXP_TILE, P - 1);
  uint32_t l = 0, h = nx;
  while (h - l > 1) {
    const uint32_t mid = l + ((h - l) >> 1);
    if (po[mid] <= p) l = mid; else h = mid;
  }
  tile_x[t] = l;
}

// the same for EVERY vote pass of a sub-batch in one launch: pass p covers the votes [pv[2p], pv[2p+1]) and owns the
// entries [toff[p], toff[p+1]) of tile_x (its tiles + 1)
__global__ void m_tile_start_all_kernel(const uint64_t* __restrict__ po, uint32_t nx, const uint64_t* __restrict__ pv,
                                        const uint32_t* __restrict__ toff, uint32_t n_pass, uint32_t* __restrict__ tile_x) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= toff[n_pass]) return;
  uint32_t lo = 0, hi = n_pass;   // last pass with toff[p] <= e
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (toff[mid] <= e) lo = mid; else hi = mid;
  }
  const uint64_t v_lo = pv[2 * lo], P = pv[2 * lo + 1];
  const uint64_t p = min(v_lo + (uint64_t)(e - toff[lo]) * M_EXP_TILE, P - 1);
  uint32_t l = 0, h = nx;
  while (h - l > 1) {
    const uint32_t mid = l + ((h - l) >> 1);
    if (po[mid] <= p) l = mid; else h = mid;
  }
  tile_x[e] = l;
}

#define M_NO_QUERY_BITS INT64_MIN   // q_base of a pass whose votes carry no query index (segmented sort: position says it)
__device__ __forceinline__ uint64_t m_vote(uint64_t e, uint32_t sid, uint32_t off, uint32_t first, m_bits mb, int64_t q_base) {
  const uint64_t q = q_base == M_NO_QUERY_BITS ? 0ull : (uint64_t)((int64_t)(e >> QIDX_SHIFT) + q_base);
  const uint32_t qo = (uint32_t)e & ((1u << QOFF_BITS) - 1);
  const uint64_t dprime = (uint64_t)off + mb.bias - qo;  // delta + bias >= 0
  return ((((q << mb.sb) | sid) << mb.dbits | dprime) << 1) | first;
}

// Votes [v_lo, P) of the sub-batch go to v[0 ..): VT = uint64_t, or uint32_t when a pass's queries, song ids and deltas
// fit 31 bits (q_base then shifts the query index to the pass's first query).
// The votes [base, end) of the sub-batch, whose sub-groups are [xA, xB], go to v[base - v_lo ..).  HIST: the digit
// (vote >> shift) & dmask of every vote is counted in h (LDS, 256 counters of the caller).  May be called in a loop: the
// tables below are rewritten per call, behind a barrier.
template <typename VT, bool HIST>
__device__ __forceinline__ void m_expand_chunk(const uint64_t* __restrict__ E, const uint32_t* __restrict__ gs,
                                               const uint64_t* __restrict__ po, const uint32_t* __restrict__ g_lo,
                                               const shz_seg_dev* __restrict__ segs, uint32_t nseg, uint64_t v_lo,
                                               uint64_t base, uint64_t end, uint32_t xA, uint32_t xB, m_bits mb,
                                               int64_t q_base, VT* __restrict__ v, uint32_t* h, int shift, uint32_t dmask) {
  __shared__ uint64_t s_po[M_EXP_SUB], s_e[M_EXP_SUB];
  __shared__ uint32_t s_lo[M_EXP_SUB], s_e0[M_EXP_SUB], s_noff[M_EXP_SUB], s_sg[M_EXP_SUB];
  __shared__ const uint32_t* s_sid[SHZ_MAX_SEGS];
  __shared__ const uint32_t* s_off[SHZ_MAX_SEGS];
  if (HIST) __syncthreads();   // (the previous chunk's tables are still being read)
  const uint64_t last = end - 1;
  const uint32_t cnt = xB - xA + 1;
  int iters = 0;
  while ((1u << iters) < cnt) ++iters;  // uniform
  uint64_t p[M_EXP_PER];
  uint32_t l[M_EXP_PER], hh[M_EXP_PER];
#pragma unroll
  for (int j = 0; j < M_EXP_PER; ++j)
    p[j] = min(base + (uint64_t)j * 256 + threadIdx.x, last);  // clamped: lanes past the end redo the last pair, unstored
  if (threadIdx.x < nseg) { s_sid[threadIdx.x] = segs[threadIdx.x].sid; s_off[threadIdx.x] = segs[threadIdx.x].off; }
  if (cnt <= M_EXP_SUB) {  // uniform
    for (uint32_t i = threadIdx.x; i < cnt; i += 256) {
      const uint32_t x = xA + i, g = x / nseg;
      const uint32_t e0 = gs[g];
      s_po[i] = po[x];
      s_lo[i] = g_lo[x];
      s_e0[i] = e0;
      s_noff[i] = gs[g + 1] - e0;
      s_sg[i] = x - g * nseg;
      s_e[i] = E[e0];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < M_EXP_PER; ++j) { l[j] = 0; hh[j] = cnt; }  // last i in [0, cnt) with s_po[i] <= p
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < M_EXP_PER; ++j) {
        const uint32_t mid = (l[j] + hh[j]) >> 1;   // == l once h - l <= 1: s_po[l] <= p keeps l
        if (s_po[mid] <= p[j]) l[j] = mid; else hh[j] = mid;
      }
    }
#pragma unroll
    for (int j = 0; j < M_EXP_PER; ++j) {
      const uint32_t i = l[j];
      const uint32_t noff = s_noff[i];
      const uint32_t r = (uint32_t)(p[j] - s_po[i]);  // < P < 2^32
      uint32_t ridx = r, oi = 0;                      // row number inside the sub-group's rows, offset of the hash
      uint64_t e = s_e[i];
      if (noff != 1) { ridx = r / noff; oi = r - ridx * noff; e = E[s_e0[i] + oi]; }
      const uint32_t row = s_lo[i] + ridx, sg = s_sg[i];
      const uint64_t out = m_vote(e, s_sid[sg][row], s_off[sg][row], oi == 0 ? 1u : 0u, mb, q_base);
      const uint64_t pj = base + (uint64_t)j * 256 + threadIdx.x;
      if (pj < end) {
        v[pj - v_lo] = (VT)out;
        if (HIST) atomicAdd(&h[((uint32_t)out >> shift) & dmask], 1u);
      }
    }
    return;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < M_EXP_PER; ++j) { l[j] = xA; hh[j] = xB + 1; }  // last x in [xA, xB] with po[x] <= p
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < M_EXP_PER; ++j) {
      const uint32_t mid = (l[j] + hh[j]) >> 1;
      if (po[mid] <= p[j]) l[j] = mid; else hh[j] = mid;
    }
  }
#pragma unroll 2
  for (int j = 0; j < M_EXP_PER; ++j) {
    const uint32_t x = l[j], g = x / nseg, sg = x - g * nseg;
    const uint32_t e0 = gs[g], noff = gs[g + 1] - e0;
    const uint32_t r = (uint32_t)(p[j] - po[x]);
    uint32_t ridx = r, oi = 0;
    if (noff != 1) { ridx = r / noff; oi = r - ridx * noff; }
    const uint32_t row = g_lo[x] + ridx;
    const uint64_t out = m_vote(E[e0 + oi], s_sid[sg][row], s_off[sg][row], oi == 0 ? 1u : 0u, mb, q_base);
    const uint64_t pj = base + (uint64_t)j * 256 + threadIdx.x;
    if (pj < end) {
      v[pj - v_lo] = (VT)out;
      if (HIST) atomicAdd(&h[((uint32_t)out >> shift) & dmask], 1u);
    }
  }
}

// one tile of M_EXP_TILE votes per workgroup
template <typename VT>
__device__ __forceinline__ void m_expand_tile(const uint64_t* __restrict__ E, const uint32_t* __restrict__ gs,
                                              const uint32_t* __restrict__ tile_x, const uint64_t* __restrict__ po,
                                              const uint32_t* __restrict__ g_lo,
                                              const shz_seg_dev* __restrict__ segs, uint32_t nseg, uint64_t v_lo,
                                              uint64_t P, m_bits mb, int64_t q_base, VT* __restrict__ v) {
  const uint64_t base = v_lo + (uint64_t)blockIdx.x * M_EXP_TILE;
  m_expand_chunk<VT, false>(E, gs, po, g_lo, segs, nseg, v_lo, base, min(base + M_EXP_TILE, P), tile_x[blockIdx.x],
                            tile_x[blockIdx.x + 1], mb, q_base, v, nullptr, 0, 0u);
}

template <typename VT>
__global__ __launch_bounds__(256) void m_expand_kernel(const uint64_t* __restrict__ E, const uint32_t* __restrict__ gs,
                                                       const uint32_t* __restrict__ tile_x, const uint64_t* __restrict__ po,
                                                       const uint32_t* __restrict__ g_lo,
                                                       const shz_seg_dev* __restrict__ segs, uint32_t nseg, uint64_t v_lo,
                                                       uint64_t P, m_bits mb, int64_t q_base, VT* __restrict__ v) {
  m_expand_tile<VT>(E, gs, tile_x, po, g_lo, segs, nseg, v_lo, P, mb, q_base, v);
}

// ---- expand by the BLOCKS of the segmented vote sort (shz_sort_u32_seg): workgroup b produces the votes of sort block b
// (<= 4,096 or 8,192 votes of one query, M_EXP_TILE at a time) and counts their first-pass digits on the way, in the
// sort's table layout -- the sort's first counting kernel (a pass over all votes at 3.4 TB/s, 5 % of a match at 1M songs)
// is not run.  cx[b * (cpb + 1) + c], c = 0 .. cpb: the sub-group of the first vote of chunk c of block b (cpb chunks of
// M_EXP_TILE votes per block; positions past the block's end name its end).
__device__ __forceinline__ void m_block_range(const shz_seg_plan& sp, uint32_t tile, uint32_t b, uint32_t& seg, uint32_t& bl,
                                              uint32_t& bs, uint32_t& be) {
  uint32_t i = 0, hi = sp.nq;                        // last segment with bq[i] <= b (empty segments have no blocks)
  while (hi - i > 1) {
    const uint32_t mid = (i + hi) >> 1;
    if (sp.bq[mid] <= b) i = mid; else hi = mid;
  }
  seg = i;
  bl = b - sp.bq[i];
  bs = sp.qv[i] + bl * tile;
  be = min(bs + tile, sp.qv[i + 1]);
}

__global__ void m_chunk_start_kernel(const uint64_t* __restrict__ po, uint32_t nx, shz_seg_plan sp, uint32_t tile, uint64_t v_lo,
                                     uint64_t v_hi, uint32_t* __restrict__ cx) {
  const uint32_t cpb = tile / M_EXP_TILE, nb = sp.bq[sp.nq];
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nb * (cpb + 1)) return;
  const uint32_t b = e / (cpb + 1), c = e - b * (cpb + 1);
  uint32_t seg, bl, bs, be;
  m_block_range(sp, tile, b, seg, bl, bs, be);
  const uint64_t p = min(v_lo + min(bs + c * M_EXP_TILE, be), v_hi - 1);
  uint32_t l = 0, h = nx;
  while (h - l > 1) {
    const uint32_t mid = l + ((h - l) >> 1);
    if (po[mid] <= p) l = mid; else h = mid;
  }
  cx[e] = l;
}

__global__ __launch_bounds__(256, 8) void m_expand_blocks_kernel(const uint64_t* __restrict__ E, const uint32_t* __restrict__ gs,
                                                              const uint32_t* __restrict__ cx, const uint64_t* __restrict__ po,
                                                              const uint32_t* __restrict__ g_lo,
                                                              const shz_seg_dev* __restrict__ segs, uint32_t nseg, uint64_t v_lo,
                                                              shz_seg_plan sp, uint32_t tile, m_bits mb, int shift, uint32_t dmask,
                                                              uint32_t* __restrict__ v, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;   // (every chunk starts with a barrier)
  const uint32_t cpb = tile / M_EXP_TILE;
  uint32_t seg, bl, bs, be;
  m_block_range(sp, tile, blockIdx.x, seg, bl, bs, be);
  const uint32_t* cxb = cx + (uint64_t)blockIdx.x * (cpb + 1);
  for (uint32_t c = 0; c < cpb; ++c) {   // uniform
    const uint32_t cb = bs + c * M_EXP_TILE;
    if (cb >= be) break;
    m_expand_chunk<uint32_t, true>(E, gs, po, g_lo, segs, nseg, v_lo, v_lo + cb, v_lo + min(cb + (uint32_t)M_EXP_TILE, be), cxb[c],
                                   cxb[c + 1], mb, M_NO_QUERY_BITS, v, h, shift, dmask);
  }
  __syncthreads();
  const uint32_t nbs = sp.bq[seg + 1] - sp.bq[seg];
  hist[((uint64_t)sp.bq[seg] << 8) + bl + (uint64_t)threadIdx.x * nbs] = h[threadIdx.x];
}

// ---- one small query, votes queued BEFORE their number is known on the host
// A single query against a small or medium table has a few thousand votes, and its match is a chain of short launches: the
// read-back of the vote count in the middle of it (a stream sync, ~30 us) is a sixth of the whole call.  For one query
// whose hashes went through m_head_small_kernel the host therefore queues the one-workgroup vote path behind the probe
// at once, sized for VT_ONE_WG_MAX votes; these kernels read the vote count from the control block and do nothing when it
// is larger -- the host then sees that count in the one read-back and continues as if nothing had been queued.
//
// m_spec_plan_kernel: tile_x of the expand tiles (as m_tile_start_kernel) + the single vote range (as vt_one_range_kernel)
// + the zeroed result block
__global__ void m_spec_plan_kernel(const uint64_t* __restrict__ po, const mctl* __restrict__ ctl, uint32_t nseg, uint64_t cap,
                                   uint32_t* __restrict__ tile_x, uint32_t* __restrict__ n_heavy, uint2* __restrict__ heavy,
                                   uint32_t* __restrict__ heavy_q, uint32_t* __restrict__ rb, uint32_t rb_words) {
  for (uint32_t i = threadIdx.x; i < rb_words; i += blockDim.x) rb[i] = 0u;   // the result block (the host's fill otherwise)
  const uint64_t P = ctl->P;
  const bool ok = P > 0 && P <= cap;
  if (threadIdx.x == 0) {
    *n_heavy = ok ? 1u : 0u;
    heavy[0] = make_uint2(0u, ok ? (uint32_t)P : 0u);
    heavy_q[0] = 0u;
  }
  if (!ok) return;
  const uint32_t nx = (uint32_t)ctl->ng * nseg, ntiles = (uint32_t)((P + M_EXP_TILE - 1) / M_EXP_TILE);
  for (uint32_t t = threadIdx.x; t <= ntiles; t += blockDim.x) {
    const uint64_t p = min((uint64_t)t * M_EXP_TILE, P - 1);
    uint32_t l = 0, h = nx;
    while (h - l > 1) {
      const uint32_t mid = l + ((h - l) >> 1);
      if (po[mid] <= p) l = mid; else h = mid;
    }
    tile_x[t] = l;
  }
}

__global__ __launch_bounds__(256) void m_expand_spec_kernel(const uint64_t* __restrict__ E, const uint32_t* __restrict__ gs,
                                                            const uint32_t* __restrict__ tile_x, const uint64_t* __restrict__ po,
                                                            const uint32_t* __restrict__ g_lo,
                                                            const shz_seg_dev* __restrict__ segs, uint32_t nseg,
                                                            const mctl* __restrict__ ctl, uint64_t cap, m_bits mb,
                                                            uint32_t* __restrict__ v) {
  const uint64_t P = ctl->P;
  if (P == 0 || P > cap || (uint64_t)blockIdx.x * M_EXP_TILE >= P) return;   // uniform, before any barrier
  m_expand_tile<uint32_t>(E, gs, tile_x, po, g_lo, segs, nseg, 0ull, P, mb, (int64_t)0, v);
}

// ---- fold of the sorted votes into one record per (query, sid) group
// The votes are sorted by (query, sid, delta, flag).  A thread owns RG_PER consecutive votes and every group whose FIRST
// vote lies among them; it follows such a group past its own votes to the group's end (groups average a few votes, the
// true match a few hundred).  Record = (best count of one delta, smallest delta reaching it, votes with the
// first-offset flag = rows matched).  Records are written densely in group order: m_gcount_kernel counts the heads
// per wave, a scan of those counts gives every wave its first slot.
#define RG_PER 8
#define RG_LONG_ITERS 16   // loads of RG_PER votes a thread follows a group before handing it over
#define RG_LONG_CAP 16384  // long groups per vote pass that get a workgroup of their own

__device__ __forceinline__ int rg_load(const uint64_t* __restrict__ v, uint64_t i0, uint64_t P, uint64_t (&e)[RG_PER]) {
  if (i0 + RG_PER <= P) {
    const ulonglong2* p2 = (const ulonglong2*)(v + i0);
#pragma unroll
    for (int j = 0; j < RG_PER / 2; ++j) { const ulonglong2 x = p2[j]; e[2 * j] = x.x; e[2 * j + 1] = x.y; }
    return RG_PER;
  }
  const int n = i0 < P ? (int)(P - i0) : 0;
#pragma unroll
  for (int j = 0; j < RG_PER; ++j) e[j] = j < n ? v[i0 + j] : 0;
  return n;
}

// the vote before a thread's first one: the neighbour lane's last vote, or memory for lane 0 (0 = "none" for i0 == 0)
__device__ __forceinline__ uint64_t rg_prev(const uint64_t* __restrict__ v, uint64_t i0, uint64_t P, uint64_t e_last) {
  uint64_t prev = (uint64_t)__shfl_up((long long)e_last, 1, 64);
  if ((threadIdx.x & 63) == 0) prev = (i0 > 0 && i0 <= P) ? v[i0 - 1] : 0;
  return prev;
}

__global__ __launch_bounds__(256) void m_gcount_kernel(const uint64_t* __restrict__ v, uint64_t P, int gshift,
                                                       uint32_t* __restrict__ wcnt) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t i0 = t * RG_PER;
  uint64_t e[RG_PER];
  const int n = rg_load(v, i0, P, e);
  uint64_t prev = rg_prev(v, i0, P, e[RG_PER - 1]);
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < RG_PER; ++j) {
    if (j < n && (i0 + j == 0 || (e[j] >> gshift) != (prev >> gshift))) ++c;
    prev = e[j];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor((int)c, d, 64);
  if ((threadIdx.x & 63) == 0) wcnt[t >> 6] = c;
}

struct rg_state {
  uint64_t grp, curd;
  uint32_t best, bestd, cur, dedup;
};

__device__ __forceinline__ void rg_add(rg_state& s, uint64_t val, uint64_t dmask) {
  const uint64_t d = (val >> 1) & dmask;
  if (d != s.curd) {
    if (s.cur > s.best) { s.best = s.cur; s.bestd = (uint32_t)s.curd; }
    s.curd = d;
    s.cur = 0;
  }
  ++s.cur;
  s.dedup += (uint32_t)(val & 1);
}

__global__ __launch_bounds__(256) void m_reduce_kernel(const uint64_t* __restrict__ v, uint64_t P, m_bits mb,
                                                       const uint32_t* __restrict__ wbase, uint64_t* __restrict__ g_pack,
                                                       uint32_t* __restrict__ g_delta, uint32_t* __restrict__ g_dedup,
                                                       uint32_t* __restrict__ qstart, uint32_t* __restrict__ long_cnt,
                                                       uint2* __restrict__ long_list) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t i0 = t * RG_PER;
  const int gshift = mb.dbits + 1, qshift = mb.sb + mb.dbits + 1;
  const uint64_t dmask = (1ull << mb.dbits) - 1, smask = (1ull << mb.sb) - 1;
  uint64_t e[RG_PER];
  const int n = rg_load(v, i0, P, e);
  const uint64_t before = rg_prev(v, i0, P, e[RG_PER - 1]);
  uint32_t hm = 0;  // bit j: vote j opens a group
  {
    uint64_t prev = before;
#pragma unroll
    for (int j = 0; j < RG_PER; ++j) {
      if (j < n && (i0 + j == 0 || (e[j] >> gshift) != (prev >> gshift))) hm |= 1u << j;
      prev = e[j];
    }
  }
  const uint32_t hc = (uint32_t)__popc(hm);
  uint32_t incl = hc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
    if ((int)(threadIdx.x & 63) >= d) incl += o;
  }
  if (hm == 0) return;
  uint32_t slot = wbase[t >> 6] + incl - hc;
  rg_state s;
  bool owning = false;
  auto emit = [&]() {
    if (s.cur > s.best) { s.best = s.cur; s.bestd = (uint32_t)s.curd; }
    g_pack[slot] = ((uint64_t)s.best << 32) | (0xFFFFFFFFu - (uint32_t)(s.grp & smask));   // rank: count desc, then sid asc
    g_delta[slot] = s.bestd;
    g_dedup[slot] = s.dedup;
    ++slot;
  };
  {
    uint64_t prev = before;
#pragma unroll
    for (int j = 0; j < RG_PER; ++j) {
      if (j < n) {
        if ((hm >> j) & 1u) {
          if (owning) emit();
          owning = true;
          s.grp = e[j] >> gshift;
          s.curd = ~0ull;
          s.best = s.bestd = s.cur = s.dedup = 0;
          if (i0 + j == 0 || (e[j] >> qshift) != (prev >> qshift)) qstart[e[j] >> qshift] = slot;  // first group of a query
        }
        if (owning) rg_add(s, e[j], dmask);
      }
      prev = e[j];
    }
  }
  // the last group may run on past this thread's votes.  A group still open after RG_LONG_ITERS more loads (a clean
  // query puts thousands of votes on its own song) goes to m_reduce_long_kernel, a workgroup per group, instead of
  // being walked by this one thread; the list is bounded, a group that does not get a place is walked here after all.
  bool open = n == RG_PER;
  int iters = 0;
  for (uint64_t i = i0 + RG_PER; open && i < P; i += RG_PER) {
    if (++iters == RG_LONG_ITERS + 1) {
      const uint32_t idx = atomicAdd(long_cnt, 1u);
      if (idx < RG_LONG_CAP) {
        long_list[idx] = make_uint2(slot, (uint32_t)(i0 + (31 - __clz((int)hm))));   // record slot, first vote of the group
        return;
      }
    }
    uint64_t x[RG_PER];
    const int m = rg_load(v, i, P, x);
#pragma unroll
    for (int j = 0; j < RG_PER; ++j) {
      if (open && j < m) {
        if ((x[j] >> gshift) != s.grp) open = false; else rg_add(s, x[j], dmask);
      }
    }
    if (m < RG_PER) open = false;
  }
  emit();
}

// One workgroup per long group handed over by m_reduce_kernel: wave 0 finds the group's end (64-ary search, the votes
// are sorted), every thread folds a contiguous piece into (first run, last run, best run between them), thread 0 joins
// the pieces in order -- runs of one delta that cross piece borders are added up, ties keep the smaller delta.
struct rg_part {
  uint64_t hd, td;   // delta of the first / last run of the piece
  uint32_t hl, tl;   // their lengths (hl == 0: empty piece)
  uint32_t best, bestd, dedup, single;   // best run strictly between them; flag votes; single: the piece is one run
};

__global__ __launch_bounds__(256) void m_reduce_long_kernel(const uint64_t* __restrict__ v, uint64_t P, m_bits mb,
                                                            const uint32_t* __restrict__ long_cnt,
                                                            const uint2* __restrict__ long_list,
                                                            uint64_t* __restrict__ g_pack, uint32_t* __restrict__ g_delta,
                                                            uint32_t* __restrict__ g_dedup) {
  __shared__ rg_part parts[256];
  __shared__ uint64_t s_end;
  const int gshift = mb.dbits + 1;
  const uint64_t dmask = (1ull << mb.dbits) - 1, smask = (1ull << mb.sb) - 1;
  const uint32_t nl = min(*long_cnt, (uint32_t)RG_LONG_CAP);
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t L = blockIdx.x; L < nl; L += gridDim.x) {
    const uint32_t slot = long_list[L].x;
    const uint64_t h0 = long_list[L].y;
    const uint64_t grp = v[h0] >> gshift;
    if (threadIdx.x < 64) {  // last index of the group in [h0, P)
      uint64_t lo = h0, n = P - h0;
      while (n > 1) {
        const uint64_t step = (n + 63) / 64;
        const uint64_t idx = lo + lane * step;
        const bool ok = lane * step < n && (v[idx] >> gshift) == grp;   // a prefix of the lanes (lane 0 always)
        const unsigned long long b = __ballot(ok);
        const uint32_t last = 63u - (uint32_t)__clzll(b);
        n = min(step, n - last * step);
        lo += last * step;
      }
      if (lane == 0) s_end = lo + 1;
    }
    __syncthreads();
    const uint64_t end = s_end;
    const uint64_t per = (end - h0 + 255) / 256;
    const uint64_t a = min(h0 + threadIdx.x * per, end), b = min(a + per, end);
    rg_part p;
    p.hd = p.td = 0;
    p.hl = p.tl = p.best = p.bestd = p.dedup = 0;
    p.single = 1;
    uint64_t curd = ~0ull;
    uint32_t cur = 0, nruns = 0;
    for (uint64_t i = a; i < b; ++i) {
      const uint64_t val = v[i];
      const uint64_t d = (val >> 1) & dmask;
      if (d != curd) {
        if (cur) {  // a run ended inside the piece
          if (nruns == 0) { p.hd = curd; p.hl = cur; }
          else if (cur > p.best) { p.best = cur; p.bestd = (uint32_t)curd; }
          ++nruns;
        }
        curd = d;
        cur = 0;
      }
      ++cur;
      p.dedup += (uint32_t)(val & 1);
    }
    if (cur) {
      if (nruns == 0) { p.hd = curd; p.hl = cur; }          // the piece is one run
      else { p.td = curd; p.tl = cur; p.single = 0; }      // last run of several
    }
    parts[threadIdx.x] = p;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t best = 0, bestd = 0, run = 0, dedup = 0;
      uint64_t rd = ~0ull;
      for (int k = 0; k < 256; ++k) {
        const rg_part& q = parts[k];
        if (q.hl == 0) continue;
        dedup += q.dedup;
        if (q.hd == rd) run += q.hl;
        else {
          if (run > best) { best = run; bestd = (uint32_t)rd; }
          rd = q.hd;
          run = q.hl;
        }
        if (!q.single) {
          if (run > best) { best = run; bestd = (uint32_t)rd; }
          if (q.best > best) { best = q.best; bestd = q.bestd; }
          rd = q.td;
          run = q.tl;
        }
      }
      if (run > best) { best = run; bestd = (uint32_t)rd; }
      g_pack[slot] = ((uint64_t)best << 32) | (0xFFFFFFFFu - (uint32_t)(grp & smask));
      g_delta[slot] = bestd;
      g_dedup[slot] = dedup;
    }
    __syncthreads();
  }
}

// group records [r0, r1) of query q: qstart[] holds the first record of every query that has any (0xFFFFFFFF = none)
__device__ __forceinline__ void m_query_groups(const uint32_t* __restrict__ qstart, uint32_t nq,
                                               const unsigned long long* __restrict__ G, uint32_t q, uint32_t& r0,
                                               uint32_t& r1) {
  r0 = qstart[q];
  if (r0 == 0xFFFFFFFFu) { r0 = r1 = 0; return; }
  r1 = (uint32_t)*G;
  for (uint32_t k = q + 1; k < nq; ++k) {
    const uint32_t x = qstart[k];
    if (x != 0xFFFFFFFFu) { r1 = x; break; }
  }
}

// qstart[0..nq) = "no group yet", the counter of long groups behind it = 0
__global__ void m_tail_init_kernel(uint32_t* __restrict__ qstart, uint32_t nq) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) qstart[i] = 0xFFFFFFFFu;
  else if (i == nq) qstart[nq] = 0u;
}

// one workgroup per query: top-n groups by (count desc, sid asc) over the packed group summaries
__global__ __launch_bounds__(256) void m_topn_kernel(const uint32_t* __restrict__ qstart,
                                                     const unsigned long long* __restrict__ G, m_bits mb,
                                                     const uint64_t* __restrict__ g_pack,
                                                     const uint32_t* __restrict__ g_delta,
                                                     const uint32_t* __restrict__ g_dedup, uint32_t nq, uint32_t topn,
                                                     uint32_t* __restrict__ out_sid, int32_t* __restrict__ out_delta,
                                                     uint32_t* __restrict__ out_aligned, uint32_t* __restrict__ out_dedup,
                                                     uint32_t* __restrict__ out_nres) {
  __shared__ uint64_t s_best[4];
  __shared__ uint32_t s_r[4];
  const uint32_t q = blockIdx.x;
  if (q >= nq) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t r0, r1;
  m_query_groups(qstart, nq, G, q, r0, r1);
  uint64_t prev = ~0ull;  // packed rank of the previous winner; candidates must rank strictly below it
  uint32_t found = 0;
  for (uint32_t n = 0; n < topn; ++n) {
    uint64_t best = 0;
    uint32_t bestr = 0xFFFFFFFFu;
    for (uint32_t r = r0 + threadIdx.x; r < r1; r += 256) {
      const uint64_t packed = g_pack[r];
      if (packed < prev && packed > best) { best = packed; bestr = r; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint64_t ob = (uint64_t)__shfl_xor((long long)best, d, 64);
      const uint32_t orr = (uint32_t)__shfl_xor((int)bestr, d, 64);
      if (ob > best) { best = ob; bestr = orr; }
    }
    if (lane == 0) { s_best[wave] = best; s_r[wave] = bestr; }
    __syncthreads();
    best = s_best[0]; bestr = s_r[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (s_best[w] > best) { best = s_best[w]; bestr = s_r[w]; }
    __syncthreads();
    if (best == 0) break;  // uniform
    if (threadIdx.x == 0) {
      const uint64_t o = (uint64_t)q * topn + n;
      out_sid[o] = 0xFFFFFFFFu - (uint32_t)best;
      out_aligned[o] = (uint32_t)(best >> 32);
      out_delta[o] = (int32_t)((int64_t)g_delta[bestr] - (int64_t)mb.bias);
      out_dedup[o] = g_dedup[bestr];
    }
    prev = best;
    ++found;
  }
  if (threadIdx.x == 0) out_nres[q] = found;
}

// Two-level form of m_topn_kernel for queries with millions of (query, song) groups (a 1M-song table yields ~9M per
// 10 s query, and only a few dozen queries fit one vote pass): with one workgroup per query the scan of g_pack ran on
// a few dozen workgroups.  Level 1: workgroup (c, q) finds the top-n of slice c of query q's groups; level 2: one
// workgroup per query ranks the C x topn candidates.  g_pack is unique per group, so "strictly below the previous
// winner" selects the same groups as the single-level kernel.
__device__ __forceinline__ void topn_block_max(uint64_t& best, uint32_t& bestr, uint64_t* s_best, uint32_t* s_r) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint64_t ob = (uint64_t)__shfl_xor((long long)best, d, 64);
    const uint32_t orr = (uint32_t)__shfl_xor((int)bestr, d, 64);
    if (ob > best) { best = ob; bestr = orr; }
  }
  if (lane == 0) { s_best[wave] = best; s_r[wave] = bestr; }
  __syncthreads();
  best = s_best[0]; bestr = s_r[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (s_best[w] > best) { best = s_best[w]; bestr = s_r[w]; }
  __syncthreads();
}

__global__ __launch_bounds__(256) void m_topn_partial_kernel(const uint32_t* __restrict__ qstart,
                                                             const unsigned long long* __restrict__ G, uint32_t nq,
                                                             const uint64_t* __restrict__ g_pack, uint32_t topn, uint64_t* __restrict__ part_pack,
                                                             uint32_t* __restrict__ part_r) {
  __shared__ uint64_t s_best[4];
  __shared__ uint32_t s_r[4];
  const uint32_t q = blockIdx.y, c = blockIdx.x, C = gridDim.x;
  uint32_t r0, r1;
  m_query_groups(qstart, nq, G, q, r0, r1);
  const uint32_t S = (r1 - r0 + C - 1) / C;
  const uint32_t a = r0 + (uint32_t)min((uint64_t)c * S, (uint64_t)(r1 - r0));
  const uint32_t b = (uint32_t)min((uint64_t)a + S, (uint64_t)r1);
  uint64_t prev = ~0ull;
  const uint64_t o = ((uint64_t)q * C + c) * topn;
  for (uint32_t n = 0; n < topn; ++n) {
    uint64_t best = 0;
    uint32_t bestr = 0xFFFFFFFFu;
    if (prev) {
      for (uint32_t r = a + threadIdx.x; r < b; r += 256) {
        const uint64_t packed = g_pack[r];
        if (packed < prev && packed > best) { best = packed; bestr = r; }
      }
    }
    topn_block_max(best, bestr, s_best, s_r);
    if (threadIdx.x == 0) { part_pack[o + n] = best; part_r[o + n] = bestr; }
    prev = best;  // 0 once the slice is exhausted: the remaining slots are written as 0
  }
}

__global__ __launch_bounds__(256) void m_topn_final_kernel(const uint64_t* __restrict__ part_pack,
                                                           const uint32_t* __restrict__ part_r, uint32_t ncand, m_bits mb,
                                                           const uint32_t* __restrict__ g_delta,
                                                           const uint32_t* __restrict__ g_dedup, uint32_t nq, uint32_t topn,
                                                           uint32_t* __restrict__ out_sid, int32_t* __restrict__ out_delta,
                                                           uint32_t* __restrict__ out_aligned, uint32_t* __restrict__ out_dedup,
                                                           uint32_t* __restrict__ out_nres) {
  __shared__ uint64_t s_best[4];
  __shared__ uint32_t s_r[4];
  const uint32_t q = blockIdx.x;
  if (q >= nq) return;
  const uint64_t* pp = part_pack + (uint64_t)q * ncand;
  const uint32_t* pr = part_r + (uint64_t)q * ncand;
  uint64_t prev = ~0ull;
  uint32_t found = 0;
  for (uint32_t n = 0; n < topn; ++n) {
    uint64_t best = 0;
    uint32_t bestr = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < ncand; i += 256) {
      const uint64_t packed = pp[i];
      if (packed < prev && packed > best) { best = packed; bestr = pr[i]; }
    }
    topn_block_max(best, bestr, s_best, s_r);
    if (best == 0) break;  // uniform
    if (threadIdx.x == 0) {
      const uint64_t o = (uint64_t)q * topn + n;
      out_sid[o] = 0xFFFFFFFFu - (uint32_t)best;
      out_aligned[o] = (uint32_t)(best >> 32);
      out_delta[o] = (int32_t)((int64_t)g_delta[bestr] - (int64_t)mb.bias);
      out_dedup[o] = g_dedup[bestr];
    }
    prev = best;
    ++found;
  }
  if (threadIdx.x == 0) out_nres[q] = found;
}


// ---------------------------------------------------------------------------------------------------------------
// Vote tiles: the fold of 4-byte votes without sorting them all the way.
// A record (best count of one delta, smallest delta reaching it, rows matched) is a function of the MULTISET of a
// (query, song) group's votes, and the top-n of a query is a function of its records, so the full order the radix
// sort produced (four passes + run lengths + scans + dense records) is not needed.  Two radix passes order the votes
// by their upper 16 bits only (query | upper song-id bits); the array is then cut into tiles of ~VT_TILE votes at
// places where those bits change, never inside a (query, song) group and never across queries; one workgroup per
// tile counts (song, delta) in an LDS hash table, reduces that to one entry per song in a second table (atomicMax of
// count << 32 | ~delta, atomicAdd of the flags) and picks the tile's top-n; a last kernel ranks the tiles' candidates
// of every query.  g_pack (count << 32 | ~sid) is the rank of align_matches as in m_reduce_kernel: count descending,
// song id ascending, and the delta of a record is the smallest one reaching the count (recognizer.py:305-322 as
// restated in oracle/cpu_ref.py: vote / align_matches).
#define VT_THREADS 1024
#define VT_SLOTS 8192          // per LDS table (4 arrays of 4 bytes x VT_SLOTS = 128 KB)
#define VT_TILE 2048           // votes per tile before the cut is moved to the next group border
#define VT_LIMIT 6144          // distinct keys a table may hold: VT_TILE + 2^12 deltas of one song always fit one sweep
#define VT_MAXQ SHZ_SEG_MAX     // queries per vote pass (their votes stay apart as segments of the sort: shz_sort_u32_seg)
#define VT_MAXTOPN 8
#define VT_MAX_DBITS 12
#define VT_MAX_DSPLIT 8        // delta bits above those a sweep may split by (vt_fold_kernel): tracks of up to 2^20 frames
#define VT_EMPTY 0xFFFFFFFFu
#ifndef VT_ORDERED_BITS
#define VT_ORDERED_BITS 16     // upper bits of a vote the radix passes order (two passes of 8)
#endif
#define VT_ONE_WG_MAX 32768     // votes of a single query that one workgroup folds without any radix pass

struct vt_plan {
  uint32_t nq;                 // queries of the pass
  uint32_t qv[VT_MAXQ + 1];    // first vote of every query in the ordered pass (prefix sums of the per-query counts)
  uint32_t tb[VT_MAXQ + 1];    // first tile of every query; tb[nq] = number of tiles
  int g_lo;                    // lowest bit the radix passes ordered: tiles are cut where bits >= g_lo change
  int dbits, sb;               // layout of a vote: flag | delta (dbits) | song id (sb) | query
  uint32_t tile;               // nominal votes per tile
  uint32_t flush;              // votes a batch of vt_stream_kernel holds before it may end at a group border
};

__device__ __forceinline__ uint32_t vt_query_of_tile(const vt_plan& pl, uint32_t g) {
  uint32_t i = 0, hi = pl.nq;                        // last query with tb[i] <= g (queries without votes have no tiles)
  while (hi - i > 1) {
    const uint32_t mid = (i + hi) >> 1;
    if (pl.tb[mid] <= g) i = mid; else hi = mid;
  }
  return i;
}

// tile_start[g], g = 0 .. ntiles: the first tile of a query starts at the query's first vote, the others at the first
// group border at or behind their nominal start.  One WAVE per tile: the border is a few dozen votes away as a rule (a
// group is 2^slb songs), so the wave looks at the next 63 votes at once, then at three more rows, and only a group longer
// than that is searched by halving (the votes are ordered by the bits >= g_lo) -- one or two dependent loads instead of the
// ~23 of a binary search over the whole query, which was 10 us of a single query's 300.
__global__ __launch_bounds__(256) void vt_bounds_kernel(const uint32_t* __restrict__ k, vt_plan pl, uint32_t* __restrict__ tile_start,
                                                        uint32_t* __restrict__ n_heavy, unsigned long long* __restrict__ qbar) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = pl.tb[pl.nq];
  if (qbar && t <= VT_MAXQ) qbar[t] = 0ull;          // the queries' bars start at nothing (vt_stream2_kernel)
  const uint32_t g = t >> 6, lane = t & 63u;
  if (g > nt) return;
  if (g == nt) { if (lane == 0) { tile_start[g] = pl.qv[pl.nq]; *n_heavy = 0; } return; }
  const uint32_t i = vt_query_of_tile(pl, g), l = g - pl.tb[i];
  const uint32_t a = pl.qv[i], b = pl.qv[i + 1];
  if (l == 0) { if (lane == 0) tile_start[g] = a; return; }
  const uint32_t p = a + l * pl.tile;   // < b: the query has ceil((b - a) / tile) tiles
  uint32_t pos = p - 1, found = b;      // lane 0 of the first row holds the vote in front of the nominal start
  uint32_t h = 0;
  bool done = false;
  for (int r = 0; r < 4 && pos < b; ++r) {           // uniform
    const uint32_t j = pos + lane;
    const uint32_t v = j < b ? k[j] >> pl.g_lo : 0u;
    if (r == 0) h = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    const unsigned long long m = __ballot(j < b && v > h);   // (lane 0 of the first row is the vote itself: not above itself)
    if (m) { found = pos + (uint32_t)__ffsll((long long)m) - 1; done = true; break; }
    pos += 64;
  }
  if (!done && pos < b) {
    uint32_t lo = pos, hi = b;          // first position in [pos, b) whose upper bits exceed h
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if ((k[mid] >> pl.g_lo) > h) hi = mid; else lo = mid + 1;
    }
    found = lo;
  }
  if (lane == 0) tile_start[g] = found;
}

#define VT_SLOTS2 4096         // table 2 (one entry per song)
#define VT_LIMIT2 3000         // songs a sweep may hold: VT_LIMIT2 + one row of new ones stays below VT_SLOTS2
static_assert(VT_SLOTS == 1 << 13 && VT_SLOTS2 == 1 << 12, "vt_hash<13> / vt_hash<12>");
static_assert(VT_LIMIT + VT_THREADS < VT_SLOTS && VT_LIMIT2 + VT_THREADS < VT_SLOTS2, "a probe must find a free slot");
static_assert(VT_TILE + (1 << VT_MAX_DBITS) <= VT_LIMIT && VT_TILE < VT_LIMIT2, "the finest sweep always fits");

template <int BITS>
__device__ __forceinline__ uint32_t vt_hash(uint32_t x) { return (x * 2654435761u) >> (32 - BITS); }

// Slots of N (song | delta) keys in table 1 and of their N songs in table 2, open addressing, inserted if absent.
// All compare-and-swaps of a round are in flight together: a round costs one LDS round trip, not 2 N.
// A probe sequence is BOUNDED by `probe_limit` rounds (the table size: after that many steps every slot has been seen; a
// debug switch lowers it).  The fill limits of the callers keep a free slot within reach, so the bound is never met -- but
// if a table ever is full (a caller's limit broken, a zero-trip clear: the hang of round 2) the lanes give up, *err is
// set, the pass's results are void and the host repeats the pass through the full sort.
template <int N, int B1, int B2>
__device__ __forceinline__ void vt_slots(uint32_t* key1, uint32_t* key2, const uint32_t (&x1)[N], const uint32_t (&x2)[N],
                                         const bool (&valid)[N], uint32_t (&s1)[N], uint32_t (&s2)[N], bool (&fresh1)[N],
                                         bool (&fresh2)[N], uint32_t probe_limit, uint32_t* __restrict__ err) {
  bool p1[N], p2[N];
#pragma unroll
  for (int r = 0; r < N; ++r) {
    s1[r] = vt_hash<B1>(x1[r]);
    s2[r] = vt_hash<B2>(x2[r]);
    p1[r] = p2[r] = valid[r];
    fresh1[r] = fresh2[r] = false;
  }
  auto round = [&]() {
    uint32_t o1[N], o2[N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
      o1[r] = p1[r] ? atomicCAS(&key1[s1[r]], VT_EMPTY, x1[r]) : 0u;
      o2[r] = p2[r] ? atomicCAS(&key2[s2[r]], VT_EMPTY, x2[r]) : 0u;
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
      const bool e1 = p1[r] && o1[r] == VT_EMPTY, e2 = p2[r] && o2[r] == VT_EMPTY;
      fresh1[r] = fresh1[r] || e1;
      fresh2[r] = fresh2[r] || e2;
      p1[r] = p1[r] && !e1 && o1[r] != x1[r];
      p2[r] = p2[r] && !e2 && o2[r] != x2[r];
      if (p1[r]) s1[r] = (s1[r] + 1) & ((1u << B1) - 1u);
      if (p2[r]) s2[r] = (s2[r] + 1) & ((1u << B2) - 1u);
    }
  };
  auto pending = [&]() {
    bool any = false;
#pragma unroll
    for (int r = 0; r < N; ++r) any |= p1[r] | p2[r];
    return __ballot(any) != 0ull;   // uniform
  };
  // The first rounds are straight-line code: at the fill levels the callers keep, two or three rounds settle a row, and
  // outside a loop the lanes' states (pending, fresh) stay lane masks in scalar registers -- as loop-carried values of a
  // divergent loop they were re-materialised per round (35 VALU instructions per round; vt_stream_kernel is bound by VALU
  // issue).  Whatever is still pending goes round a wave-uniform loop.
  constexpr uint32_t STRAIGHT = 3;
  round();                                   // (probe_limit >= 1)
  if (probe_limit > 1) round();
  if (probe_limit > 2) round();
  for (uint32_t done = probe_limit < STRAIGHT ? probe_limit : STRAIGHT; pending(); ++done) {
    if (done >= probe_limit) { atomicOr(err, 1u); break; }   // (the slots of the lanes that gave up are valid indices of the wrong entries)
    round();
  }
}

// N votes of a lane: count of (song, delta) in table 1; the song's entry of table 2 keeps the largest count seen and,
// among equal counts, the smallest delta (counts only grow, so the maximum over all increments is the maximum of the
// final counts), and the number of flagged votes.  Songs entered for the first time are appended to lst (length *n2),
// one LDS atomic per wave; *n1 counts table 1's entries when `count1`.  Called by whole waves.
template <int N>
__device__ __forceinline__ void vt_votes(uint32_t* key1, uint32_t* cnt, uint32_t* key2, unsigned long long* best,
                                         uint32_t* ded, uint16_t* lst, uint32_t* n1, uint32_t* n2, bool count1,
                                         const uint32_t (&v)[N], const bool (&valid)[N], int dbits, uint32_t dmask,
                                         uint32_t probe_limit, uint32_t* __restrict__ err) {
  uint32_t x1[N], x2[N], s1[N], s2[N];
  bool f1[N], f2[N];
#pragma unroll
  for (int r = 0; r < N; ++r) { x1[r] = v[r] >> 1; x2[r] = x1[r] >> dbits; }
  vt_slots<N, 13, 12>(key1, key2, x1, x2, valid, s1, s2, f1, f2, probe_limit, err);
  uint32_t c[N];
#pragma unroll
  for (int r = 0; r < N; ++r) c[r] = valid[r] ? atomicAdd(&cnt[s1[r]], 1u) : 0u;
#pragma unroll
  for (int r = 0; r < N; ++r)
    if (valid[r]) {
      atomicMax(&best[s2[r]], ((unsigned long long)(c[r] + 1u) << 32) | (dmask - (x1[r] & dmask)));
      if (v[r] & 1u) atomicAdd(&ded[s2[r]], 1u);
    }
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned long long fb[N];
  uint32_t tot = 0, tot1 = 0;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    fb[r] = __ballot(f2[r]);
    tot += (uint32_t)__popcll(fb[r]);
    if (count1) tot1 += (uint32_t)__popcll(__ballot(f1[r]));
  }
  if (count1 && tot1 && lane == 0) atomicAdd(n1, tot1);
  if (tot == 0) return;   // uniform
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(n2, tot);
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
  for (int r = 0; r < N; ++r) {
    if (f2[r]) lst[base + (uint32_t)__popcll(fb[r] & lt)] = (uint16_t)s2[r];
    base += (uint32_t)__popcll(fb[r]);
  }
}

// maximum of v over the wave (all lanes get it): DPP inside the rows of 16 lanes, then the four row results
__device__ __forceinline__ uint32_t vt_wave_max(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141 /* row_half_mirror */, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140 /* row_mirror */, 0xF, 0xF, true));
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(r0, r1), max(r2, r3));
}


// Persistent workgroups, tile g = blockIdx.x, + gridDim.x, ...  Per tile (and per sweep of an over-full tile):
//   A  every vote: vt_votes (both tables at once)
//   C  wave 0: top-n of the listed songs (+ the candidates of earlier sweeps)
//   D  both tables cleared for the next tile
// The first VT_ROWS rows of the NEXT tile's votes are requested after A, so that they arrive during C and D.
#define VT_ROWS 3

__global__ __launch_bounds__(VT_THREADS) void vt_fold_kernel(const uint32_t* __restrict__ k,
                                                             const uint2* __restrict__ ranges,
                                                             const uint32_t* __restrict__ n_ranges, uint32_t cap, vt_plan pl,
                                                             uint32_t topn, uint64_t* __restrict__ c_pack,
                                                             uint32_t* __restrict__ c_delta, uint32_t* __restrict__ c_dedup,
                                                             uint32_t probe_limit, uint32_t* __restrict__ err) {
  __shared__ uint4 t1[VT_SLOTS / 2];                  // table 1: key1[VT_SLOTS] | cnt[VT_SLOTS]
  __shared__ uint32_t key2[VT_SLOTS2];
  __shared__ unsigned long long best[VT_SLOTS2];     // count << 32 | dmask - delta
  __shared__ uint32_t ded[VT_SLOTS2];
  __shared__ uint16_t lst[VT_SLOTS2];
  __shared__ uint64_t s_cpack[2][VT_MAXTOPN];
  __shared__ uint32_t s_cdelta[2][VT_MAXTOPN], s_cdedup[2][VT_MAXTOPN];
  __shared__ uint32_t s_n1, s_n2;
  uint32_t* const key1 = (uint32_t*)t1;
  uint32_t* const cnt = key1 + VT_SLOTS;
  const uint32_t j = threadIdx.x, lane = j & 63, nt = min(*n_ranges, cap);
  if (blockIdx.x >= nt) return;   // the usual case: nothing was handed over
  const uint32_t dmask = (1u << pl.dbits) - 1u, smask = (pl.sb >= 32 ? ~0u : (1u << pl.sb) - 1u);
  const int slb = pl.g_lo - 1 - pl.dbits;           // song-id bits below the ordered ones: what sweeps may split by
  // ... and behind them the delta bits above the low VT_MAX_DBITS: one song of a long track (> 2^12 frames) may vote for
  // more distinct deltas than table 1 holds; its delta range is then swept in parts.  The song's entry of table 2 lives
  // through all parts of its deltas (counts of one (song, delta) pair are complete inside one part, so the maximum over
  // the parts is the maximum over all deltas; ties keep the smallest delta; the flagged votes add up).
  const int dsb = pl.dbits > VT_MAX_DBITS ? pl.dbits - VT_MAX_DBITS : 0;
  auto clear1 = [&]() {
    for (uint32_t i = j; i < VT_SLOTS / 4; i += VT_THREADS) t1[i] = make_uint4(VT_EMPTY, VT_EMPTY, VT_EMPTY, VT_EMPTY);
    for (uint32_t i = VT_SLOTS / 4 + j; i < VT_SLOTS / 2; i += VT_THREADS) t1[i] = make_uint4(0, 0, 0, 0);
  };
  clear1();
  for (uint32_t i = j; i < VT_SLOTS2; i += VT_THREADS) { key2[i] = VT_EMPTY; best[i] = 0; ded[i] = 0; }
  if (j == 0) { s_n1 = 0; s_n2 = 0; }
  uint32_t g = blockIdx.x;
  uint32_t a = 0, b = 0, pre[VT_ROWS];
  if (g < nt) { a = ranges[g].x; b = ranges[g].y; }
#pragma unroll
  for (int r = 0; r < VT_ROWS; ++r) { const uint32_t i = a + r * VT_THREADS + j; pre[r] = i < b ? k[i] : 0u; }
  __syncthreads();
  for (; g < nt; g += gridDim.x) {
    const uint32_t gn = g + gridDim.x;
    uint32_t na = 0, nb = 0;                         // (both 0 behind the last tile: nothing is loaded)
    if (gn < nt) { na = ranges[gn].x; nb = ranges[gn].y; }
    const bool checked = b - a > VT_LIMIT2;         // fewer votes than either table may hold: no sweep can overflow
    int cl = 0, ns = 0, nd = 0;                      // candidate list in use; song bits / delta bits the sweeps split by
    // more votes than table 1 holds keys (a single query's unsorted votes, mostly one per (song, delta)): start with
    // the sweeps that many distinct pairs need, instead of finding out at the end of a first sweep over everything
    if (b - a > VT_LIMIT)
      while (ns < slb && ((uint32_t)VT_LIMIT << ns) < b - a) ++ns;
    bool prefetched = false;
    for (bool done = (a >= b); !done;) {             // until a sweep count is found under which every sweep fits
      bool over = false;
      uint32_t over_songs = 0;                         // songs in table 2 when a sweep overflowed
      cl = 0;
      const int nsl = ns + nd;
      for (uint32_t sw = 0; sw < (1u << nsl) && !over; ++sw) {
        const uint32_t song_part = sw >> nd, delta_part = sw & ((1u << nd) - 1u);
        const bool last_of_song = delta_part == (1u << nd) - 1u;
        // ---- A
        if (!checked && nsl == 0 && b - a <= VT_ROWS * VT_THREADS) {   // the usual tile: its rows are in registers
          bool ok[VT_ROWS];
#pragma unroll
          for (int r = 0; r < VT_ROWS; ++r) ok[r] = a + r * VT_THREADS + j < b;
          vt_votes<VT_ROWS>(key1, cnt, key2, best, ded, lst, &s_n1, &s_n2, false, pre, ok, pl.dbits, dmask, probe_limit, err);
        } else {
          for (uint32_t row = 0, base = a; base < b; base += VT_THREADS, ++row) {
            const uint32_t i = base + j;
            uint32_t v[1] = {0};
            if (row < VT_ROWS && nsl == 0) v[0] = row == 0 ? pre[0] : row == 1 ? pre[1] : pre[2];
            else if (i < b) v[0] = k[i];
            const uint32_t x = v[0] >> 1;
            const bool ok[1] = {i < b && (ns == 0 || (((x >> pl.dbits) & ((1u << slb) - 1u)) >> (slb - ns)) == song_part) &&
                                (nd == 0 || ((x & dmask) >> (pl.dbits - nd)) == delta_part)};
            vt_votes<1>(key1, cnt, key2, best, ded, lst, &s_n1, &s_n2, true, v, ok, pl.dbits, dmask, probe_limit, err);
            if (checked) {
              __syncthreads();
              over = s_n1 > VT_LIMIT || s_n2 > VT_LIMIT2;      // uniform: read between two barriers
              over_songs = s_n2;
              __syncthreads();
              if (over) break;
            }
          }
        }
        static_assert(VT_ROWS == 3, "the row select above names pre[0..2]");
        __syncthreads();
        const uint32_t n2 = s_n2;
        if (!prefetched) {               // the next tile's first rows: in flight during C and D
          prefetched = true;
#pragma unroll
          for (int r = 0; r < VT_ROWS; ++r) { const uint32_t i = na + r * VT_THREADS + j; pre[r] = i < nb ? k[i] : 0u; }
        }
        // ---- C: rank = (count descending, song id ascending); the smallest delta reaching the count is in best
        if (j < 64 && !over && last_of_song) {
          constexpr int CE = 12;                     // songs per lane whose rank stays in registers (n2 <= 768: the usual tile)
          uint64_t cpk[CE];
          const bool cached = n2 <= 64 * CE;         // uniform
          if (cached) {
            uint32_t cs[CE];
#pragma unroll
            for (int u = 0; u < CE; ++u) { const uint32_t e = lane + 64 * u; cs[u] = e < n2 ? lst[e] : 0xFFFFFFFFu; }
#pragma unroll
            for (int u = 0; u < CE; ++u)
              cpk[u] = cs[u] == 0xFFFFFFFFu ? 0ull : ((best[cs[u]] >> 32) << 32) | (0xFFFFFFFFu - (key2[cs[u]] & smask));
          }
          uint64_t prev = ~0ull;
          const uint64_t old = (song_part > 0 && lane < topn) ? s_cpack[cl][lane] : 0ull;
          for (uint32_t n = 0; n < topn; ++n) {
            uint64_t m = old < prev ? old : 0ull;
            uint32_t ms = 0xFFFFFFFFu;               // slot of m (0xFFFFFFFF: the old candidate), index of the entry if cached
            if (cached) {
#pragma unroll
              for (int u = 0; u < CE; ++u)
                if (cpk[u] < prev && cpk[u] > m) { m = cpk[u]; ms = lane + 64 * u; }
            } else {
              for (uint32_t e = lane; e < n2; e += 64) {
                const uint32_t s = lst[e];
                const uint64_t pk = ((best[s] >> 32) << 32) | (0xFFFFFFFFu - (key2[s] & smask));
                if (pk < prev && pk > m) { m = pk; ms = s; }
              }
            }
            const uint32_t cmax = vt_wave_max((uint32_t)(m >> 32));
            const uint32_t lo = vt_wave_max((uint32_t)(m >> 32) == cmax ? (uint32_t)m : 0u);
            const uint64_t w = ((uint64_t)cmax << 32) | lo;     // the wave's best below prev (0: none left)
            if (w != 0 && m == w) {                              // one lane: packs are unique
              if (cached && ms != 0xFFFFFFFFu) ms = lst[ms];
              s_cpack[cl ^ 1][n] = w;
              s_cdelta[cl ^ 1][n] = ms == 0xFFFFFFFFu ? s_cdelta[cl][lane] : dmask - ((uint32_t)best[ms] & dmask);
              s_cdedup[cl ^ 1][n] = ms == 0xFFFFFFFFu ? s_cdedup[cl][lane] : ded[ms];
            }
            if (w == 0 && lane == 0) { s_cpack[cl ^ 1][n] = 0; s_cdelta[cl ^ 1][n] = 0; s_cdedup[cl ^ 1][n] = 0; }
            prev = w;   // 0 once exhausted: nothing ranks below it
          }
        }
        __syncthreads();
        // ---- D (also what an overflowing sweep leaves behind)
        clear1();
        if (last_of_song || over) {                  // table 2 lives through the parts of a song's delta range
          for (uint32_t e = j; e < n2; e += VT_THREADS) { const uint32_t s = lst[e]; key2[s] = VT_EMPTY; best[s] = 0; ded[s] = 0; }
          if (j == 0) s_n2 = 0;
        }
        if (j == 0) s_n1 = 0;
        if (!over && last_of_song) cl ^= 1;
        __syncthreads();
      }
      if (!over) { done = true; break; }
      // at nsl == slb + dsb a sweep is 2^VT_MAX_DBITS deltas of one song of the last group + less than VT_TILE other
      // votes, which fits.  Should it not (a broken invariant), the range is given up and the pass flagged: the host
      // repeats it through the full sort.
      if (nsl >= slb + dsb) {
        if (j < VT_MAXTOPN) s_cpack[0][j] = 0;
        if (j == 0) atomicOr(err, 2u);
        cl = 0;
        done = true;
      }
      // split by song ids while many songs share the sweep; a sweep of a handful of songs that still overflows holds one
      // long track's deltas: split those (a query against a stationary 10-minute track would otherwise walk through
      // every song-id level, each a full pass over the range, before the first delta split)
      if (ns < slb && (over_songs > 8 || nd >= dsb)) ++ns;
      else if (nd < dsb) ++nd;
      else ++ns;
      __syncthreads();
    }
    if (!prefetched) {   // an empty tile, or one given up
#pragma unroll
      for (int r = 0; r < VT_ROWS; ++r) { const uint32_t i = na + r * VT_THREADS + j; pre[r] = i < nb ? k[i] : 0u; }
    }
    if (j < topn) {
      const uint64_t o = (uint64_t)g * topn + j;
      const bool any = a < b;
      c_pack[o] = any ? s_cpack[cl][j] : 0ull;
      c_delta[o] = any ? s_cdelta[cl][j] : 0u;
      c_dedup[o] = any ? s_cdedup[cl][j] : 0u;
    }
    __syncthreads();   // the candidate lists are rewritten by the next tile
    a = na;
    b = nb;
  }
}

// ---- the usual path: one WAVE per tile of ~VW_CHUNK votes, streaming through it with wave-private tables.
// No barriers, no LDS counters: a CU runs nine such waves side by side, each hiding the others' LDS round trips.
// The wave adds whole groups (votes that share the ordered bits) to its tables until VW_FLUSH votes are in, and at
// the next group border merges the songs of that batch into its running top-n (lane n holds candidate n) and clears
// the tables.  A batch whose distinct (song, delta) pairs or songs outgrow the tables -- one group with hundreds of
// pairs: copies of one recording, stationary tones -- is handed to vt_fold_kernel as the range [batch start, next
// group border): at most VW_FLUSH + 64 votes in front of its last group, so that kernel's bound holds for it too.
#define VW_B1 9
#define VW_S1 (1 << VW_B1)     // (song | delta) entries of a batch
#define VW_LIMIT1 384
#define VW_FLUSH 64
#define VW_CHUNK 2048
// songs of a batch: 2^B2 slots, B2 = 7 (2 KB, 26 waves per CU) where a batch is expected to hold few songs, else 8
#define VW_HEAVY_PER_TILE 40   // a range handed over has more than 2^7 - 68 votes: at most 34 inside the nominal tile + the one around its last group
static_assert(VW_LIMIT1 + 64 < VW_S1, "a probe must find a free slot");
static_assert(VW_FLUSH + 64 <= VT_TILE, "a range handed to vt_fold_kernel has less than VT_TILE votes before its last group");

template <int VW_B2, bool QR = true>
__global__ __launch_bounds__(64) void vt_stream_kernel(const uint32_t* __restrict__ k, const uint32_t* __restrict__ tile_start,
                                                       vt_plan pl, uint32_t topn, uint64_t* __restrict__ c_pack,
                                                       uint32_t* __restrict__ c_delta, uint32_t* __restrict__ c_dedup,
                                                       uint32_t* __restrict__ n_heavy, uint2* __restrict__ heavy,
                                                       uint32_t* __restrict__ heavy_q, uint32_t heavy_cap,
                                                       uint32_t probe_limit, uint32_t* __restrict__ err) {
  constexpr int VW_S2 = 1 << VW_B2;
  constexpr uint32_t VW_LIMIT2 = VW_S2 - 68;        // + one row of new songs stays below VW_S2
  static_assert(VW_LIMIT2 + 64 < (uint32_t)VW_S2 && VW_LIMIT1 + 64 < VW_S1, "a probe must find a free slot in either table");
  __shared__ uint4 t1[VW_S1 / 2];                    // key1[VW_S1] | cnt[VW_S1]
  __shared__ uint4 t2[VW_S2];                        // key2[VW_S2] | ded[VW_S2] | best[VW_S2] (8 bytes each)
  uint32_t* const key1 = (uint32_t*)t1;
  uint32_t* const cnt = key1 + VW_S1;
  uint32_t* const key2 = (uint32_t*)t2;
  uint32_t* const ded = key2 + VW_S2;
  unsigned long long* const best = (unsigned long long*)(ded + VW_S2);
  const uint32_t g = blockIdx.x, lane = threadIdx.x;
  const uint32_t a = tile_start[g], b = tile_start[g + 1];
  const uint32_t dmask = (1u << pl.dbits) - 1u, smask = (pl.sb >= 32 ? ~0u : (1u << pl.sb) - 1u);
  uint64_t cp = 0;                                   // lane n < topn: candidate n of this tile
  uint32_t cdl = 0, cdd = 0;
  auto clear = [&]() {
#pragma unroll
    for (int i = 0; i < VW_S1 / 4 / 64; ++i) t1[lane + 64 * i] = make_uint4(VT_EMPTY, VT_EMPTY, VT_EMPTY, VT_EMPTY);
#pragma unroll
    for (int i = 0; i < VW_S1 / 4 / 64; ++i) t1[VW_S1 / 4 + lane + 64 * i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < (VW_S2 + 63) / 64; ++i) {     // key2 (VW_S2 / 4 vectors of EMPTY), then ded and best (zero)
      const uint32_t e = lane + 64 * i;
      if (e < VW_S2) t2[e] = e < VW_S2 / 4 ? make_uint4(VT_EMPTY, VT_EMPTY, VT_EMPTY, VT_EMPTY) : make_uint4(0, 0, 0, 0);
    }
  };
  if (a < b) {
    clear();
    uint32_t n1 = 0, n2 = 0, batch_start = a, batch_votes = 0, last_hi = 0xFFFFFFFFu;   // wave-uniform
    bool skip = false;                               // the current batch goes to vt_fold_kernel: look for its end
    auto insert = [&](bool act, uint32_t v) {        // whole wave
      const uint32_t x1[1] = {v >> 1}, x2[1] = {(v >> 1) >> pl.dbits};
      uint32_t s1[1], s2[1];
      bool f1[1], f2[1];
      const bool ok[1] = {act};
      vt_slots<1, VW_B1, VW_B2>(key1, key2, x1, x2, ok, s1, s2, f1, f2, probe_limit, err);
      if (act) {
        const uint32_t c = atomicAdd(&cnt[s1[0]], 1u);
        atomicMax(&best[s2[0]], ((unsigned long long)(c + 1u) << 32) | (dmask - (x1[0] & dmask)));
        if (v & 1u) atomicAdd(&ded[s2[0]], 1u);
      }
      n2 += (uint32_t)__popcll(__ballot(f2[0]));
      n1 += (uint32_t)__popcll(__ballot(f1[0]));
      batch_votes += (uint32_t)__popcll(__ballot(act));
    };
    auto flush = [&]() {                             // the batch's songs into the running top-n; tables emptied
      if (n2 != 0) {
        constexpr int CE = VW_S2 / 64;               // every slot of table 2: CE per lane
        uint64_t pk[CE];
#pragma unroll
        for (int u = 0; u < CE; ++u) {
          const uint32_t s = lane + 64 * u, kk = key2[s];
          pk[u] = kk == VT_EMPTY ? 0ull : ((best[s] >> 32) << 32) | (0xFFFFFFFFu - (kk & smask));
        }
        // the usual batch changes nothing: its best song ranks below the n-th candidate (noise votes count 1 each, and
        // the votes arrive by ascending song id, so among equal counts the earlier batches win) -- one maximum says so
        // (a song lies in one group, hence in one batch: the candidates are other songs, packs are unique)
        if (QR) {
          uint64_t bm = pk[0];
#pragma unroll
          for (int u = 1; u < CE; ++u) bm = pk[u] > bm ? pk[u] : bm;
          const uint32_t bh = vt_wave_max((uint32_t)(bm >> 32));
          const uint32_t bl = vt_wave_max((uint32_t)(bm >> 32) == bh ? (uint32_t)bm : 0u);
          const uint32_t nth_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cp >> 32), (int)topn - 1);
          const uint32_t nth_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cp, (int)topn - 1);
          if ((((uint64_t)bh << 32) | bl) < (((uint64_t)nth_hi << 32) | nth_lo)) { clear(); n1 = n2 = 0; return; }
        }
        uint64_t prev = ~0ull, ncp = 0;
        uint32_t ncdl = 0, ncdd = 0;
        for (uint32_t n = 0; n < topn; ++n) {
          uint64_t m = cp < prev ? cp : 0ull;        // (lanes >= topn hold 0)
          int mu = -1;                               // entry of m, -1: the old candidate
#pragma unroll
          for (int u = 0; u < CE; ++u)
            if (pk[u] < prev && pk[u] > m) { m = pk[u]; mu = u; }
          const uint32_t cmax = vt_wave_max((uint32_t)(m >> 32));
          const uint32_t lo = vt_wave_max((uint32_t)(m >> 32) == cmax ? (uint32_t)m : 0u);
          const uint64_t w = ((uint64_t)cmax << 32) | lo;
          if (w == 0) break;                         // uniform: nothing ranks below prev
          const int win = __ffsll((long long)__ballot(m == w)) - 1;   // one lane: packs are unique
          uint32_t wdl = cdl, wdd = cdd;
          if ((int)lane == win && mu >= 0) { const uint32_t s = lane + 64 * mu; wdl = dmask - ((uint32_t)best[s] & dmask); wdd = ded[s]; }
          wdl = (uint32_t)__shfl((int)wdl, win, 64);
          wdd = (uint32_t)__shfl((int)wdd, win, 64);
          if (lane == n) { ncp = w; ncdl = wdl; ncdd = wdd; }
          prev = w;
        }
        cp = ncp; cdl = ncdl; cdd = ncdd;
      }
      clear();
      n1 = n2 = 0;
    };
    auto hand_over = [&](uint32_t e) {               // [batch_start, e) to vt_fold_kernel
      if (lane == 0) {
        const uint32_t idx = atomicAdd(n_heavy, 1u);
        if (idx < heavy_cap) { heavy[idx] = make_uint2(batch_start, e); heavy_q[idx] = vt_query_of_tile(pl, g); }
        else atomicOr(err, 4u);                      // no room in the list: the range's votes would be lost -> the pass is repeated
      }
    };
    uint32_t r0, r1, r2, r3;                         // the next four rows of votes: loads in flight
    { uint32_t i = a + lane; r0 = i < b ? k[i] : 0u; i += 64; r1 = i < b ? k[i] : 0u; i += 64; r2 = i < b ? k[i] : 0u; i += 64; r3 = i < b ? k[i] : 0u; }
    for (uint32_t base = a; base < b; base += 64) {
      const uint32_t v = r0;
      r0 = r1; r1 = r2; r2 = r3;
      { const uint32_t i = base + 256 + lane; r3 = i < b ? k[i] : 0u; }
      const bool valid = base + lane < b;
      const uint32_t hi = v >> pl.g_lo;
      // the neighbour lane's upper bits (lane 0: the previous row's last) -- a DPP wave shift, not a trip through the LDS crossbar
      const uint32_t hp = (uint32_t)__builtin_amdgcn_update_dpp((int)last_hi, (int)hi, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
      const unsigned long long mg = __ballot(valid && hi != hp);     // lanes that open a group
      last_hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);    // (the last row is the only partial one)
      uint32_t lo = 0;                               // first lane of the row not dealt with yet
      while (lo < 64) {                              // uniform
        const unsigned long long rest = mg & ~((1ull << lo) - 1ull);
        if (skip) {
          if (!rest) break;
          lo = (uint32_t)__ffsll((long long)rest) - 1;
          hand_over(base + lo);
          skip = false;
          batch_start = base + lo;
          batch_votes = 0;
          continue;
        }
        uint32_t cut = 64;                           // the batch may end at the next group border once it is large enough
        if (batch_votes >= pl.flush && rest) cut = (uint32_t)__ffsll((long long)rest) - 1;
        if (cut == lo) { flush(); batch_start = base + lo; batch_votes = 0; continue; }
        insert(valid && lane >= lo && lane < cut, v);
        const bool over = n1 > VW_LIMIT1 || n2 > VW_LIMIT2;
        if (over) {
          clear();
          n1 = n2 = 0;
          if (cut < 64) { hand_over(base + cut); batch_start = base + cut; batch_votes = 0; }
          else skip = true;
        } else if (cut < 64) { flush(); batch_start = base + cut; batch_votes = 0; }
        lo = cut;
      }
    }
    if (skip) hand_over(b); else flush();
  }
  if (lane < topn) {
    const uint64_t o = (uint64_t)g * topn + lane;
    c_pack[o] = cp;
    c_delta[o] = cdl;
    c_dedup[o] = cdd;
  }
}

// ---- the same tile, filter first (round 4).  What decides a query's top-n are the (song, delta) pairs that many votes
// agree on; of the ~140 votes of a batch (a group of 16 songs against 1M songs: ~9 votes a song over ~1,000 deltas) nearly
// all are alone with their pair or share it with one other.  vt_stream_kernel pays two compare-and-swap probe chains, a
// 64-bit maximum and a table scan for every one of them, and is bound by instruction issue (171 vector + 211 scalar
// instructions per row of 64 votes, both ports ~70 % busy: profiles/r03f_pmc_match.json).  Here a batch first goes through
// a KEYLESS COUNTING FILTER: table 1's 4 KB as 4,096 eight-bit counters, one ds_add_rtn per vote on the counter its
// (song | delta) hashes to.  A counter is an upper bound of the count of every pair that maps to it -- never below -- so
// "no counter of the batch reached the bar" proves that no pair of the batch did: the batch is finished, with no probe
// loop, no key compare, no per-lane state (the test is one compare per row, OR-ed into a scalar mask).
// The bar is the larger of the tile's own n-th candidate and the QUERY's bar: every tile publishes its n-th candidate
// (atomicMax on a word per query), and the n-th best of any subset of a query's songs is a lower bound of the n-th best of
// all of them -- a song below it is not in the query's top-n whatever its tile's list holds.  So a tile does not climb from
// zero: after a query's first tiles the bar stands at the noise ceiling.  A batch that MAY reach it (no bar yet; a counter
// got there -- by a real pair or by pairs sharing a counter; the filter says nothing about ties, so "reached" includes
// "equalled") is read again and folded by the exact code of vt_stream_kernel: every candidate's count, smallest delta
// and row sum come from there, so the results are the same arrays (which tile reports a song, and whether a tile reports
// songs that end below the final top-n, depends on timing; the ranking of vt_rank_kernel does not).
__device__ __forceinline__ uint64_t vt_wave_max64(uint64_t v) {
  const uint32_t h = vt_wave_max((uint32_t)(v >> 32));
  const uint32_t l = vt_wave_max((uint32_t)(v >> 32) == h ? (uint32_t)v : 0u);
  return ((uint64_t)h << 32) | l;
}
// counter of a (song | delta) key, 12 bits: the multiply is 24-bit (full rate); the bits above 24 are folded in first
#ifndef VW_SEED_TILES
#define VW_SEED_TILES 64u      // tiles at the head of a query that fold undecided batches on the spot (vt_stream2_kernel)
#endif
#define VW_FILTER_MAX 200u     // a counter that gets here sends its batch to the exact fold whatever the bar (8 bits wrap at 256)
__device__ __forceinline__ uint32_t vw_hash_filter(uint32_t x) { return (__umul24((x ^ (x >> 13)) & 0xFFFFFFu, 0x9E3779u) >> 11) & 4095u; }
static_assert(VW_S1 * 8 == 4096, "table 1 (keys + counts) is 4 KB: 4,096 eight-bit counters");

template <int VW_B2>
__global__ __launch_bounds__(64) void vt_stream2_kernel(const uint32_t* __restrict__ k, const uint32_t* __restrict__ tile_start,
                                                        vt_plan pl, uint32_t topn, uint64_t* __restrict__ c_pack,
                                                        uint32_t* __restrict__ c_delta, uint32_t* __restrict__ c_dedup,
                                                        uint32_t* __restrict__ n_heavy, uint2* __restrict__ heavy,
                                                        uint32_t* __restrict__ heavy_q, uint32_t heavy_cap,
                                                        uint32_t probe_limit, unsigned long long* __restrict__ qbar,
                                                        uint32_t* __restrict__ err, unsigned long long* __restrict__ stats) {
  constexpr int VW_S2 = 1 << VW_B2;
  constexpr uint32_t VW_LIMIT2 = VW_S2 - 68;        // + one row of new songs stays below VW_S2
  static_assert(VW_LIMIT2 + 64 < (uint32_t)VW_S2 && VW_LIMIT1 + 64 < VW_S1, "a probe must find a free slot in either table");
  uint32_t st_b = 0, st_seed = 0, st_beat = 0, st_votes_redo = 0;   // SHZ_VT_STATS
  __shared__ uint4 t1[VW_S1 / 2];                    // key1[VW_S1] | cnt[VW_S1]
  __shared__ uint4 t2[VW_S2];                        // key2[VW_S2] | ded[VW_S2] | best[VW_S2] (8 bytes each)
  uint32_t* const key1 = (uint32_t*)t1;
  uint32_t* const cnt = key1 + VW_S1;
  uint32_t* const key2 = (uint32_t*)t2;
  uint32_t* const ded = key2 + VW_S2;
  unsigned long long* const best = (unsigned long long*)(ded + VW_S2);
  const uint32_t g = blockIdx.x, lane = threadIdx.x;
  const uint32_t a = tile_start[g], b = tile_start[g + 1];
  const uint32_t dmask = (1u << pl.dbits) - 1u, smask = (pl.sb >= 32 ? ~0u : (1u << pl.sb) - 1u);
  uint64_t cp = 0;                                   // lane n < topn: candidate n of this tile
  uint32_t cdl = 0, cdd = 0;
  auto clear1 = [&]() {                              // table 1
#pragma unroll
    for (int i = 0; i < VW_S1 / 4 / 64; ++i) t1[lane + 64 * i] = make_uint4(VT_EMPTY, VT_EMPTY, VT_EMPTY, VT_EMPTY);
#pragma unroll
    for (int i = 0; i < VW_S1 / 4 / 64; ++i) t1[VW_S1 / 4 + lane + 64 * i] = make_uint4(0, 0, 0, 0);
  };
  auto clear2 = [&]() {
#pragma unroll
    for (int i = 0; i < (VW_S2 + 63) / 64; ++i) {     // key2 (VW_S2 / 4 vectors of EMPTY), then ded and best (zero)
      const uint32_t e = lane + 64 * i;
      if (e < VW_S2) t2[e] = e < VW_S2 / 4 ? make_uint4(VT_EMPTY, VT_EMPTY, VT_EMPTY, VT_EMPTY) : make_uint4(0, 0, 0, 0);
    }
  };
  auto clear_filter = [&]() {                        // table 1 as the filter's counters
#pragma unroll
    for (int i = 0; i < VW_S1 / 2 / 64; ++i) t1[lane + 64 * i] = make_uint4(0, 0, 0, 0);
  };
  uint32_t* const filt = (uint32_t*)t1;
  if (a < b) {
    const uint32_t qi = vt_query_of_tile(pl, g);
    unsigned long long bar_q = qbar[qi];             // the query's bar as this tile last saw it (it only rises)
    clear_filter();
    clear2();                                        // (table 2 is left empty by whoever used it)
    uint32_t n1 = 0, n2 = 0;                         // wave-uniform (exact fold only)
    // ---------------- the exact fold of one batch [s, e): vt_stream_kernel's, on a range that is known
    auto insert = [&](bool act, uint32_t v) {        // whole wave
      const uint32_t x1[1] = {v >> 1}, x2[1] = {(v >> 1) >> pl.dbits};
      uint32_t s1[1], s2[1];
      bool f1[1], f2[1];
      const bool ok[1] = {act};
      vt_slots<1, VW_B1, VW_B2>(key1, key2, x1, x2, ok, s1, s2, f1, f2, probe_limit, err);
      if (act) {
        const uint32_t c = atomicAdd(&cnt[s1[0]], 1u);
        atomicMax(&best[s2[0]], ((unsigned long long)(c + 1u) << 32) | (dmask - (x1[0] & dmask)));
        if (v & 1u) atomicAdd(&ded[s2[0]], 1u);
      }
      n2 += (uint32_t)__popcll(__ballot(f2[0]));
      n1 += (uint32_t)__popcll(__ballot(f1[0]));
    };
    auto merge = [&]() {                             // the batch's songs into the running top-n
      if (n2 == 0) return;
      constexpr int CE = VW_S2 / 64;                 // every slot of table 2: CE per lane
      uint64_t pk[CE];
#pragma unroll
      for (int u = 0; u < CE; ++u) {
        const uint32_t s = lane + 64 * u, kk = key2[s];
        pk[u] = kk == VT_EMPTY ? 0ull : ((best[s] >> 32) << 32) | (0xFFFFFFFFu - (kk & smask));
      }
      uint64_t prev = ~0ull, ncp = 0;
      uint32_t ncdl = 0, ncdd = 0;
      for (uint32_t n = 0; n < topn; ++n) {
        uint64_t m = cp < prev ? cp : 0ull;          // (lanes >= topn hold 0)
        int mu = -1;                                 // entry of m, -1: the old candidate
#pragma unroll
        for (int u = 0; u < CE; ++u)
          if (pk[u] < prev && pk[u] > m) { m = pk[u]; mu = u; }
        const uint64_t w = vt_wave_max64(m);
        if (w == 0) break;                           // uniform: nothing ranks below prev
        const int win = __ffsll((long long)__ballot(m == w)) - 1;   // one lane: packs are unique
        uint32_t wdl = cdl, wdd = cdd;
        if ((int)lane == win && mu >= 0) { const uint32_t s = lane + 64 * mu; wdl = dmask - ((uint32_t)best[s] & dmask); wdd = ded[s]; }
        wdl = (uint32_t)__shfl((int)wdl, win, 64);
        wdd = (uint32_t)__shfl((int)wdd, win, 64);
        if (lane == n) { ncp = w; ncdl = wdl; ncdd = wdd; }
        prev = w;
      }
      cp = ncp; cdl = ncdl; cdd = ncdd;
    };
    auto hand_over = [&](uint32_t s, uint32_t e) {   // [s, e) to vt_fold_kernel
      if (lane == 0) {
        const uint32_t idx = atomicAdd(n_heavy, 1u);
        if (idx < heavy_cap) { heavy[idx] = make_uint2(s, e); heavy_q[idx] = qi; }
        else atomicOr(err, 4u);                      // no room in the list: the range's votes would be lost -> the pass is repeated
      }
    };
    auto nth_own = [&]() -> uint64_t {
      const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cp >> 32), (int)topn - 1);
      const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cp, (int)topn - 1);
      return ((uint64_t)h << 32) | l;
    };
    auto exact_batch = [&](uint32_t s, uint32_t e) { // table 2 empty on entry and on return; table 1 is handed back as the filter
      clear1();
      n1 = n2 = 0;
      bool over = false;
      for (uint32_t base = s; base < e && !over; base += 64) {
        const uint32_t i = base + lane;
        const bool act = i < e;
        insert(act, act ? k[i] : 0u);
        over = n1 > VW_LIMIT1 || n2 > VW_LIMIT2;
      }
      if (over) hand_over(s, e); else merge();
      clear_filter();
      clear2();
      // the tile's n-th candidate raises the query's bar for every tile still running; and this tile looks at the bar again
      const uint64_t mine = nth_own();
      if (mine > bar_q) {
        unsigned long long seen = 0;
        if (lane == 0) seen = atomicMax(&qbar[qi], (unsigned long long)mine);
        seen = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(seen >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)seen);
        bar_q = seen > mine ? seen : mine;
      }
    };
    // ---------------- the stream
    uint32_t batch_start = a, batch_votes = 0, last_hi = 0xFFFFFFFFu;   // wave-uniform
    unsigned long long hit = 0;                      // wave-uniform: lanes whose counter reached the bar in this batch
    uint32_t cm = 0;                                 // per lane: the largest counter value it saw in this batch
    uint32_t cur_hi = 0;                             // the ordered bits of the batch's first vote
    uint32_t thr = 2u;                               // a counter at or above this marks the batch (`hit`)
    uint32_t thr1 = 0u;                              // the count a pair of the batch needs to get past the bar (0: anything does)
    uint64_t own = 0;                                // the tile's n-th candidate (0: the list is not full)
    const int slb = pl.g_lo - 1 - pl.dbits;          // bits of a song id below the ordered ones: song id >= (hi << slb)
    // the smallest count that lets a pair of a batch whose lowest possible id is first_hi << slb past `bar`
    auto thr_of = [&](uint64_t bar, uint32_t first_hi) -> uint32_t {
      uint32_t barc = (uint32_t)(bar >> 32);
      // a pair that only EQUALS the bar's count wins on the smaller song id: never, when every id of the batch (ids ascend
      // along a tile, and tile after tile) lies above the bar's song -- then a counter must EXCEED the bar's count
      const uint32_t bar_sid = 0xFFFFFFFFu - (uint32_t)bar;
      if (barc != 0u && ((uint64_t)first_hi << slb) > (uint64_t)bar_sid) ++barc;
      return barc;
    };
    auto set_thr = [&](uint32_t first_hi) {          // first_hi: the ordered bits of the batch's first vote
      cur_hi = first_hi;
      own = nth_own();
      thr1 = thr_of(own > bar_q ? own : bar_q, first_hi);
      thr = thr1 < 2u ? 2u : (thr1 > VW_FILTER_MAX ? VW_FILTER_MAX : thr1);   // (count 1 is not the filter's business: thr1 < 2 is)
    };
    auto refresh_bar = [&]() {                       // the query's bar as it stands now
      unsigned long long now = 0;
      if (lane == 0) now = __hip_atomic_load(&qbar[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      now = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(now >> 32)) << 32) |
            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)now);
      if (now > bar_q) bar_q = now;
    };
    // does a batch whose largest counter was cmax (>= 1: every vote is a pair of count 1) still need the exact fold, with
    // the tile's list and the query's bar as they are NOW?  (Exact batches of a tile run in ascending order of ids, so the
    // tile's own candidates come from lower ids than the batch's.)
    auto needed = [&](uint32_t cmax, uint32_t first_hi) -> bool {
      own = nth_own();
      return cmax >= VW_FILTER_MAX || cmax >= thr_of(own > bar_q ? own : bar_q, first_hi);
    };
    // DEFERRED batches.  All tiles of a query start together with no bar at all, and a bar worth having (count 2 at a low
    // id, which makes every later batch need count 3) only exists once a few tiles have folded a few batches exactly; a tile
    // that decided every batch on the spot paid one exact fold for its first batch (no bar) and more for the counter
    // collisions of the next ones (count-1 bar) -- 60 % of the time of a single query's fold (measured with the bar left
    // over from the previous call: 0.345 -> 0.283 ms a query at 1M songs).  So only the query's first VW_SEED_TILES tiles
    // (the lowest ids: their bars are the ones that bind everybody else) decide on the spot; the others note an undecided
    // batch -- its range, first ordered bits and largest counter, in the registers of lane i for note i -- and decide
    // when the tile's stream has ended (or 64 notes are there), in order, against the bar of that moment.
    // (a sixteenth of the query's tiles, between 4 and VW_SEED_TILES: 100k songs, ~110 tiles a query: 9.95 us a query with 64
    // seeds -- nearly every tile folding on the spot --, 9.25 with 8; 1M songs, 1,100-4,400 tiles: 4 / 16 / 64 within 2 %)
    const uint32_t q_tiles = pl.tb[qi + 1] - pl.tb[qi];
    const uint32_t n_seed = min(max(q_tiles >> 4, 4u), VW_SEED_TILES);
    const bool seed_tile = g - pl.tb[qi] < n_seed;
    uint32_t npend = 0;                              // wave-uniform
    uint32_t pd_s = 0, pd_e = 0, pd_hi = 0, pd_c = 0;   // lane i: note i
    auto run_pending = [&]() {
      if (npend == 0) return;
      refresh_bar();
      for (uint32_t i = 0; i < npend; ++i) {         // uniform
        const uint32_t ps = (uint32_t)__builtin_amdgcn_readlane((int)pd_s, (int)i), pe = (uint32_t)__builtin_amdgcn_readlane((int)pd_e, (int)i);
        const uint32_t ph = (uint32_t)__builtin_amdgcn_readlane((int)pd_hi, (int)i), pc = (uint32_t)__builtin_amdgcn_readlane((int)pd_c, (int)i);
        if (needed(pc, ph)) { ++st_beat; st_votes_redo += pe - ps; exact_batch(ps, pe); }   // (which looks at the bar again)
      }
      npend = 0;
    };
    // ONE place ends a batch (the row loop below) and ONE place folds exactly (run_pending): the exact fold is a long piece
    // of code -- probe loops, the top-n merge -- and the seven inlined copies it had made a 16,000-instruction kernel.
    auto end_batch = [&](uint32_t e, uint32_t next_hi, bool last) {   // next_hi: the ordered bits of the vote at e (the next batch's first)
      ++st_b;
      bool noted = false;
      if (hit != 0ull || thr1 < 2u) {                // a counter reached the batch's threshold, or songs of count 1 may still enter
        const uint32_t w = vt_wave_max(cm), cmax = w < 1u ? 1u : w;
        ++st_seed;
        if (lane == npend) { pd_s = batch_start; pd_e = e; pd_hi = cur_hi; pd_c = cmax; }
        ++npend;
        noted = true;
      }
      clear_filter();
      // decide now: a seed tile's note (on the spot), 64 notes, the end of the tile.  Otherwise a tile that keeps noting batches
      // has a stale bar (none, if it started with its query): it looks again now and then -- with a bar worth having, notes
      // become rare and so do these loads
      if (npend != 0u && (seed_tile || npend == 64u || last)) run_pending();
      else if (noted && (npend & 7u) == 0u) refresh_bar();
      set_thr(next_hi);
      hit = 0ull;
      cm = 0;
      batch_start = e;
      batch_votes = 0;
    };
    // one row's votes of the lanes `act` through the filter: +1 on the counter of (song | delta); lanes outside add nothing
    auto fast_insert = [&](bool act, uint32_t v) {
      const uint32_t h = vw_hash_filter(v >> 1), sh = (h & 3u) << 3;
      const uint32_t old = atomicAdd(&filt[h >> 2], act ? 1u << sh : 0u);
      const uint32_t c = act ? ((old >> sh) & 0xFFu) + 1u : 0u;
      cm = c > cm ? c : cm;
      hit |= __ballot(c >= thr);
    };
    // the same in two halves: a whole row's update is ISSUED, and its counter values are looked at one row later -- after
    // the next row's update has been issued -- so that an LDS round trip is always in flight beside the next row's work
    uint32_t p_old = 0, p_sh = 0;
    bool p_any = false;                              // wave-uniform: a row's values are still to be looked at
    auto consume = [&]() {
      if (!p_any) return;
      const uint32_t c = ((p_old >> p_sh) & 0xFFu) + 1u;   // (only whole rows are deferred: every lane took part)
      cm = c > cm ? c : cm;
      hit |= __ballot(c >= thr);
      p_any = false;
    };
    // the next VW_AHEAD rows of votes: loads in flight.  The wave waits for memory, not for issue slots (SQ counters: 90 % of
    // its cycles) -- 26 waves a CU with 4 rows each in flight are 6.8 MB on the whole chip, what ~3 us of loaded latency
    // turn into ~2 TB/s at best
#ifndef VW_AHEAD_N
#define VW_AHEAD_N 8
#endif
    constexpr int VW_AHEAD = VW_AHEAD_N;
    uint32_t rr[VW_AHEAD];
#pragma unroll
    for (int j = 0; j < VW_AHEAD; ++j) { const uint32_t i = a + lane + 64u * j; rr[j] = i < b ? k[i] : 0u; }
    for (uint32_t base = a;; base += 64) {
      // behind the last vote comes a pseudo-row with a border at lane 0: it ends the last batch where every batch ends
      const bool tail = base >= b;
      uint32_t v = 0, hi = 0;
      unsigned long long mg = 1ull;
      bool valid = false;
      if (!tail) {
        v = rr[0];
#pragma unroll
        for (int j = 0; j + 1 < VW_AHEAD; ++j) rr[j] = rr[j + 1];
        { const uint32_t i = base + 64u * VW_AHEAD + lane; rr[VW_AHEAD - 1] = i < b ? k[i] : 0u; }
        valid = base + lane < b;
        hi = v >> pl.g_lo;
        const uint32_t hp = (uint32_t)__builtin_amdgcn_update_dpp((int)last_hi, (int)hi, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        mg = __ballot(valid && hi != hp);                            // lanes that open a group
        last_hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);  // (the last row is the only partial one)
        if (base == a) set_thr((uint32_t)__builtin_amdgcn_readfirstlane((int)hi));
        if (mg == 0ull || batch_votes < pl.flush) {                  // no border in the row, or none that may end the batch (it is
          // still below its flush size when the row begins: the whole row joins it, as in vt_stream_kernel): one piece
          if (base + 64u <= b) {                                     // a whole row: issued now, looked at with the next row
            const uint32_t h = vw_hash_filter(v >> 1), sh = (h & 3u) << 3;
            const uint32_t old = atomicAdd(&filt[h >> 2], 1u << sh);
            consume();
            p_old = old; p_sh = sh; p_any = true;
            batch_votes += 64u;
          } else {
            consume();
            fast_insert(valid, v);
            batch_votes += b - base;
          }
          continue;
        }
      }
      consume();
      uint32_t lo = 0;                               // first lane of the row not dealt with yet
      while (lo < 64) {                              // uniform
        const unsigned long long rest = mg & ~((1ull << lo) - 1ull);
        uint32_t cut = 64;                           // the batch may end at the next group border once it is large enough
        if ((batch_votes >= pl.flush || tail) && rest) cut = (uint32_t)__ffsll((long long)rest) - 1;
        if (cut == lo) {
          end_batch(tail ? b : base + lo, tail ? 0xFFFFFFFFu : (uint32_t)__builtin_amdgcn_readlane((int)hi, (int)lo), tail);
          if (tail) break;
          continue;
        }
        const bool act = valid && lane >= lo && lane < cut;
        fast_insert(act, v);
        batch_votes += (uint32_t)__popcll(__ballot(act));
        lo = cut;                                    // (a border at `cut` ends the batch at the top of this loop)
      }
      if (tail) break;
    }
    if (stats && lane == 0) {
      atomicAdd(stats + 0, (unsigned long long)st_b);
      atomicAdd(stats + 2, (unsigned long long)st_seed); atomicAdd(stats + 3, (unsigned long long)st_beat);
      atomicAdd(stats + 6, (unsigned long long)st_votes_redo); atomicAdd(stats + 7, (unsigned long long)(b - a));
    }
  }
  if (lane < topn) {
    const uint64_t o = (uint64_t)g * topn + lane;
    c_pack[o] = cp;
    c_delta[o] = cdl;
    c_dedup[o] = cdd;
  }
}

// one query whose votes one workgroup can fold as they come out of the expand (no radix pass at all): the whole pass is
// the one range handed to vt_fold_kernel
__global__ void vt_one_range_kernel(uint32_t* __restrict__ n_heavy, uint2* __restrict__ heavy, uint32_t* __restrict__ heavy_q,
                                    uint32_t n_votes) {
  *n_heavy = 1u;
  heavy[0] = make_uint2(0u, n_votes);
  heavy_q[0] = 0u;
}

// one workgroup per query of the pass: top-n of its tiles' candidates
#define VR_THREADS 1024
#define VR_CACHE 12            // candidates a thread keeps in registers: 12,288 per query before it re-reads them

__global__ __launch_bounds__(VR_THREADS) void vt_rank_kernel(vt_plan pl, uint32_t topn, m_bits mb, const uint64_t* __restrict__ c_pack,
                                                             const uint32_t* __restrict__ c_delta,
                                                             const uint32_t* __restrict__ c_dedup,
                                                             const uint32_t* __restrict__ n_heavy,
                                                             const uint32_t* __restrict__ heavy_q, uint32_t heavy_cap,
                                                             uint32_t* __restrict__ out_sid,
                                                             int32_t* __restrict__ out_delta, uint32_t* __restrict__ out_aligned,
                                                             uint32_t* __restrict__ out_dedup, uint32_t* __restrict__ out_nres) {
  __shared__ uint64_t s_best[2][VR_THREADS / 64];
  const uint32_t q = blockIdx.x, j = threadIdx.x;
  const uint32_t c0 = pl.tb[q] * topn, nc = (pl.tb[q + 1] - pl.tb[q]) * topn;
  const uint32_t nt = pl.tb[pl.nq], nh = min(*n_heavy, heavy_cap) * topn;   // candidates of range r: slots (nt + r) * topn ...
  const bool cached = nc <= VR_CACHE * VR_THREADS;   // uniform
  uint64_t ck[VR_CACHE];
  if (cached) {
#pragma unroll
    for (int u = 0; u < VR_CACHE; ++u) { const uint32_t i = j + u * VR_THREADS; ck[u] = i < nc ? c_pack[c0 + i] : 0ull; }
  }
  uint64_t prev = ~0ull;
  uint32_t found = 0;
  for (uint32_t n = 0; n < topn; ++n) {
    uint64_t best = 0;
    uint32_t bestr = 0xFFFFFFFFu;
    if (cached) {
#pragma unroll
      for (int u = 0; u < VR_CACHE; ++u)
        if (ck[u] < prev && ck[u] > best) { best = ck[u]; bestr = c0 + j + u * VR_THREADS; }
    } else {
      for (uint32_t i = j; i < nc; i += VR_THREADS) {
        const uint64_t packed = c_pack[c0 + i];
        if (packed < prev && packed > best) { best = packed; bestr = c0 + i; }
      }
    }
    for (uint32_t i = j; i < nh; i += VR_THREADS) {    // (usually none)
      if (heavy_q[i / topn] != q) continue;
      const uint64_t packed = c_pack[(uint64_t)nt * topn + i];
      if (packed < prev && packed > best) { best = packed; bestr = nt * topn + i; }
    }
    // the block's best: count first, then the low word, inside the waves; the 16 wave results through LDS
    const uint32_t cmax = vt_wave_max((uint32_t)(best >> 32));
    const uint32_t lo = vt_wave_max((uint32_t)(best >> 32) == cmax ? (uint32_t)best : 0u);
    if ((j & 63) == 0) s_best[n & 1][j >> 6] = ((uint64_t)cmax << 32) | lo;
    __syncthreads();
    uint64_t w = 0;
#pragma unroll
    for (int i = 0; i < VR_THREADS / 64; ++i) w = s_best[n & 1][i] > w ? s_best[n & 1][i] : w;
    if (w == 0) break;  // uniform
    if (best == w) {    // one thread: packs are unique
      const uint64_t o = (uint64_t)q * topn + n;
      out_sid[o] = 0xFFFFFFFFu - (uint32_t)w;
      out_aligned[o] = (uint32_t)(w >> 32);
      out_delta[o] = (int32_t)((int64_t)c_delta[bestr] - (int64_t)mb.bias);
      out_dedup[o] = c_dedup[bestr];
    }
    prev = w;
    ++found;
  }
  if (j == 0) out_nres[q] = found;
}

// shz_match_pairs: the packed votes themselves leave match_core (device buffer of `cap` entries), in a key layout
// the caller chose for ALL shards: ((q << sb | sid) << dbits | delta + bias) << 1 | first-offset flag, q global
struct pair_sink {
  uint64_t* d_pairs;
  uint64_t cap, count;
  m_bits lay;
  uint32_t shard, nshards;  // nshards > 1: only the query hashes this shard owns are looked up
};

// sort the packed votes, fold every (query, sid) group into one record and pick the top-n per query into device
// result arrays (r_*: nq*topn / nq entries, zeroed by the caller).  v0 holds the P votes, v1 is scratch of the same size.
static int32_t vote_fold(shz_ctx* ctx, const uint64_t* vs, uint64_t P, uint32_t nq, m_bits mb, uint32_t topn,
                         uint64_t* d_tot, uint32_t* r_sid, int32_t* r_delta, uint32_t* r_al, uint32_t* r_dd, uint32_t* r_n);

static int32_t vote_tail(shz_ctx* ctx, uint64_t* v0, uint64_t* v1, uint64_t P, uint32_t nq, m_bits mb, uint32_t topn,
                         uint64_t* d_tot, uint32_t* r_sid, int32_t* r_delta, uint32_t* r_al, uint32_t* r_dd, uint32_t* r_n) {
  int sel = 0;
  // bit 0 (the first-offset flag) is only counted by the fold, never compared: it stays out of the sort
  if (P <= MH_MAX) {
    hipLaunchKernelGGL(m_sort_small_kernel, dim3(1), dim3(MH_THREADS), 0, ctx->stream, (const uint64_t*)v0, v1, (uint32_t)P, 1,
                       mb.qb + mb.sb + mb.dbits + 1);
    SHZ_HIP(ctx, hipGetLastError());
    sel = 1;
  } else {
    SHZ_TRY(shz_sort_u64(ctx, v0, v1, nullptr, nullptr, 0, P, 1, mb.qb + mb.sb + mb.dbits + 1, &sel));
  }
  return vote_fold(ctx, sel ? v1 : v0, P, nq, mb, topn, d_tot, r_sid, r_delta, r_al, r_dd, r_n);
}

// sorted votes -> one record per (query, song) group -> top-n per query
static int32_t vote_fold(shz_ctx* ctx, const uint64_t* vs, uint64_t P, uint32_t nq, m_bits mb, uint32_t topn,
                         uint64_t* d_tot, uint32_t* r_sid, int32_t* r_delta, uint32_t* r_al, uint32_t* r_dd, uint32_t* r_n) {
  // group heads per wave -> first record slot of every wave -> one record per (query, sid) group
  const uint32_t nb = nblk((P + RG_PER - 1) / RG_PER), nw = nb * 4;
  void *wc, *gh, *gd, *gdd;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, ((uint64_t)nw * 2 + nq + 2) * 4 + (uint64_t)RG_LONG_CAP * 8, &wc));
  uint32_t *wcnt = (uint32_t*)wc, *wbase = wcnt + nw, *qstart = wbase + nw;
  uint32_t* long_cnt = qstart + nq;
  uint2* long_list = (uint2*)(long_cnt + 1 + ((nw * 2 + nq + 1) & 1));   // 8-byte aligned behind the counter
  hipLaunchKernelGGL(m_gcount_kernel, dim3(nb), dim3(256), 0, ctx->stream, vs, P, mb.dbits + 1, wcnt);
  SHZ_HIP(ctx, hipGetLastError());
  SHZ_TRY(shz_scan_u32(ctx, wcnt, wbase, nw, d_tot));   // *d_tot = G, the number of (query, song) groups: stays on the device
  hipLaunchKernelGGL(m_tail_init_kernel, dim3(nblk((uint64_t)nq + 1)), dim3(256), 0, ctx->stream, qstart, nq);   // qstart = none, long_cnt = 0
  // G <= P, and a group needs a vote per song: the record arrays are sized by that bound instead of a read-back of G
  const uint64_t Gb = P;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M2, Gb * 8, &gh));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M5, Gb * 4, &gd));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M6, Gb * 4, &gdd));
  const unsigned long long* d_G = (const unsigned long long*)d_tot;
  hipLaunchKernelGGL(m_reduce_kernel, dim3(nb), dim3(256), 0, ctx->stream, vs, P, mb, (const uint32_t*)wbase, (uint64_t*)gh,
                     (uint32_t*)gd, (uint32_t*)gdd, qstart, long_cnt, long_list);
  hipLaunchKernelGGL(m_reduce_long_kernel, dim3(std::min<uint32_t>(1024, nb)), dim3(256), 0, ctx->stream, vs, P, mb,
                     (const uint32_t*)long_cnt, (const uint2*)long_list, (uint64_t*)gh, (uint32_t*)gd, (uint32_t*)gdd);
  // groups per query decide the shape: one workgroup per query, or C slices per query and a final ranking.  The bound
  // votes / queries stands in for groups / queries (more slices than needed cost a few idle workgroups).
  const uint32_t C = (uint32_t)std::min<uint64_t>(512, (P / nq + 16383) / 16384);
  if (C <= 1) {
    hipLaunchKernelGGL(m_topn_kernel, dim3(nq), dim3(256), 0, ctx->stream, (const uint32_t*)qstart, d_G, mb, (const uint64_t*)gh,
                       (const uint32_t*)gd, (const uint32_t*)gdd, nq, topn, r_sid, r_delta, r_al, r_dd, r_n);
  } else {
    void *pp, *pr;
    const uint64_t ncand = (uint64_t)C * topn;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M8, (uint64_t)nq * ncand * 8, &pp));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M9, (uint64_t)nq * ncand * 4, &pr));
    hipLaunchKernelGGL(m_topn_partial_kernel, dim3(C, nq), dim3(256), 0, ctx->stream, (const uint32_t*)qstart, d_G, nq,
                       (const uint64_t*)gh, topn, (uint64_t*)pp, (uint32_t*)pr);
    hipLaunchKernelGGL(m_topn_final_kernel, dim3(nq), dim3(256), 0, ctx->stream, (const uint64_t*)pp, (const uint32_t*)pr,
                       (uint32_t)ncand, mb, (const uint32_t*)gd, (const uint32_t*)gdd, nq, topn, r_sid, r_delta, r_al, r_dd, r_n);
  }
  SHZ_HIP(ctx, hipGetLastError());
  return SHZ_OK;
}

// ---- one vote-tile pass, in two steps so that the producer of the votes may use the plan (m_expand_blocks_kernel counts the
// first radix pass by the sort's blocks)
// plan from the votes per query of the pass: vote and tile offsets per query, the bits the radix passes order
static void vt_make_plan(const uint64_t* counts, uint32_t nqp, const m_bits& mbp, vt_plan& pl, shz_seg_plan& sp) {
  const int Bt = mbp.sb + mbp.dbits + 1;   // the votes of a tile pass carry no query bits
  pl.nq = sp.nq = nqp;
  pl.dbits = mbp.dbits;
  pl.sb = mbp.sb;
  pl.g_lo = std::max(1 + mbp.dbits, Bt - VT_ORDERED_BITS);
  // votes per tile: a tile pays a fixed price (its place in the plan, the query's bar, the first loads' latency, its
  // candidates' way through vt_rank_kernel) that 2,048 votes do not amortise -- 1M songs, batches of 200: 0.147 ms a query at
  // 2,048, 0.141 at 4,096, 0.138 at 8,192, 0.136 at 16,384 -- but the chip wants ~2 rounds of tiles (256 CUs x 26 waves).
  // (Those were the fold of mid round 4; with deferred batches and batches of 128: 0.1365 at 2,048, 0.1348 at 4,096, 0.132 at
  // 8,192, 0.1348 at 16,384, 0.1418 at 32,768: the cap is 8,192.)
  static const uint32_t chunk_env = [] { const char* e = getenv("SHZ_VW_CHUNK"); const int v = e ? atoi(e) : 0; return v >= 1024 && v <= 65536 ? (uint32_t)v : 0u; }();
  uint64_t all_votes = 0;
  for (uint32_t i = 0; i < nqp; ++i) all_votes += counts[i];
  uint32_t chunk = VW_CHUNK;
  while (chunk < 8192u && all_votes / (2ull * chunk) >= 13312ull) chunk *= 2;
  // (one query of 8.9 M votes, fewer tiles than wave slots: 1,024 votes a tile 0.379 ms, 2,048: 0.354, 4,096: 0.349 -- smaller tiles
  // do not shorten the pass, every tile pays its start)
  if (chunk_env) chunk = chunk_env;
  pl.tile = chunk;
  static const uint32_t flush_env = [] { const char* e = getenv("SHZ_VW_FLUSH"); const int v = e ? atoi(e) : 0; return v >= 16 && v <= 256 ? (uint32_t)v : 0u; }();
  // votes a batch holds before it may end at a group border: 64 where tiles are small (one query: more, smaller batches let
  // the bar settle sooner: 0.281 ms against 0.300 with 128), 128 in the large passes of a batch of queries (fewer batch ends
  // to pay for: 0.1365 -> 0.1339 ms a query at 1M songs; 192: the same)
  pl.flush = flush_env ? flush_env : (chunk >= 8192 ? 2 * VW_FLUSH : VW_FLUSH);
  pl.qv[0] = pl.tb[0] = sp.qv[0] = sp.bq[0] = 0;
  for (uint32_t i = 0; i < nqp; ++i) {
    const uint64_t c = counts[i];
    pl.qv[i + 1] = sp.qv[i + 1] = pl.qv[i] + (uint32_t)c;
    pl.tb[i + 1] = pl.tb[i] + (uint32_t)((c + chunk - 1) / chunk);
    sp.bq[i + 1] = sp.bq[i] + (uint32_t)((c + 4095) / 4096);   // (the sort fills in its own block size)
  }
  for (uint32_t i = nqp; i < VT_MAXQ; ++i) {
    pl.qv[i + 1] = sp.qv[i + 1] = pl.qv[nqp];
    pl.tb[i + 1] = pl.tb[nqp];
    sp.bq[i + 1] = sp.bq[nqp];
  }
}

// the pass itself: the pp votes in k32 (4 bytes each, no query bits, query by query as the plan says; k32_alt = the second
// buffer of the sort) -> two radix passes (the first one already counted: hist0) -> tiles -> per-query top-n in
// rs / rdl / ra / rd / rn (rows of the pass's first query).  *d_vt_err is set when a tile gave up (the caller repeats the
// votes through the full sort).
static int32_t vt_run_pass(shz_ctx* ctx, uint32_t* k32, uint32_t* k32_alt, uint64_t pp, const vt_plan& pl, const shz_seg_plan& sp,
                           const m_bits& mbp, uint32_t topn, bool hist0, uint32_t max_sid, uint32_t* d_vt_err, uint32_t* rs,
                           int32_t* rdl, uint32_t* ra, uint32_t* rd, uint32_t* rn) {
  const uint32_t nqp = pl.nq;
  const int Bt = mbp.sb + mbp.dbits + 1;
  const uint32_t vt_probe_limit_1 = (ctx->debug & SHZ_DEBUG_VT_PROBE1) ? 1u : 0u;   // debug: a probe gives up after one round
  const uint32_t nt = pl.tb[nqp], hcap = (ctx->debug & SHZ_DEBUG_VT_TINY_HEAVY) ? 1u : nt * (VW_HEAVY_PER_TILE * ((pl.tile + VW_CHUNK - 1) / VW_CHUNK));
  int sel = 0;
  SHZ_TRY(shz_sort_u32_seg(ctx, k32, k32_alt, pp, pl.g_lo, Bt, sp, &sel, hist0));
  const uint32_t* ks = sel ? k32_alt : k32;
  // tile starts | counter of handed-over ranges | the ranges | their queries; candidates of tiles, then of ranges
  void *ts, *cp, *cd, *cdd;
  const uint64_t ncand = ((uint64_t)nt + hcap) * topn;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT0, ((uint64_t)nt + 4 + (uint64_t)hcap * 3) * 4, &ts));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT1, ncand * 8, &cp));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT2, ncand * 4, &cd));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT3, ncand * 4, &cdd));
  uint32_t* tile_start = (uint32_t*)ts;
  uint32_t* n_heavy = tile_start + nt + 1;
  uint2* heavy = (uint2*)(tile_start + ((nt + 2 + 1) & ~1u));   // 8-byte aligned
  uint32_t* heavy_q = (uint32_t*)(heavy + hcap);
  static const int vt_fast = [] { const char* e = getenv("SHZ_VT_FAST"); return e ? atoi(e) : 1; }();   // 0: the fold of round 3 (A/B)
  static const bool no_qr = [] { const char* e = getenv("SHZ_VT_NO_REJECT"); return e && atoi(e) != 0; }();
  unsigned long long* d_qbar = nullptr;   // the bar of every query of the pass (its tiles' n-th candidates, atomicMax); zeroed by vt_bounds_kernel
  if (vt_fast && !no_qr) {
    void* p;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT6, 8ull * (VT_MAXQ + 1), &p));
    d_qbar = (unsigned long long*)p;
  }
  static const bool keep_bar = [] { const char* e = getenv("SHZ_VT_KEEPBAR"); return e && atoi(e) != 0; }();   // EXPERIMENT
  hipLaunchKernelGGL(vt_bounds_kernel, dim3(nblk(std::max<uint64_t>(((uint64_t)nt + 1) * 64, (uint64_t)VT_MAXQ + 1))), dim3(256), 0, ctx->stream, ks, pl, tile_start,
                     n_heavy, keep_bar ? nullptr : d_qbar);
  // songs a batch is expected to hold: the 2^slb ids of a group + the ids that fill 64 votes
  const int slb_ = pl.g_lo - 1 - mbp.dbits;
  const double per_song = std::max(1.0, (double)pp / nqp / std::max<uint32_t>(max_sid, 1u));
  const bool few_songs = (double)(1u << slb_) + (double)pl.flush / per_song <= 40.0;
  const uint32_t plim = vt_probe_limit_1 ? vt_probe_limit_1 : (uint32_t)VW_S1;
  static const bool vt_stats = [] { const char* e = getenv("SHZ_VT_STATS"); return e && atoi(e) != 0; }();
  unsigned long long* d_stats = nullptr;
  if (vt_stats) {
    void* p;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC3, 64, &p));
    d_stats = (unsigned long long*)p;
    SHZ_HIP(ctx, hipMemsetAsync(d_stats, 0, 64, ctx->stream));
  }
  if (vt_fast && !no_qr) {
    if (few_songs)
      hipLaunchKernelGGL(vt_stream2_kernel<7>, dim3(nt), dim3(64), 0, ctx->stream, ks, (const uint32_t*)tile_start, pl, topn,
                         (uint64_t*)cp, (uint32_t*)cd, (uint32_t*)cdd, n_heavy, heavy, heavy_q, hcap, plim, d_qbar, d_vt_err, d_stats);
    else
      hipLaunchKernelGGL(vt_stream2_kernel<8>, dim3(nt), dim3(64), 0, ctx->stream, ks, (const uint32_t*)tile_start, pl, topn,
                         (uint64_t*)cp, (uint32_t*)cd, (uint32_t*)cdd, n_heavy, heavy, heavy_q, hcap, plim, d_qbar, d_vt_err, d_stats);
  } else if (no_qr)
    hipLaunchKernelGGL((vt_stream_kernel<7, false>), dim3(nt), dim3(64), 0, ctx->stream, ks, (const uint32_t*)tile_start, pl, topn,
                       (uint64_t*)cp, (uint32_t*)cd, (uint32_t*)cdd, n_heavy, heavy, heavy_q, hcap, plim, d_vt_err);
  else if (few_songs)
    hipLaunchKernelGGL(vt_stream_kernel<7>, dim3(nt), dim3(64), 0, ctx->stream, ks, (const uint32_t*)tile_start, pl, topn,
                       (uint64_t*)cp, (uint32_t*)cd, (uint32_t*)cdd, n_heavy, heavy, heavy_q, hcap, plim, d_vt_err);
  else
    hipLaunchKernelGGL(vt_stream_kernel<8>, dim3(nt), dim3(64), 0, ctx->stream, ks, (const uint32_t*)tile_start, pl, topn,
                       (uint64_t*)cp, (uint32_t*)cd, (uint32_t*)cdd, n_heavy, heavy, heavy_q, hcap, plim, d_vt_err);
  if (d_stats) {
    unsigned long long h[8];
    SHZ_HIP(ctx, hipMemcpyAsync(h, d_stats, 64, hipMemcpyDeviceToHost, ctx->stream));
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    fprintf(stderr, "[vt_stats] tiles %u few_songs %d batches %llu count1 %llu beat %llu votes_redone %llu votes %llu\n",
            nt, (int)few_songs, h[0], h[2], h[3], h[6], h[7]);
  }
  hipLaunchKernelGGL(vt_fold_kernel, dim3(std::min<uint32_t>(hcap, 64u)),
                     dim3(VT_THREADS), 0, ctx->stream, ks, (const uint2*)heavy, (const uint32_t*)n_heavy, hcap, pl, topn,
                     (uint64_t*)cp + (uint64_t)nt * topn, (uint32_t*)cd + (uint64_t)nt * topn,
                     (uint32_t*)cdd + (uint64_t)nt * topn, vt_probe_limit_1 ? vt_probe_limit_1 : (uint32_t)VT_SLOTS, d_vt_err);
  hipLaunchKernelGGL(vt_rank_kernel, dim3(nqp), dim3(VR_THREADS), 0, ctx->stream, pl, topn, mbp, (const uint64_t*)cp,
                     (const uint32_t*)cd, (const uint32_t*)cdd, (const uint32_t*)n_heavy, (const uint32_t*)heavy_q, hcap,
                     rs, rdl, ra, rd, rn);
  SHZ_HIP(ctx, hipGetLastError());
  return SHZ_OK;
}

static int32_t match_core(shz_ctx* ctx, shz_table* t, const uint32_t* key32, const uint32_t* q_off,
                          const uint64_t* query_off, uint32_t n_queries, uint32_t topn, uint32_t flags,
                          uint32_t* out_sid, int32_t* out_delta, uint32_t* out_aligned, uint32_t* out_dedup,
                          uint32_t* out_nres, uint32_t* out_nhash, uint64_t* out_npairs, pair_sink* vs_out) {
  if (!ctx || !t) return SHZ_E_INVALID;
  if (t->ctx != ctx) SHZ_FAIL(ctx, SHZ_E_INVALID, "table belongs to another ctx");
  if (pending_rows(t) || (!t->bucket && t->done.empty())) SHZ_FAIL(ctx, SHZ_E_STATE, "table not finalized");
  if (n_queries == 0) return SHZ_OK;
  if (!query_off) SHZ_FAIL(ctx, SHZ_E_INVALID, "match: query_off is NULL");
  if (!vs_out && (!out_sid || !out_delta || !out_aligned || !out_dedup || !out_nres))
    SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_match_batch: NULL buffer");
  if (!vs_out && (topn < 1 || topn > 64)) SHZ_FAIL(ctx, SHZ_E_INVALID, "topn must be in [1,64]");
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  ctx->st_rows = ctx->st_pairs = ctx->st_keys = 0;
  const uint64_t P_BUDGET = 1ull << 28;     // votes of one vote pass (its buffers: 2 x 4 or 8 bytes per vote)
  // votes of one sub-batch of queries (one head: compose, sort, probe, one round trip): up to 2^30 where the device has
  // the memory for it (a sub-batch that ends up on the 8-byte path needs ~32 bytes per vote of workspace)
  uint64_t SUB_BUDGET = P_BUDGET;
  if (n_queries > 32) {
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess)
      SUB_BUDGET = std::min<uint64_t>(1ull << 30, std::max<uint64_t>(P_BUDGET, (uint64_t)mem_free / 64));
  }
  // segment descriptors for the kernels (an empty table probes one empty segment)
  std::vector<shz_seg_dev> hsegs;
  for (const shz_seg& g : all_segs(t)) hsegs.push_back(seg_dev_of(g));
  if (hsegs.empty()) hsegs.push_back(shz_seg_dev{nullptr, nullptr, nullptr, t->bucket, 0u, 0u, 0ull});
  const int nseg = (int)hsegs.size();
  m_bits mb;
  mb.sb = bits_for(t->max_sid);
  mb.dbits = 0;
  mb.bias = 0;
  // the query set on the device once.  Host input of moderate size travels as ONE copy from pinned memory:
  // segment descriptors | query offsets | keys | offsets (every separate small copy costs ~20 us of latency)
  const uint64_t h0 = query_off[0], h1 = query_off[n_queries];
  const uint32_t *d_key = key32, *d_qo = q_off;
  void* d_segs;
  const uint64_t* d_qoff_all = nullptr;   // query_off[0 .. n_queries] on the device, when it went with the packed upload
  mctl* d_ctl_packed = nullptr;           // ... and a zeroed control block for the first sub-batch
  const uint64_t seg_bytes = (sizeof(shz_seg_dev) * (uint64_t)nseg + 255) & ~255ull;
  const uint64_t qoff_bytes = (((uint64_t)n_queries + 1) * 8 + 255) & ~255ull;
  const uint64_t pk_bytes = seg_bytes + qoff_bytes + ((h1 * 4 + 255) & ~255ull) * 2 + 256;   // + a zeroed mctl at the end
  if (!(flags & SHZ_IN_DEVICE) && pk_bytes <= (4ull << 20)) {
    void *hm, *dm;
    SHZ_TRY(shz_mailbox(ctx, pk_bytes, &hm));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_KEY, pk_bytes, &dm));
    char* hp = (char*)hm;
    memcpy(hp, hsegs.data(), sizeof(shz_seg_dev) * nseg);
    memcpy(hp + seg_bytes, query_off, ((uint64_t)n_queries + 1) * 8);
    const uint64_t ko = seg_bytes + qoff_bytes, oo = ko + ((h1 * 4 + 255) & ~255ull);
    if (h1) {
      memcpy(hp + ko, key32, h1 * 4);
      memcpy(hp + oo, q_off, h1 * 4);
    }
    memset(hp + pk_bytes - 256, 0, 256);
    d_ctl_packed = (mctl*)((char*)dm + pk_bytes - 256);
    SHZ_HIP(ctx, hipMemcpyAsync(dm, hm, pk_bytes, hipMemcpyHostToDevice, ctx->stream));
    d_segs = dm;
    d_qoff_all = (const uint64_t*)((char*)dm + seg_bytes);
    d_key = (const uint32_t*)((char*)dm + ko);
    d_qo = (const uint32_t*)((char*)dm + oo);
  } else {
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_META, sizeof(shz_seg_dev) * SHZ_MAX_SEGS, &d_segs));
    SHZ_HIP(ctx, shz_memcpy(ctx, d_segs, hsegs.data(), sizeof(shz_seg_dev) * nseg, hipMemcpyHostToDevice));
    if (!(flags & SHZ_IN_DEVICE) && h1 > 0) {
      void *a, *b;
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_KEY, h1 * 4, &a));
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_T1, h1 * 4, &b));
      SHZ_HIP(ctx, shz_memcpy(ctx, a, key32, h1 * 4, hipMemcpyHostToDevice));
      SHZ_HIP(ctx, shz_memcpy(ctx, b, q_off, h1 * 4, hipMemcpyHostToDevice));
      d_key = (const uint32_t*)a;
      d_qo = (const uint32_t*)b;
    }
  }
  (void)h0;
  uint32_t q0 = 0;
  uint32_t step = std::min<uint32_t>(n_queries, MAX_Q_SUB);
  bool redo_full_sort = false;   // the vote tiles of this sub-batch flagged an overflow: once more, through the full sort
  const uint32_t vt_probe_limit_1 = (ctx->debug & SHZ_DEBUG_VT_PROBE1) ? 1u : 0u;   // debug: a probe gives up after one round
  while (q0 < n_queries) {
    const uint32_t flags_sub = flags | (redo_full_sort ? SHZ_MATCH_FULL_SORT : 0u);
    uint32_t nq = std::min<uint32_t>(step, n_queries - q0);
    // the vote budget: sized from what a hash yielded in the last sub-batch, so that the head (compose, sort, probe: a
    // round trip) is not run on 200 queries, then 100, then 50 to find that 25 fit
    if (ctx->m_votes_per_hash > 0.0 && nq > 1) {
      const double hashes = (double)SUB_BUDGET / (1.05 * ctx->m_votes_per_hash);
      uint32_t lo = 1, hi = nq;   // largest count whose hashes stay under the estimate
      while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if ((double)(query_off[q0 + mid] - query_off[q0]) <= hashes) lo = mid; else hi = mid - 1;
      }
      nq = lo;
    }
    // shrink so the element count stays sortable
    while (nq > 1 && query_off[q0 + nq] - query_off[q0] >= (1ull << 31)) nq /= 2;
    const uint64_t m = query_off[q0 + nq] - query_off[q0];
    if (m >= (1ull << 31)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "query %u has too many hashes", q0);
    mb.qb = bits_for(nq - 1);
    static const bool trace = [] { const char* e = getenv("SHZ_TRACE_MATCH"); return e && atoi(e) != 0; }();
    const double tr0 = trace ? now_s() : 0.0;
    double tr1 = 0, tr2 = 0, tr3 = 0;
    void *d_qoff, *c0, *c1, *fl, *ps, *ctl_p;
    if (d_qoff_all) {
      d_qoff = (void*)(d_qoff_all + q0);
    } else {
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, (uint64_t)(nq + 1) * 8, &d_qoff));
      SHZ_HIP(ctx, shz_memcpy(ctx, d_qoff, query_off + q0, (uint64_t)(nq + 1) * 8, hipMemcpyHostToDevice));
    }
    if (d_ctl_packed && q0 == 0) {
      ctl_p = d_ctl_packed;               // arrived zeroed with the upload
      d_ctl_packed = nullptr;             // (a retry with fewer queries, or a later sub-batch, zeroes its own)
    } else {
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC0, 256, &ctl_p));
      SHZ_HIP(ctx, hipMemsetAsync(ctl_p, 0, 256, ctx->stream));
    }
    mctl* d_ctl = (mctl*)ctl_p;
    uint64_t* tot = (uint64_t*)ctl_p;      // tot[0..4] = mu, ng, rows, P, G
    uint32_t* err = d_ctl->err;
    if (m == 0) {
      for (uint32_t q = 0; q < nq; ++q) {
        if (out_nres) out_nres[q0 + q] = 0;
        if (out_nhash) out_nhash[q0 + q] = 0;
        if (out_npairs) out_npairs[q0 + q] = 0;
      }
      q0 += nq;
      continue;
    }
    const uint64_t nx_bound = m * (uint64_t)nseg + 1;   // sub-groups (group x segment) + sentinel, from the bound groups <= m
    if (nx_bound >= (1ull << 32)) {
      if (nq > 1) { step = nq / 2; continue; }
      SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "query %u: %llu hashes x %d segments", q0, (unsigned long long)m, nseg);
    }
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_A, m * 8, &c0));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_B, m * 8, &c1));
    const uint32_t f_nsh = vs_out ? vs_out->nshards : 1u, f_sh = vs_out ? vs_out->shard : 0u;
    void *gs, *glo, *gpairs, *po;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M3, (m + 1) * 4, &gs));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M4, nx_bound * 4, &glo));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M6, nx_bound * 8, &gpairs));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M7, nx_bound * 8, &po));
    uint64_t* E;
    static const bool no_small_head = [] { const char* e = getenv("SHZ_MATCH_NO_SMALL_HEAD"); return e && atoi(e) != 0; }();
    if (f_nsh == 1 && m <= MH_MAX && !no_small_head) {
      E = (uint64_t*)c1;
      hipLaunchKernelGGL(m_head_small_kernel, dim3(1), dim3(MH_THREADS), 0, ctx->stream, d_key, d_qo, (const uint64_t*)d_qoff,
                         nq, (uint32_t)m, QIDX_SHIFT + mb.qb, E, (uint32_t*)gs, d_ctl);
      SHZ_HIP(ctx, hipGetLastError());
    } else {
    hipLaunchKernelGGL(m_compose_kernel, dim3(nblk(m)), dim3(256), 0, ctx->stream, d_key, d_qo, (const uint64_t*)d_qoff, nq,
                       m, f_nsh, f_sh, (uint64_t*)c0, err);
    SHZ_HIP(ctx, hipGetLastError());
    int sel = 0;
    SHZ_TRY(shz_sort_u64(ctx, (uint64_t*)c0, (uint64_t*)c1, nullptr, nullptr, 0, m, 0,
                         QIDX_SHIFT + (f_nsh > 1 ? bits_for(nq) : mb.qb), &sel));
    uint64_t* cs = sel ? (uint64_t*)c1 : (uint64_t*)c0;   // sorted
    E = sel ? (uint64_t*)c0 : (uint64_t*)c1;              // unique elements go to the other buffer
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, m * 4, &fl));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M2, m * 4, &ps));
    // unique (query, key, off): mu of them; groups = distinct (query, key): ng of them; probe; pairs -- queued without a
    // host round trip in between: every launch is sized by the bound m, the counts are read on the device (mctl)
    hipLaunchKernelGGL(m_head_flag_kernel, dim3(nblk(m)), dim3(256), 0, ctx->stream, (const uint64_t*)cs, m, 0, (uint32_t*)fl);
    SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)fl, (uint32_t*)ps, m, tot));
    hipLaunchKernelGGL(m_compact_vals_kernel, dim3(nblk(m)), dim3(256), 0, ctx->stream, (const uint64_t*)cs,
                       (const uint32_t*)fl, (const uint32_t*)ps, m, E);
    hipLaunchKernelGGL(m_fix_mu_kernel, dim3(1), dim3(1), 0, ctx->stream, d_ctl, (unsigned long long)m);
    hipLaunchKernelGGL(m_head_flag_dn_kernel, dim3(nblk(m)), dim3(256), 0, ctx->stream, (const uint64_t*)E, &d_ctl->mu, m,
                       QKEY_SHIFT, (uint32_t*)fl);
    SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)fl, (uint32_t*)ps, m, tot + 1));
    hipLaunchKernelGGL(m_compact_idx_kernel, dim3(nblk(m)), dim3(256), 0, ctx->stream, (const uint32_t*)fl,
                       (const uint32_t*)ps, &d_ctl->mu, m, &d_ctl->ng, (uint32_t*)gs);
    }
    // results and per-query counters in ONE device block: zeroed by the probe, one copy after.
    // layout: npairs[nq] u64 | sid, delta, aligned, dedup [nq * topn] u32 each | nres[nq] | nhash[nq]
    const uint64_t nres = (uint64_t)nq * (vs_out ? 0 : topn);
    const uint64_t rb_bytes = (uint64_t)nq * 8 + nres * 16 + (uint64_t)nq * 8 + 8;   // + the vote tiles' flag word (and a pad)
    void* rb;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_F, rb_bytes + 64, &rb));
    hipLaunchKernelGGL(m_probe_kernel, dim3(nblk(nx_bound)), dim3(256), 0, ctx->stream, (const uint64_t*)E,
                       (const uint32_t*)gs, &d_ctl->ng, nx_bound, (const shz_seg_dev*)d_segs, (uint32_t)nseg, (uint32_t*)glo,
                       (uint64_t*)gpairs, (unsigned long long*)ctl_p + 16, (uint32_t*)rb, (uint32_t)((rb_bytes + 3) / 4));
    SHZ_HIP(ctx, hipGetLastError());
    SHZ_TRY(shz_scan_u64(ctx, (const uint64_t*)gpairs, (uint64_t*)po, nx_bound, tot + 3));
    uint64_t* d_np = (uint64_t*)rb;
    uint32_t* r_sid = (uint32_t*)(d_np + nq);
    uint32_t *r_delta = r_sid + nres, *r_al = r_delta + nres, *r_dd = r_al + nres, *r_n = r_dd + nres, *d_nh = r_n + nq;
    uint32_t* d_vt_err = d_nh + nq;   // set by a vote-tile kernel whose LDS table or range list overflowed: the sub-batch is voted again by the full sort
    // ONE small query from host memory: its votes are queued now, before their number is known (m_spec_plan_kernel),
    // through the one-workgroup path below and with that path's conditions; the layout needs the largest query offset,
    // which the host has
    static const int tiles_env = [] { const char* e = getenv("SHZ_VOTE_TILES"); return e ? atoi(e) : -1; }();   // 0 never
    static const int force32 = [] { const char* e = getenv("SHZ_VOTE32"); return e ? atoi(e) : -1; }();   // 0 never, 1 whenever it fits
    static const bool no_spec = [] { const char* e = getenv("SHZ_MATCH_NO_SPEC"); return e && atoi(e) != 0; }();
    bool spec = false;
    uint32_t spec_bias = 0;
    if (!vs_out && nq == 1 && f_nsh == 1 && m <= MH_MAX && !no_small_head && !no_spec && !(flags & SHZ_IN_DEVICE) &&
        !(flags_sub & SHZ_MATCH_FULL_SORT) && tiles_env != 0 && force32 != 1 && topn <= VT_MAXTOPN &&
        (double)m * t->votes_per_hash <= 2.0 * VT_ONE_WG_MAX) {   // (a table whose last query had far more votes: not worth queueing)
      const uint64_t a0 = query_off[q0], a1 = query_off[q0 + 1];
      bool wide = false;
      for (uint64_t i = a0; i < a1; ++i) {
        spec_bias = std::max(spec_bias, q_off[i]);
        wide |= q_off[i] >= (1u << QOFF_BITS);
      }
      m_bits ms = mb;
      ms.qb = 0;
      ms.bias = spec_bias;
      ms.dbits = bits_for((uint64_t)t->max_off + spec_bias);
      if (!wide && 31 - ms.sb - ms.dbits >= 0 && ms.dbits <= VT_MAX_DBITS + VT_MAX_DSPLIT) {
        spec = true;
        ++ctx->st_spec_queued;
        const uint64_t cap = VT_ONE_WG_MAX;
        const uint32_t cap_tiles = (uint32_t)((cap + M_EXP_TILE - 1) / M_EXP_TILE);
        void *tx, *k32, *ts, *cp, *cd, *cdd;
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_HCNT, ((uint64_t)cap_tiles + 1) * 4, &tx));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_C, cap * 8 + 16, &k32));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT0, 64, &ts));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT1, (uint64_t)topn * 8, &cp));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT2, (uint64_t)topn * 4, &cd));
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT3, (uint64_t)topn * 4, &cdd));
        uint32_t* n_heavy = (uint32_t*)ts;
        uint2* heavy = (uint2*)(n_heavy + 2);
        uint32_t* heavy_q = n_heavy + 4;
        vt_plan pl;
        pl.nq = 1;
        pl.dbits = ms.dbits;
        pl.sb = ms.sb;
        pl.g_lo = ms.sb + ms.dbits + 1;   // no bit is ordered: a sweep may split by any song-id bit
        pl.tile = VW_CHUNK;
        pl.flush = VW_FLUSH;
        for (uint32_t i = 0; i <= VT_MAXQ; ++i) { pl.qv[i] = 0u; pl.tb[i] = 0; }   // no tiles, one range (its end: heavy[0])
        hipLaunchKernelGGL(m_spec_plan_kernel, dim3(1), dim3(64), 0, ctx->stream, (const uint64_t*)po, (const mctl*)d_ctl,
                           (uint32_t)nseg, cap, (uint32_t*)tx, n_heavy, heavy, heavy_q, (uint32_t*)rb, (uint32_t)(rb_bytes / 4));
        hipLaunchKernelGGL(m_expand_spec_kernel, dim3(cap_tiles), dim3(256), 0, ctx->stream, (const uint64_t*)E,
                           (const uint32_t*)gs, (const uint32_t*)tx, (const uint64_t*)po, (const uint32_t*)glo,
                           (const shz_seg_dev*)d_segs, (uint32_t)nseg, (const mctl*)d_ctl, cap, ms, (uint32_t*)k32);
        hipLaunchKernelGGL(vt_fold_kernel, dim3(1), dim3(VT_THREADS), 0, ctx->stream, (const uint32_t*)k32, (const uint2*)heavy,
                           (const uint32_t*)n_heavy, 1u, pl, topn, (uint64_t*)cp, (uint32_t*)cd, (uint32_t*)cdd,
                           vt_probe_limit_1 ? vt_probe_limit_1 : (uint32_t)VT_SLOTS, d_vt_err);
        hipLaunchKernelGGL(vt_rank_kernel, dim3(1), dim3(VR_THREADS), 0, ctx->stream, pl, topn, ms, (const uint64_t*)cp,
                           (const uint32_t*)cd, (const uint32_t*)cdd, (const uint32_t*)n_heavy, (const uint32_t*)heavy_q, 1u,
                           r_sid, (int32_t*)r_delta, r_al, r_dd, r_n);
        SHZ_HIP(ctx, hipGetLastError());
      }
    }
    if (nq > 1) {   // one query: its counts are the totals
      hipLaunchKernelGGL(m_query_stats_kernel, dim3(nblk(nq)), dim3(256), 0, ctx->stream, (const uint64_t*)E, (const mctl*)d_ctl,
                         (const uint32_t*)gs, (const uint64_t*)po, (uint32_t)nseg, nq, d_nh, d_np);
      SHZ_HIP(ctx, hipGetLastError());
    }
    // the one read-back before the votes: their number sizes the vote buffers and the sort, their number per query
    // plans the vote passes (behind a queued small query: its results come along)
    void* mailp;
    SHZ_TRY(shz_mailbox(ctx, 256 + (uint64_t)nq * 8 + (spec ? rb_bytes : 0), &mailp));
    SHZ_HIP(ctx, hipMemcpyAsync(mailp, d_ctl, 256, hipMemcpyDeviceToHost, ctx->stream));
    if (nq > 1) SHZ_HIP(ctx, hipMemcpyAsync((char*)mailp + 256, d_np, (uint64_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (spec) SHZ_HIP(ctx, hipMemcpyAsync((char*)mailp + 256 + 8, rb, rb_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (trace) tr1 = now_s();
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (trace) tr2 = now_s();
    const mctl h = *(const mctl*)mailp;
    if (h.err[0]) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "query offsets must be < 2^%d frames", QOFF_BITS);
    const uint64_t mu = h.mu;
    if (mu == 0) {          // nothing of this sub-batch belongs to this shard
      for (uint32_t q = 0; q < nq; ++q) {
        if (out_nhash) out_nhash[q0 + q] = 0;
        if (out_npairs) out_npairs[q0 + q] = 0;
      }
      q0 += nq;
      continue;
    }
    mb.bias = h.err[1];
    mb.dbits = bits_for((uint64_t)t->max_off + mb.bias);
    if (vs_out) {  // the caller's layout must hold this table's ids and offsets and these queries' offsets
      const m_bits& L = vs_out->lay;
      if (h.err[1] > L.bias || bits_for(t->max_sid) > L.sb || bits_for((uint64_t)t->max_off + L.bias) > L.dbits)
        SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_match_pairs: layout (sid_bits %d, delta_bits %d, bias %u) too small for this table / these queries", L.sb, L.dbits, L.bias);
    } else if (mb.qb + mb.sb + mb.dbits + 1 > 64) {
      if (nq > 1) { step = nq / 2; continue; }
      SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "song id / offset range too wide for the packed vote key");
    }
    const uint32_t ng = (uint32_t)h.ng;
    const uint64_t nx = (uint64_t)ng * nseg;   // sub-groups: (query, key) group x segment
    const uint64_t P = h.P;
    uint64_t rows_total = 0;
    for (int i = 0; i < M_ROW_STRIPES; ++i) rows_total += ((const uint64_t*)mailp)[16 + i];
    if (mu) ctx->m_votes_per_hash = t->votes_per_hash = (double)P / (double)mu;
    if (P > SUB_BUDGET && nq > 1) {  // too many pairs for one sub-batch: retry with fewer queries
      step = std::max<uint32_t>(1, nq / 2);
      continue;
    }
    if (P >= (1ull << 32)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "query %u alone produces %llu matches (limit 2^32)", q0, (unsigned long long)P);
    std::vector<uint64_t> h_votes(nq, P);   // votes per query (read back with the counts)
    if (nq > 1) memcpy(h_votes.data(), (const char*)mailp + 256, (uint64_t)nq * 8);
    void* tile_x = nullptr;
    // the queued small query ran when its votes fit (else its kernels did nothing, and the passes below run as ever)
    const bool spec_done = spec && P <= VT_ONE_WG_MAX && mb.bias == spec_bias;
    if (spec_done) {
    } else if (P > 0 && vs_out) {
      // hand the votes over: expand straight into the caller's buffer, in the shared layout, with global query indices
      if (vs_out->count + P <= vs_out->cap) {
        const uint32_t ntiles = (uint32_t)((P + M_EXP_TILE - 1) / M_EXP_TILE);
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_HCNT, ((uint64_t)ntiles + 1) * 4, &tile_x));
        hipLaunchKernelGGL(m_tile_start_kernel, dim3(nblk((uint64_t)ntiles + 1)), dim3(256), 0, ctx->stream, (const uint64_t*)po,
                           (uint32_t)nx, (uint64_t)0, P, ntiles, (uint32_t*)tile_x);
        hipLaunchKernelGGL(m_expand_kernel<uint64_t>, dim3(ntiles), dim3(256), 0, ctx->stream, (const uint64_t*)E,
                           (const uint32_t*)gs, (const uint32_t*)tile_x, (const uint64_t*)po, (const uint32_t*)glo,
                           (const shz_seg_dev*)d_segs, (uint32_t)nseg, (uint64_t)0, P, vs_out->lay, (int64_t)q0,
                           vs_out->d_pairs + vs_out->count);
        SHZ_HIP(ctx, hipGetLastError());
      }
      vs_out->count += P;  // keeps counting past cap: the caller learns the size it needs
    } else if (P > 0) {
      // expand -> sort -> groups -> top-n, in vote passes over ranges of queries.  4-BYTE votes when a pass's query
      // index, the song id and the biased delta fit 31 bits (1M songs x 10 s queries: 1 + 20 + 10): half the bytes
      // through expand and the sort passes, whose last pass widens to the 64-bit layout the fold reads.  Worth it only
      // while a pass still has millions of votes; else one 8-byte pass over all queries.
      struct vpass { uint32_t qa, qb; uint64_t v_lo, v_hi; };
      std::vector<vpass> passes;
      const int qbits32 = 31 - mb.sb - mb.dbits;
      bool use32 = qbits32 >= 0 && P > MH_MAX && force32 != 0 && !(flags_sub & SHZ_MATCH_FULL_SORT);
      // vote tiles (vt_fold_kernel): two radix passes + an LDS fold per tile instead of four passes + the record chain
      const bool tiles = use32 && tiles_env != 0 && topn <= VT_MAXTOPN && mb.dbits <= VT_MAX_DBITS + VT_MAX_DSPLIT;
      if (use32) {
        const uint32_t q_per_pass = tiles ? (uint32_t)VT_MAXQ : (1u << std::min(qbits32, 30));   // tiles: no query bits in the vote
        uint64_t v = 0;
        for (uint32_t qa = 0; qa < nq;) {
          uint32_t qb = qa;
          uint64_t pv = 0;
          while (qb < nq && (qb - qa) < q_per_pass && (qb == qa || pv + h_votes[qb] <= P_BUDGET)) pv += h_votes[qb++];
          if (pv) passes.push_back(vpass{qa, qb, v, v + pv});
          v += pv;
          qa = qb;
        }
        if (force32 != 1 && P / std::max<size_t>(passes.size(), 1) < (1ull << 22)) use32 = false;   // small passes: launch-bound
      }
      if (!use32) { passes.clear(); passes.push_back(vpass{0, nq, 0, P}); }
      // ONE query with few votes (a 5-10 s query against thousands of songs): expand, then one workgroup folds the
      // unordered votes (vt_fold_kernel over the single range [0, P)) -- 5 launches instead of the 9 of sort + fold,
      // and the launches are what such a match costs
      const bool one_wg = !use32 && !(flags_sub & SHZ_MATCH_FULL_SORT) && nq == 1 && P <= VT_ONE_WG_MAX && qbits32 >= 0 && tiles_env != 0 && topn <= VT_MAXTOPN &&
                          mb.dbits <= VT_MAX_DBITS + VT_MAX_DSPLIT;
      uint64_t pmax = 0;
      for (const vpass& vp : passes) pmax = std::max(pmax, vp.v_hi - vp.v_lo);
      void *v0, *v1;   // E lives in one of SORT_A/B; the vote buffers use SORT_C/D
      const uint64_t pmax4 = (pmax + 3) & ~3ull;   // second 4-byte vote buffer of a pass: 16-byte aligned behind the first
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_C, pmax * 8 + 16, &v0));
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_D, pmax * 8, &v1));
      // vote tiles: the expand runs by the blocks of the segmented sort and counts the first pass's digits (no tile starts
      // here: m_chunk_start_kernel per pass)
      static const bool no_fuse = [] { const char* e = getenv("SHZ_NO_EXPAND_HIST"); return e && atoi(e) != 0; }();
      const int Bt_all = mb.sb + mb.dbits + 1, g_lo_all = std::max(1 + mb.dbits, Bt_all - VT_ORDERED_BITS);
      uint32_t fuse_dmask = 0;
      const bool fuse = tiles && use32 && !no_fuse && shz_seg_first_pass(g_lo_all, Bt_all, &fuse_dmask) == 8;
      // the sub-group of every expand tile's first pair, for all passes in one launch: pass table (votes, first entry)
      // up, one kernel
      std::vector<uint32_t> toff(passes.size() + 1, 0);
      std::vector<uint64_t> ptab(passes.size() * 2 + (passes.size() + 2) / 2);   // pv[2 n_pass] | toff[n_pass + 1] (u32)
      for (size_t i = 0; i < passes.size(); ++i) {
        ptab[2 * i] = passes[i].v_lo;
        ptab[2 * i + 1] = passes[i].v_hi;
        toff[i + 1] = toff[i] + (uint32_t)((passes[i].v_hi - passes[i].v_lo + M_EXP_TILE - 1) / M_EXP_TILE) + 1;
      }
      memcpy(ptab.data() + passes.size() * 2, toff.data(), toff.size() * 4);
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_HCNT, (uint64_t)toff.back() * 4, &tile_x));   // (M5 belongs to the fold)
      if (fuse) {
      } else if (passes.size() == 1) {   // no table to send up
        hipLaunchKernelGGL(m_tile_start_kernel, dim3(nblk(toff.back())), dim3(256), 0, ctx->stream, (const uint64_t*)po, (uint32_t)nx,
                           passes[0].v_lo, passes[0].v_hi, toff.back() - 1, (uint32_t*)tile_x);
      } else {
        void* d_ptab;
        SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT4, ptab.size() * 8, &d_ptab));
        SHZ_HIP(ctx, shz_memcpy(ctx, d_ptab, ptab.data(), ptab.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(m_tile_start_all_kernel, dim3(nblk(toff.back())), dim3(256), 0, ctx->stream, (const uint64_t*)po, (uint32_t)nx,
                           (const uint64_t*)d_ptab, (const uint32_t*)((const uint64_t*)d_ptab + passes.size() * 2),
                           (uint32_t)passes.size(), (uint32_t*)tile_x);
      }
      SHZ_HIP(ctx, hipGetLastError());
      void* const tile_x_all = tile_x;
      size_t pass_i = 0;
      for (const vpass& vp : passes) {
        const uint64_t pp = vp.v_hi - vp.v_lo;
        const uint32_t nqp = vp.qb - vp.qa;
        const uint32_t ntiles = (uint32_t)((pp + M_EXP_TILE - 1) / M_EXP_TILE);
        m_bits mbp = mb;
        if (use32) mbp.qb = (nqp > 1 && !tiles) ? bits_for(nqp - 1) : 0;
        tile_x = (uint32_t*)tile_x_all + toff[pass_i++];
        uint32_t *rs = r_sid + (uint64_t)vp.qa * topn, *ra = r_al + (uint64_t)vp.qa * topn, *rd = r_dd + (uint64_t)vp.qa * topn;
        int32_t* rdl = (int32_t*)r_delta + (uint64_t)vp.qa * topn;
        if (one_wg) {
          uint32_t* k32 = (uint32_t*)v0;
          hipLaunchKernelGGL(m_expand_kernel<uint32_t>, dim3(ntiles), dim3(256), 0, ctx->stream, (const uint64_t*)E,
                             (const uint32_t*)gs, (const uint32_t*)tile_x, (const uint64_t*)po, (const uint32_t*)glo,
                             (const shz_seg_dev*)d_segs, (uint32_t)nseg, vp.v_lo, vp.v_hi, mbp, (int64_t)0, k32);
          vt_plan pl;
          pl.nq = 1;
          pl.dbits = mbp.dbits;
          pl.sb = mbp.sb;
          pl.g_lo = mbp.sb + mbp.dbits + 1;   // no bit is ordered: a sweep may split by any song-id bit
          pl.tile = VW_CHUNK;
          pl.flush = VW_FLUSH;
          for (uint32_t i = 0; i <= VT_MAXQ; ++i) { pl.qv[i] = i ? (uint32_t)pp : 0u; pl.tb[i] = 0; }   // no tiles, one range
          void *ts, *cp, *cd, *cdd;
          SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT0, 64, &ts));
          SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT1, (uint64_t)topn * 8, &cp));
          SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT2, (uint64_t)topn * 4, &cd));
          SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT3, (uint64_t)topn * 4, &cdd));
          uint32_t* n_heavy = (uint32_t*)ts;
          uint2* heavy = (uint2*)(n_heavy + 2);
          uint32_t* heavy_q = n_heavy + 4;
          hipLaunchKernelGGL(vt_one_range_kernel, dim3(1), dim3(1), 0, ctx->stream, n_heavy, heavy, heavy_q, (uint32_t)pp);
          hipLaunchKernelGGL(vt_fold_kernel, dim3(1), dim3(VT_THREADS), 0, ctx->stream, (const uint32_t*)k32, (const uint2*)heavy,
                             (const uint32_t*)n_heavy, 1u, pl, topn, (uint64_t*)cp, (uint32_t*)cd, (uint32_t*)cdd,
                             vt_probe_limit_1 ? vt_probe_limit_1 : (uint32_t)VT_SLOTS, d_vt_err);
          hipLaunchKernelGGL(vt_rank_kernel, dim3(1), dim3(VR_THREADS), 0, ctx->stream, pl, topn, mbp, (const uint64_t*)cp,
                             (const uint32_t*)cd, (const uint32_t*)cdd, (const uint32_t*)n_heavy, (const uint32_t*)heavy_q, 1u,
                             rs, rdl, ra, rd, r_n + vp.qa);
          SHZ_HIP(ctx, hipGetLastError());
        } else if (use32) {
          uint32_t* k32 = (uint32_t*)v0;   // two 4-byte buffers in SORT_C, the widened result in SORT_D
          if (!fuse) {
            hipLaunchKernelGGL(m_expand_kernel<uint32_t>, dim3(ntiles), dim3(256), 0, ctx->stream, (const uint64_t*)E,
                               (const uint32_t*)gs, (const uint32_t*)tile_x, (const uint64_t*)po, (const uint32_t*)glo,
                               (const shz_seg_dev*)d_segs, (uint32_t)nseg, vp.v_lo, vp.v_hi, mbp,
                               tiles ? M_NO_QUERY_BITS : -(int64_t)vp.qa, k32);
            SHZ_HIP(ctx, hipGetLastError());
          }
          const int B = mbp.qb + mbp.sb + mbp.dbits + 1;
          if (tiles) {
            vt_plan pl;
            shz_seg_plan sp;
            std::vector<uint64_t> counts(nqp);
            for (uint32_t i = 0; i < nqp; ++i) counts[i] = nq > 1 ? h_votes[vp.qa + i] : pp;
            vt_make_plan(counts.data(), nqp, mbp, pl, sp);
            const int Bt = mbp.sb + mbp.dbits + 1;   // the votes of a tile pass carry no query bits
            (void)Bt;
            if (fuse) {
              const uint32_t stile = shz_seg_tile(pp), cpb = stile / M_EXP_TILE;
              shz_seg_blocks(&sp, stile);
              const uint32_t nsb = sp.bq[nqp];
              void *cxp, *hist;
              SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_VT5, (uint64_t)nsb * (cpb + 1) * 4, &cxp));
              SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_H, ((uint64_t)nsb << 8) * 4, &hist));   // (the sort asks for the same)
              hipLaunchKernelGGL(m_chunk_start_kernel, dim3(nblk((uint64_t)nsb * (cpb + 1))), dim3(256), 0, ctx->stream,
                                 (const uint64_t*)po, (uint32_t)nx, sp, stile, vp.v_lo, vp.v_hi, (uint32_t*)cxp);
              hipLaunchKernelGGL(m_expand_blocks_kernel, dim3(nsb), dim3(256), 0, ctx->stream, (const uint64_t*)E,
                                 (const uint32_t*)gs, (const uint32_t*)cxp, (const uint64_t*)po, (const uint32_t*)glo,
                                 (const shz_seg_dev*)d_segs, (uint32_t)nseg, vp.v_lo, sp, stile, mbp, pl.g_lo, fuse_dmask, k32,
                                 (uint32_t*)hist);
              SHZ_HIP(ctx, hipGetLastError());
            }
            SHZ_TRY(vt_run_pass(ctx, k32, k32 + pmax4, pp, pl, sp, mbp, topn, fuse, t->max_sid, d_vt_err, rs, rdl, ra, rd, r_n + vp.qa));
          } else {
            SHZ_TRY(shz_sort_u32_widen(ctx, k32, k32 + pmax4, (uint64_t*)v1, pp, 1, B, 0, nullptr));
            SHZ_TRY(vote_fold(ctx, (const uint64_t*)v1, pp, nqp, mbp, topn, tot + 4, rs, rdl, ra, rd, r_n + vp.qa));
          }
        } else {
          hipLaunchKernelGGL(m_expand_kernel<uint64_t>, dim3(ntiles), dim3(256), 0, ctx->stream, (const uint64_t*)E,
                             (const uint32_t*)gs, (const uint32_t*)tile_x, (const uint64_t*)po, (const uint32_t*)glo,
                             (const shz_seg_dev*)d_segs, (uint32_t)nseg, vp.v_lo, vp.v_hi, mbp, (int64_t)0, (uint64_t*)v0);
          SHZ_HIP(ctx, hipGetLastError());
          SHZ_TRY(vote_tail(ctx, (uint64_t*)v0, (uint64_t*)v1, pp, nqp, mbp, topn, tot + 4, rs, rdl, ra, rd, r_n + vp.qa));
        }
      }
    }
    {
      void* hb;
      if (spec_done) {
        ++ctx->st_spec_used;
        hb = (char*)mailp + 256 + 8;   // came with the first read-back
        if (trace) tr3 = now_s();
      } else {
        SHZ_TRY(shz_mailbox(ctx, rb_bytes, &hb));
        SHZ_HIP(ctx, hipMemcpyAsync(hb, rb, rb_bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (trace) tr3 = now_s();
        SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
      }
      if (trace)
        fprintf(stderr, "match trace: head enqueue %.1f us, wait %.1f us, tail enqueue %.1f us, wait %.1f us (P %llu)\n",
                (tr1 - tr0) * 1e6, (tr2 - tr1) * 1e6, (tr3 - tr2) * 1e6, (now_s() - tr3) * 1e6, (unsigned long long)P);
      const uint64_t* h_np = (const uint64_t*)hb;
      const uint32_t* h_sid = (const uint32_t*)(h_np + nq);
      const uint32_t *h_delta = h_sid + nres, *h_al = h_delta + nres, *h_dd = h_al + nres, *h_n = h_dd + nres, *h_nh = h_n + nq;
      if (h_nh[nq] != 0 && !redo_full_sort) {   // a vote tile overflowed (its results are void): the same queries again, full sort
        ++ctx->st_vt_redo;
        redo_full_sort = true;
        continue;
      }
      redo_full_sort = false;
      ctx->st_rows += rows_total;
      ctx->st_pairs += P;
      ctx->st_keys += ng;
      if (nq == 1) {
        if (out_npairs) out_npairs[q0] = P;
        if (out_nhash) out_nhash[q0] = (uint32_t)mu;
      } else {
        if (out_npairs) memcpy(out_npairs + q0, h_np, (uint64_t)nq * 8);
        if (out_nhash) memcpy(out_nhash + q0, h_nh, (uint64_t)nq * 4);
      }
      if (!vs_out) {
        const uint64_t o0 = (uint64_t)q0 * topn;
        memcpy(out_sid + o0, h_sid, nres * 4);
        memcpy(out_delta + o0, h_delta, nres * 4);
        memcpy(out_aligned + o0, h_al, nres * 4);
        memcpy(out_dedup + o0, h_dd, nres * 4);
        memcpy(out_nres + q0, h_n, (uint64_t)nq * 4);
      }
    }
    q0 += nq;
  }
  return SHZ_OK;
}

extern "C" int32_t shz_match_batch(shz_ctx* ctx, shz_table* t, const uint32_t* key32, const uint32_t* q_off,
                                   const uint64_t* query_off, uint32_t n_queries, uint32_t topn, uint32_t flags,
                                   uint32_t* out_sid, int32_t* out_delta, uint32_t* out_aligned, uint32_t* out_dedup,
                                   uint32_t* out_nres, uint32_t* out_nhash, uint64_t* out_npairs) {
  return match_core(ctx, t, key32, q_off, query_off, n_queries, topn, flags, out_sid, out_delta, out_aligned, out_dedup,
                    out_nres, out_nhash, out_npairs, nullptr);
}

extern "C" int32_t shz_table_maxima(shz_table* t, uint32_t* max_sid, uint32_t* max_off) {
  if (!t) return SHZ_E_INVALID;
  if (max_sid) *max_sid = t->max_sid;
  if (max_off) *max_off = t->max_off;
  return SHZ_OK;
}

extern "C" int32_t shz_match_pairs(shz_ctx* ctx, shz_table* t, const uint32_t* key32, const uint32_t* q_off,
                                   const uint64_t* query_off, uint32_t n_queries, uint32_t flags, uint32_t shard,
                                   uint32_t nshards, uint32_t sid_bits, uint32_t delta_bits, uint32_t bias,
                                   uint64_t* d_pairs, uint64_t cap, uint64_t* count, uint32_t* out_nhash,
                                   uint64_t* out_npairs) {
  if (!ctx || !count) return SHZ_E_INVALID;
  if (cap && !d_pairs) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_match_pairs: NULL buffer");
  if (nshards == 0 || shard >= nshards) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_match_pairs: shard %u of %u", shard, nshards);
  if (sid_bits < 1 || delta_bits < 1 || bits_for(n_queries ? n_queries - 1 : 0) + sid_bits + delta_bits + 1 > 64)
    SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "shz_match_pairs: %u queries x %u sid bits x %u delta bits do not fit 64 bits", n_queries, sid_bits, delta_bits);
  pair_sink sink{d_pairs, cap, 0, m_bits{(int)sid_bits, (int)delta_bits, 0, bias}, shard, nshards};
  SHZ_TRY(match_core(ctx, t, key32, q_off, query_off, n_queries, 1, flags, nullptr, nullptr, nullptr, nullptr, nullptr,
                     out_nhash, out_npairs, &sink));
  *count = sink.count;
  if (sink.count > cap) SHZ_FAIL(ctx, SHZ_E_CAPACITY, "shz_match_pairs: %llu votes, capacity %llu", (unsigned long long)sink.count, (unsigned long long)cap);
  return SHZ_OK;
}

extern "C" int32_t shz_set_debug(shz_ctx* ctx, uint32_t flags) {
  if (!ctx) return SHZ_E_INVALID;
  ctx->debug = flags;
  return SHZ_OK;
}
extern "C" int32_t shz_match_vt_redo(shz_ctx* ctx, uint64_t* count) {
  if (!ctx || !count) return SHZ_E_INVALID;
  *count = ctx->st_vt_redo;
  return SHZ_OK;
}

extern "C" int32_t shz_match_spec_stats(shz_ctx* ctx, uint64_t* queued, uint64_t* used) {
  if (!ctx || !queued || !used) return SHZ_E_INVALID;
  *queued = ctx->st_spec_queued;
  *used = ctx->st_spec_used;
  return SHZ_OK;
}

extern "C" int32_t shz_match_stats(shz_ctx* ctx, uint64_t* rows_scanned, uint64_t* pairs, uint64_t* distinct_keys) {
  if (!ctx) return SHZ_E_INVALID;
  if (rows_scanned) *rows_scanned = ctx->st_rows;
  if (pairs) *pairs = ctx->st_pairs;
  if (distinct_keys) *distinct_keys = ctx->st_keys;
  return SHZ_OK;
}

// ---------------------------------------------------------------------------------------- votes: gather + rank
extern "C" int32_t shz_pairs_allgather(shz_comm* c, uint64_t n_local, const uint64_t* d_pairs, uint64_t* d_all, uint64_t cap,
                                       uint64_t* n_total) {
  if (!c || !n_total) return SHZ_E_INVALID;
  int rank, nranks;
  shz_comm_info(c, &rank, &nranks);
  shz_ctx* ctx = shz_comm_ctx(c);
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  void* d_cnt;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC2, 8ull * (nranks + 1), &d_cnt));
  SHZ_HIP(ctx, shz_memcpy(ctx, d_cnt, &n_local, 8, hipMemcpyHostToDevice));
  SHZ_TRY(shz_comm_allgather_bytes(c, d_cnt, (uint64_t*)d_cnt + 1, 8));
  std::vector<uint64_t> cnt(nranks), bytes(nranks), displ(nranks);
  SHZ_HIP(ctx, shz_memcpy(ctx, cnt.data(), (uint64_t*)d_cnt + 1, 8ull * nranks, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t total = 0;
  for (int r = 0; r < nranks; ++r) { displ[r] = total * 8; bytes[r] = cnt[r] * 8; total += cnt[r]; }
  *n_total = total;
  if (total > cap) SHZ_FAIL(ctx, SHZ_E_CAPACITY, "shz_pairs_allgather: %llu votes, capacity %llu", (unsigned long long)total, (unsigned long long)cap);
  if (total == 0) return SHZ_OK;
  if (!d_all || (n_local && !d_pairs)) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_pairs_allgather: NULL buffer");
  SHZ_TRY(shz_comm_allgatherv_bytes(c, d_pairs, d_all, bytes.data(), displ.data()));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}

// ---- the gathered votes of a sharded match through the vote tiles
// first vote of every query in the pairs ordered by query: qstart[q] = first i with (pairs[i] >> shift) >= q, q = 0 .. nq
__global__ void pv_qstart_kernel(const uint64_t* __restrict__ pairs, uint64_t n, int shift, uint32_t nq,
                                 unsigned long long* __restrict__ qstart) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q > nq) return;
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if ((pairs[mid] >> shift) < q) lo = mid + 1; else hi = mid;
  }
  qstart[q] = lo;
}
// the 4-byte vote of the tiles: a pair without its query bits
__global__ void pv_narrow_kernel(const uint64_t* __restrict__ pairs, uint64_t n, uint32_t mask, uint32_t* __restrict__ k32) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    k32[i] = (uint32_t)pairs[i] & mask;
}

// Votes of shz_match_pairs (8 bytes, query index on top) -> results, by the path of the unsharded match: ONE stable sort
// by the query bits (one or two passes over 8 bytes instead of the five of the full key), the query bits dropped, then
// the vote-tile passes over 4-byte votes.  Returns false (nothing written) when the layout does not fit the tiles; *redo is
// set when a tile gave up and the votes have to go through the full sort after all.
static int32_t pairs_vote_tiles(shz_ctx* ctx, uint64_t* d_pairs, uint64_t* d_alt, uint64_t n, uint32_t nq, const m_bits& mb,
                                uint32_t topn, uint32_t* r_sid, int32_t* r_delta, uint32_t* r_al, uint32_t* r_dd, uint32_t* r_n,
                                bool* done, bool* redo) {
  *done = *redo = false;
  const int Bt = mb.sb + mb.dbits + 1;
  static const int tiles_env = [] { const char* e = getenv("SHZ_VOTE_TILES"); return e ? atoi(e) : -1; }();
  static const int force32 = [] { const char* e = getenv("SHZ_VOTE32"); return e ? atoi(e) : -1; }();   // 0 never, 1 whenever it fits
  if (tiles_env == 0 || force32 == 0 || Bt > 31 || topn > VT_MAXTOPN || mb.dbits > VT_MAX_DBITS + VT_MAX_DSPLIT) return SHZ_OK;
  if (force32 != 1 && n < (1ull << 22)) return SHZ_OK;   // few votes: the passes' launches cost more than the full sort's extra bytes
  // 1. by query (stable: the order inside a query does not matter to the tiles)
  int sel = 0;
  if (mb.qb > 0) SHZ_TRY(shz_sort_u64(ctx, d_pairs, d_alt, nullptr, nullptr, 0, n, Bt, Bt + mb.qb, &sel));
  const uint64_t* sorted = sel ? d_alt : d_pairs;
  // 2. votes per query
  void *qs, *flag;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, ((uint64_t)nq + 1) * 8, &qs));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, 64, &flag));
  SHZ_HIP(ctx, hipMemsetAsync(flag, 0, 64, ctx->stream));
  hipLaunchKernelGGL(pv_qstart_kernel, dim3(nblk((uint64_t)nq + 1)), dim3(256), 0, ctx->stream, sorted, n, Bt, nq,
                     (unsigned long long*)qs);
  SHZ_HIP(ctx, hipGetLastError());
  std::vector<uint64_t> h_qs((size_t)nq + 1);
  SHZ_HIP(ctx, shz_memcpy(ctx, h_qs.data(), qs, ((uint64_t)nq + 1) * 8, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // 3. 4-byte votes; the other half of the buffer is the sort's second one
  void* kb;
  const uint64_t n4 = (n + 3) & ~3ull;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_C, (n4 * 2) * 4 + 64, &kb));
  uint32_t* k32 = (uint32_t*)kb;
  hipLaunchKernelGGL(pv_narrow_kernel, dim3((unsigned)std::min<uint64_t>(nblk(n), 65536)), dim3(256), 0, ctx->stream, sorted, n,
                     (uint32_t)((1ull << Bt) - 1ull), k32);
  SHZ_HIP(ctx, hipGetLastError());
  // 4. passes of up to VT_MAXQ queries and 2^28 votes
  m_bits mbp = mb;
  mbp.qb = 0;
  const uint64_t P_PASS = 1ull << 28;
  for (uint32_t qa = 0; qa < nq;) {
    uint32_t qb = qa;
    while (qb < nq && qb - qa < (uint32_t)VT_MAXQ && (qb == qa || h_qs[qb + 1] - h_qs[qa] <= P_PASS)) ++qb;
    const uint64_t v_lo = h_qs[qa], pp = h_qs[qb] - v_lo;
    if (pp >= (1ull << 32)) return SHZ_OK;   // (one query with 2^32 votes: the full sort's error message)
    if (pp) {
      const uint32_t nqp = qb - qa;
      std::vector<uint64_t> counts(nqp);
      for (uint32_t i = 0; i < nqp; ++i) counts[i] = h_qs[qa + i + 1] - h_qs[qa + i];
      vt_plan pl;
      shz_seg_plan sp;
      vt_make_plan(counts.data(), nqp, mbp, pl, sp);
      // the pass sorts inside [v_lo, v_lo + pp) of k32 and of the second buffer; both offsets 16-byte aligned by address
      // is not required (the counting kernel aligns by address)
      SHZ_TRY(vt_run_pass(ctx, k32 + v_lo, k32 + n4 + v_lo, pp, pl, sp, mbp, topn, false, mb.sb >= 31 ? 0x7FFFFFFFu : (1u << mb.sb),
                          (uint32_t*)flag, r_sid + (uint64_t)qa * topn, r_delta + (uint64_t)qa * topn, r_al + (uint64_t)qa * topn,
                          r_dd + (uint64_t)qa * topn, r_n + qa));
    }
    qa = qb;
  }
  uint32_t h_flag = 0;
  SHZ_HIP(ctx, shz_memcpy(ctx, &h_flag, flag, 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h_flag) { ++ctx->st_vt_redo; *redo = true; return SHZ_OK; }
  *done = true;
  return SHZ_OK;
}

extern "C" int32_t shz_pairs_vote(shz_ctx* ctx, uint64_t* d_pairs, uint64_t n, uint32_t n_queries, uint32_t sid_bits,
                                  uint32_t delta_bits, uint32_t bias, uint32_t topn, uint32_t* out_sid, int32_t* out_delta,
                                  uint32_t* out_aligned, uint32_t* out_dedup, uint32_t* out_nres) {
  if (!ctx) return SHZ_E_INVALID;
  if (n_queries == 0) return SHZ_OK;
  if (!out_sid || !out_delta || !out_aligned || !out_dedup || !out_nres) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_pairs_vote: NULL output");
  if (topn < 1 || topn > 64) SHZ_FAIL(ctx, SHZ_E_INVALID, "topn must be in [1,64]");
  if (n >= (1ull << 32)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "shz_pairs_vote: %llu votes (limit 2^32); vote fewer queries per call", (unsigned long long)n);
  m_bits mb{(int)sid_bits, (int)delta_bits, bits_for(n_queries - 1), bias};
  if (sid_bits < 1 || delta_bits < 1 || mb.qb + mb.sb + mb.dbits + 1 > 64) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "shz_pairs_vote: key layout does not fit 64 bits");
  const uint64_t nres = (uint64_t)n_queries * topn;
  memset(out_sid, 0, nres * 4); memset(out_delta, 0, nres * 4); memset(out_aligned, 0, nres * 4);
  memset(out_dedup, 0, nres * 4); memset(out_nres, 0, (uint64_t)n_queries * 4);
  if (n == 0) return SHZ_OK;
  if (!d_pairs) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_pairs_vote: NULL votes");
  if ((uintptr_t)d_pairs & 15) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_pairs_vote: the vote buffer must be 16-byte aligned");
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  void *v1, *tot, *r_sid, *r_delta, *r_al, *r_dd, *r_n;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_D, n * 8, &v1));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC0, 64, &tot));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_F, nres * 4, &r_sid));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_T, nres * 4, &r_delta));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_CLIP, nres * 4, &r_al));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, nres * 4, &r_dd));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC2, (uint64_t)n_queries * 4, &r_n));
  SHZ_HIP(ctx, hipMemsetAsync(r_sid, 0, nres * 4, ctx->stream));
  SHZ_HIP(ctx, hipMemsetAsync(r_delta, 0, nres * 4, ctx->stream));
  SHZ_HIP(ctx, hipMemsetAsync(r_al, 0, nres * 4, ctx->stream));
  SHZ_HIP(ctx, hipMemsetAsync(r_dd, 0, nres * 4, ctx->stream));
  SHZ_HIP(ctx, hipMemsetAsync(r_n, 0, (uint64_t)n_queries * 4, ctx->stream));
  bool done = false, redo = false;
  SHZ_TRY(pairs_vote_tiles(ctx, d_pairs, (uint64_t*)v1, n, n_queries, mb, topn, (uint32_t*)r_sid, (int32_t*)r_delta, (uint32_t*)r_al,
                           (uint32_t*)r_dd, (uint32_t*)r_n, &done, &redo));
  if (redo) {   // a tile gave up: its rows of the result arrays are void
    SHZ_HIP(ctx, hipMemsetAsync(r_sid, 0, nres * 4, ctx->stream));
    SHZ_HIP(ctx, hipMemsetAsync(r_delta, 0, nres * 4, ctx->stream));
    SHZ_HIP(ctx, hipMemsetAsync(r_al, 0, nres * 4, ctx->stream));
    SHZ_HIP(ctx, hipMemsetAsync(r_dd, 0, nres * 4, ctx->stream));
    SHZ_HIP(ctx, hipMemsetAsync(r_n, 0, (uint64_t)n_queries * 4, ctx->stream));
  }
  // (the full sort orders any permutation of the votes: whatever the sort by query left in d_pairs / v1 is fine for it)
  if (!done)
    SHZ_TRY(vote_tail(ctx, d_pairs, (uint64_t*)v1, n, n_queries, mb, topn, (uint64_t*)tot, (uint32_t*)r_sid, (int32_t*)r_delta,
                      (uint32_t*)r_al, (uint32_t*)r_dd, (uint32_t*)r_n));
  SHZ_HIP(ctx, shz_memcpy(ctx, out_sid, r_sid, nres * 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, shz_memcpy(ctx, out_delta, r_delta, nres * 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, shz_memcpy(ctx, out_aligned, r_al, nres * 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, shz_memcpy(ctx, out_dedup, r_dd, nres * 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, shz_memcpy(ctx, out_nres, r_n, (uint64_t)n_queries * 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}

