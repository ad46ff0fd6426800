// Device-wide primitives used by the extraction and match paths: exclusive scans and a
// stable LSD radix sort.  Wave = 64 lanes; blocks of 256 threads (4 waves).
#include "shz_internal.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

struct in_u32 {
  const uint32_t* p;
  __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return p[i]; }
};
struct in_popc64 {
  const uint64_t* p;
  __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return (uint32_t)__popcll(p[i]); }
};
struct in_u64 {
  const uint64_t* p;
  __device__ __forceinline__ uint64_t operator()(uint64_t i) const { return p[i]; }
};

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// exclusive scan across the block of one value per thread; returns exclusive prefix, *total = block sum
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T* total, T* lds /* >= 5 entries */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T inc = wave_incl_scan(v, lane);
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  T woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    T s = lds[w];
    if (w < wave) woff += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return woff + inc - v;
}

template <typename T, typename In>
__global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(In in, T* __restrict__ sums, uint64_t n) {
  __shared__ T lds[8];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  T s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) s += (T)in(base + i);
  T tot;
  block_excl_scan(s, &tot, lds);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

template <typename T, typename In>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(In in, T* __restrict__ out,
                                                                   const T* __restrict__ block_off, uint64_t n,
                                                                   uint64_t* __restrict__ total) {
  __shared__ T lds[8];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  T v[SCAN_ITEMS];
  T s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? (T)in(base + i) : (T)0;
    s += v[i];
  }
  T tot;
  T off = block_excl_scan(s, &tot, lds) + (block_off ? block_off[blockIdx.x] : (T)0);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = off;
    off += v[i];
  }
  if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) *total = (uint64_t)off;
}

// a few tiles in ONE workgroup that carries the running sum from tile to tile: one launch instead of three where the
// launches are what a scan costs (the (group x segment) counts of a single query)
#define SCAN_LOOP_TILES 4
template <typename T, typename In>
__global__ __launch_bounds__(SCAN_THREADS) void scan_loop_kernel(In in, T* __restrict__ out, uint64_t n,
                                                                  uint64_t* __restrict__ total) {
  __shared__ T lds[8];
  T carry = 0;
  for (uint64_t t0 = 0; t0 < n; t0 += SCAN_TILE) {   // uniform
    const uint64_t base = t0 + (uint64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
      v[i] = (base + i < n) ? (T)in(base + i) : (T)0;
      s += v[i];
    }
    T tot;
    T off = block_excl_scan(s, &tot, lds) + carry;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
      if (base + i < n) out[base + i] = off;
      off += v[i];
    }
    carry += tot;
  }
  if (total && threadIdx.x == 0) *total = (uint64_t)carry;
}

// Two launches instead of three for up to SCAN_FLAT_TILES tiles: every workgroup adds up the sums of the tiles in front of
// it itself (at most a few thousand values from L2) instead of waiting for a launch that scans them.
#define SCAN_FLAT_TILES 2048
template <typename T, typename In>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_flat_kernel(In in, T* __restrict__ out, const T* __restrict__ sums,
                                                                        uint64_t n, uint64_t* __restrict__ total) {
  __shared__ T lds[8];
  T pre = 0;
  for (uint32_t i = threadIdx.x; i < blockIdx.x; i += SCAN_THREADS) pre += sums[i];
  T before;
  block_excl_scan(pre, &before, lds);   // before = sum of the tiles in front of this one
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  T v[SCAN_ITEMS];
  T s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? (T)in(base + i) : (T)0;
    s += v[i];
  }
  T tot;
  T off = block_excl_scan(s, &tot, lds) + before;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = off;
    off += v[i];
  }
  if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) *total = (uint64_t)off;
}

__global__ void set_u64_kernel(uint64_t* p, uint64_t v) { *p = v; }

template <typename T, typename In>
static int32_t scan_impl(shz_ctx* ctx, In in, T* d_out, uint64_t n, uint64_t* d_total, T* tmp, uint64_t tmp_elems) {
  if (n == 0) {
    if (d_total) hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(1), 0, ctx->stream, d_total, 0ull);
    return SHZ_OK;
  }
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 1) {
    hipLaunchKernelGGL((scan_apply_kernel<T, In>), dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, in, d_out,
                       (const T*)nullptr, n, d_total);
    SHZ_HIP(ctx, hipGetLastError());
    return SHZ_OK;
  }
  if (nb <= SCAN_LOOP_TILES) {
    hipLaunchKernelGGL((scan_loop_kernel<T, In>), dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, in, d_out, n, d_total);
    SHZ_HIP(ctx, hipGetLastError());
    return SHZ_OK;
  }
  if (nb > tmp_elems) SHZ_FAIL(ctx, SHZ_E_INVALID, "scan: temp too small");
  hipLaunchKernelGGL((scan_sums_kernel<T, In>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, tmp, n);
  SHZ_HIP(ctx, hipGetLastError());
  if (nb <= SCAN_FLAT_TILES) {
    hipLaunchKernelGGL((scan_apply_flat_kernel<T, In>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, d_out,
                       (const T*)tmp, n, d_total);
    SHZ_HIP(ctx, hipGetLastError());
    return SHZ_OK;
  }
  // scan the block sums in place (recursive), using the tail of tmp as the next level's scratch
  if (sizeof(T) == 4) {
    in_u32 nin{(const uint32_t*)tmp};
    SHZ_TRY((scan_impl<uint32_t, in_u32>(ctx, nin, (uint32_t*)tmp, nb, nullptr, (uint32_t*)tmp + nb, tmp_elems - nb)));
  } else {
    in_u64 nin{(const uint64_t*)tmp};
    SHZ_TRY((scan_impl<uint64_t, in_u64>(ctx, nin, (uint64_t*)tmp, nb, nullptr, (uint64_t*)tmp + nb, tmp_elems - nb)));
  }
  hipLaunchKernelGGL((scan_apply_kernel<T, In>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, d_out,
                     (const T*)tmp, n, d_total);
  SHZ_HIP(ctx, hipGetLastError());
  return SHZ_OK;
}

static uint64_t scan_tmp_elems(uint64_t n) {
  uint64_t t = 0;
  while (n > SCAN_TILE) {
    n = (n + SCAN_TILE - 1) / SCAN_TILE;
    t += n;
  }
  return t + 16;
}

int32_t shz_scan_u32(shz_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, uint64_t n, uint64_t* d_total) {
  if (n >= (1ull << 41)) SHZ_FAIL(ctx, SHZ_E_INVALID, "scan too large");
  uint64_t te = scan_tmp_elems(n);
  void* tmp;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SCAN_TMP, te * 8, &tmp));
  return scan_impl<uint32_t, in_u32>(ctx, in_u32{d_in}, d_out, n, d_total, (uint32_t*)tmp, te);
}

int32_t shz_scan_popc64(shz_ctx* ctx, const uint64_t* d_in, uint32_t* d_out, uint64_t n, uint64_t* d_total) {
  uint64_t te = scan_tmp_elems(n);
  void* tmp;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SCAN_TMP, te * 8, &tmp));
  return scan_impl<uint32_t, in_popc64>(ctx, in_popc64{d_in}, d_out, n, d_total, (uint32_t*)tmp, te);
}

int32_t shz_scan_u64(shz_ctx* ctx, const uint64_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* d_total) {
  uint64_t te = scan_tmp_elems(n);
  void* tmp;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SCAN_TMP, te * 8, &tmp));
  return scan_impl<uint64_t, in_u64>(ctx, in_u64{d_in}, d_out, n, d_total, (uint64_t*)tmp, te);
}

// ---------------------------------------------------------------------------------------
// Stable LSD radix sort, 8 or 9 bits per pass (9 only where it saves a whole pass: e.g. 35 key bits = 9+9+9+8).
// Tile = 4096 keys per workgroup of 256 threads.
#define SORT_THREADS 256
#define SORT_ROUNDS 16
#define SORT_TILE (SORT_THREADS * SORT_ROUNDS)

// Rank of a lane among the valid lanes of its wave row that hold the same digit d (in lane order: stable), and their
// number.  One ballot per digit bit; the mask of a lane's peers is kept as two 32-bit halves so that every bit costs a
// sign-extended bit field, a compare and two three-input bit operations (v_bitop3_b32: peers & ~(ballot ^ L), L = ~0 where
// the lane's bit is set, 0 where it is clear) -- the 64-bit select the obvious form compiles to cost eleven instructions
// per bit, and the scatter kernels are bound by instruction issue (110 VALU instructions per row of 64 keys before).
template <int BITS>
__device__ __forceinline__ void wave_digit_rank(uint32_t d, bool valid, uint32_t& rk, uint32_t& cnt) {
  const unsigned long long v = __ballot(valid);
  uint32_t plo = (uint32_t)v, phi = (uint32_t)(v >> 32);
#pragma unroll
  for (int b = 0; b < BITS; ++b) {
    const uint32_t L = (uint32_t)(((int32_t)(d << (31 - b))) >> 31);   // ~0 where the lane's bit is set, 0 where it is clear
    const unsigned long long m = __builtin_amdgcn_uicmp(L, 0u, 33 /* ICMP_NE */);   // the ballot, as ONE compare on L
    plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)m, L, 0x90);          // plo & ~(m ^ L)
    phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(m >> 32), L, 0x90);
  }
  rk = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
  cnt = (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
}

template <int BITS>
__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const uint64_t* __restrict__ keys, uint64_t n, int shift,
                                                                  uint32_t dmask /*digit mask: the last digit may be narrower*/,
                                                                  uint32_t* __restrict__ hist /*[2^BITS][nblocks]*/,
                                                                  uint32_t nblocks) {
  constexpr uint32_t DIG = 1u << BITS;
  __shared__ uint32_t h[DIG];
#pragma unroll
  for (uint32_t d = threadIdx.x; d < DIG; d += SORT_THREADS) h[d] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * SORT_TILE;
  if (base + SORT_TILE <= n) {  // whole tile: all loads in flight (two keys per lane and load) before the first count
    const ulonglong2* k2 = (const ulonglong2*)(keys + base);
    ulonglong2 x[SORT_ROUNDS / 2];
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS / 2; ++r) x[r] = k2[r * SORT_THREADS + threadIdx.x];
    // keys that arrive clustered on this digit (votes grouped by shard and query, sorted on their query bits; rows of one
    // song sorted on their song bits) put all 64 lanes of a wave on ONE counter, which the LDS then serves one lane at a
    // time: a wave whose lanes agree adds 64 once
    auto count = [&](uint32_t d) {
      const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
      if (__ballot(d != d0) == 0ull) { if ((threadIdx.x & 63) == 0) atomicAdd(&h[d0], 64u); }
      else atomicAdd(&h[d], 1u);
    };
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS / 2; ++r) {
      count((uint32_t)(x[r].x >> shift) & dmask);
      count((uint32_t)(x[r].y >> shift) & dmask);
    }
  } else {
    for (int r = 0; r < SORT_ROUNDS; ++r) {
      uint64_t i = base + (uint64_t)r * SORT_THREADS + threadIdx.x;
      if (i < n) atomicAdd(&h[(keys[i] >> shift) & dmask], 1u);
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t d = threadIdx.x; d < DIG; d += SORT_THREADS) hist[(uint64_t)d * nblocks + blockIdx.x] = h[d];
}

template <int VB> struct val_t { typedef uint32_t type; };
template <> struct val_t<8> { typedef uint64_t type; };

// Scatter of one pass.  The tile (4096 keys) is first re-ordered by digit inside LDS, then written out so that each
// digit's run is one contiguous, coalesced global segment instead of 4096 scattered 8-byte stores.  Stable: wave w owns
// the w-th quarter of the tile in memory order (16 rows of 64 keys, all loaded up front so that one memory latency
// covers the tile); a key's slot = first slot of its digit + keys of that digit in earlier waves + its rank among the
// wave's own keys of that digit (ballot rank inside a row + the wave's running count, which only that wave touches:
// LDS operations of one wave execute in order, so the rows need no barrier).
template <int VB, int BITS>
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(const uint64_t* __restrict__ keys,
                                                                     const void* __restrict__ vals_,
                                                                     uint64_t* __restrict__ okeys,
                                                                     void* __restrict__ ovals_, uint64_t n, int shift,
                                                                     uint32_t dmask,
                                                                     const uint32_t* __restrict__ offs /*[2^BITS][nblocks]*/,
                                                                     uint32_t nblocks) {
  typedef typename val_t<VB>::type V;
  constexpr uint32_t DIG = 1u << BITS;
  constexpr int DPT = DIG / SORT_THREADS;   // digits per thread in the bookkeeping steps: thread t owns t*DPT ...
  const V* vals = (const V*)vals_;
  V* ovals = (V*)ovals_;
  __shared__ uint64_t skey[SORT_TILE];
  __shared__ V sval[VB ? SORT_TILE : 1];
  __shared__ uint32_t gbase[DIG];        // first global slot of each digit's run of this tile
  __shared__ uint16_t lstart[DIG];       // first LDS slot of each digit's run (all LDS slots < 4096)
  __shared__ uint16_t wrun[4][DIG];      // per wave: running digit counts, then the first slot of the wave's keys per digit
  __shared__ uint32_t scan_tmp[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t base = (uint64_t)blockIdx.x * SORT_TILE;
  const uint32_t tile_n = (uint32_t)(n - base < SORT_TILE ? n - base : SORT_TILE);
  constexpr int ROWS = SORT_TILE / SORT_THREADS;   // rows of 64 keys per wave
  uint64_t k[ROWS];
  V pv[VB ? ROWS : 1];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const uint32_t li = (uint32_t)wave * (ROWS * 64) + (uint32_t)r * 64 + lane;
    k[r] = li < tile_n ? keys[base + li] : 0;
    if (VB != 0) pv[r] = li < tile_n ? vals[base + li] : 0;
  }
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int i = 0; i < DPT; ++i) wrun[w][threadIdx.x * DPT + i] = 0;
  {  // digit counts of this tile from the scanned histogram: next flattened entry minus this one
    uint32_t g0[DPT], c[DPT], sum = 0;
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      const uint64_t f = (uint64_t)(threadIdx.x * DPT + i) * nblocks + blockIdx.x;
      g0[i] = offs[f];
      const uint32_t g1 = (f + 1 < (uint64_t)DIG * nblocks) ? offs[f + 1] : (uint32_t)n;
      c[i] = g1 - g0[i];
      sum += c[i];
    }
    uint32_t tot;
    uint32_t ls = block_excl_scan<uint32_t>(sum, &tot, scan_tmp);
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      gbase[threadIdx.x * DPT + i] = g0[i];
      lstart[threadIdx.x * DPT + i] = (uint16_t)ls;
      ls += c[i];
    }
  }
  __syncthreads();
  uint32_t rank[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const uint32_t li = (uint32_t)wave * (ROWS * 64) + (uint32_t)r * 64 + lane;
    const bool valid = li < tile_n;
    const uint32_t d = (uint32_t)(k[r] >> shift) & dmask;
    uint32_t rk, same;
    wave_digit_rank<BITS>(d, valid, rk, same);
    const uint32_t run = wrun[wave][d];
    rank[r] = run + rk;
    if (valid && rk == 0) wrun[wave][d] = (uint16_t)(run + same);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < DPT; ++i) {
    const uint32_t d = threadIdx.x * DPT + i;
    const uint32_t c0 = wrun[0][d], c1 = wrun[1][d], c2 = wrun[2][d];
    const uint32_t ls = lstart[d];
    wrun[0][d] = (uint16_t)ls;
    wrun[1][d] = (uint16_t)(ls + c0);
    wrun[2][d] = (uint16_t)(ls + c0 + c1);
    wrun[3][d] = (uint16_t)(ls + c0 + c1 + c2);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const uint32_t li = (uint32_t)wave * (ROWS * 64) + (uint32_t)r * 64 + lane;
    if (li < tile_n) {
      const uint32_t d = (uint32_t)(k[r] >> shift) & dmask;
      const uint32_t pos = wrun[wave][d] + rank[r];
      skey[pos] = k[r];
      if (VB != 0) sval[pos] = pv[r];
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < tile_n; i += SORT_THREADS) {
    const uint64_t kk = skey[i];
    const uint32_t d = (uint32_t)(kk >> shift) & dmask;
    const uint32_t g = gbase[d] + (i - lstart[d]);
    okeys[g] = kk;
    if (VB != 0) ovals[g] = sval[i];
  }
}

// Up to one tile of keys: the whole sort in ONE workgroup, all passes inside LDS (ping-pong), 8 bits per pass.  A
// single query sorts ~2,000 hash elements on ~52 bits: that was 6 passes x 5 launches; the launches, not the work,
// were a quarter of the match latency of one query.  Same ranking scheme as sort_scatter_kernel.
template <int VB>
__global__ __launch_bounds__(SORT_THREADS) void sort_small_kernel(uint64_t* __restrict__ keys, void* __restrict__ vals_,
                                                                   uint32_t n, int bit_lo, int bit_hi) {
  typedef typename val_t<VB>::type V;
  V* vals = (V*)vals_;
  const int npass = (bit_hi - bit_lo + 7) / 8;
  __shared__ uint64_t sk[2][SORT_TILE];
  __shared__ V sv[2][VB ? SORT_TILE : 1];
  __shared__ uint16_t wrun[4][256];
  __shared__ uint32_t scan_tmp[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int ROWS = SORT_TILE / SORT_THREADS;
  for (uint32_t i = threadIdx.x; i < n; i += SORT_THREADS) {
    sk[0][i] = keys[i];
    if (VB != 0) sv[0][i] = vals[i];
  }
  for (int p = 0; p < npass; ++p) {
    const int cur = p & 1, shift = bit_lo + 8 * p;
    const uint32_t dmask = (1u << min(8, bit_hi - shift)) - 1u;   // the last digit may be narrower
#pragma unroll
    for (int w = 0; w < 4; ++w) wrun[w][threadIdx.x] = 0;
    __syncthreads();   // also: the previous pass (or the load) has filled sk[cur]
    uint64_t k[ROWS];
    uint32_t rank[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const uint32_t li = (uint32_t)wave * (ROWS * 64) + (uint32_t)r * 64 + lane;
      const bool valid = li < n;
      k[r] = valid ? sk[cur][li] : 0;
      const uint32_t d = (uint32_t)(k[r] >> shift) & dmask;
      uint32_t rk, same;
      wave_digit_rank<8>(d, valid, rk, same);
      const uint32_t run = wrun[wave][d];
      rank[r] = run + rk;
      if (valid && rk == 0) wrun[wave][d] = (uint16_t)(run + same);
    }
    __syncthreads();
    {
      const uint32_t c0 = wrun[0][threadIdx.x], c1 = wrun[1][threadIdx.x], c2 = wrun[2][threadIdx.x], c3 = wrun[3][threadIdx.x];
      uint32_t tot;
      const uint32_t ls = block_excl_scan<uint32_t>(c0 + c1 + c2 + c3, &tot, scan_tmp);
      wrun[0][threadIdx.x] = (uint16_t)ls;
      wrun[1][threadIdx.x] = (uint16_t)(ls + c0);
      wrun[2][threadIdx.x] = (uint16_t)(ls + c0 + c1);
      wrun[3][threadIdx.x] = (uint16_t)(ls + c0 + c1 + c2);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const uint32_t li = (uint32_t)wave * (ROWS * 64) + (uint32_t)r * 64 + lane;
      if (li < n) {
        const uint32_t d = (uint32_t)(k[r] >> shift) & dmask;
        const uint32_t pos = wrun[wave][d] + rank[r];
        sk[cur ^ 1][pos] = k[r];
        if (VB != 0) sv[cur ^ 1][pos] = sv[cur][li];
      }
    }
    __syncthreads();
  }
  const int fin = npass & 1;
  for (uint32_t i = threadIdx.x; i < n; i += SORT_THREADS) {
    keys[i] = sk[fin][i];
    if (VB != 0) vals[i] = sv[fin][i];
  }
}

template <int BITS>
static int32_t sort_pass(shz_ctx* ctx, uint32_t nblocks, const uint64_t* kin, const void* vin, uint64_t* kout, void* vout,
                      int vbytes, uint64_t n, int shift, uint32_t dmask, uint32_t* hist) {
  hipLaunchKernelGGL(sort_hist_kernel<BITS>, dim3(nblocks), dim3(SORT_THREADS), 0, ctx->stream, kin, n, shift, dmask, hist, nblocks);
  SHZ_TRY(shz_scan_u32(ctx, hist, hist, (uint64_t)nblocks << BITS, nullptr));
  if (vbytes == 4)
    hipLaunchKernelGGL((sort_scatter_kernel<4, BITS>), dim3(nblocks), dim3(SORT_THREADS), 0, ctx->stream, kin, vin, kout, vout,
                       n, shift, dmask, (const uint32_t*)hist, nblocks);
  else if (vbytes == 8)
    hipLaunchKernelGGL((sort_scatter_kernel<8, BITS>), dim3(nblocks), dim3(SORT_THREADS), 0, ctx->stream, kin, vin, kout, vout,
                       n, shift, dmask, (const uint32_t*)hist, nblocks);
  else
    hipLaunchKernelGGL((sort_scatter_kernel<0, BITS>), dim3(nblocks), dim3(SORT_THREADS), 0, ctx->stream, kin, vin, kout, vout,
                       n, shift, dmask, (const uint32_t*)hist, nblocks);
  SHZ_HIP(ctx, hipGetLastError());
  return SHZ_OK;
}

int32_t shz_sort_u64(shz_ctx* ctx, uint64_t* k0, uint64_t* k1, void* v0, void* v1, int vbytes, uint64_t n, int bit_lo,
                     int bit_hi, int* out_sel) {
  *out_sel = 0;
  if (n <= 1 || bit_hi <= bit_lo) return SHZ_OK;
  if (n >= (1ull << 32)) SHZ_FAIL(ctx, SHZ_E_INVALID, "sort: n must be < 2^32 (got %llu)", (unsigned long long)n);
  if (vbytes != 0 && vbytes != 4 && vbytes != 8) SHZ_FAIL(ctx, SHZ_E_INVALID, "sort: payload must be 0, 4 or 8 bytes");
  if (n <= SORT_TILE) {  // one workgroup, every pass in LDS, sorted in place (*out_sel stays 0)
    if (vbytes == 4)
      hipLaunchKernelGGL(sort_small_kernel<4>, dim3(1), dim3(SORT_THREADS), 0, ctx->stream, k0, v0, (uint32_t)n, bit_lo, bit_hi);
    else if (vbytes == 8)
      hipLaunchKernelGGL(sort_small_kernel<8>, dim3(1), dim3(SORT_THREADS), 0, ctx->stream, k0, v0, (uint32_t)n, bit_lo, bit_hi);
    else
      hipLaunchKernelGGL(sort_small_kernel<0>, dim3(1), dim3(SORT_THREADS), 0, ctx->stream, k0, v0, (uint32_t)n, bit_lo, bit_hi);
    SHZ_HIP(ctx, hipGetLastError());
    return SHZ_OK;
  }
  const uint32_t nblocks = (uint32_t)((n + SORT_TILE - 1) / SORT_TILE);
  const int bits = bit_hi - bit_lo;
  const int np8 = (bits + 7) / 8, np9 = (bits + 8) / 9;
  const bool wide = np9 < np8;   // 9-bit digits only where they save a pass
  void* hist;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_H, ((uint64_t)nblocks << (wide ? 9 : 8)) * 4, &hist));
  uint64_t* kin = k0;
  uint64_t* kout = k1;
  void* vin = v0;
  void* vout = v1;
  int sel = 0, left = wide ? np9 : np8;
  for (int shift = bit_lo; shift < bit_hi; --left) {
    const int w = wide ? (bit_hi - shift + left - 1) / left : 8;   // wide: spread the bits evenly, each digit <= 9
    const uint32_t dmask = (1u << std::min(w == 9 ? 9 : 8, bit_hi - shift)) - 1u;   // exactly the bits [bit_lo, bit_hi)
    if (w == 9)
      SHZ_TRY(sort_pass<9>(ctx, nblocks, kin, vin, kout, vout, vbytes, n, shift, dmask, (uint32_t*)hist));
    else
      SHZ_TRY(sort_pass<8>(ctx, nblocks, kin, vin, kout, vout, vbytes, n, shift, dmask, (uint32_t*)hist));
    shift += w == 9 ? 9 : 8;
    uint64_t* tk = kin; kin = kout; kout = tk;
    void* tv = vin; vin = vout; vout = tv;
    sel ^= 1;
  }
  *out_sel = sel;
  return SHZ_OK;
}

// ---------------------------------------------------------------------------------------
// The same sort for 4-BYTE keys (the votes of one or two queries fit 32 bits: song 20 + delta 10 + flag 1), keys only.
// Half the bytes per pass; the LAST pass writes 8-byte keys, `key + add`, so that what follows the sort (the fold of the
// votes, which wants the query index above the song id) reads the usual 64-bit layout.
template <int BITS>
__global__ __launch_bounds__(SORT_THREADS) void sort_hist32_kernel(const uint32_t* __restrict__ keys, uint64_t n, int shift,
                                                                    uint32_t dmask, uint32_t* __restrict__ hist,
                                                                    uint32_t nblocks) {
  constexpr uint32_t DIG = 1u << BITS;
  __shared__ uint32_t h[DIG];
#pragma unroll
  for (uint32_t d = threadIdx.x; d < DIG; d += SORT_THREADS) h[d] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * SORT_TILE;
  if (base + SORT_TILE <= n) {  // whole tile: four keys per lane and load, all loads in flight before the first count
    const uint4* k4 = (const uint4*)(keys + base);
    uint4 x[SORT_ROUNDS / 4];
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS / 4; ++r) x[r] = k4[r * SORT_THREADS + threadIdx.x];
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS / 4; ++r) {
      atomicAdd(&h[(x[r].x >> shift) & dmask], 1u);
      atomicAdd(&h[(x[r].y >> shift) & dmask], 1u);
      atomicAdd(&h[(x[r].z >> shift) & dmask], 1u);
      atomicAdd(&h[(x[r].w >> shift) & dmask], 1u);
    }
  } else {
    for (int r = 0; r < SORT_ROUNDS; ++r) {
      const uint64_t i = base + (uint64_t)r * SORT_THREADS + threadIdx.x;
      if (i < n) atomicAdd(&h[(keys[i] >> shift) & dmask], 1u);
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t d = threadIdx.x; d < DIG; d += SORT_THREADS) hist[(uint64_t)d * nblocks + blockIdx.x] = h[d];
}

// OT = uint32_t: an ordinary pass; OT = uint64_t: the last pass, out = key + add
template <typename OT, int BITS>
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter32_kernel(const uint32_t* __restrict__ keys, OT* __restrict__ okeys,
                                                                       uint64_t n, int shift, uint32_t dmask,
                                                                       const uint32_t* __restrict__ offs, uint32_t nblocks,
                                                                       uint64_t add) {
  constexpr uint32_t DIG = 1u << BITS;
  constexpr int DPT = DIG / SORT_THREADS;
  __shared__ uint32_t skey[SORT_TILE];
  __shared__ uint32_t gbase[DIG];
  __shared__ uint16_t lstart[DIG];
  __shared__ uint16_t wrun[4][DIG];
  __shared__ uint32_t scan_tmp[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t base = (uint64_t)blockIdx.x * SORT_TILE;
  const uint32_t tile_n = (uint32_t)(n - base < SORT_TILE ? n - base : SORT_TILE);
  constexpr int ROWS = SORT_TILE / SORT_THREADS;
  uint32_t k[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const uint32_t li = (uint32_t)wave * (ROWS * 64) + (uint32_t)r * 64 + lane;
    k[r] = li < tile_n ? keys[base + li] : 0;
  }
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int i = 0; i < DPT; ++i) wrun[w][threadIdx.x * DPT + i] = 0;
  {
    uint32_t g0[DPT], c[DPT], sum = 0;
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      const uint64_t f = (uint64_t)(threadIdx.x * DPT + i) * nblocks + blockIdx.x;
      g0[i] = offs[f];
      const uint32_t g1 = (f + 1 < (uint64_t)DIG * nblocks) ? offs[f + 1] : (uint32_t)n;
      c[i] = g1 - g0[i];
      sum += c[i];
    }
    uint32_t tot;
    uint32_t ls = block_excl_scan<uint32_t>(sum, &tot, scan_tmp);
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      gbase[threadIdx.x * DPT + i] = g0[i];
      lstart[threadIdx.x * DPT + i] = (uint16_t)ls;
      ls += c[i];
    }
  }
  __syncthreads();
  uint32_t rank[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const uint32_t li = (uint32_t)wave * (ROWS * 64) + (uint32_t)r * 64 + lane;
    const bool valid = li < tile_n;
    const uint32_t d = (k[r] >> shift) & dmask;
    uint32_t rk, same;
    wave_digit_rank<BITS>(d, valid, rk, same);
    const uint32_t run = wrun[wave][d];
    rank[r] = run + rk;
    if (valid && rk == 0) wrun[wave][d] = (uint16_t)(run + same);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < DPT; ++i) {
    const uint32_t d = threadIdx.x * DPT + i;
    const uint32_t c0 = wrun[0][d], c1 = wrun[1][d], c2 = wrun[2][d];
    const uint32_t ls = lstart[d];
    wrun[0][d] = (uint16_t)ls;
    wrun[1][d] = (uint16_t)(ls + c0);
    wrun[2][d] = (uint16_t)(ls + c0 + c1);
    wrun[3][d] = (uint16_t)(ls + c0 + c1 + c2);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const uint32_t li = (uint32_t)wave * (ROWS * 64) + (uint32_t)r * 64 + lane;
    if (li < tile_n) {
      const uint32_t d = (k[r] >> shift) & dmask;
      skey[wrun[wave][d] + rank[r]] = k[r];
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < tile_n; i += SORT_THREADS) {
    const uint32_t kk = skey[i];
    const uint32_t d = (kk >> shift) & dmask;
    okeys[gbase[d] + (i - lstart[d])] = (OT)((uint64_t)kk + add);
  }
}

// widen in place of a sort: out[i] = in[i] + add (when no bit needs sorting)
__global__ void widen32_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t n, uint64_t add) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint64_t)in[i] + add;
}

// out64 == nullptr: every pass writes 4-byte keys and *sel says which of k0 / k1 holds the result
int32_t shz_sort_u32_widen(shz_ctx* ctx, uint32_t* k0, uint32_t* k1, uint64_t* out64, uint64_t n, int bit_lo, int bit_hi,
                           uint64_t add, int* sel) {
  if (sel) *sel = 0;
  if (n == 0) return SHZ_OK;
  if (n >= (1ull << 32)) SHZ_FAIL(ctx, SHZ_E_INVALID, "sort: n must be < 2^32 (got %llu)", (unsigned long long)n);
  if (bit_hi > 32) SHZ_FAIL(ctx, SHZ_E_INVALID, "sort32: bits [%d, %d)", bit_lo, bit_hi);
  if (bit_hi <= bit_lo) {
    if (!out64) return SHZ_OK;
    hipLaunchKernelGGL(widen32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)k0, out64, n, add);
    SHZ_HIP(ctx, hipGetLastError());
    return SHZ_OK;
  }
  const uint32_t nblocks = (uint32_t)((n + SORT_TILE - 1) / SORT_TILE);
  const int bits = bit_hi - bit_lo;
  const int np8 = (bits + 7) / 8, np9 = (bits + 8) / 9;
  const bool wide = np9 < np8;
  void* hist;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_H, ((uint64_t)nblocks << (wide ? 9 : 8)) * 4, &hist));
  uint32_t *kin = k0, *kout = k1;
  int left = wide ? np9 : np8;
  for (int shift = bit_lo; shift < bit_hi; --left) {
    const int w = wide ? (bit_hi - shift + left - 1) / left : 8;
    const int wb = w == 9 ? 9 : 8;
    const uint32_t dmask = (1u << std::min(wb, bit_hi - shift)) - 1u;
    const bool last = out64 && shift + wb >= bit_hi;
    if (wb == 9) {
      hipLaunchKernelGGL(sort_hist32_kernel<9>, dim3(nblocks), dim3(SORT_THREADS), 0, ctx->stream, (const uint32_t*)kin, n, shift, dmask, (uint32_t*)hist, nblocks);
      SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)hist, (uint32_t*)hist, (uint64_t)nblocks << 9, nullptr));
      if (last) hipLaunchKernelGGL((sort_scatter32_kernel<uint64_t, 9>), dim3(nblocks), dim3(SORT_THREADS), 0, ctx->stream, (const uint32_t*)kin, out64, n, shift, dmask, (const uint32_t*)hist, nblocks, add);
      else hipLaunchKernelGGL((sort_scatter32_kernel<uint32_t, 9>), dim3(nblocks), dim3(SORT_THREADS), 0, ctx->stream, (const uint32_t*)kin, kout, n, shift, dmask, (const uint32_t*)hist, nblocks, (uint64_t)0);
    } else {
      hipLaunchKernelGGL(sort_hist32_kernel<8>, dim3(nblocks), dim3(SORT_THREADS), 0, ctx->stream, (const uint32_t*)kin, n, shift, dmask, (uint32_t*)hist, nblocks);
      SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)hist, (uint32_t*)hist, (uint64_t)nblocks << 8, nullptr));
      if (last) hipLaunchKernelGGL((sort_scatter32_kernel<uint64_t, 8>), dim3(nblocks), dim3(SORT_THREADS), 0, ctx->stream, (const uint32_t*)kin, out64, n, shift, dmask, (const uint32_t*)hist, nblocks, add);
      else hipLaunchKernelGGL((sort_scatter32_kernel<uint32_t, 8>), dim3(nblocks), dim3(SORT_THREADS), 0, ctx->stream, (const uint32_t*)kin, kout, n, shift, dmask, (const uint32_t*)hist, nblocks, (uint64_t)0);
    }
    SHZ_HIP(ctx, hipGetLastError());
    shift += wb;
    std::swap(kin, kout);
  }
  if (sel) *sel = kin == k1 ? 1 : 0;
  return SHZ_OK;
}

// ---------------------------------------------------------------------------------------
// SEGMENTED form of the 4-byte sort: the keys of segment i stay inside [qv[i], qv[i+1]) and are ordered among
// themselves.  The votes of a pass come out of the expand query by query, so the query index needs no bits of the
// key: a pass holds as many queries as the vote budget allows, not the two that a 32-bit key has a spare bit for, and
// every per-pass launch (scans of the digit tables, tile bounds, rank) is paid once per ~30 queries instead of once
// per two.  Blocks of <= 4,096 keys never cross a segment border; the digit table is laid out segment-major
// ([segment][digit][block of the segment]), so ONE linear exclusive scan yields every block's destinations: keys of
// earlier segments, then smaller digits of the own segment, then the same digit in earlier blocks of the segment.
__device__ __forceinline__ void seg_of_block(const shz_seg_plan& sp, uint32_t tile, uint32_t b, uint32_t& seg, uint32_t& lo,
                                             uint32_t& len, uint32_t& hbase, uint32_t& nbs) {
  uint32_t i = 0, hi = sp.nq;                        // last segment with bq[i] <= b (empty segments have no blocks)
  while (hi - i > 1) {
    const uint32_t mid = (i + hi) >> 1;
    if (sp.bq[mid] <= b) i = mid; else hi = mid;
  }
  seg = i;
  const uint32_t bl = b - sp.bq[i];
  lo = sp.qv[i] + bl * tile;
  len = min(tile, sp.qv[i + 1] - lo);
  nbs = sp.bq[i + 1] - sp.bq[i];
  hbase = bl;                                       // + (bq[i] << BITS) + digit * nbs
}

template <int BITS, int ROWS>
__global__ __launch_bounds__(SORT_THREADS) void sort_hist32_seg_kernel(const uint32_t* __restrict__ keys, shz_seg_plan sp, int shift,
                                                                        uint32_t dmask, uint32_t* __restrict__ hist) {
  constexpr uint32_t DIG = 1u << BITS;
  __shared__ uint32_t h[DIG];
#pragma unroll
  for (uint32_t d = threadIdx.x; d < DIG; d += SORT_THREADS) h[d] = 0;
  __syncthreads();
  uint32_t seg, lo, len, hb, nbs;
  seg_of_block(sp, (uint32_t)ROWS * SORT_THREADS, blockIdx.x, seg, lo, len, hb, nbs);
  // a segment starts wherever its query's votes start: up to three keys in front of the first 16-byte boundary and
  // behind the last one are counted singly, the rest four per load
  // (alignment by ADDRESS: the key buffer itself may start anywhere, e.g. the second half of a ping-pong pair)
  const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(keys + lo) >> 2) & 3u);
  const uint32_t head = min((4u - mis) & 3u, len), nv = (len - head) >> 2, tail0 = head + 4u * nv;
  {
    const uint4* k4 = (const uint4*)(keys + lo + head);
    uint4 x[ROWS / 4];
#pragma unroll
    for (int r = 0; r < ROWS / 4; ++r) {
      const uint32_t i = (uint32_t)r * SORT_THREADS + threadIdx.x;
      x[r] = i < nv ? k4[i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < ROWS / 4; ++r) {
      if ((uint32_t)r * SORT_THREADS + threadIdx.x < nv) {
        atomicAdd(&h[(x[r].x >> shift) & dmask], 1u);
        atomicAdd(&h[(x[r].y >> shift) & dmask], 1u);
        atomicAdd(&h[(x[r].z >> shift) & dmask], 1u);
        atomicAdd(&h[(x[r].w >> shift) & dmask], 1u);
      }
    }
    if (threadIdx.x < head) atomicAdd(&h[(keys[lo + threadIdx.x] >> shift) & dmask], 1u);
    if (tail0 + threadIdx.x < len) atomicAdd(&h[(keys[lo + tail0 + threadIdx.x] >> shift) & dmask], 1u);
  }
  __syncthreads();
  const uint64_t base = ((uint64_t)sp.bq[seg] << BITS) + hb;
#pragma unroll
  for (uint32_t d = threadIdx.x; d < DIG; d += SORT_THREADS) hist[base + (uint64_t)d * nbs] = h[d];
}

// NW waves per workgroup, ROWS keys per thread: a block of 64 * NW * ROWS keys.  (8,192 keys as 8 waves x 16 rows instead
// of 4 x 32 -- twice the waves per CU for the same runs: 0.1765 -> 0.1785 ms/query at 1M songs, same box; not used.)
template <int BITS, int ROWS, int NW>
__global__ __launch_bounds__(64 * NW) void sort_scatter32_seg_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ okeys,
                                                                      uint64_t n, shz_seg_plan sp, int shift, uint32_t dmask,
                                                                      const uint32_t* __restrict__ offs, uint64_t n_hist) {
  constexpr uint32_t DIG = 1u << BITS, THREADS = 64u * NW;
  constexpr int DPT = DIG >= THREADS ? DIG / THREADS : 1;   // digits a thread owns (threads past the last digit: none)
  __shared__ uint32_t skey[ROWS * THREADS];
  __shared__ uint32_t gbase[DIG];
  __shared__ uint16_t lstart[DIG];
  __shared__ uint16_t wrun[NW][DIG];
  __shared__ uint32_t scan_tmp[NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool owner = threadIdx.x * DPT < DIG;
  uint32_t seg, lo, tile_n, hb, nbs;
  seg_of_block(sp, (uint32_t)ROWS * THREADS, blockIdx.x, seg, lo, tile_n, hb, nbs);
  uint32_t k[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const uint32_t li = (uint32_t)wave * (ROWS * 64) + (uint32_t)r * 64 + lane;
    k[r] = li < tile_n ? keys[lo + li] : 0;
  }
  if (owner) {
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int i = 0; i < DPT; ++i) wrun[w][threadIdx.x * DPT + i] = 0;
  }
  {
    uint32_t g0[DPT], c[DPT], sum = 0;
    const uint64_t base = ((uint64_t)sp.bq[seg] << BITS) + hb;
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      g0[i] = c[i] = 0;
      if (owner) {
        const uint64_t f = base + (uint64_t)(threadIdx.x * DPT + i) * nbs;
        g0[i] = offs[f];
        const uint32_t g1 = (f + 1 < n_hist) ? offs[f + 1] : (uint32_t)n;   // the next entry of the scan: own count behind g0
        c[i] = g1 - g0[i];
        sum += c[i];
      }
    }
    // exclusive scan of `sum` over the workgroup
    const uint32_t inc = wave_incl_scan(sum, lane);
    if (lane == 63) scan_tmp[wave] = inc;
    __syncthreads();
    uint32_t ls = inc - sum;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (w < wave) ls += scan_tmp[w];
    if (owner) {
#pragma unroll
      for (int i = 0; i < DPT; ++i) {
        gbase[threadIdx.x * DPT + i] = g0[i] - ls;   // destination of local position i: gbase[d] + i (mod 2^32)
        lstart[threadIdx.x * DPT + i] = (uint16_t)ls;
        ls += c[i];
      }
    }
  }
  __syncthreads();
  uint32_t rank[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const uint32_t li = (uint32_t)wave * (ROWS * 64) + (uint32_t)r * 64 + lane;
    const bool valid = li < tile_n;
    const uint32_t d = (k[r] >> shift) & dmask;
    uint32_t rk, same;
    wave_digit_rank<BITS>(d, valid, rk, same);
    const uint32_t run = wrun[wave][d];
    rank[r] = run + rk;
    if (valid && rk == 0) wrun[wave][d] = (uint16_t)(run + same);
  }
  __syncthreads();
  if (owner) {
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      const uint32_t d = threadIdx.x * DPT + i;
      uint32_t at = lstart[d];
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const uint32_t cw = wrun[w][d];
        wrun[w][d] = (uint16_t)at;
        at += cw;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const uint32_t li = (uint32_t)wave * (ROWS * 64) + (uint32_t)r * 64 + lane;
    if (li < tile_n) {
      const uint32_t d = (k[r] >> shift) & dmask;
      skey[wrun[wave][d] + rank[r]] = k[r];
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < tile_n; i += THREADS) {
    const uint32_t kk = skey[i];
    const uint32_t d = (kk >> shift) & dmask;
    okeys[gbase[d] + i] = kk;
  }
}

// HR: keys per thread of the counting kernel (256 threads); the scatter covers the same block as NW waves x SR rows
template <int HR, int SR, int NW>
static void seg_pass(shz_ctx* ctx, int wb, uint32_t nblocks, const uint32_t* kin, uint32_t* kout, uint64_t n, const shz_seg_plan& sp,
                     int shift, uint32_t dmask, uint32_t* hist, uint64_t nh, bool counted, int32_t* rc) {
  static_assert(HR * SORT_THREADS == SR * 64 * NW, "one block size");
  if (wb == 9) {
    if (!counted) hipLaunchKernelGGL((sort_hist32_seg_kernel<9, HR>), dim3(nblocks), dim3(SORT_THREADS), 0, ctx->stream, kin, sp, shift, dmask, hist);
    *rc = shz_scan_u32(ctx, (const uint32_t*)hist, hist, nh, nullptr);
    if (*rc != SHZ_OK) return;
    hipLaunchKernelGGL((sort_scatter32_seg_kernel<9, SR, NW>), dim3(nblocks), dim3(64 * NW), 0, ctx->stream, kin, kout, n, sp, shift, dmask, (const uint32_t*)hist, nh);
  } else {
    if (!counted) hipLaunchKernelGGL((sort_hist32_seg_kernel<8, HR>), dim3(nblocks), dim3(SORT_THREADS), 0, ctx->stream, kin, sp, shift, dmask, hist);
    *rc = shz_scan_u32(ctx, (const uint32_t*)hist, hist, nh, nullptr);
    if (*rc != SHZ_OK) return;
    hipLaunchKernelGGL((sort_scatter32_seg_kernel<8, SR, NW>), dim3(nblocks), dim3(64 * NW), 0, ctx->stream, kin, kout, n, sp, shift, dmask, (const uint32_t*)hist, nh);
  }
}

// keys per block of a segmented sort of n keys: blocks of 8,192 keys where the segments are long: a block's keys of one
// digit leave as one run, 128 bytes on average instead of 64 (the scatter is bound by its partial-line writes)
// (16,384: 64 keys per thread in registers, 0.185 -> 0.222 ms/query at 1M songs)
uint32_t shz_seg_tile(uint64_t n) {
  static const int tile_env = [] { const char* e = getenv("SHZ_SEG_TILE"); return e ? atoi(e) : 0; }();
  // (fewer than ~2,000 large blocks leave CUs idle: one 10 s query at 1M songs 0.393 -> 0.399 ms)
  const bool big = tile_env ? tile_env >= 8192 : n >= (1ull << 24);
  return big ? 8192u : 4096u;
}

void shz_seg_blocks(shz_seg_plan* sp, uint32_t tile) {
  sp->bq[0] = 0;
  for (uint32_t i = 0; i < SHZ_SEG_MAX; ++i)
    sp->bq[i + 1] = sp->bq[i] + (i < sp->nq ? (sp->qv[i + 1] - sp->qv[i] + tile - 1) / tile : 0u);
}

// width of the first pass over the bits [bit_lo, bit_hi), and the digit mask of that pass
int shz_seg_first_pass(int bit_lo, int bit_hi, uint32_t* dmask) {
  const int bits = bit_hi - bit_lo;
  const int np8 = (bits + 7) / 8, np9 = (bits + 8) / 9;
  const bool wide = np9 < np8;
  const int wb = wide ? 9 : 8;   // (a plan with a 9-bit pass is not counted ahead: its table has 512 columns per block)
  if (dmask) *dmask = (1u << std::min(wb, bits)) - 1u;
  return wb;
}

// `hist0`: the digit table of the FIRST pass is already in the SHZ_WS_SORT_H workspace (the producer of the keys counted
// them, block by block, in this function's layout -- m_expand_blocks_kernel); sp_in then carries the blocks (shz_seg_blocks
// with shz_seg_tile(n)) the producer used
int32_t shz_sort_u32_seg(shz_ctx* ctx, uint32_t* k0, uint32_t* k1, uint64_t n, int bit_lo, int bit_hi, const shz_seg_plan& sp_in,
                         int* sel, bool hist0) {
  if (sel) *sel = 0;
  if (n == 0 || bit_hi <= bit_lo) return SHZ_OK;
  if (n >= (1ull << 32) || bit_hi > 32) SHZ_FAIL(ctx, SHZ_E_INVALID, "sort32: n %llu, bits [%d, %d)", (unsigned long long)n, bit_lo, bit_hi);
  const uint32_t tile = shz_seg_tile(n);
  const bool big = tile == 8192u;
  shz_seg_plan sp = sp_in;
  shz_seg_blocks(&sp, tile);
  const uint32_t nblocks = sp.bq[sp.nq];
  const int bits = bit_hi - bit_lo;
  const int np8 = (bits + 7) / 8, np9 = (bits + 8) / 9;
  const bool wide = np9 < np8;
  if (hist0 && wide) SHZ_FAIL(ctx, SHZ_E_INVALID, "sort32: a counted first pass is 8 bits wide");
  void* hist;
  const uint64_t n_hist = (uint64_t)nblocks << (wide ? 9 : 8);
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_H, n_hist * 4, &hist));   // (same size as the producer's reservation: same block)
  uint32_t *kin = k0, *kout = k1;
  int left = wide ? np9 : np8;
  for (int shift = bit_lo; shift < bit_hi; --left) {
    const int w = wide ? (bit_hi - shift + left - 1) / left : 8;
    const int wb = w == 9 ? 9 : 8;
    const uint32_t dmask = (1u << std::min(wb, bit_hi - shift)) - 1u;
    const uint64_t nh = (uint64_t)nblocks << wb;
    int32_t rc = SHZ_OK;
    const bool counted = hist0 && shift == bit_lo;
    if (big) seg_pass<32, 32, 4>(ctx, wb, nblocks, kin, kout, n, sp, shift, dmask, (uint32_t*)hist, nh, counted, &rc);
    else seg_pass<16, 16, 4>(ctx, wb, nblocks, kin, kout, n, sp, shift, dmask, (uint32_t*)hist, nh, counted, &rc);
    SHZ_TRY(rc);
    SHZ_HIP(ctx, hipGetLastError());
    shift += wb;
    std::swap(kin, kout);
  }
  if (sel) *sel = kin == k1 ? 1 : 0;
  return SHZ_OK;
}

extern "C" int32_t shz_sort_keys32_seg(shz_ctx* ctx, const uint32_t* keys, const uint64_t* seg_off, uint32_t n_segs,
                                       uint32_t bit_lo, uint32_t bit_hi, uint32_t* out) {
  if (!ctx) return SHZ_E_INVALID;
  if (!seg_off || n_segs == 0 || n_segs > SHZ_SEG_MAX) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_sort_keys32_seg: 1..%d segments", SHZ_SEG_MAX);
  const uint64_t n = seg_off[n_segs];
  if (n == 0) return SHZ_OK;
  if (!keys || !out || seg_off[0] != 0 || n >= (1ull << 32) || bit_hi > 32 || bit_lo > bit_hi)
    SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_sort_keys32_seg: arguments");
  shz_seg_plan sp;
  sp.nq = n_segs;
  sp.qv[0] = sp.bq[0] = 0;
  for (uint32_t i = 0; i < SHZ_SEG_MAX; ++i) {
    const uint64_t a = i < n_segs ? seg_off[i] : n, b = i < n_segs ? seg_off[i + 1] : n;
    if (b < a) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_sort_keys32_seg: offsets must not decrease");
    sp.qv[i + 1] = (uint32_t)b;
    sp.bq[i + 1] = sp.bq[i] + (uint32_t)((b - a + SORT_TILE - 1) / SORT_TILE);
  }
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  void* k0;
  const uint64_t n4 = (n + 3) & ~3ull;   // the second buffer of the ping-pong pair starts 16-byte aligned
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_A, (n4 + n) * 4, &k0));
  SHZ_HIP(ctx, shz_memcpy(ctx, k0, keys, n * 4, hipMemcpyHostToDevice));
  int sel = 0;
  SHZ_TRY(shz_sort_u32_seg(ctx, (uint32_t*)k0, (uint32_t*)k0 + n4, n, (int)bit_lo, (int)bit_hi, sp, &sel, false));
  SHZ_HIP(ctx, shz_memcpy(ctx, out, (uint32_t*)k0 + (sel ? n4 : 0), n * 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}

extern "C" int32_t shz_sort_keys32(shz_ctx* ctx, const uint32_t* keys, uint64_t n, uint32_t bit_lo, uint32_t bit_hi,
                                   uint64_t add, uint64_t* out64) {
  if (!ctx) return SHZ_E_INVALID;
  if (n == 0) return SHZ_OK;
  if (!keys || !out64) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_sort_keys32: NULL buffer");
  if (bit_hi > 32 || bit_lo > bit_hi) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_sort_keys32: bits [%u, %u)", bit_lo, bit_hi);
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  void *k0, *o;
  const uint64_t n4 = (n + 3) & ~3ull;   // the second buffer of the ping-pong pair starts 16-byte aligned
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_A, (n4 + n) * 4, &k0));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_B, n * 8, &o));
  SHZ_HIP(ctx, shz_memcpy(ctx, k0, keys, n * 4, hipMemcpyHostToDevice));
  SHZ_TRY(shz_sort_u32_widen(ctx, (uint32_t*)k0, (uint32_t*)k0 + n4, (uint64_t*)o, n, (int)bit_lo, (int)bit_hi, add, nullptr));
  SHZ_HIP(ctx, shz_memcpy(ctx, out64, o, n * 8, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}

extern "C" int32_t shz_sort_pairs(shz_ctx* ctx, uint64_t* keys, void* vals, uint32_t val_bytes, uint64_t n, uint32_t bit_lo,
                                  uint32_t bit_hi) {
  if (!ctx) return SHZ_E_INVALID;
  if (n == 0) return SHZ_OK;
  if (!keys || (val_bytes && !vals)) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_sort_pairs: NULL buffer");
  if (bit_hi > 64 || bit_lo > bit_hi) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_sort_pairs: bits [%u, %u)", bit_lo, bit_hi);
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  void *k0, *k1, *v0 = nullptr, *v1 = nullptr;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_A, n * 8, &k0));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_B, n * 8, &k1));
  if (val_bytes) {
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_C, n * val_bytes, &v0));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SORT_D, n * val_bytes, &v1));
    SHZ_HIP(ctx, shz_memcpy(ctx, v0, vals, n * val_bytes, hipMemcpyHostToDevice));
  }
  SHZ_HIP(ctx, shz_memcpy(ctx, k0, keys, n * 8, hipMemcpyHostToDevice));
  int sel = 0;
  SHZ_TRY(shz_sort_u64(ctx, (uint64_t*)k0, (uint64_t*)k1, v0, v1, (int)val_bytes, n, (int)bit_lo, (int)bit_hi, &sel));
  SHZ_HIP(ctx, shz_memcpy(ctx, keys, sel ? k1 : k0, n * 8, hipMemcpyDeviceToHost));
  if (val_bytes) SHZ_HIP(ctx, shz_memcpy(ctx, vals, sel ? v1 : v0, n * val_bytes, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}
