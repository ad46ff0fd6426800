// Inverse of the reference's hash on the GPU: which (f1, f2, dt) produced a given 10-byte digest?
// The preimage space of sha1(f"{f1}|{f2}|{dt}")[:20] (__init__.py:207-208) is tiny -- f <= 2048,
// dt <= 200 (MAX_HASH_TIME_DELTA), 8.4e8 strings -- so a hex hash that arrives from outside (a MySQL
// dump, another process) is resolved by hashing the whole space once and probing the sorted targets.
// This is what lets the key32-indexed device table serve the hex-keyed reference API
// (insert_hashes / SELECT_MULTIPLE, mysql_database.py:62-68, 82-86) for hashes it never produced.
#include <algorithm>

#include "shz_internal.h"

__device__ __forceinline__ uint32_t rol(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

// append the decimal digits of v (< 10000) to a big-endian 128-bit string accumulator
__device__ __forceinline__ void put_dec128(uint64_t& hi, uint64_t& lo, int& len, uint32_t v) {
  const uint32_t d3 = v / 1000, d2 = (v / 100) % 10, d1 = (v / 10) % 10, d0 = v % 10;
  const int nd = v >= 1000 ? 4 : (v >= 100 ? 3 : (v >= 10 ? 2 : 1));
  uint32_t digs = ('0' + d3) << 24 | ('0' + d2) << 16 | ('0' + d1) << 8 | ('0' + d0);
  digs &= nd == 4 ? 0xFFFFFFFFu : (0xFFFFFFFFu >> (8 * (4 - nd)));
  hi = (hi << (8 * nd)) | (lo >> (64 - 8 * nd));
  lo = (lo << (8 * nd)) | digs;
  len += nd;
}
__device__ __forceinline__ void put_chr128(uint64_t& hi, uint64_t& lo, int& len, uint32_t c) {
  hi = (hi << 8) | (lo >> 56);
  lo = (lo << 8) | c;
  len += 1;
}

__device__ __forceinline__ void sha1_of_key(uint32_t f1, uint32_t f2, uint32_t dt, uint32_t& h0, uint32_t& h1, uint32_t& h2) {
  uint64_t hi = 0, lo = 0;
  int len = 0;
  put_dec128(hi, lo, len, f1);
  put_chr128(hi, lo, len, '|');
  put_dec128(hi, lo, len, f2);
  put_chr128(hi, lo, len, '|');
  put_dec128(hi, lo, len, dt);
  const int mlen = len;            // message bytes (<= 13)
  put_chr128(hi, lo, len, 0x80);   // padding marker, len <= 14
  const int sh = 8 * (16 - len);   // left-align the string in the 128-bit block prefix (16..104 bits)
  if (sh >= 64) { hi = lo << (sh - 64); lo = 0; }
  else if (sh > 0) { hi = (hi << sh) | (lo >> (64 - sh)); lo <<= sh; }
  uint32_t w[16];
  w[0] = (uint32_t)(hi >> 32); w[1] = (uint32_t)hi; w[2] = (uint32_t)(lo >> 32); w[3] = (uint32_t)lo;
#pragma unroll
  for (int q = 4; q < 15; ++q) w[q] = 0;
  w[15] = (uint32_t)mlen * 8;
  uint32_t a = 0x67452301u, b = 0xEFCDAB89u, c = 0x98BADCFEu, d = 0x10325476u, e = 0xC3D2E1F0u;
#pragma unroll
  for (int r = 0; r < 80; ++r) {
    uint32_t wt;
    if (r < 16) wt = w[r];
    else {
      wt = rol(w[(r + 13) & 15] ^ w[(r + 8) & 15] ^ w[(r + 2) & 15] ^ w[r & 15], 1);
      w[r & 15] = wt;
    }
    uint32_t f, k;
    if (r < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
    else if (r < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
    else if (r < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
    else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
    const uint32_t tmp = rol(a, 5) + f + e + k + wt;
    e = d; d = c; c = rol(b, 30); b = a; a = tmp;
  }
  h0 = 0x67452301u + a; h1 = 0xEFCDAB89u + b; h2 = 0x98BADCFEu + c;
}

// grid: x over (f2, dt) pairs, y = f1.  targets sorted by their first 8 digest bytes (big-endian u64).
__global__ __launch_bounds__(256) void sha1_invert_kernel(const uint64_t* __restrict__ tgt_hi, const uint16_t* __restrict__ tgt_lo,
                                                          uint32_t n, uint32_t* __restrict__ out_key) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (SHZ_NBINS) * (SHZ_MAX_DT + 1)) return;
  const uint32_t f1 = blockIdx.y, f2 = i / (SHZ_MAX_DT + 1), dt = i % (SHZ_MAX_DT + 1);
  uint32_t h0, h1, h2;
  sha1_of_key(f1, f2, dt, h0, h1, h2);
  const uint64_t p = ((uint64_t)h0 << 32) | h1;
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (tgt_hi[mid] < p) lo = mid + 1; else hi = mid;
  }
  for (; lo < n && tgt_hi[lo] == p; ++lo)
    if (tgt_lo[lo] == (uint16_t)(h2 >> 16)) out_key[lo] = (f1 << 20) | (f2 << 8) | dt;
}

extern "C" int32_t shz_sha1_invert(shz_ctx* ctx, const uint8_t* digests10, uint64_t n, uint32_t* key32_out) {
  if (!ctx) return SHZ_E_INVALID;
  if (n == 0) return SHZ_OK;
  if (!digests10 || !key32_out) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_sha1_invert: NULL buffer");
  if (n >= (1ull << 31)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "too many digests in one call");
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> order(n);
  std::vector<uint64_t> hi(n), shi(n);
  std::vector<uint16_t> lo(n), slo(n);
  for (uint64_t i = 0; i < n; ++i) {
    const uint8_t* d = digests10 + 10 * i;
    uint64_t v = 0;
    for (int b = 0; b < 8; ++b) v = (v << 8) | d[b];
    hi[i] = v;
    lo[i] = (uint16_t)((d[8] << 8) | d[9]);
    order[i] = (uint32_t)i;
  }
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hi[a] < hi[b]; });
  for (uint64_t i = 0; i < n; ++i) { shi[i] = hi[order[i]]; slo[i] = lo[order[i]]; }
  void *dh, *dl, *dk;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M0, n * 8, &dh));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M1, n * 2 + 64, &dl));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_M2, n * 4, &dk));
  SHZ_HIP(ctx, shz_memcpy(ctx, dh, shi.data(), n * 8, hipMemcpyHostToDevice));
  SHZ_HIP(ctx, shz_memcpy(ctx, dl, slo.data(), n * 2, hipMemcpyHostToDevice));
  SHZ_HIP(ctx, hipMemsetAsync(dk, 0xFF, n * 4, ctx->stream));
  const uint32_t per_f1 = SHZ_NBINS * (SHZ_MAX_DT + 1);
  hipLaunchKernelGGL(sha1_invert_kernel, dim3((per_f1 + 255) / 256, SHZ_NBINS), dim3(256), 0, ctx->stream,
                     (const uint64_t*)dh, (const uint16_t*)dl, (uint32_t)n, (uint32_t*)dk);
  SHZ_HIP(ctx, hipGetLastError());
  std::vector<uint32_t> sk(n);
  SHZ_HIP(ctx, shz_memcpy(ctx, sk.data(), dk, n * 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint64_t i = 0; i < n; ++i) key32_out[order[i]] = sk[i];
  return SHZ_OK;
}
