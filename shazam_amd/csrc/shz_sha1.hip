// sha1(f"{f1}|{f2}|{dt}")[:10 bytes] per packed key -- the reference's hash identity
// (__init__.py:207-208) and the BINARY(10) column of mysql_database.py:48.  One thread per key;
// the decimal ASCII message is at most 13 bytes so it is always a single 64-byte SHA-1 block.
#include "shz_internal.h"

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

__device__ __forceinline__ int put_dec(uint8_t* m, int pos, uint32_t v) {
  char tmp[4];
  int n = 0;
  do {
    tmp[n++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) m[pos++] = (uint8_t)tmp[--n];
  return pos;
}

__global__ __launch_bounds__(256) void sha1_prefix_kernel(const uint32_t* __restrict__ key32, uint64_t n,
                                                          uint8_t* __restrict__ out10) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t key = key32[i];
  uint8_t msg[64];
#pragma unroll
  for (int q = 0; q < 64; ++q) msg[q] = 0;
  int len = put_dec(msg, 0, key >> 20);
  msg[len++] = '|';
  len = put_dec(msg, len, (key >> 8) & 0xFFFu);
  msg[len++] = '|';
  len = put_dec(msg, len, key & 0xFFu);
  msg[len] = 0x80;
  msg[62] = (uint8_t)((len * 8) >> 8);
  msg[63] = (uint8_t)(len * 8);
  uint32_t w[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
    w[q] = ((uint32_t)msg[4 * q] << 24) | ((uint32_t)msg[4 * q + 1] << 16) | ((uint32_t)msg[4 * q + 2] << 8) | msg[4 * q + 3];
  uint32_t a = 0x67452301u, b = 0xEFCDAB89u, c = 0x98BADCFEu, d = 0x10325476u, e = 0xC3D2E1F0u;
#pragma unroll
  for (int r = 0; r < 80; ++r) {
    uint32_t wt;
    if (r < 16) wt = w[r];
    else {
      wt = rotl32(w[(r + 13) & 15] ^ w[(r + 8) & 15] ^ w[(r + 2) & 15] ^ w[r & 15], 1);
      w[r & 15] = wt;
    }
    uint32_t f, k;
    if (r < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
    else if (r < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
    else if (r < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
    else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
    const uint32_t tmp = rotl32(a, 5) + f + e + k + wt;
    e = d; d = c; c = rotl32(b, 30); b = a; a = tmp;
  }
  const uint32_t h0 = 0x67452301u + a, h1 = 0xEFCDAB89u + b, h2 = 0x98BADCFEu + c;
  uint8_t* o = out10 + i * 10;
  o[0] = (uint8_t)(h0 >> 24); o[1] = (uint8_t)(h0 >> 16); o[2] = (uint8_t)(h0 >> 8); o[3] = (uint8_t)h0;
  o[4] = (uint8_t)(h1 >> 24); o[5] = (uint8_t)(h1 >> 16); o[6] = (uint8_t)(h1 >> 8); o[7] = (uint8_t)h1;
  o[8] = (uint8_t)(h2 >> 24); o[9] = (uint8_t)(h2 >> 16);
}

extern "C" int32_t shz_sha1_prefix(shz_ctx* ctx, const uint32_t* key32, uint64_t n, uint32_t flags, uint8_t* out10) {
  if (!ctx) return SHZ_E_INVALID;
  if (n == 0) return SHZ_OK;
  if (!key32 || !out10) SHZ_FAIL(ctx, SHZ_E_INVALID, "shz_sha1_prefix: NULL buffer");
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t chunk = 1ull << 26;
  for (uint64_t s = 0; s < n; s += chunk) {
    const uint64_t m = n - s < chunk ? n - s : chunk;
    const uint32_t* d_key = key32 + s;
    void *pk, *po;
    if (!(flags & SHZ_IN_DEVICE)) {
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_KEY, m * 4, &pk));
      SHZ_HIP(ctx, shz_memcpy(ctx, pk, key32 + s, m * 4, hipMemcpyHostToDevice));
      d_key = (const uint32_t*)pk;
    }
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC3, m * 10, &po));
    hipLaunchKernelGGL(sha1_prefix_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, d_key, m,
                       (uint8_t*)po);
    SHZ_HIP(ctx, hipGetLastError());
    SHZ_HIP(ctx, shz_memcpy(ctx, out10 + s * 10, po, m * 10, hipMemcpyDeviceToHost));
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return SHZ_OK;
}
