// Fingerprint extraction on gfx950: PCM -> STFT power (fp64) -> dB -> 21x21 local-max peaks ->
// ordered peak list -> peak-pair keys.  Restates (from the spec in SURVEY.md 8a, not from code)
//   mlab.specgram(x, 4096, Fs, window_hanning, 2048)[0]         __init__.py:232-237, mlab:213-373
//   10*log10 where != 0                                          __init__.py:241
//   get_2D_peaks (21x21 max filter == value, > amp_min)          __init__.py:116-177
//   generate_hashes (time-major order, 4 partners, dt <= 200)    __init__.py:179-210
//
// Data layout in HBM (per sub-batch):
//   pcm      int16, clips concatenated                                        (input)
//   pw       f64 [frame][DB_STRIDE=2056] POWER (0 stored as 1.0), all clips' frames  (stage buffer)
//   mask     u64 [frame][n_slabs][4]  one bit per (frame, bin): peak          (stage buffer)
//   peak_f/t u16/u32 in (clip, t asc, f asc) order                             (stage buffer)
//   key32/t1 u32 in the reference's generation order                           (output)
#include <type_traits>
#include <algorithm>

#include "shz_internal.h"
#include "shz_log10.h"

#define DB_STRIDE 2056  // 2049 bins padded so every row starts 64-byte aligned

// ======================================================================================
// K1: stft_psd.  One workgroup (256 threads) per frame, persistent over frames.
// real 4096-FFT = complex 2048-FFT (Stockham radix 8,8,8,4 through LDS) + split post-pass.
// LDS: 2048 double2 data (XOR-swizzled index against bank conflicts) + twiddle tables (1025 + 16 + 128).
// ======================================================================================
typedef double2 cplx;

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ cplx cmul_negi(cplx a) { return make_double2(a.y, -a.x); }  // a * (-i)
// c ? a : b as four 32-bit selects.  Written on the halves so that the compiler keeps the operands in registers: with
// plain `c ? a : b` on eight live complex values it parks them in scratch and selects by address.
__device__ __forceinline__ double dsel(bool c, double a, double b) {
  const int lo = c ? __double2loint(a) : __double2loint(b);
  const int hi = c ? __double2hiint(a) : __double2hiint(b);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ cplx csel(bool c, cplx a, cplx b) { return make_double2(dsel(c, a.x, b.x), dsel(c, a.y, b.y)); }

__device__ __forceinline__ void dft4(cplx& c0, cplx& c1, cplx& c2, cplx& c3) {
  cplx d0 = cadd(c0, c2), d2 = csub(c0, c2), d1 = cadd(c1, c3), d3 = cmul_negi(csub(c1, c3));
  c0 = cadd(d0, d1);
  c1 = cadd(d2, d3);
  c2 = csub(d0, d1);
  c3 = csub(d2, d3);
}

// forward 8-point DFT, in place, natural output order
__device__ __forceinline__ void dft8(cplx* v) {
  const double s = 0.70710678118654752440;
  cplx b0 = cadd(v[0], v[4]), b4 = csub(v[0], v[4]);
  cplx b1 = cadd(v[1], v[5]), b5 = csub(v[1], v[5]);
  cplx b2 = cadd(v[2], v[6]), b6 = csub(v[2], v[6]);
  cplx b3 = cadd(v[3], v[7]), b7 = csub(v[3], v[7]);
  b5 = make_double2((b5.x + b5.y) * s, (b5.y - b5.x) * s);   // * W8^1
  b6 = cmul_negi(b6);                                        // * W8^2
  b7 = make_double2((b7.y - b7.x) * s, -(b7.x + b7.y) * s);  // * W8^3
  dft4(b0, b1, b2, b3);
  dft4(b4, b5, b6, b7);
  v[0] = b0; v[1] = b4; v[2] = b1; v[3] = b5; v[4] = b2; v[5] = b6; v[6] = b3; v[7] = b7;
}

// LDS index swizzle: conflict-free for the stride-8 writes of pass 1 without padding
#define SWZ(i) ((i) ^ (((i) >> 3) & 7))

struct stft_args {
  const int16_t* pcm;
  const uint64_t* clip_soff;   // [n_clips] first sample of each clip (into pcm)
  const uint64_t* clip_len;    // [n_clips] samples
  const uint32_t* clip_foff;   // [n_clips+1] first frame of each clip (sub-batch numbering)
  uint32_t n_clips;
  uint32_t total_frames;
  uint32_t frames_per_wg;      // 0: persistent workgroups striding over the frames; k: workgroup b takes frames [k b, k b + k)
  void* out;                   // [total_frames][stride] power (f64, stride DB_STRIDE; or f32, stride P32_STRIDE),
                               // exact zeros stored as 1.0 (= 0 dB)
  const double* window;        // [4096]
  const cplx* tw;              // [1025] W4096^k
  double scale;                // 0.25 / (Fs * sum(w^2))
  double wscale;               // sqrt(2 scale): the kernel scales its window registers once, |X|^2 then IS the doubled, scaled power
  uint32_t opt;                // experiment switches (SHZ_STFT_OPT): 1 = no rotation of the special wave, 2 = one frame loop for all waves
  uint32_t hop;                // new samples per frame = NFFT - noverlap (mlab:307-308); 2,048 unless shz_set_overlap says otherwise
  // numpy's arithmetic (stft_np_kernel, peak_verify_kernel): np.hanning(4096), pocketfft's twiddles, 1 / Fs, 1 / sum(w^2)
  const double* np_window;
  const cplx* np_comp;
  double r_fs, r_s;
  uint32_t np_unfused;         // 1: numpy's complex product without FMA (re*re + im*im): a host whose numpy has no FMA3 (shz_set_numpy_product)
};

// The staged spectrogram holds POWER, not dB.  10*log10 is non-decreasing, so the window maximum of the dB values
// is the dB value of the window's power maximum M -- but it is NOT injective in fp64 (8-37 adjacent doubles of P share
// one dB value), and the reference tests equality on dB (`maximum_filter(arr2D) == arr2D`, __init__.py:143, after the
// log at :241).  So peak picking keeps every cell close to M as a candidate and decides candidates with P != M
// by the exact comparison 10*log10(P) == 10*log10(M) (shz_log10.h); the log is evaluated nowhere else on the hot
// path (the `> amp_min` test at :161 needs it only within 1e-9 of the threshold).  Exact-zero power maps to 1.0
// because the reference maps it to 0 dB (__init__.py:241), which keeps its rank against sub-unity cells.
//
// Default staging is fp32 (P32_STRIDE floats per frame): rounding to fp32 is monotone, so a cell that alone holds
// the fp32 maximum of its window holds the fp64 maximum too, and a cell more than one fp32 step below it cannot tie
// with it in dB.  What fp32 cannot decide -- cells that share the top two fp32 steps of their window, cells within
// 1e-7 of the amp_min threshold -- goes to peak_verify_kernel, which recomputes the fp64 values of just those cells
// with the arithmetic of this kernel (stft_frame below is the one definition of it).

#define TW_MAIN 513                        // W4096^k, k in [0, 512]
#define TW_P2 (7 * 8)                      // pass 2: [t-1][k] = W_64^(k t)
#define TW_P3 (7 * 64)                     // pass 3: [t-1][k] = W_512^(k t)
#define TW_ALL (TW_MAIN + TW_P2 + TW_P3)   // entries of ctx->d_twiddle (shz_ctx.hip)
#define LDS_CPLX (2048 + TW_ALL)

// twiddle tables into LDS
__device__ __forceinline__ void stft_tables(cplx* lds, const cplx* gtw, int j, int nthreads) {
  for (int i = j; i < TW_ALL; i += nthreads) lds[2048 + i] = gtw[i];
}
// W^(1024 - k) from W^k = (c, s):  W^1024 conj(W^k) = -i (c, -s) = (-s, -c)
__device__ __forceinline__ cplx tw_mirror(cplx w) { return make_double2(-w.y, -w.x); }

// forward 8-point DFT, in place, natural output order.  The factor sqrt(1/2) of the rotations W8^1, W8^3 is not applied
// to the rotated terms but folded into the additions that consume them (fma), four multiplications less.
__device__ __forceinline__ void dft8f(cplx* v) {
  const double s = 0.70710678118654752440;
  cplx b0 = cadd(v[0], v[4]), b4 = csub(v[0], v[4]);
  cplx b1 = cadd(v[1], v[5]), b5 = csub(v[1], v[5]);
  cplx b2 = cadd(v[2], v[6]), b6 = csub(v[2], v[6]);
  cplx b3 = cadd(v[3], v[7]), b7 = csub(v[3], v[7]);
  dft4(b0, b1, b2, b3);
  // odd half: p5 = sqrt2 b5 W8^1, p7 = sqrt2 b7 W8^3 (unscaled), b6 W8^2 = -i b6
  const cplx p5 = make_double2(b5.x + b5.y, b5.y - b5.x);
  const cplx p7 = make_double2(b7.y - b7.x, -(b7.x + b7.y));
  b6 = cmul_negi(b6);
  const cplx d0 = cadd(b4, b6), d2 = csub(b4, b6);
  const cplx q1 = cadd(p5, p7), q3 = cmul_negi(csub(p5, p7));
  const cplx c0 = make_double2(fma(s, q1.x, d0.x), fma(s, q1.y, d0.y));
  const cplx c2 = make_double2(fma(-s, q1.x, d0.x), fma(-s, q1.y, d0.y));
  const cplx c1 = make_double2(fma(s, q3.x, d2.x), fma(s, q3.y, d2.y));
  const cplx c3 = make_double2(fma(-s, q3.x, d2.x), fma(-s, q3.y, d2.y));
  v[0] = b0; v[1] = c0; v[2] = b1; v[3] = c1; v[4] = b2; v[5] = c2; v[6] = b3; v[7] = c3;
}

// pass 1 takes its inputs straight from the PCM and the window: v[t] = (x[2n] w[2n], x[2n+1] w[2n+1]), n = j + 256 t, and the
// first stage v[t] +- v[t+4] is one product and two fused steps a component instead of two products, a sum and a difference
// (STFT_FUSED, below)
__device__ __forceinline__ void dft8f_tail(cplx b0, cplx b1, cplx b2, cplx b3, cplx b4, cplx b5, cplx b6, cplx b7, cplx* v) {
  const double s = 0.70710678118654752440;
  dft4(b0, b1, b2, b3);
  const cplx p5 = make_double2(b5.x + b5.y, b5.y - b5.x);
  const cplx p7 = make_double2(b7.y - b7.x, -(b7.x + b7.y));
  b6 = cmul_negi(b6);
  const cplx d0 = cadd(b4, b6), d2 = csub(b4, b6);
  const cplx q1 = cadd(p5, p7), q3 = cmul_negi(csub(p5, p7));
  const cplx c0 = make_double2(fma(s, q1.x, d0.x), fma(s, q1.y, d0.y));
  const cplx c2 = make_double2(fma(-s, q1.x, d0.x), fma(-s, q1.y, d0.y));
  const cplx c1 = make_double2(fma(s, q3.x, d2.x), fma(s, q3.y, d2.y));
  const cplx c3 = make_double2(fma(-s, q3.x, d2.x), fma(-s, q3.y, d2.y));
  v[0] = b0; v[1] = c0; v[2] = b1; v[3] = c1; v[4] = b2; v[5] = c2; v[6] = b3; v[7] = c3;
}

// Twiddle products FUSED into the first additions behind them (round 4).  A butterfly whose inputs carry twiddles first
// forms a + w b and a - w b; written as product-then-sum that is 4 + 2 + 2 operations, written as
//   r = fma(-w.y, b.y, fma(w.x, b.x, a.x)) | fma(w.y, b.x, fma(w.x, b.y, a.y))  and  s = 2 a - r   (one fma a component)
// it is 6: two fewer per twiddled pair -- 32 of the 373 fp64 operations of a frame (passes 2 and 3: the four pairs of each
// radix-8 butterfly, pass 4: two per radix-4 butterfly, the post-pass: one per bin pair).  The values differ from the
// unfused form in the last bits, as any two correct transforms do; what decides ties is the reference's own arithmetic
// (np_fft4096), not this kernel's.  STFT_FUSED=0 builds the unfused form (A/B).
#ifndef STFT_FUSED
#define STFT_FUSED 1
#endif
#ifndef STFT_TW3_DERIVE
#define STFT_TW3_DERIVE 0   // 1: four of pass 3's seven twiddles as products of the other three (4 LDS reads less, 16 operations more): 3.806 -> 3.827 ms, off
#endif
__device__ __forceinline__ void cmul_pm(cplx a, cplx w, cplx b, cplx& r, cplx& s) {   // r = a + w b, s = a - w b
#if STFT_FUSED
  r = make_double2(fma(-w.y, b.y, fma(w.x, b.x, a.x)), fma(w.y, b.x, fma(w.x, b.y, a.y)));
  s = make_double2(fma(2.0, a.x, -r.x), fma(2.0, a.y, -r.y));
#else
  const cplx t = cmul(b, w);
  r = cadd(a, t);
  s = csub(a, t);
#endif
}
// dft8f of v[0], w[0] v[1], ..., w[6] v[7]
__device__ __forceinline__ void dft8f_tw(cplx* v, const cplx (&w)[7]) {
  const double s = 0.70710678118654752440;
  cplx b0, b1, b2, b3, b4, b5, b6, b7;
  cmul_pm(v[0], w[3], v[4], b0, b4);
  cmul_pm(cmul(v[1], w[0]), w[4], v[5], b1, b5);
  cmul_pm(cmul(v[2], w[1]), w[5], v[6], b2, b6);
  cmul_pm(cmul(v[3], w[2]), w[6], v[7], b3, b7);
  dft4(b0, b1, b2, b3);
  const cplx p5 = make_double2(b5.x + b5.y, b5.y - b5.x);
  const cplx p7 = make_double2(b7.y - b7.x, -(b7.x + b7.y));
  b6 = cmul_negi(b6);
  const cplx d0 = cadd(b4, b6), d2 = csub(b4, b6);
  const cplx q1 = cadd(p5, p7), q3 = cmul_negi(csub(p5, p7));
  const cplx c0 = make_double2(fma(s, q1.x, d0.x), fma(s, q1.y, d0.y));
  const cplx c2 = make_double2(fma(-s, q1.x, d0.x), fma(-s, q1.y, d0.y));
  const cplx c1 = make_double2(fma(s, q3.x, d2.x), fma(s, q3.y, d2.y));
  const cplx c3 = make_double2(fma(-s, q3.x, d2.x), fma(-s, q3.y, d2.y));
  v[0] = b0; v[1] = c0; v[2] = b1; v[3] = c1; v[4] = b2; v[5] = c2; v[6] = b3; v[7] = c3;
}
// dft4 of c0, w1 c1, w2 c2, w3 c3
__device__ __forceinline__ void dft4_tw(cplx& c0, cplx& c1, cplx& c2, cplx& c3, cplx w1, cplx w2, cplx w3) {
  cplx d0, d2, d1, d3;
  cmul_pm(c0, w2, c2, d0, d2);
  cmul_pm(cmul(c1, w1), w3, c3, d1, d3);
  d3 = cmul_negi(d3);
  c0 = cadd(d0, d1);
  c1 = cadd(d2, d3);
  c2 = csub(d0, d1);
  c3 = csub(d2, d3);
}

// dft8f of the windowed samples of one thread: pw[t] = packed (x[2n], x[2n+1]), ww[t] = (w[2n], w[2n+1])
__device__ __forceinline__ void dft8f_win(const int (&pw)[8], const double2 (&ww)[8], cplx* v) {
  cplx b[8];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const double x0 = (double)(short)(pw[t] & 0xFFFF), y0 = (double)(pw[t] >> 16);
    const double x4 = (double)(short)(pw[t + 4] & 0xFFFF), y4 = (double)(pw[t + 4] >> 16);
#if STFT_FUSED
    const double px = x0 * ww[t].x, py = y0 * ww[t].y;
    b[t] = make_double2(fma(x4, ww[t + 4].x, px), fma(y4, ww[t + 4].y, py));
    b[t + 4] = make_double2(fma(-x4, ww[t + 4].x, px), fma(-y4, ww[t + 4].y, py));
#else
    const cplx v0 = make_double2(x0 * ww[t].x, y0 * ww[t].y), v4 = make_double2(x4 * ww[t + 4].x, y4 * ww[t + 4].y);
    b[t] = cadd(v0, v4);
    b[t + 4] = csub(v0, v4);
#endif
  }
  dft8f_tail(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], v);
}

// PCM of frame `g` (sub-batch numbering) of clip `lo` as 8 packed sample pairs per thread
// (lo16 = x[2n], hi16 = x[2n+1], n = j + 256 t)
__device__ __forceinline__ void stft_load_frame(const stft_args& a, uint32_t lo, uint32_t g, int j, int (&pw)[8]) {
  const uint64_t clen = a.clip_len[lo];
  const uint64_t s_in_clip = (uint64_t)(g - a.clip_foff[lo]) * a.hop;
  const int16_t* src = a.pcm + a.clip_soff[lo] + s_in_clip;
  const uint64_t avail = clen > s_in_clip ? clen - s_in_clip : 0;  // samples readable from src
  if (avail >= SHZ_NFFT && (((uintptr_t)src) & 3) == 0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) pw[t] = reinterpret_cast<const int*>(src)[j + 256 * t];
  } else if (avail >= SHZ_NFFT) {  // clip starts at an odd sample of the packed PCM buffer
    int x0[8], x1[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      x0[t] = (int)src[2 * (j + 256 * t)];
      x1[t] = (int)src[2 * (j + 256 * t) + 1];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) pw[t] = (x0[t] & 0xFFFF) | (x1[t] << 16);
  } else {  // zero padding of inputs shorter than one window (mlab:268-271)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const uint64_t n0 = 2 * (uint64_t)(j + 256 * t);
      const int x0 = n0 < avail ? (int)src[n0] : 0;
      const int x1 = n0 + 1 < avail ? (int)src[n0 + 1] : 0;
      pw[t] = (x0 & 0xFFFF) | (x1 << 16);
    }
  }
}

// One frame: windowed samples v[8] (complex point j + 256 t = samples 2n, 2n+1) -> power of the 2049 bins.
// `before_out()` runs between the last butterflies and the outputs (the persistent kernel issues the next frame's
// loads there); `out(k, p)` receives every bin k once with its scaled power p (exact zero already mapped to 1.0).
// Ends with a barrier: buf may be rewritten on return.
// value staged for a bin: the power itself, an exact zero as 1.0 (see the comment above stft_args).  The test reads
// the bits (p >= 0: zero <=> no bit set), two 32-bit operations instead of an fp64 compare and a 64-bit select.
template <typename T>
__device__ __forceinline__ T stage_value(double p) {
  return ((__double2hiint(p) | __double2loint(p)) != 0) ? (T)p : (T)1.0;
}
// fp32 staging tests the CONVERTED value (one 32-bit compare): a power so small that it rounds to 0.0f (< 1e-45, -450 dB)
// is staged as 1.0 like an exact zero.  Either way the cell is below 0 dB: it cannot be a peak for amp_min >= 0 (negative
// amp_min takes the fp64 pass), and it cannot outrank a cell that can.
template <>
__device__ __forceinline__ float stage_value<float>(double p) {
  const float f = (float)p;
  return f == 0.0f ? 1.0f : f;
}

// One frame: windowed samples v[8] (complex point j + 256 t = samples 2n, 2n+1) -> power of the 2049 bins.
// `before_out()` runs between the last butterflies and the outputs (the persistent kernel issues the next frame's
// loads there); `out(k, p)` receives every bin k once with its scaled power p (>= 0; stage_value maps it).
// On return buf may be rewritten (the barrier that says so sits right behind the last LDS reads).
//
// The passes are those of the Stockham formulation (pass p: inputs [j + 256 t], twiddles W_{8 Ns}^{k t}, outputs
// [(j div Ns) 8 Ns + k + Ns r]) -- same operations on the same values in the same order, so every bit of the output
// is what it was -- but the ARRAYS BETWEEN THE PASSES LIVE IN PLACE: a butterfly writes its eight results into the eight
// LDS slots it read its inputs from.  Nobody else reads or writes those slots during the pass, so the barrier "everybody
// has read" between a pass's reads and its writes is gone; and with the butterflies of passes 2 and 3 numbered so that a
// wave's pass-3 inputs are exactly the slots that wave wrote in pass 2, the barrier between those two passes is gone as
// well (the LDS executes one wave's instructions in order).  Three workgroup barriers per frame instead of seven.
//   slot of A1[i] (after pass 1) = i;   slot of A2[i] = i[2:0] | i[10:6] << 3 | i[5:3] << 8;
//   slot of A3[i] = i[2:0] | i[10:9] << 3 | i[8:6] << 5 | i[5:3] << 8          (i[a:b] = bits a..b of i)
// Bank swizzle SZ: among 16 neighbouring lanes the slots vary in bits {3,4,5,6} (pass 1 stores), {0,1,2,5} (pass 2),
// {0,1,2,8} (passes 3 and 4); XOR-ing bit 4 -> 1, bit 5 -> 3 and 0, bit 6 -> 2, bit 8 -> 3 makes each of those sets hit 16
// different 16-byte bank groups.
__device__ __forceinline__ int stft_sz(int s) {
  return s ^ (((s >> 4) & 1) << 1) ^ (((s >> 5) & 1) * 9) ^ (((s >> 6) & 1) << 2) ^ (((s >> 8) & 1) << 3);
}
// LDS accesses are written on BYTE offsets (slot << 4, the constants of a pass scaled alike): with element indices the
// compiler shifted every one of the 66 addresses of a frame by four again (and the kernel is bound by VALU issue).
__device__ __forceinline__ int stft_szb(int s) { return stft_sz(s) << 4; }
__device__ __forceinline__ cplx& lds_at(cplx* buf, int byte_off) {
  return *reinterpret_cast<cplx*>(reinterpret_cast<char*>(buf) + byte_off);
}
__device__ __forceinline__ const cplx& lds_at(const cplx* buf, int byte_off) {
  return *reinterpret_cast<const cplx*>(reinterpret_cast<const char*>(buf) + byte_off);
}
// ---- the pieces of a frame (stft_frame = one frame; stft_psd2_kernel interleaves the pieces of two) ----
struct stft_tabs { const cplx *tw, *tw2, *tw3; };
__device__ __forceinline__ stft_tabs stft_tabs_at(const cplx* tw) { return stft_tabs{tw, tw + TW_MAIN, tw + TW_MAIN + TW_P2}; }

// pass 1: Ns = 1 (no twiddles); A1[8j + r] at slot 8j + r
__device__ __forceinline__ void stft_p1(const int (&pw)[8], const double2 (&ww)[8], cplx (&v)[8], cplx* buf, int j) {
  dft8f_win(pw, ww, v);
  const int a = stft_szb(8 * j);   // the low three slot bits hold swizzle terms only
#pragma unroll
  for (int r = 0; r < 8; ++r) lds_at(buf, a ^ (r << 4)) = v[r];
}
// pass 2: Ns = 8, twiddles W_64^(k t) from the table.  Butterfly j2 = L[2:0] | W << 3 | L[5:3] << 5 (a wave takes the
// butterflies whose bits 3,4 spell its number): inputs A1[j2 + 256 t] at slots j2 + 256 t, results written back there.
__device__ __forceinline__ void stft_p2_load(cplx (&v)[8], const cplx* buf, int j) {
  const int W = j >> 6, L = j & 63;
  const int j2 = (L & 7) | (W << 3) | ((L >> 3) << 5);
  const int a0 = stft_szb(j2), a1 = stft_szb(j2 | 256);   // t even / odd (slot bit 8 enters the swizzle)
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = lds_at(buf, ((t & 1) ? a1 : a0) + (t >> 1) * (512 << 4));
}
__device__ __forceinline__ void stft_p2_rest(cplx (&v)[8], cplx* buf, const stft_tabs& T, int j) {
  const int W = j >> 6, L = j & 63;
  const int j2 = (L & 7) | (W << 3) | ((L >> 3) << 5);
  const int a0 = stft_szb(j2), a1 = stft_szb(j2 | 256);
  const int k = L & 7;
  cplx w[7];
#pragma unroll
  for (int t = 1; t < 8; ++t) w[t - 1] = T.tw2[(t - 1) * 8 + k];
  dft8f_tw(v, w);
#pragma unroll
  for (int r = 0; r < 8; ++r) lds_at(buf, ((r & 1) ? a1 : a0) + (r >> 1) * (512 << 4)) = v[r];
}
// pass 3: Ns = 64, twiddles W_512^(k t).  Butterfly j: inputs A2[j + 256 t] at slots j[2:0] | W << 3 | t << 5 | j[5:3] << 8
// -- all written by this wave in pass 2: no barrier in front.  Slot bits 5 and 6 (t & 1, t & 2) enter the swizzle, bit 7
// (t & 4) is a plain offset of 128 elements.
__device__ __forceinline__ void stft_p3_load(cplx (&v)[8], const cplx* buf, int j) {
  const int b3 = stft_szb((j & 7) | ((j >> 6) << 3) | (((j >> 3) & 7) << 8));
  const int q0 = b3, q1 = (b3 ^ (9 << 4)) + (32 << 4), q2 = (b3 ^ (4 << 4)) + (64 << 4), q3 = (b3 ^ (13 << 4)) + (96 << 4);
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = lds_at(buf, ((t & 3) == 0 ? q0 : (t & 3) == 1 ? q1 : (t & 3) == 2 ? q2 : q3) + (t >> 2) * (128 << 4));
}
__device__ __forceinline__ void stft_p3_rest(cplx (&v)[8], cplx* buf, const stft_tabs& T, int j) {
  const int b3 = stft_szb((j & 7) | ((j >> 6) << 3) | (((j >> 3) & 7) << 8));
  const int q0 = b3, q1 = (b3 ^ (9 << 4)) + (32 << 4), q2 = (b3 ^ (4 << 4)) + (64 << 4), q3 = (b3 ^ (13 << 4)) + (96 << 4);
  const int k = j & 63;
  cplx w[7];
#if STFT_TW3_DERIVE
  // three of the seven twiddles from the table, the others as their products (four LDS reads less, sixteen operations more)
  w[0] = T.tw3[k]; w[1] = T.tw3[64 + k]; w[3] = T.tw3[3 * 64 + k];
  w[2] = cmul(w[0], w[1]); w[4] = cmul(w[0], w[3]); w[5] = cmul(w[1], w[3]); w[6] = cmul(w[2], w[3]);
#else
#pragma unroll
  for (int t = 1; t < 8; ++t) w[t - 1] = T.tw3[(t - 1) * 64 + k];
#endif
  dft8f_tw(v, w);
#pragma unroll
  for (int r = 0; r < 8; ++r) lds_at(buf, ((r & 3) == 0 ? q0 : (r & 3) == 1 ? q1 : (r & 3) == 2 ? q2 : q3) + (r >> 2) * (128 << 4)) = v[r];
}
// pass 4 + split post-pass, no trip through LDS between them.  Pass 4: Ns = 512, radix 4, twiddles
// W_2048^(b t) = W4096^(2 b t); butterfly b yields Z[b + 512 c], c = 0..3.  The post-pass pairs Z[k] with
// Z[2048 - k]: X[k] = E + W^k O, X[2048-k] = conj(E - W^k O).  2048 - (b + 512 c) = (512 - b) + 512 (3 - c), so a
// thread that computes the butterflies b = j and 512 - j holds both members of its four pairs (thread 0: b = 0 and
// b = 256, which pair with themselves) -- 32 KB of LDS stores, the reads behind them and two barriers less per frame.
// Inputs: A3[b + 512 c] at slot b[2:0] | c << 3 | b[8:6] << 5 | b[5:3] << 8 (c = 1 sets slot bit 3, c = 2 bit 4, which
// flips swizzle bit 1); v[0..3] = butterfly j, v[4..7] = butterfly 512 - j (thread 0: 256).
__device__ __forceinline__ void stft_p4_load(cplx (&v)[8], const cplx* buf, int j) {
  const int bb = j == 0 ? 256 : 512 - j;
  auto slot4 = [](int b) { return stft_szb((b & 7) | (((b >> 6) & 7) << 5) | (((b >> 3) & 7) << 8)); };
  const int sa = slot4(j), sb = slot4(bb);
  v[0] = lds_at(buf, sa); v[1] = lds_at(buf, sa ^ (8 << 4)); v[2] = lds_at(buf, sa ^ (18 << 4)); v[3] = lds_at(buf, sa ^ (26 << 4));
  v[4] = lds_at(buf, sb); v[5] = lds_at(buf, sb ^ (8 << 4)); v[6] = lds_at(buf, sb ^ (18 << 4)); v[7] = lds_at(buf, sb ^ (26 << 4));
}
// Twiddles: butterfly 512 - j uses W^(1024 - 2j) and its powers, mirrors of butterfly j's (tw_mirror): W1' = -i conj(W1),
// W2' = -conj(W2), W3' = i conj(W3); the post-pass needs W^j, W^(512-j) and their mirrors W^(1024-j), W^(512+j).
// Thread 0's second butterfly is b = 256, not 512: its twiddles are constants, selected below.
// SPECIAL = false: the caller knows that no lane of the wave is thread 0 (waves 1-3): the selects fold away.
template <bool SPECIAL, class PRE, class OUT>
__device__ __forceinline__ void stft_p4_rest(cplx (&v)[8], const stft_tabs& T, int j, PRE&& before_out, OUT&& out) {
  const cplx* tw = T.tw;
  const bool t0 = SPECIAL && j == 0;
  const int bb = t0 ? 256 : 512 - j;
  cplx A0 = v[0], A1 = v[1], A2 = v[2], A3 = v[3], B0 = v[4], B1 = v[5], B2 = v[6], B3 = v[7];
  const double h = 0.70710678118654752440;
  {
    const cplx w1 = tw[2 * j];
    const cplx w2 = cmul(w1, w1), w3 = cmul(w1, w2);
    // b = 256: W^512 = (h, -h), W^1024 = (0, -1), W^1536 = (-h, -h)
    const cplx v1 = csel(t0, make_double2(h, -h), tw_mirror(w1));
    const cplx v2 = csel(t0, make_double2(0.0, -1.0), make_double2(-w2.x, w2.y));
    const cplx v3 = csel(t0, make_double2(-h, -h), make_double2(w3.y, w3.x));
    dft4_tw(A0, A1, A2, A3, w1, w2, w3);
    dft4_tw(B0, B1, B2, B3, v1, v2, v3);
  }
  before_out();
  // the window carries sqrt(2 scale) (stft_args::wscale): |X|^2 is the power of bins 1..2047, doubled as mlab doubles them
  // (mlab:339-345); bins 0 and 2048 -- the pair k = 0, thread 0's -- are not doubled: half of it
  // bins k and 2048 - k from zk = Z[k], zm = Z[2048 - k], k in [0, 1024], wk = W^k
  auto pair_out = [&](int k, cplx wk, cplx zk, cplx zm) {
    const cplx e = make_double2(zk.x + zm.x, zk.y - zm.y);
    const cplx o = make_double2(zk.y + zm.y, zm.x - zk.x);
    cplx xa, xb;
    cmul_pm(e, wk, o, xa, xb);
    double pa = fma(xa.x, xa.x, xa.y * xa.y);
    double pb = fma(xb.x, xb.x, xb.y * xb.y);
    if (SPECIAL && k == 0) { pa *= 0.5; pb *= 0.5; }
    out(k, pa);
    if (k != 1024) out(2048 - k, pb);
  };
  // thread j >= 1: (j, 2048 - j), (j + 512, 1536 - j), (512 - j, 1536 + j), (1024 - j, 1024 + j);
  // thread 0 (butterflies 0 and 256): (0, 2048), (512, 1536), (256, 1792), (768, 1280) and bin 1024 alone
  const cplx wj = tw[j], wa = tw[bb];                                  // W^j, W^(512-j)   [thread 0: W^0, W^256]
  const cplx wma = tw_mirror(wa);                                      // W^(512+j)        [thread 0: W^768]
  const cplx w3rd = csel(t0, make_double2(h, -h), wma);                // k = j + 512      [thread 0: W^512]
  const cplx w4th = csel(t0, wma, tw_mirror(wj));                      // k = 1024 - j     [thread 0: W^768]
  pair_out(bb, wa, B0, csel(t0, B3, A3));   // ordered so that each pair frees its operands early
  pair_out(j, wj, A0, csel(t0, A0, B3));
  pair_out(j + 512, w3rd, A1, csel(t0, A3, B2));
  pair_out(t0 ? 768 : 1024 - j, w4th, B1, csel(t0, B2, A2));
  if (t0) pair_out(1024, make_double2(0.0, -1.0), A2, A2);
}

// SPECIAL = false: the caller knows that its wave does not hold thread 0 (see stft_psd_kernel)
template <bool SPECIAL = true, class PRE, class OUT>
__device__ __forceinline__ void stft_frame(const int (&pw)[8], const double2 (&ww)[8], cplx* lds, int j, PRE&& before_out, OUT&& out) {
  cplx* buf = lds;
  const stft_tabs T = stft_tabs_at(lds + 2048);
  cplx v[8];
  stft_p1(pw, ww, v, buf, j);
  __syncthreads();
  stft_p2_load(v, buf, j);
  stft_p2_rest(v, buf, T, j);
  // (no barrier: the slots this wave reads next are the ones it has just written)
  stft_p3_load(v, buf, j);
  stft_p3_rest(v, buf, T, j);
  __syncthreads();
  stft_p4_load(v, buf, j);
  __syncthreads();  // everybody holds its inputs: buf may be rewritten by the next frame's pass 1
  // thread 0's butterflies pair differently: the selects that say so cost 36 instructions per frame
  // (a wave-uniform branch around this call alone was tried: everything live in the frame crosses the split, 76 spilled
  // registers against 16 -- the kernel branches once, outside the frame loop, instead)
  stft_p4_rest<SPECIAL>(v, T, j, before_out, out);
}

// ---------------------------------------------------------------------------------------------------------------
// THE REFERENCE'S OWN ARITHMETIC, operation by operation (round 4).  Where a decision hangs on the last bits of a power --
// fp64 staging (clips fp32 cannot decide, whole passes with amp_min < 0, shz_stft_db) and the cells peak_verify_kernel
// recomputes -- the power is computed the way the call of __init__.py:232-237 computes it on a host with numpy 2.x:
//   result = x[frame] * np.hanning(4096)                  one product a sample
//   np.fft.fft(result)                                    pocketfft's COMPLEX transform of the real frame: for 4096 = 8^4
//                                                         four radix-8 passes (its `pass8`), twiddles from `sincos_2pibyn`,
//                                                         no fused multiply-add anywhere
//   np.conj(result) * result                              real part = fma(re, re, im * im): numpy's complex product on a
//                                                         host with FMA3 (x86-64 AVX2 / AVX-512: `npyv_muladdsub`)
//   result[1:-1] *= 2; result /= Fs; result /= (w**2).sum()   the two divisions are complex / real: numpy multiplies by
//                                                         the rounded reciprocals 1.0 / Fs and 1.0 / sum
// numpy and pocketfft are third-party code that is not in /root/reference; the passes below restate the
// published algorithm and are pinned by digests of the reference's own spectrograms (tests/golden/psd_digests.json, all
// 1.3 M values of a 30 s clip bit for bit) and by the click-train fixture, whose 1,519 peaks hang on exactly these bits.
// The dataflow of a pass is pocketfft's (butterfly (k, i) of pass p reads cc[i + ido (b + 8 k)], b = 0..7, and writes
// ch[i + ido (k + l1 c)], c = 0..7, times comp[c l1 i] for i > 0); the array lives in ONE LDS buffer (all reads of a pass,
// barrier, all writes).  Built for correctness: ~2x the time of stft_psd_kernel, and only ever run on what fp32 left open.
__device__ __forceinline__ cplx np_smul(cplx v, cplx w) {   // special_mul<fwd>: v * conj(w), products and sums rounded one by one
  return make_double2(v.x * w.x + v.y * w.y, v.y * w.x - v.x * w.y);
}
__device__ __forceinline__ void np_bfly8(const cplx (&c)[8], cplx (&o)[8]) {
  const double hsqt2 = 0.707106781186547524400844362104849;
  cplx a0, a1, a2, a3, a4, a5, a6, a7;
  a1 = cadd(c[1], c[5]); a5 = csub(c[1], c[5]);                       // PM(a1, a5, c1, c5)
  a3 = cadd(c[3], c[7]); a7 = csub(c[3], c[7]);                       // PM(a3, a7, c3, c7)
  { const cplx t = a1; a1 = cadd(a1, a3); a3 = csub(t, a3); }         // PMINPLACE(a1, a3)
  a3 = make_double2(a3.y, -a3.x);                                     // ROTX90<fwd>(a3)
  a7 = make_double2(a7.y, -a7.x);                                     // ROTX90<fwd>(a7)
  { const cplx t = a5; a5 = cadd(a5, a7); a7 = csub(t, a7); }         // PMINPLACE(a5, a7)
  a5 = make_double2(hsqt2 * (a5.x + a5.y), hsqt2 * (a5.y - a5.x));    // ROTX45<fwd>(a5)
  a7 = make_double2(hsqt2 * (a7.y - a7.x), hsqt2 * (-a7.x - a7.y));   // ROTX135<fwd>(a7)
  a0 = cadd(c[0], c[4]); a4 = csub(c[0], c[4]);                       // PM(a0, a4, c0, c4)
  a2 = cadd(c[2], c[6]); a6 = csub(c[2], c[6]);                       // PM(a2, a6, c2, c6)
  { const cplx t = a0; a0 = cadd(a0, a2); a2 = csub(t, a2); }         // PMINPLACE(a0, a2)
  o[0] = cadd(a0, a1); o[4] = csub(a0, a1);
  o[2] = cadd(a2, a3); o[6] = csub(a2, a3);
  a6 = make_double2(a6.y, -a6.x);                                     // ROTX90<fwd>(a6)
  { const cplx t = a4; a4 = cadd(a4, a6); a6 = csub(t, a6); }         // PMINPLACE(a4, a6)
  o[1] = cadd(a4, a5); o[5] = csub(a4, a5);
  o[3] = cadd(a6, a7); o[7] = csub(a6, a7);
}
// buf: the 4096 complex inputs in natural order; 256 threads; the caller has put a barrier behind its stores.  Returns with
// the transform in buf, natural order, behind a barrier.
// tw: the twiddles pass by pass, [c - 1][i] = comp[c l1 i] (pocketfft's per-pass tables; shz_ctx_create copies them there)
__device__ __forceinline__ void np_fft4096(cplx* buf, const cplx* __restrict__ tw, int j) {
#pragma unroll 1
  for (int p = 0; p < 4; ++p) {
    const int ls = 3 * p, is = 9 - 3 * p;             // l1 = 8^p butterflies groups, ido = 512 / 8^p
    const int ido = 1 << is;
    const cplx* __restrict__ twp = tw + 7 * (p == 0 ? 0 : p == 1 ? 512 : p == 2 ? 576 : 584);   // behind the tables of the passes in front (7 ido entries each)
    cplx o[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int bf = j + 256 * u, k = bf >> is, i = bf & (ido - 1);
      cplx c[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) c[b] = buf[i + ((b + 8 * k) << is)];
      np_bfly8(c, o[u]);
      if (i != 0) {
#pragma unroll
        for (int cc = 1; cc < 8; ++cc) o[u][cc] = np_smul(o[u][cc], twp[((cc - 1) << is) + i]);
      }
    }
    __syncthreads();   // every butterfly holds its inputs
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int bf = j + 256 * u, k = bf >> is, i = bf & (ido - 1);
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) buf[i + ((k + (cc << ls)) << is)] = o[u][cc];
    }
    __syncthreads();
  }
}
// power of bin k from its transform value, scaled as mlab scales it
__device__ __forceinline__ double np_power(cplx X, int k, double r_fs, double r_s, uint32_t unfused = 0) {
  double p = unfused ? X.x * X.x + X.y * X.y : fma(X.x, X.x, X.y * X.y);
  if (k != 0 && k != SHZ_NFFT / 2) p *= 2.0;
  p = p * r_fs;
  return p * r_s;
}
// windowed samples of a frame into buf (real parts; thread j holds samples 2n, 2n + 1, n = j + 256 t)
__device__ __forceinline__ void np_stage_frame(cplx* buf, const int (&pw)[8], const double2 (&ww)[8], int j) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int n = j + 256 * t;
    *reinterpret_cast<double4*>(&buf[2 * n]) = make_double4((double)(short)(pw[t] & 0xFFFF) * ww[t].x, 0.0, (double)(pw[t] >> 16) * ww[t].y, 0.0);
  }
}

#define NP_FRAMES_PER_WG 8
__global__ __launch_bounds__(256) void stft_np_kernel(stft_args a) {
  __shared__ cplx buf[SHZ_NFFT];
  const int j = threadIdx.x;
  double2 ww[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) ww[t] = *reinterpret_cast<const double2*>(a.np_window + 2 * (j + 256 * t));
  const uint32_t g0 = blockIdx.x * a.frames_per_wg, gend = min(g0 + a.frames_per_wg, a.total_frames);
  uint32_t lo = 0;
  for (uint32_t g = g0; g < gend; ++g) {
    if (g == g0) {
      uint32_t hi = a.n_clips;
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.clip_foff[mid] <= g) lo = mid; else hi = mid;
      }
    } else {
      while (lo + 1 < a.n_clips && a.clip_foff[lo + 1] <= g) ++lo;
    }
    int pw[8];
    stft_load_frame(a, lo, g, j, pw);
    np_stage_frame(buf, pw, ww, j);
    __syncthreads();
    np_fft4096(buf, a.np_comp, j);
    double* orow = reinterpret_cast<double*>(a.out) + (uint64_t)g * DB_STRIDE;
#pragma unroll
    for (int t = 0; t <= 8; ++t) {
      const int k = j + 256 * t;
      if (k <= SHZ_NFFT / 2) orow[k] = stage_value<double>(np_power(buf[k], k, a.r_fs, a.r_s, a.np_unfused));
    }
    __syncthreads();   // buf is rewritten by the next frame
  }
}

#define P32_STRIDE 2064  // floats per fp32 row: 2049 bins padded so every row starts 64-byte aligned
#ifndef STFT_OCC
#define STFT_OCC 3
#endif

template <typename T>
__global__ __launch_bounds__(256, STFT_OCC) void stft_psd_kernel(stft_args a) {
  __shared__ cplx lds[LDS_CPLX];
  // logical thread number: the wave that plays "wave 0" (thread 0's extra selects, see the end of this kernel) differs from
  // workgroup to workgroup, so that the three workgroups of a CU do not put their heavier wave on the same SIMD
  const int j = (int)threadIdx.x ^ (int)((a.opt & 1u) ? 0u : (blockIdx.x & 3u) << 6);
  stft_tables(lds, a.tw, j, 256);
  __syncthreads();
  constexpr uint32_t STRIDE = sizeof(T) == 8 ? DB_STRIDE : P32_STRIDE;

  // the Hann window values this thread multiplies with are the same for every frame: registers.  (Re-reading them per
  // frame -- 32 KB per workgroup from L2 -- frees 32 registers, 8 spilled instead of 24, and costs 4.73 ms against 4.38:
  // the loads sit at the head of the frame's dependency chain.)
  double2 ww[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    ww[t] = *reinterpret_cast<const double2*>(a.window + 2 * (j + 256 * t));
    ww[t].x *= a.wscale;   // (once per workgroup: the scale of the power rides on the window, stft_p4_rest)
    ww[t].y *= a.wscale;
  }

  // The loads of frame g + gridDim are issued BEFORE frame g's output stores: vmcnt retires in
  // order, so loads issued behind the stores would wait for the stores' HBM acknowledgements.
  int pw[8];
  uint32_t clip = 0xFFFFFFFFu;  // clip of the frame loaded last: a workgroup's frames ascend, so the next one is near
  auto issue_loads = [&](uint32_t g) {
    uint32_t lo;
    if (clip == 0xFFFFFFFFu) {  // first frame: uniform binary search over the frame offsets
      lo = 0;
      uint32_t hi = a.n_clips;
      while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (a.clip_foff[mid] <= g) lo = mid; else hi = mid;
      }
    } else {                    // later frames: walk on from the last clip (one or two scalar loads, not ten in a chain)
      lo = clip;
      while (lo + 1 < a.n_clips && a.clip_foff[lo + 1] <= g) ++lo;
    }
    clip = lo;
    stft_load_frame(a, lo, g, j, pw);
  };
  // XCD-aware frame map: workgroups b, b+8, b+16, ... share an XCD (round-robin dispatch), so each
  // group of gridDim/8 workgroups walks ONE contiguous eighth of the frames and the 50 % overlap of
  // neighbouring frames is served by that XCD's L2 (a speed choice only; any placement is correct).
  uint32_t gstep = gridDim.x >> 3;                             // host launches a multiple of 8 workgroups
  const uint32_t chunk = (a.total_frames + 7) >> 3;
  uint32_t g0 = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  uint32_t gend = min(((blockIdx.x & 7) + 1) * chunk, a.total_frames);
  if (a.frames_per_wg) {   // short-lived workgroups: k consecutive frames each, so that CU slots turn over every few dozen
    gstep = 1;             // microseconds and another stream's small kernels are not held up until this grid ends
    g0 = min(blockIdx.x * a.frames_per_wg, a.total_frames);
    gend = min(g0 + a.frames_per_wg, a.total_frames);
  }
  if (g0 < gend) issue_loads(g0);

  // Two copies of the frame loop: wave 0 (it holds thread 0, whose last butterflies pair differently) runs the one with the
  // selects, waves 1-3 the one without (36 of ~490 vector instructions per frame).  The branch is taken ONCE, on a scalar
  // condition, so nothing of a frame's state is live across it; both copies meet at the same three barriers per frame.
  auto frames = [&](auto special_c) {
    constexpr bool SPECIAL = decltype(special_c)::value;
    for (uint32_t g = g0; g < gend; g += gstep) {
      T* orow = reinterpret_cast<T*>(a.out) + (uint64_t)g * STRIDE;
      stft_frame<SPECIAL>(
          pw, ww, lds, j, [&] { if (g + gstep < gend) issue_loads(g + gstep); /* in flight across the stores */ },
          [&](int k, double p) { orow[k] = stage_value<T>(p); });
    }
  };
  if (__builtin_amdgcn_readfirstlane(j) < 64 || (a.opt & 2u)) frames(std::true_type{});
  else frames(std::false_type{});
}

// ======================================================================================
// K2: peak_pick.  Workgroup = (clip segment, slab of PK_SW bins); streams frames in time.
// peak <=> A == max over the 21x21 window clipped to the array, and A > amp_min (ties all count).
// Frequency direction: the row goes through LDS as pair maxima p2[i] = max(x[i], x[i+1]) only (one store per
// value; LDS stores cost three times what reads cost per byte): a 21-column window = 10 disjoint pairs + the pair
// (c+19, c+20).  Time direction: van Herk / Gil-Werman with blocks of 21 frames held
// in registers -- prefix max R of the current block, suffix maxima prevS of the previous one, so
// the 21-frame window max costs 3 max ops per frame instead of 20.
// POWER = true: values are power (see stft_psd_kernel) and the threshold is applied to
// 10*log10(value); POWER = false: values are compared with amp_min directly (get_2D_peaks API).
// ======================================================================================
#define PK_SW 228
#define PK_COLS 248
#define PK_PF 7

struct peak_seg {
  uint32_t gframe0;  // global (sub-batch) frame index of the clip's frame 0
  uint32_t nframes;  // frames in the clip
  uint32_t t0, t1;   // output frames [t0, t1)
};

// exact threshold test of the reference, `10*log10(P) > amp_min` (__init__.py:161,241), kept out of
// line: it runs only for local maxima whose power lies within 1e-9 of the threshold
__device__ __noinline__ bool db_above(double p, double amp_min) { return shz_db_of(p) > amp_min; }
// exact tie test of the reference: the cell's dB value equals the window's dB maximum (__init__.py:143 on the
// array of :241).  Out of line: runs only for cells within 2^-19 of their window's power maximum M that are not M.
__device__ __noinline__ bool db_equal(double p, double m) { return shz_db_of(p) == shz_db_of(m); }
// candidates of the tie test: the high words of p and m (sign, exponent, 20 mantissa bits) are equal or adjacent,
// i.e. p > m (1 - 2^-19).  A dB tie needs p within ~40 ulp of m.  Positive doubles order like their high words, and
// -inf (the padding) has a negative one; two 32-bit integer operations instead of an fp64 multiply and compare.
__device__ __forceinline__ bool pk_near(double p, double m) { return __double2hiint(p) >= __double2hiint(m) - 1; }

// --- LDS reads of peak_pick as explicit single ds_read_b64 (see the comment at their use) ---
typedef __attribute__((address_space(3))) const double pk_lds_cd;
template <int OFF>
__device__ __forceinline__ double pk_lds_ld(uint32_t a) {
  double x;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x) : "v"(a), "n"(OFF) : "memory");
  return x;
}
// row ROW of the group: x[0..9] = pr2[ROW][c + 2d], x[10] = pr2[ROW][c + 19] (columns c+19, c+20: the overlap with x[9]
// is harmless for a maximum, and no row of single values has to be stored), c = output column (rows are 256 doubles apart)
template <int ROW>
__device__ __forceinline__ void pk_row_reads(uint32_t a_pr2, double (&x)[11]) {
  x[0] = pk_lds_ld<ROW * 2048 + 0>(a_pr2);
  x[1] = pk_lds_ld<ROW * 2048 + 16>(a_pr2);
  x[2] = pk_lds_ld<ROW * 2048 + 32>(a_pr2);
  x[3] = pk_lds_ld<ROW * 2048 + 48>(a_pr2);
  x[4] = pk_lds_ld<ROW * 2048 + 64>(a_pr2);
  x[5] = pk_lds_ld<ROW * 2048 + 80>(a_pr2);
  x[6] = pk_lds_ld<ROW * 2048 + 96>(a_pr2);
  x[7] = pk_lds_ld<ROW * 2048 + 112>(a_pr2);
  x[8] = pk_lds_ld<ROW * 2048 + 128>(a_pr2);
  x[9] = pk_lds_ld<ROW * 2048 + 144>(a_pr2);
  x[10] = pk_lds_ld<ROW * 2048 + 152>(a_pr2);
}
// wait until at most N of this wave's LDS operations are outstanding; the operands tie the values to the wait so
// that no use of them is scheduled above it
#define PK_WAIT(x, N)                                                                                              \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                         \
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),    \
                 "+v"(x[8]), "+v"(x[9]), "+v"(x[10])                                                                \
               :                                                                                                   \
               : "memory")
// value of the next lane (lane + 1) of the wave; lane 63 gets an unspecified value
__device__ __forceinline__ double pk_wave_shl1(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pk_row_max(const double (&x)[11]) {
  const double a = fmax(fmax(x[0], x[1]), fmax(x[2], x[3]));
  const double b = fmax(fmax(x[4], x[5]), fmax(x[6], x[7]));
  const double c = fmax(fmax(x[8], x[9]), x[10]);
  return fmax(fmax(a, b), c);
}

template <bool POWER>
__global__ __launch_bounds__(256, 3) void peak_pick_kernel(const double* __restrict__ A, uint32_t row_stride,
                                                        uint32_t n_bins, const peak_seg* __restrict__ segs,
                                                        uint32_t n_slabs, double amp_min, double p_lo, double p_hi,
                                                        uint64_t* __restrict__ mask) {
  __shared__ double pr2[PK_PF][256];   // pair maxima max(x[c], x[c+1]) of PK_PF consecutive frames' rows of this slab
  const peak_seg sg = segs[blockIdx.y];
  const uint32_t slab = blockIdx.x;
  const int j = threadIdx.x, lane = j & 63, wave = j >> 6;
  const double NEG = -__builtin_inf();
  // thread -> column of the LDS row: wave w, lane l holds column li = 63 w + l, so lane 63 repeats lane 0 of the next
  // wave.  With that one column of overlap every lane below 63 finds its right neighbour in its own wave: the pair
  // maxima come from a DPP wave shift instead of a second trip through LDS (one barrier and seven LDS reads less
  // per group).  Lane 63 loads the value like its twin but produces no pair maximum and no output.
  const int li = 63 * wave + lane;  // = column - (slab*PK_SW - 10); li >= PK_COLS (wave 3, lanes 59..63) is idle
  const long long col = (long long)slab * PK_SW - 10 + li;
  const bool loads = li < PK_COLS && col >= 0 && col < (long long)n_bins;
  const bool reads = li >= 10 && li < PK_SW + 10 && lane < 63;  // owns output column li - 10
  const bool is_out = reads && col < (long long)n_bins;
  const int lo = (int)sg.t0 - 10 > 0 ? (int)sg.t0 - 10 : 0;
  const int hi = (int)(sg.t1 + 10 < sg.nframes ? sg.t1 + 10 : sg.nframes);
  const int end = (int)sg.t1 + 10;  // last iteration decides frame t1-1
  const double* src = A + (uint64_t)sg.gframe0 * row_stride + (loads ? col : 0);
  const uint32_t a_pr2 = (uint32_t)(uintptr_t)(pk_lds_cd*)&pr2[0][reads ? li - 10 : 0];   // window = columns li-10 .. li+10

  double prevS[21], cur[21], pre[PK_PF];
  uint32_t fc = 0, fp = 0, sp = 0;  // row-max flags of the current / previous block, suffix-max flags
  double R = NEG;
#pragma unroll
  for (int u = 0; u < 21; ++u) { prevS[u] = NEG; cur[u] = NEG; }
#pragma unroll
  for (int p = 0; p < PK_PF; ++p) {
    const int t = lo + p;
    pre[p] = (loads && t < hi) ? src[(uint64_t)t * row_stride] : NEG;
  }
  // blocks of 21 frames = 3 groups of PK_PF = 7 frames; one barrier triple per group.  Iterations
  // past `end` in the last block run predicated (no loads, no output).
  for (int tb = lo; tb < end; tb += 21) {
#pragma unroll
    for (int g3 = 0; g3 < 21 / PK_PF; ++g3) {
      double v[PK_PF], m1[PK_PF];
#pragma unroll
      for (int i = 0; i < PK_PF; ++i) {
        v[i] = pre[i];
        const int tn = tb + PK_PF * (g3 + 1) + i;  // next group's frame: keeps PK_PF rows in flight
        pre[i] = (loads && tn < hi) ? src[(uint64_t)tn * row_stride] : NEG;
      }
      __syncthreads();  // every wave is done reading the previous group's rows
      if (li < PK_COLS) {
#pragma unroll
        for (int i = 0; i < PK_PF; ++i) {
          const double nb = pk_wave_shl1(v[i]);        // column li + 1 (idle lanes hold -inf)
          if (lane < 63) pr2[i][li] = fmax(v[i], nb);
        }
      }
      __syncthreads();
      // Row maxima over columns j .. j+20 = pairs (j, j+1) .. (j+18, j+19) and the pair (j+19, j+20).  The eleven reads of a
      // row are single ds_read_b64 (2 LDS-array cycles each); left to the compiler they are fused in pairs into
      // ds_read2_b64 at 8 cycles per pair (PMC: 2,201 -> 1,403 LDS-array cycles per frame).  Two rows are in
      // flight: the reads of row i+1 are issued before row i is reduced.
#pragma unroll
      for (int i = 0; i < PK_PF; ++i) m1[i] = NEG;
      if (reads) {
        double xa[11], xb[11];
        pk_row_reads<0>(a_pr2, xa);
        pk_row_reads<1>(a_pr2, xb); PK_WAIT(xa, 11); m1[0] = pk_row_max(xa);
        pk_row_reads<2>(a_pr2, xa); PK_WAIT(xb, 11); m1[1] = pk_row_max(xb);
        pk_row_reads<3>(a_pr2, xb); PK_WAIT(xa, 11); m1[2] = pk_row_max(xa);
        pk_row_reads<4>(a_pr2, xa); PK_WAIT(xb, 11); m1[3] = pk_row_max(xb);
        pk_row_reads<5>(a_pr2, xb); PK_WAIT(xa, 11); m1[4] = pk_row_max(xa);
        pk_row_reads<6>(a_pr2, xa); PK_WAIT(xb, 11); m1[5] = pk_row_max(xb);
        PK_WAIT(xa, 0); m1[6] = pk_row_max(xa);
      }
      // time direction (registers only)
#pragma unroll
      for (int i = 0; i < PK_PF; ++i) {
        const int u = PK_PF * g3 + i;
        const int t = tb + u;
        R = (u == 0) ? m1[i] : fmax(R, m1[i]);
        cur[u] = m1[i];
        // POWER: flags mark candidates of the dB tie test (value within 2^-19 of the maximum), decided below
        if (POWER ? pk_near(v[i], m1[i]) : v[i] == m1[i]) fc |= 1u << u;
        const double m2 = (u < 20) ? fmax(prevS[(u + 1) % 21], R) : R;
        // centre frame t - 10: position u-10 of the current block or u+11 of the previous one
        bool cand;
        if (u >= 10) {
          const double ck = cur[(u + 11) % 21];
          cand = ((fc >> ((u + 11) % 21)) & 1u) && (POWER ? pk_near(ck, m2) : ck == m2);
        } else {
          const double sk = prevS[(u + 11) % 21];
          cand = ((fp >> ((u + 11) % 21)) & 1u) && ((sp >> ((u + 11) % 21)) & 1u) && (POWER ? pk_near(sk, m2) : sk == m2);
        }
        const int tc = t - 10;
        const bool in_seg = tc >= (int)sg.t0 && tc < (int)sg.t1;
        bool pk = is_out && in_seg && cand;
        if (POWER) {
          if (pk) {  // a few cells per frame: reload the centre value, which the streaming part does not keep
            const double vc = src[(uint64_t)tc * row_stride];
            pk = (vc == m2 || db_equal(vc, m2)) &&
                 // power > p_hi: surely above amp_min dB; <= p_lo: surely not; between: exact test (dB(vc) == dB(m2))
                 (m2 > p_hi || (m2 > p_lo && db_above(m2, amp_min)));
          }
        } else {
          pk = pk && (m2 > amp_min);
        }
        const unsigned long long bal = __ballot(pk);
        if (in_seg && lane == 0)
          mask[((uint64_t)(sg.gframe0 + tc) * n_slabs + slab) * 4 + wave] = bal;
        if (u == 20) {  // block complete: suffix maxima + "is the suffix max" flags
          fp = fc;
          fc = 0;
          sp = 1u << 20;
          prevS[20] = cur[20];
#pragma unroll
          for (int k = 19; k >= 0; --k) {
            if (POWER ? pk_near(cur[k], prevS[k + 1]) : cur[k] >= prevS[k + 1]) sp |= 1u << k;
            prevS[k] = fmax(cur[k], prevS[k + 1]);
          }
        }
      }
    }
  }
}

#include "shz_peak32.inc"

// amp_min < 0 only: the reference drops local maxima that sit inside a region of exact zeros.  get_2D_peaks marks
// `local_max != binary_erosion(arr2D == 0, 21x21, border_value=1)` (__init__.py:147-151): a zero-valued cell whose whole
// 21x21 window (clipped to the clip; outside counts as zero) is zero is a local maximum AND eroded background, the XOR
// removes it.  For amp_min >= 0 such cells never pass `> amp_min`; below zero they would, so their mask bits are cleared
// here, before the scan.  ZERO is 0.0 on a dB array and 1.0 on the power array (zero power is stored as 1.0 = 0 dB).
__global__ __launch_bounds__(256) void peak_zero_plateau_kernel(uint64_t* __restrict__ mask, uint64_t n_words,
                                                                mask_geom mg, const double* __restrict__ A,
                                                                uint32_t row_stride, uint32_t n_bins, double zero,
                                                                const uint32_t* __restrict__ clip_foff,
                                                                uint32_t n_clips) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  const uint64_t m0 = mask[w];
  if (!m0) return;
  const uint32_t per_frame = mg.n_slabs * mg.nw;
  const uint32_t g = (uint32_t)(w / per_frame);
  const uint32_t rem = (uint32_t)(w % per_frame);
  const uint32_t slab = rem / mg.nw, wv = rem % mg.nw;
  uint32_t lo = 0, hi = n_clips;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (clip_foff[mid] <= g) lo = mid; else hi = mid;
  }
  const int t = (int)(g - clip_foff[lo]), F = (int)(clip_foff[lo + 1] - clip_foff[lo]);
  uint64_t m = m0, keep = m0;
  while (m) {
    const int b = __ffsll((long long)m) - 1;
    m &= m - 1;
    const int col = (int)(slab * mg.sw + wv * mg.lane_stride) + b - 10;
    const double* centre = A + (uint64_t)g * row_stride + col;
    if (*centre != zero) continue;
    bool all_zero = true;
    for (int dt = -10; dt <= 10 && all_zero; ++dt) {
      if (t + dt < 0 || t + dt >= F) continue;
      for (int df = -10; df <= 10; ++df) {
        const int c = col + df;
        if (c < 0 || c >= (int)n_bins) continue;
        if (centre[(long long)dt * row_stride + df] != zero) { all_zero = false; break; }
      }
    }
    if (all_zero) keep &= ~(1ull << b);
  }
  if (keep != m0) mask[w] = keep;
}

// K3: expand peak masks into the ordered peak list.  One thread per mask word.
// frame_t[g] = index of frame g inside its clip (one binary search per frame instead of one per mask word)
__global__ void frame_time_kernel(const uint32_t* __restrict__ clip_foff, uint32_t n_clips, uint32_t frames,
                                  uint32_t* __restrict__ frame_t) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= frames) return;
  uint32_t lo = 0, hi = n_clips;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (clip_foff[mid] <= g) lo = mid; else hi = mid;
  }
  frame_t[g] = g - clip_foff[lo];
}

__global__ __launch_bounds__(256) void peak_expand_kernel(const uint64_t* __restrict__ mask,
                                                          const uint32_t* __restrict__ word_off, uint32_t n_words,
                                                          mask_geom mg, const uint32_t* __restrict__ frame_t,
                                                          uint16_t* __restrict__ peak_f, uint32_t* __restrict__ peak_t,
                                                          uint32_t cap) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  uint64_t m = mask[w];
  if (!m) return;
  const uint32_t per_frame = mg.n_slabs * mg.nw;
  const uint32_t g = w / per_frame;
  const uint32_t rem = w - g * per_frame;
  const uint32_t slab = rem / mg.nw, wv = rem % mg.nw;
  const uint32_t t = frame_t[g];
  uint32_t o = word_off[w];
  while (m) {
    const int b = __ffsll((long long)m) - 1;
    m &= m - 1;
    if (o < cap) {  // a list too small for the sub-batch is reported through xctl (XF_PEAK_CAP), never overrun
      peak_f[o] = (uint16_t)(slab * mg.sw + wv * mg.lane_stride + b - 10);  // wave wv, lane b holds that slab column
      peak_t[o] = t;
    }
    ++o;
  }
}

// The same from a per-FRAME exclusive scan of the peak counts (fp32 staging: peak_pick32 and peak_verify count the
// peaks of every frame as they set mask bits, so the 25.8M-word popcount scan of a 1,000-clip batch -- three passes over
// 206 MB and a 103 MB prefix array -- shrinks to a scan of 644,000 counts).  A workgroup owns 256 consecutive mask
// words; all but one in a hundred hold no bit at all and end here.  Otherwise: prefix of the words inside the
// workgroup, plus the peaks of the first word's frame that lie in front of the workgroup, plus that frame's offset.
#define PXF_PER 8   // mask words per thread: a workgroup reads 16 KB before it decides whether there is anything to do
__global__ __launch_bounds__(256) void peak_expand_frames_kernel(const uint64_t* __restrict__ mask,
                                                                 const uint32_t* __restrict__ frame_off, uint32_t n_words,
                                                                 mask_geom mg, const uint32_t* __restrict__ frame_t,
                                                                 uint16_t* __restrict__ peak_f, uint32_t* __restrict__ peak_t,
                                                                 uint32_t cap) {
  __shared__ uint32_t s_w[4], s_before;
  const uint32_t w0 = blockIdx.x * (256u * PXF_PER), w1 = w0 + threadIdx.x * PXF_PER;   // the thread's first word
  uint64_t m[PXF_PER];
  if (w1 + PXF_PER <= n_words) {
    const ulonglong2* p2 = (const ulonglong2*)(mask + w1);   // the mask buffer and w1 * 8 are 16-byte aligned
#pragma unroll
    for (int i = 0; i < PXF_PER / 2; ++i) { const ulonglong2 x = p2[i]; m[2 * i] = x.x; m[2 * i + 1] = x.y; }
  } else {
#pragma unroll
    for (int i = 0; i < PXF_PER; ++i) m[i] = w1 + i < n_words ? mask[w1 + i] : 0ull;
  }
  uint64_t any = 0;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < PXF_PER; ++i) { any |= m[i]; c += (uint32_t)__popcll(m[i]); }
  if (!__syncthreads_or(any != 0)) return;   // uniform
  const uint32_t per_frame = mg.n_slabs * mg.nw;
  const uint32_t g0 = w0 / per_frame, r0 = w0 - g0 * per_frame;   // r0 < 256 words of frame g0 precede the workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t before = threadIdx.x < r0 ? (uint32_t)__popcll(mask[(uint64_t)g0 * per_frame + threadIdx.x]) : 0u;
  uint32_t inc = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
    if (lane >= d) inc += o;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) before += (uint32_t)__shfl_xor((int)before, d, 64);
  if (threadIdx.x == 0) s_before = 0;
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  if (lane == 0 && before) atomicAdd(&s_before, before);
  __syncthreads();
  if (!any) return;
  uint32_t o = frame_off[g0] + s_before + inc - c;
  for (int k = 0; k < wave; ++k) o += s_w[k];
#pragma unroll
  for (int i = 0; i < PXF_PER; ++i) {
    uint64_t mm = m[i];
    if (!mm) continue;
    const uint32_t w = w1 + i;
    const uint32_t g = w / per_frame;
    const uint32_t rem = w - g * per_frame;
    const uint32_t slab = rem / mg.nw, wv = rem % mg.nw;
    const uint32_t t = frame_t[g];
    while (mm) {
      const int bb = __ffsll((long long)mm) - 1;
      mm &= mm - 1;
      if (o < cap) {
        peak_f[o] = (uint16_t)(slab * mg.sw + wv * mg.lane_stride + bb - 10);
        peak_t[o] = t;
      }
      ++o;
    }
  }
}

// per-clip CSR offsets from a per-element exclusive scan: out[c] = scan[first[c]*mult] (or total at the end)
__global__ void gather_offsets_kernel(const uint32_t* __restrict__ scan, const uint64_t* __restrict__ total,
                                      const uint32_t* __restrict__ first, uint64_t mult, uint64_t n_elems,
                                      uint32_t n_clips, uint32_t* __restrict__ out, uint32_t clamp) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > n_clips) return;
  const uint64_t idx = (uint64_t)first[c] * mult;
  const uint32_t v = (c == n_clips || idx >= n_elems) ? (uint32_t)*total : scan[idx];
  out[c] = v < clamp ? v : clamp;
}

// clip of peak i: the workgroup's first peak is searched once (peaks are clip-major), every thread walks on from there
__device__ __forceinline__ uint32_t pair_clip_of(const uint32_t* __restrict__ peak_coff, uint32_t n_clips, uint32_t i) {
  __shared__ uint32_t s_lo;
  if (threadIdx.x == 0) {
    const uint32_t first = blockIdx.x * blockDim.x;
    uint32_t lo = 0, hi = n_clips;
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (peak_coff[mid] <= first) lo = mid; else hi = mid;
    }
    s_lo = lo;
  }
  __syncthreads();
  uint32_t lo = s_lo;
  while (lo + 1 < n_clips && peak_coff[lo + 1] <= i) ++lo;
  return lo;
}

// K4a: number of valid partners of each peak (prefix of the next fan-1 peaks of the same clip with dt <= 200).
// The number of peaks is read on the device (d_n, clamped to the list capacity n_cap); slots beyond it count 0.
__global__ __launch_bounds__(256) void pair_count_kernel(const uint32_t* __restrict__ peak_t,
                                                         const uint32_t* __restrict__ peak_coff, uint32_t n_clips,
                                                         const unsigned long long* __restrict__ d_n, uint32_t n_cap,
                                                         uint32_t fan, uint32_t* __restrict__ cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long nn = *d_n;
  const uint32_t n_peaks = nn < n_cap ? (uint32_t)nn : n_cap;
  if (blockIdx.x * blockDim.x >= n_peaks) {  // whole workgroup beyond the list (uniform)
    if (i < n_cap) cnt[i] = 0;
    return;
  }
  const uint32_t lo = pair_clip_of(peak_coff, n_clips, i < n_peaks ? i : n_peaks - 1);
  if (i >= n_peaks) {
    if (i < n_cap) cnt[i] = 0;
    return;
  }
  const uint32_t endp = peak_coff[lo + 1];
  const uint32_t t1 = peak_t[i];
  uint32_t c = 0;
  for (uint32_t jn = 1; jn < fan; ++jn) {
    if (i + jn >= endp) break;
    const uint32_t dt = peak_t[i + jn] - t1;  // sorted by time: dt >= 0 = MIN_HASH_TIME_DELTA
    if (dt <= SHZ_MAX_DT) ++c;
  }
  cnt[i] = c;
}

// K4b: write (key32, t1) in (i, j) generation order at out_base0 + *d_base (d_base may be null)
__global__ __launch_bounds__(256) void pair_write_kernel(const uint16_t* __restrict__ peak_f,
                                                         const uint32_t* __restrict__ peak_t,
                                                         const uint32_t* __restrict__ peak_coff, uint32_t n_clips,
                                                         const unsigned long long* __restrict__ d_n, uint32_t n_cap,
                                                         uint32_t fan, const uint32_t* __restrict__ hoff,
                                                         uint32_t* __restrict__ key32, uint32_t* __restrict__ t1out,
                                                         uint64_t out_base0, const unsigned long long* __restrict__ d_base,
                                                         uint64_t cap) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long nn = *d_n;
  const uint32_t n_peaks = nn < n_cap ? (uint32_t)nn : n_cap;
  if (blockIdx.x * blockDim.x >= n_peaks) return;  // uniform
  const uint32_t lo = pair_clip_of(peak_coff, n_clips, i < n_peaks ? i : n_peaks - 1);
  if (i >= n_peaks) return;
  const uint32_t endp = peak_coff[lo + 1];
  const uint32_t t1 = peak_t[i], f1 = peak_f[i];
  uint64_t o = out_base0 + (d_base ? *d_base : 0ull) + hoff[i];
  for (uint32_t jn = 1; jn < fan; ++jn) {
    if (i + jn >= endp) break;
    const uint32_t dt = peak_t[i + jn] - t1;
    if (dt <= SHZ_MAX_DT) {
      if (o < cap) {
        key32[o] = (f1 << 20) | ((uint32_t)peak_f[i + jn] << 8) | dt;
        t1out[o] = t1;
      }
      ++o;
    }
  }
}

// ---- bookkeeping of the device-driven pipeline (xctl, shz_peak32.inc): tiny single-thread kernels ----
__global__ void xctl_begin_sub_kernel(xctl* c) { c->und_count = 0; }
// after the popcount scan: was the peak list large enough?
__global__ void xctl_after_peaks_kernel(xctl* c, uint32_t cap_peaks, uint32_t und_cap) {
  if (c->sub_peaks > cap_peaks) c->flags |= XF_PEAK_CAP;
  if (c->sub_peaks > c->max_sub_peaks) c->max_sub_peaks = c->sub_peaks;
  if (c->und_count > und_cap) c->flags |= XF_UND_CAP | XF_FALLBACK;
}
// per-clip output offsets of this sub-batch: out64[c0 + i + 1] = base + rel[i + 1]  (rel clamped by the producer)
__global__ void xctl_offsets_kernel(const xctl* c, bool hashes, const uint32_t* __restrict__ rel, uint32_t nc,
                                    uint32_t c0, unsigned long long* __restrict__ out64) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && c0 == 0) out64[0] = 0;
  if (i >= nc) return;
  out64[c0 + i + 1] = (hashes ? c->hash_base : c->peak_base) + rel[i + 1];
}
__global__ void xctl_advance_kernel(xctl* c, uint32_t cap_peaks, bool hashes) {
  c->peak_base += c->sub_peaks < cap_peaks ? c->sub_peaks : cap_peaks;
  if (hashes) c->hash_base += c->sub_hashes;
}
// ---- the whole tail of a SMALL sub-batch in one workgroup ------------------------------------------------------------
// One 5-10 s query is ~100-200 frames, ~500-1,000 peaks, ~2,000-4,000 hashes: the thirteen launches between peak_verify and
// the read-back (scan of the per-frame counts, bookkeeping, frame times, mask expansion, per-clip offsets, partner counts,
// their scan, the hash write, per-clip hash offsets, offsets out, advance) cost ~4.7 us each and did microseconds of work.
// Here one workgroup walks the mask in word order -- the exclusive prefix of the words' popcounts IS the peak index
// (__init__.py:155,194-195: peaks ordered by time, then frequency, per clip) -- and does the rest on the list it wrote.
// Same results as the separate kernels, bit for bit (tests/test_gpu_extract.py runs both on the same inputs).
#define XT_THREADS 1024
#define XT_MAX_WORDS (XT_THREADS * 160)   // 4,096 frames of 40 mask words
#define XT_MAX_PEAKS 65536

__device__ __forceinline__ uint32_t xt_block_scan(uint32_t v, uint32_t* total, uint32_t* tmp) {   // exclusive, 1024 threads
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();   // tmp may still be read from the previous use
  if (lane == 63) tmp[wave] = inc;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < XT_THREADS / 64; ++w) {
    const uint32_t x = tmp[w];
    if (w < wave) woff += x;
    tot += x;
  }
  *total = tot;
  return woff + inc - v;
}
// last index lo in [0, n) with a[lo] <= x (a ascending, a[0] <= x)
__device__ __forceinline__ uint32_t xt_last_le(const uint32_t* __restrict__ a, uint32_t n, uint32_t x) {
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(XT_THREADS) void extract_tail_small_kernel(
    const uint64_t* __restrict__ mask, uint32_t n_words, mask_geom mg, const uint32_t* __restrict__ clip_foff, uint32_t nc,
    uint32_t c0, xctl* __restrict__ ctl, uint32_t cap_peaks, uint32_t und_cap, uint32_t fan, uint16_t* __restrict__ pf,
    uint32_t* __restrict__ pt, uint32_t* __restrict__ pc /* nc + 1 */, uint32_t* __restrict__ hoff /* cap_peaks */,
    uint32_t* __restrict__ key32, uint32_t* __restrict__ t1out, uint64_t o_cap, unsigned long long* __restrict__ out64) {
  __shared__ uint32_t tmp[XT_THREADS / 64];
  const uint32_t tid = threadIdx.x;
  const uint32_t per_frame = mg.n_slabs * mg.nw;
  const unsigned long long hash_base = ctl->hash_base;
  // A. mask -> ordered peak list; pc[c] = index of clip c's first peak
  uint32_t base = 0;
  for (uint32_t w0 = 0; w0 < n_words; w0 += XT_THREADS) {
    const uint32_t w = w0 + tid;
    uint64_t m = w < n_words ? mask[w] : 0ull;
    uint32_t tot;
    uint32_t o = base + xt_block_scan((uint32_t)__popcll(m), &tot, tmp);
    if (w < n_words) {
      const uint32_t g = w / per_frame, rem = w - g * per_frame;
      if (rem == 0 || m) {
        const uint32_t lo = xt_last_le(clip_foff, nc, g);
        if (rem == 0 && clip_foff[lo] == g) pc[lo] = o < cap_peaks ? o : cap_peaks;   // first word of the clip's first frame
        const uint32_t slab = rem / mg.nw, wv = rem % mg.nw, t = g - clip_foff[lo];
        while (m) {
          const int b = __ffsll((long long)m) - 1;
          m &= m - 1;
          if (o < cap_peaks) {
            pf[o] = (uint16_t)(slab * mg.sw + wv * mg.lane_stride + b - 10);
            pt[o] = t;
          }
          ++o;
        }
      }
    }
    base += tot;
  }
  const uint32_t sub_peaks = base, n_peaks = sub_peaks < cap_peaks ? sub_peaks : cap_peaks;
  if (tid == 0) pc[nc] = n_peaks;
  __syncthreads();   // the list and pc are complete (global writes of this workgroup, read below by other threads)
  // B. partners of every peak, their prefix = the hash index; (key32, t1) written in (i, j) generation order
  uint32_t hbase = 0;
  for (uint32_t i0 = 0; i0 < n_peaks; i0 += XT_THREADS) {
    const uint32_t i = i0 + tid;
    uint32_t c = 0, endp = 0, t1 = 0;
    if (i < n_peaks) {
      endp = pc[xt_last_le(pc, nc, i) + 1];   // (clips without peaks share an offset: the last of them is the peak's clip)
      t1 = pt[i];
      for (uint32_t jn = 1; jn < fan; ++jn) {
        if (i + jn >= endp) break;
        if (pt[i + jn] - t1 <= SHZ_MAX_DT) ++c;   // sorted by time: dt >= 0 = MIN_HASH_TIME_DELTA
      }
    }
    uint32_t tot;
    const uint32_t h0 = hbase + xt_block_scan(c, &tot, tmp);
    if (i < n_peaks) {
      hoff[i] = h0;
      const uint32_t f1 = pf[i];
      uint64_t o = hash_base + h0;
      for (uint32_t jn = 1; jn < fan; ++jn) {
        if (i + jn >= endp) break;
        const uint32_t dt = pt[i + jn] - t1;
        if (dt <= SHZ_MAX_DT) {
          if (o < o_cap) {
            key32[o] = (f1 << 20) | ((uint32_t)pf[i + jn] << 8) | dt;
            t1out[o] = t1;
          }
          ++o;
        }
      }
    }
    hbase += tot;
  }
  __syncthreads();
  // C. per-clip hash offsets of the call, bookkeeping
  for (uint32_t c = tid; c < nc; c += XT_THREADS) {
    const uint32_t p1 = pc[c + 1];
    out64[c0 + c + 1] = hash_base + (p1 < n_peaks ? hoff[p1] : hbase);
  }
  if (tid == 0) {
    if (c0 == 0) out64[0] = 0;
    ctl->sub_peaks = sub_peaks;
    ctl->sub_hashes = hbase;
    if (sub_peaks > cap_peaks) ctl->flags |= XF_PEAK_CAP;
    if (sub_peaks > ctl->max_sub_peaks) ctl->max_sub_peaks = sub_peaks;
    if (ctl->und_count > und_cap) ctl->flags |= XF_UND_CAP | XF_FALLBACK;
    ctl->peak_base += n_peaks;
    ctl->hash_base = hash_base + hbase;
  }
}

// append this sub-batch's peak list to the output arrays at peak_base
__global__ __launch_bounds__(256) void peaks_out_kernel(const xctl* c, const uint16_t* __restrict__ pf,
                                                        const uint32_t* __restrict__ pt, uint32_t cap_peaks,
                                                        uint16_t* __restrict__ out_f, uint32_t* __restrict__ out_t,
                                                        uint64_t cap) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t n = c->sub_peaks < cap_peaks ? c->sub_peaks : cap_peaks;
  if (i >= n) return;
  const uint64_t o = c->peak_base + i;
  if (o < cap) { out_f[o] = pf[i]; out_t[o] = pt[i]; }
}
__global__ void set_u64_dev_kernel(unsigned long long* p, unsigned long long v) { *p = v; }

// transpose [F][stride] (frame-major) -> [n_bins][F] (the reference's freq-major layout)
__global__ void transpose_db_kernel(const double* __restrict__ in, uint32_t stride, uint32_t F, uint32_t n_bins,
                                    double* __restrict__ out, bool as_power) {
  __shared__ double tile[32][33];
  const uint32_t bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx: bins, by: frames
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    uint32_t f = by + r, b = bx + threadIdx.x;
    // the staged value is power with zeros stored as 1.0: 10*log10 gives the reference's dB (0 dB for zeros)
    const double p = (f < F && b < n_bins) ? in[(uint64_t)f * stride + b] : 1.0;
    tile[r][threadIdx.x] = as_power ? p : shz_db_of(p);
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    uint32_t b = bx + r, f = by + threadIdx.x;
    if (b < n_bins && f < F) out[(uint64_t)b * F + f] = tile[threadIdx.x][r];
  }
}

// transpose back: [n_rows][n_cols] freq-major host layout -> [n_cols][stride] frame-major
__global__ void transpose_in_kernel(const double* __restrict__ in, uint32_t n_rows, uint32_t n_cols, uint32_t stride,
                                    double* __restrict__ out) {
  __shared__ double tile[32][33];
  const uint32_t bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx: cols(time), by: rows(freq)
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    uint32_t rr = by + r, cc = bx + threadIdx.x;
    tile[r][threadIdx.x] = (rr < n_rows && cc < n_cols) ? in[(uint64_t)rr * n_cols + cc] : 0.0;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    uint32_t cc = bx + r, rr = by + threadIdx.x;
    if (cc < n_cols && rr < n_rows) out[(uint64_t)cc * stride + rr] = tile[threadIdx.x][r];
  }
}

// ======================================================================================
// host orchestration
// ======================================================================================
extern "C" int32_t shz_db_values(const double* power, uint64_t n, double* out_db) {
  if (n && (!power || !out_db)) return SHZ_E_INVALID;
  for (uint64_t i = 0; i < n; ++i) out_db[i] = power[i] != 0.0 ? shz_db_of(power[i]) : 0.0;
  return SHZ_OK;
}

// frames mlab.specgram cuts n samples into with `hop` = NFFT - noverlap new samples a frame (mlab:268-271, 307-308)
static inline uint32_t frames_hop(uint64_t n, uint32_t hop) {
  if (n < SHZ_NFFT) return 1;
  return (uint32_t)std::min<uint64_t>((n - SHZ_NFFT) / hop + 1, 0xFFFFFFFFull);
}
extern "C" uint32_t shz_frame_count_hop(uint64_t n, uint32_t hop) { return hop >= 1 && hop <= SHZ_NFFT ? frames_hop(n, hop) : 0; }
extern "C" uint32_t shz_frame_count(uint64_t n) {
  if (n < SHZ_NFFT) return 1;
  return (uint32_t)((n - SHZ_NFFT) / SHZ_HOP + 1);
}

#define PK_SEG 252       // output frames per peak_pick workgroup (12 blocks of 21) when workgroups are scarce
#define PK_SEG_LONG 672  // ... and when the batch is large: 32 blocks, a 30 s clip in one piece (no time halo)
#define PK_SEG_SHORT 42  // ... and when a handful of workgroups is all there is: 2 blocks

static const mask_geom MG_F64 = {(SHZ_NBINS + PK_SW - 1) / PK_SW, 4, 63, PK_SW};
static mask_geom mg_f32(int nw) {
  const uint32_t sw = 61u * nw - 17u;
  return mask_geom{(SHZ_NBINS + sw - 1) / sw, (uint32_t)nw, 61u, sw};
}
// waves per peak_pick32 workgroup (slab = 61 nw - 17 bins).  Measured on 644,000 frames: 7 waves (5 slabs) 3.93 ms,
// 6: 4.47, 4 (10 slabs): 2.65, 3: 2.56, 2 (20 slabs of 105 bins, 19 % halo): 2.05, 1: 2.22 -- small workgroups win
// although they re-read more halo: the two barriers per 7 frames cost more than the columns
static int p32_nw() {
  static const int nw = [] {
    const char* e = getenv("SHZ_PEAK_NW");
    const int v = e ? atoi(e) : 2;
    return (v >= 1 && v <= 7 && v != 5) ? v : 2;
  }();
  return nw;
}

struct sub_batch {
  uint32_t c0, c1;        // clips [c0, c1)
  uint32_t frames;
};

static int32_t plan_sub_batches(shz_ctx* ctx, const uint64_t* clip_off, uint32_t n_clips, uint64_t bytes_per_frame,
                                std::vector<sub_batch>& out, uint64_t frame_cap = 0) {
  uint64_t max_frames = ctx->ws_limit / bytes_per_frame;
  if (frame_cap && frame_cap < max_frames) max_frames = frame_cap;
  if (max_frames > (1u << 20)) max_frames = 1u << 20;  // mask words, peaks < 2^32 (hash count checked per sub-batch)
  if (max_frames < 64) max_frames = 64;
  sub_batch cur{0, 0, 0};
  for (uint32_t c = 0; c < n_clips; ++c) {
    if (clip_off[c + 1] < clip_off[c]) SHZ_FAIL(ctx, SHZ_E_INVALID, "clip_off must be non-decreasing (clip %u)", c);
    uint64_t f = frames_hop(clip_off[c + 1] - clip_off[c], ctx->hop);
    if (f > max_frames)
      SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "clip %u has %llu frames; at most %llu fit the workspace limit", c,
               (unsigned long long)f, (unsigned long long)max_frames);
    if (cur.frames + f > max_frames && cur.c1 > cur.c0) {
      out.push_back(cur);
      cur = sub_batch{c, c, 0};
    }
    cur.c1 = c + 1;
    cur.frames += (uint32_t)f;
  }
  if (cur.c1 > cur.c0) out.push_back(cur);
  return SHZ_OK;
}

struct sub_dev {
  const int16_t* pcm;     // device pointer to sample clip_off[c0]... (base such that soff are relative)
  uint64_t* d_soff;
  uint64_t* d_len;
  uint32_t* d_foff;
  peak_seg* d_segs;
  uint32_t n_segs;
  std::vector<uint32_t> foff;
};

// Upload per-clip metadata of a sub-batch; pcm_base_off = sample offset of d_pcm[0] in clip_off units.
// No host synchronisation: the host-side image of the tables is handed to `keep`, which the caller holds until the
// call's final sync (a small pageable hipMemcpyAsync may or may not have staged its source when it returns).
static int32_t upload_meta(shz_ctx* ctx, const uint64_t* clip_off, const sub_batch& sb, uint64_t pcm_base_off,
                           uint32_t n_slabs, uint32_t wg_per_cu, sub_dev& sd, std::vector<std::vector<uint64_t>>& keep,
                           int slot = SHZ_WS_META) {
  const uint32_t nc = sb.c1 - sb.c0;
  sd.foff.assign(nc + 1, 0);
  std::vector<peak_seg> segs;
  // Each segment re-reads 10 halo frames on both sides (PMC: -4.6 % kernel time with whole-clip segments), but short
  // segments keep the chip busy on small batches: go long once that still leaves >= 4 workgroups per slot.
  const uint64_t slots = (uint64_t)ctx->prop.multiProcessorCount * wg_per_cu;
  // A single short clip (one 5 s query: 107 frames) is all latency: 42-frame segments halve the rows a workgroup walks.
  const uint32_t seg_len = (uint64_t)sb.frames * n_slabs / PK_SEG_LONG >= 4 * slots ? PK_SEG_LONG
                           : (uint64_t)sb.frames * n_slabs / PK_SEG * 8 < slots ? PK_SEG_SHORT : PK_SEG;
  // one blob: soff[nc] | len[nc] | foff[nc+1] (u32, padded to 8) | segs
  const uint64_t foff_words = (nc + 2) / 2;  // u64 words holding nc+1 u32
  for (uint32_t i = 0; i < nc; ++i) {
    const uint32_t f = frames_hop(clip_off[sb.c0 + i + 1] - clip_off[sb.c0 + i], ctx->hop);
    sd.foff[i + 1] = sd.foff[i] + f;
    for (uint32_t t0 = 0; t0 < f; t0 += seg_len)
      segs.push_back(peak_seg{sd.foff[i], f, t0, std::min(t0 + seg_len, f)});
  }
  static_assert(sizeof(peak_seg) == 16, "peak_seg is two u64 words");
  keep.emplace_back(2 * (uint64_t)nc + foff_words + 2 * segs.size() + 1);
  std::vector<uint64_t>& blob = keep.back();
  for (uint32_t i = 0; i < nc; ++i) {
    blob[i] = clip_off[sb.c0 + i] - pcm_base_off;
    blob[nc + i] = clip_off[sb.c0 + i + 1] - clip_off[sb.c0 + i];
  }
  memcpy(&blob[2 * (uint64_t)nc], sd.foff.data(), (nc + 1) * 4);
  if (!segs.empty()) memcpy(&blob[2 * (uint64_t)nc + foff_words], segs.data(), segs.size() * sizeof(peak_seg));
  void* p0;
  SHZ_TRY(shz_ws_reserve(ctx, slot, blob.size() * 8 + 64, &p0));
  sd.d_soff = (uint64_t*)p0;
  sd.d_len = sd.d_soff + nc;
  sd.d_foff = (uint32_t*)(sd.d_len + nc);
  sd.d_segs = (peak_seg*)(sd.d_soff + 2 * (uint64_t)nc + foff_words);
  sd.n_segs = (uint32_t)segs.size();
  SHZ_HIP(ctx, shz_memcpy(ctx, p0, blob.data(), blob.size() * 8, hipMemcpyHostToDevice));
  return SHZ_OK;
}

// PCM of the sub-batch on the device: either the caller's device buffer or a staged copy
static int32_t stage_pcm(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, const sub_batch& sb,
                         uint32_t flags, const int16_t** d_pcm, uint64_t* base_off, int slot = SHZ_WS_PCM) {
  const uint64_t s0 = clip_off[sb.c0], s1 = clip_off[sb.c1];
  if (flags & SHZ_PCM_DEVICE) {
    *d_pcm = pcm;
    *base_off = 0;
    return SHZ_OK;
  }
  void* p;
  SHZ_TRY(shz_ws_reserve(ctx, slot, (s1 - s0) * 2 + 64, &p));
  if (s1 > s0) SHZ_HIP(ctx, shz_memcpy(ctx, p, pcm + s0, (s1 - s0) * 2, hipMemcpyHostToDevice));
  *d_pcm = (const int16_t*)p;
  *base_off = s0;
  return SHZ_OK;
}

static stft_args make_stft_args(shz_ctx* ctx, const int16_t* d_pcm, const sub_dev& sd, uint32_t nc, uint32_t frames,
                                uint32_t fs, void* d_out) {
  stft_args a;
  a.pcm = d_pcm;
  a.clip_soff = sd.d_soff;
  a.clip_len = sd.d_len;
  a.clip_foff = sd.d_foff;
  a.n_clips = nc;
  a.total_frames = frames;
  a.out = d_out;
  a.window = ctx->d_window;
  a.hop = ctx->hop;
  a.tw = ctx->d_twiddle;
  a.scale = 0.25 / ((double)fs * ctx->win_sumsq);
  a.wscale = sqrt(2.0 * a.scale);
  a.np_window = ctx->d_np_window;
  a.np_comp = (const cplx*)ctx->d_np_comp;
  a.r_fs = 1.0 / (double)fs;          // numpy divides a complex array by a real: it multiplies by the rounded reciprocal
  a.r_s = 1.0 / ctx->np_sumsq;
  a.np_unfused = ctx->np_unfused ? 1u : 0u;
  a.frames_per_wg = 0;
  static const uint32_t opt_env = [] { const char* e = getenv("SHZ_STFT_OPT"); return e ? (uint32_t)atoi(e) : 0u; }();
  a.opt = opt_env;
  return a;
}

template <typename T>
static int32_t launch_stft(shz_ctx* ctx, const stft_args& a, int wgs_override = 0, bool persistent = false) {
  shz_prof_scope ps(ctx, 0);
  static const int wgs_per_cu = [] {  // tuning knob: resident stft workgroups per CU (LDS allows 3)
    const char* e = getenv("SHZ_STFT_WGS_PER_CU");
    const int v = e ? atoi(e) : 3;
    return v >= 1 && v <= 3 ? v : 3;
  }();
  uint32_t grid = (uint32_t)ctx->prop.multiProcessorCount * (wgs_override ? wgs_override : wgs_per_cu);
  if (grid > a.total_frames) grid = a.total_frames;
  grid = (grid + 7) & ~7u;  // multiple of 8: see the XCD-aware frame map in the kernel
  static const uint32_t chunk_frames = [] { const char* e = getenv("SHZ_STFT_CHUNK"); const int v = e ? atoi(e) : 32; return (uint32_t)(v > 0 ? v : 0); }();
  stft_args b = a;
  // One pipeline: workgroups of 32 consecutive frames (the halves neighbouring frames share stay in the workgroup's L1,
  // and CU slots turn over): 4.00 ms per 644,000 frames against 4.35 persistent (4: 4.53, 8: 4.22, 16: 4.05, 64: 4.03).
  // Two pipelines side by side prefer the persistent grid (6.48 vs 6.55 ms per step).
  if (!persistent && chunk_frames && a.total_frames > (uint64_t)grid * chunk_frames) {
    b.frames_per_wg = chunk_frames;
    grid = (a.total_frames + chunk_frames - 1) / chunk_frames;
  }
  // fp64 staging follows numpy's arithmetic (stft_np_kernel); SHZ_F64_OWN_FFT=1: this file's own transform, as before round 4
  static const bool own_f64 = [] { const char* e = getenv("SHZ_F64_OWN_FFT"); return e && atoi(e) != 0; }();
  if (sizeof(T) == 8 && !own_f64) {
    stft_args c = a;   // a clip or two (the per-clip fallback): one frame a workgroup, so that the whole chip takes part
    c.frames_per_wg = a.total_frames >= 4u * NP_FRAMES_PER_WG * (uint32_t)ctx->prop.multiProcessorCount ? NP_FRAMES_PER_WG : 1u;
    hipLaunchKernelGGL(stft_np_kernel, dim3((a.total_frames + c.frames_per_wg - 1) / c.frames_per_wg), dim3(256), 0, ctx->stream, c);
    SHZ_HIP(ctx, hipGetLastError());
    return SHZ_OK;
  }
  hipLaunchKernelGGL(stft_psd_kernel<T>, dim3(grid), dim3(256), 0, ctx->stream, b);
  SHZ_HIP(ctx, hipGetLastError());
  return SHZ_OK;
}

// fp64 threshold band around amp_min (power domain): above p_hi surely `> amp_min` dB, at or below p_lo surely not
static void threshold_band(double amp_min, double* p_lo, double* p_hi) {
  const double thr = pow(10.0, amp_min / 10.0);
  *p_lo = thr * (1.0 - 1e-9);
  *p_hi = thr * (1.0 + 1e-9);
}

// peaks of a frame-major fp64 buffer -> device peak list (ws PEAK_F / PEAK_T) + per-clip offsets (ws PEAK_CLIP).
// Host-synchronous helper of the stage APIs (shz_peaks_from_db); the batch path is extract_pass below.
static int32_t run_peaks(shz_ctx* ctx, const double* d_db, uint32_t row_stride, uint32_t n_bins, const sub_dev& sd,
                         uint32_t nc, uint32_t frames, double amp_min, bool is_power, uint16_t** d_pf, uint32_t** d_pt,
                         uint32_t** d_pcoff, uint32_t* n_peaks) {
  mask_geom mg = MG_F64;
  mg.n_slabs = (n_bins + PK_SW - 1) / PK_SW;
  const uint64_t n_words = (uint64_t)frames * mg.n_slabs * 4;
  void *d_mask, *d_woff, *d_tot;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MASK, n_words * 8, &d_mask));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SCAN, n_words * 4, &d_woff));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC0, 64, &d_tot));
  double p_lo, p_hi;
  threshold_band(amp_min, &p_lo, &p_hi);
  {
    shz_prof_scope ps(ctx, 1);
    if (is_power)
      hipLaunchKernelGGL(peak_pick_kernel<true>, dim3(mg.n_slabs, sd.n_segs), dim3(256), 0, ctx->stream, d_db, row_stride,
                         n_bins, sd.d_segs, mg.n_slabs, amp_min, p_lo, p_hi, (uint64_t*)d_mask);
    else
      hipLaunchKernelGGL(peak_pick_kernel<false>, dim3(mg.n_slabs, sd.n_segs), dim3(256), 0, ctx->stream, d_db, row_stride,
                         n_bins, sd.d_segs, mg.n_slabs, amp_min, 0.0, 0.0, (uint64_t*)d_mask);
    SHZ_HIP(ctx, hipGetLastError());
  }
  if (amp_min < 0.0 && n_words) {  // see peak_zero_plateau_kernel
    hipLaunchKernelGGL(peak_zero_plateau_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx->stream,
                       (uint64_t*)d_mask, n_words, mg, d_db, row_stride, n_bins, is_power ? 1.0 : 0.0, sd.d_foff, nc);
    SHZ_HIP(ctx, hipGetLastError());
  }
  uint64_t tot = 0;
  {
    shz_prof_scope ps(ctx, 2);
    SHZ_TRY(shz_scan_popc64(ctx, (const uint64_t*)d_mask, (uint32_t*)d_woff, n_words, (uint64_t*)d_tot));
    SHZ_HIP(ctx, shz_memcpy(ctx, &tot, d_tot, 8, hipMemcpyDeviceToHost));
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    void *pf, *pt, *pc;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_F, tot * 2 + 64, &pf));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_T, tot * 4 + 64, &pt));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_CLIP, (uint64_t)(nc + 1) * 4 + 64, &pc));
    if (n_words) {
      if (n_words >= (1ull << 32)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "too many mask words in one sub-batch");
      void* ft;
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, (uint64_t)frames * 4 + 64, &ft));
      hipLaunchKernelGGL(frame_time_kernel, dim3((frames + 255) / 256), dim3(256), 0, ctx->stream, sd.d_foff, nc, frames,
                         (uint32_t*)ft);
      hipLaunchKernelGGL(peak_expand_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx->stream,
                         (const uint64_t*)d_mask, (const uint32_t*)d_woff, (uint32_t)n_words, mg, (const uint32_t*)ft,
                         (uint16_t*)pf, (uint32_t*)pt, (uint32_t)tot);
      SHZ_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(gather_offsets_kernel, dim3((nc + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                       (const uint32_t*)d_woff, (const uint64_t*)d_tot, sd.d_foff, (uint64_t)mg.n_slabs * 4, n_words, nc,
                       (uint32_t*)pc, 0xFFFFFFFFu);
    SHZ_HIP(ctx, hipGetLastError());
    *d_pf = (uint16_t*)pf;
    *d_pt = (uint32_t*)pt;
    *d_pcoff = (uint32_t*)pc;
  }
  *n_peaks = (uint32_t)tot;
  return SHZ_OK;
}

// pair hashing of a device peak list (host-synchronous helper of shz_pair_hash); writes into (d_key, d_t1)
static int32_t run_pairs(shz_ctx* ctx, const uint16_t* d_pf, const uint32_t* d_pt, const uint32_t* d_pcoff, uint32_t nc,
                         uint32_t n_peaks, uint32_t fan, uint32_t* d_key, uint32_t* d_t1, uint64_t out_base, uint64_t cap,
                         uint64_t* n_hashes, std::vector<uint32_t>* clip_hoff /* nc+1, relative */) {
  shz_prof_scope ps(ctx, 3);
  void *d_cnt, *d_hoff, *d_tot, *d_choff;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_HCNT, (uint64_t)n_peaks * 4 + 64, &d_cnt));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_HOFF, (uint64_t)n_peaks * 4 + 64, &d_hoff));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, 64, &d_tot));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC2, (uint64_t)(nc + 1) * 4 + 64, &d_choff));
  unsigned long long* d_n = (unsigned long long*)d_tot + 1;
  hipLaunchKernelGGL(set_u64_dev_kernel, dim3(1), dim3(1), 0, ctx->stream, d_n, (unsigned long long)n_peaks);
  const unsigned nb = (n_peaks + 255) / 256;
  if (n_peaks) {
    hipLaunchKernelGGL(pair_count_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_pt, d_pcoff, nc, d_n, n_peaks, fan,
                       (uint32_t*)d_cnt);
    SHZ_HIP(ctx, hipGetLastError());
  }
  SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)d_cnt, (uint32_t*)d_hoff, n_peaks, (uint64_t*)d_tot));
  if (n_peaks) {
    hipLaunchKernelGGL(pair_write_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_pf, d_pt, d_pcoff, nc, d_n, n_peaks, fan,
                       (const uint32_t*)d_hoff, d_key, d_t1, out_base, (const unsigned long long*)nullptr, cap);
    SHZ_HIP(ctx, hipGetLastError());
  }
  // per-clip hash offsets: hoff[pcoff[c]]
  hipLaunchKernelGGL(gather_offsets_kernel, dim3((nc + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                     (const uint32_t*)d_hoff, (const uint64_t*)d_tot, d_pcoff, (uint64_t)1, (uint64_t)n_peaks, nc,
                     (uint32_t*)d_choff, 0xFFFFFFFFu);
  SHZ_HIP(ctx, hipGetLastError());
  uint64_t tot = 0;
  clip_hoff->resize(nc + 1);
  SHZ_HIP(ctx, shz_memcpy(ctx, &tot, d_tot, 8, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, shz_memcpy(ctx, clip_hoff->data(), d_choff, (uint64_t)(nc + 1) * 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n_hashes = tot;
  return SHZ_OK;
}

static int32_t check_common(shz_ctx* ctx, const void* pcm, const uint64_t* clip_off, uint32_t n_clips, uint32_t fs) {
  if (!ctx) return SHZ_E_INVALID;
  if (!clip_off) SHZ_FAIL(ctx, SHZ_E_INVALID, "clip_off is NULL");
  if (fs == 0) SHZ_FAIL(ctx, SHZ_E_INVALID, "Fs must be > 0");
  if (n_clips && !pcm && clip_off[n_clips] > clip_off[0]) SHZ_FAIL(ctx, SHZ_E_INVALID, "pcm is NULL");
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  return SHZ_OK;
}

extern "C" int32_t shz_stft_db(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t n_clips,
                               uint32_t fs, uint32_t flags, double* out_db, uint64_t cap_doubles, uint64_t* count) {
  SHZ_TRY(check_common(ctx, pcm, clip_off, n_clips, fs));
  uint64_t need = 0;
  for (uint32_t c = 0; c < n_clips; ++c) need += (uint64_t)frames_hop(clip_off[c + 1] - clip_off[c], ctx->hop) * SHZ_NBINS;
  if (count) *count = need;
  if (need > cap_doubles || (need && !out_db)) SHZ_FAIL(ctx, SHZ_E_CAPACITY, "shz_stft_db: need %llu doubles", (unsigned long long)need);
  std::vector<sub_batch> subs;
  SHZ_TRY(plan_sub_batches(ctx, clip_off, n_clips, (uint64_t)DB_STRIDE * 8, subs));
  std::vector<std::vector<uint64_t>> keep;
  uint64_t out_pos = 0;
  for (const sub_batch& sb : subs) {
    const uint32_t nc = sb.c1 - sb.c0;
    const int16_t* d_pcm;
    uint64_t base;
    SHZ_TRY(stage_pcm(ctx, pcm, clip_off, sb, flags, &d_pcm, &base));
    sub_dev sd;
    SHZ_TRY(upload_meta(ctx, clip_off, sb, base, MG_F64.n_slabs, 3, sd, keep));
    void *d_db, *d_tr;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_DB, (uint64_t)sb.frames * DB_STRIDE * 8, &d_db));
    SHZ_TRY(launch_stft<double>(ctx, make_stft_args(ctx, d_pcm, sd, nc, sb.frames, fs, d_db)));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC3, (uint64_t)sb.frames * SHZ_NBINS * 8, &d_tr));
    uint64_t tr_pos = 0;
    for (uint32_t i = 0; i < nc; ++i) {
      const uint32_t F = sd.foff[i + 1] - sd.foff[i];
      dim3 grid((SHZ_NBINS + 31) / 32, (F + 31) / 32);
      hipLaunchKernelGGL(transpose_db_kernel, grid, dim3(32, 8), 0, ctx->stream,
                         (const double*)d_db + (uint64_t)sd.foff[i] * DB_STRIDE, (uint32_t)DB_STRIDE, F,
                         (uint32_t)SHZ_NBINS, (double*)d_tr + tr_pos, (flags & SHZ_STFT_POWER) != 0);
      tr_pos += (uint64_t)F * SHZ_NBINS;
    }
    SHZ_HIP(ctx, hipGetLastError());
    SHZ_HIP(ctx, shz_memcpy(ctx, out_db + out_pos, d_tr, tr_pos * 8, hipMemcpyDeviceToHost));
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out_pos += tr_pos;
  }
  return SHZ_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Window sizes other than 4096 (fingerprint(..., wsize=...), __init__.py:212-217, 232-237): a GENERIC spectrogram, not fast
// -- nobody's hot path: the reference and all its callers use 4096 -- but with the reference's arithmetic like the fp64 path
// of the 4096 case (np_fft4096 above): numpy's window, pocketfft's plan for the size (for a power of two: radix-8 passes,
// then radix 4, and a single radix-2 pass that goes FIRST: 2048 = 8.8.8.4, 1024 = 2.8.8.8, 256 = 8.8.4, 128 = 2.8.8), its
// pass2 / pass4 / pass8 butterflies operation by operation, numpy's complex product and mlab's scaling.  One workgroup per
// frame, two LDS buffers.  Output in the reference's [bins][frames] layout, so that get_2D_peaks / generate_hashes
// (shz_peaks_from_db, shz_pair_hash) take it from there as the reference's own fingerprint() composes them.  nfft: a power
// of two in [64, 2048] (two buffers of nfft complex doubles are 64 KB of LDS at 2048; 8192 has no packed key: key32 gives a
// frequency 12 bits).
#define ANY_MAX_PASSES 6
struct any_plan {
  uint32_t npass;
  uint32_t ip[ANY_MAX_PASSES], l1[ANY_MAX_PASSES], ido[ANY_MAX_PASSES], twoff[ANY_MAX_PASSES];   // twoff: into tw, [c - 1][i] = comp[c l1 i]
};

__global__ __launch_bounds__(256) void stft_any_kernel(const int16_t* __restrict__ pcm, uint64_t n, uint32_t nfft, uint32_t hop,
                                                       uint32_t F, const double* __restrict__ window, const cplx* __restrict__ tw,
                                                       any_plan pl, double r_fs, double r_s, int as_power, uint32_t unfused,
                                                       double* __restrict__ out) {
  extern __shared__ cplx gs_lds[];
  cplx* a = gs_lds;
  cplx* b = gs_lds + nfft;
  const uint32_t f = blockIdx.x, tid = threadIdx.x, half = nfft >> 1;
  const uint64_t s0 = (uint64_t)f * hop;
  for (uint32_t i = tid; i < nfft; i += 256) {
    const uint64_t sidx = s0 + i;
    const double x = sidx < n ? (double)pcm[sidx] : 0.0;   // (only a clip shorter than one window is padded, mlab:268-271)
    a[i] = make_double2(x * window[i], 0.0);
  }
  __syncthreads();
  for (uint32_t p = 0; p < pl.npass; ++p) {   // uniform
    const uint32_t ip = pl.ip[p], l1 = pl.l1[p], ido = pl.ido[p];
    const cplx* __restrict__ twp = tw + pl.twoff[p];
    for (uint32_t bf = tid; bf < l1 * ido; bf += 256) {
      const uint32_t k = bf / ido, i = bf - k * ido;
      // inputs cc[i + ido (q + ip k)], outputs ch[i + ido (k + l1 c)], c >= 1 and i > 0 times conj(comp[c l1 i])
      if (ip == 8) {
        cplx c[8], o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = a[i + ido * (q + 8 * k)];
        np_bfly8(c, o);
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) b[i + ido * (k + l1 * cc)] = (cc && i) ? np_smul(o[cc], twp[(cc - 1) * ido + i]) : o[cc];
      } else if (ip == 4) {
        const cplx c0 = a[i + ido * (4 * k)], c1 = a[i + ido * (1 + 4 * k)], c2 = a[i + ido * (2 + 4 * k)], c3 = a[i + ido * (3 + 4 * k)];
        const cplx t2 = cadd(c0, c2), t1 = csub(c0, c2);            // PM(t2, t1, c0, c2)
        const cplx t3 = cadd(c1, c3);
        cplx t4 = csub(c1, c3);                                     // PM(t3, t4, c1, c3)
        t4 = make_double2(t4.y, -t4.x);                             // ROTX90<fwd>(t4)
        const cplx o0 = cadd(t2, t3), o1 = cadd(t1, t4), o2 = csub(t2, t3), o3 = csub(t1, t4);
        b[i + ido * k] = o0;
        b[i + ido * (k + l1)] = i ? np_smul(o1, twp[i]) : o1;
        b[i + ido * (k + 2 * l1)] = i ? np_smul(o2, twp[ido + i]) : o2;
        b[i + ido * (k + 3 * l1)] = i ? np_smul(o3, twp[2 * ido + i]) : o3;
      } else {
        const cplx c0 = a[i + ido * (2 * k)], c1 = a[i + ido * (1 + 2 * k)];
        const cplx o1 = csub(c0, c1);
        b[i + ido * k] = cadd(c0, c1);
        b[i + ido * (k + l1)] = i ? np_smul(o1, twp[i]) : o1;
      }
    }
    __syncthreads();
    cplx* t = a; a = b; b = t;
  }
  for (uint32_t k = tid; k <= half; k += 256) {
    const cplx X = a[k];
    double p = unfused ? X.x * X.x + X.y * X.y : fma(X.x, X.x, X.y * X.y);
    if (k != 0 && k != half) p *= 2.0;
    p = p * r_fs;
    p = p * r_s;
    out[(uint64_t)k * F + f] = as_power ? p : (p != 0.0 ? shz_db_of(p) : 0.0);
  }
}

extern "C" int32_t shz_stft_db_any(shz_ctx* ctx, const int16_t* pcm, uint64_t n_samples, uint32_t fs, uint32_t nfft, uint32_t noverlap,
                                   uint32_t flags, double* out_db, uint64_t cap_doubles, uint64_t* n_frames) {
  if (!ctx) return SHZ_E_INVALID;
  if (n_frames) *n_frames = 0;
  if (!pcm && n_samples) SHZ_FAIL(ctx, SHZ_E_INVALID, "pcm is NULL");
  if (fs == 0) SHZ_FAIL(ctx, SHZ_E_INVALID, "fs must be positive");
  if (nfft < 64 || nfft > 2048 || (nfft & (nfft - 1)))
    SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "the generic spectrogram takes window sizes 64, 128, ..., 2048 (4096 is shz_stft_db; got %u)", nfft);
  if (noverlap >= nfft) SHZ_FAIL(ctx, SHZ_E_INVALID, "noverlap must be less than NFFT (%u >= %u)", noverlap, nfft);   // mlab:242
  if (n_samples == 0) SHZ_FAIL(ctx, SHZ_E_INVALID, "no samples");
  const uint32_t hop = nfft - noverlap;
  const uint64_t F64 = n_samples < nfft ? 1 : (n_samples - nfft) / hop + 1;   // mlab:268-271, 307-308
  if (F64 > (1ull << 24)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "%llu frames", (unsigned long long)F64);
  const uint32_t F = (uint32_t)F64, bins = nfft / 2 + 1;
  if (n_frames) *n_frames = F;
  if ((uint64_t)F * bins > cap_doubles || !out_db) SHZ_FAIL(ctx, SHZ_E_CAPACITY, "shz_stft_db_any: need %llu doubles", (unsigned long long)F * bins);
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  // numpy's tables for this size (shz_numpy_tables), the twiddles laid out pass by pass; pocketfft's factor list of a power
  // of two: 8s, then 4s, then one 2 -- which is moved to the FRONT
  std::vector<double> win(nfft);
  std::vector<double2> comp(nfft), tw(nfft);
  double sumsq = 0.0;
  shz_numpy_tables_host(nfft, win.data(), comp.data(), &sumsq);
  any_plan pl;
  memset(&pl, 0, sizeof(pl));
  {
    uint32_t fct[ANY_MAX_PASSES], nf = 0, len = nfft;
    while ((len & 7) == 0) { fct[nf++] = 8; len >>= 3; }
    while ((len & 3) == 0) { fct[nf++] = 4; len >>= 2; }
    if ((len & 1) == 0) { len >>= 1; fct[nf++] = 2; std::swap(fct[0], fct[nf - 1]); }
    uint32_t l1 = 1, off = 0;
    for (uint32_t p = 0; p < nf; ++p) {
      const uint32_t ip = fct[p], ido = nfft / (l1 * ip);
      pl.ip[p] = ip; pl.l1[p] = l1; pl.ido[p] = ido; pl.twoff[p] = off;
      for (uint32_t c = 1; c < ip; ++c)
        for (uint32_t i = 0; i < ido; ++i) tw[off + (c - 1) * ido + i] = comp[(size_t)c * l1 * i];
      off += (ip - 1) * ido;
      l1 *= ip;
    }
    pl.npass = nf;
  }
  void *d_pcm, *d_win, *d_tw, *d_out;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PCM, n_samples * 2 + 64, &d_pcm));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC0, (uint64_t)nfft * 8 + 64, &d_win));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, (uint64_t)nfft * 16 + 64, &d_tw));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_DB, (uint64_t)F * bins * 8 + 64, &d_out));
  SHZ_HIP(ctx, shz_memcpy(ctx, d_pcm, pcm, n_samples * 2, hipMemcpyHostToDevice));
  SHZ_HIP(ctx, shz_memcpy(ctx, d_win, win.data(), (uint64_t)nfft * 8, hipMemcpyHostToDevice));
  SHZ_HIP(ctx, shz_memcpy(ctx, d_tw, tw.data(), (uint64_t)nfft * 16, hipMemcpyHostToDevice));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (the host tables go out of scope)
  // P * 2 (bins 1 .. N/2 - 1), / Fs, / sum(w^2) as numpy does them: the divisions are products with the rounded reciprocals
  hipLaunchKernelGGL(stft_any_kernel, dim3(F), dim3(256), (size_t)nfft * 2 * sizeof(cplx), ctx->stream, (const int16_t*)d_pcm, n_samples,
                     nfft, hop, F, (const double*)d_win, (const cplx*)d_tw, pl, 1.0 / (double)fs, 1.0 / sumsq,
                     (flags & SHZ_STFT_POWER) ? 1 : 0, ctx->np_unfused ? 1u : 0u, (double*)d_out);
  SHZ_HIP(ctx, hipGetLastError());
  SHZ_HIP(ctx, shz_memcpy(ctx, out_db, d_out, (uint64_t)F * bins * 8, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SHZ_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The batch path: shz_peaks / shz_fingerprint_batch.  One pass = every sub-batch queued on the stream with NO host
// read-back in between: peak counts, hash counts, output offsets and overflow flags live in the device control block
// (xctl) and come back with ONE copy + sync at the end.  List capacities are estimates; a pass whose flags say that an
// estimate was too small, or that fp32 staging cannot decide the input (XF_FALLBACK), is repeated with the measured
// sizes / with fp64 staging.
struct xparams {
  bool persistent_stft = false;  // dual pass: persistent STFT grid
  bool f32;                    // fp32 staging + verification (default) or fp64 staging with the exact test in-kernel
  uint32_t peaks_per_frame;    // peak list capacity per frame of a sub-batch
  uint64_t stage_cap;          // host output: capacity of the device staging arrays (entries)
};

#define UND_CAP (1u << 20)

template <int NW, int OCC, int AH = 1>
static void launch_pick32(shz_ctx* ctx, const p32_args& pa, uint32_t n_segs) {
  const uint32_t per_xcd = (n_segs + 7) >> 3;   // segments per XCD; 8 * per_xcd * n_slabs workgroups, see the kernel's work map
  hipLaunchKernelGGL((peak_pick32_kernel<NW, OCC, AH>), dim3(8 * per_xcd * pa.n_slabs), dim3(64 * NW), 0, ctx->stream, pa);
}
static int p32_occ() {
  static const int v = [] { const char* e = getenv("SHZ_PEAK_OCC"); const int x = e ? atoi(e) : 4; return x >= 3 && x <= 6 ? x : 4; }();
  return v;
}

// what extract_enqueue leaves for extract_finish: where the read-back lands and where the entries are
struct pass_tail {
  char* mp = nullptr;                      // pinned mailbox: xctl | offsets | (small host outputs) entries
  uint64_t off_offs = 0, off_a = 0, off_b = 0, spec = 0, b_a = 4, o_cap = 0;
  void *o_a = nullptr, *o_b = nullptr;     // device arrays the entries were written to
  uint32_t n_clips = 0;
  bool out_dev = false, want_hashes = true, stay = false;
  const uint32_t* d_fb = nullptr;          // device bitmap of the clips flagged XF_FALLBACK_CLIP
};

// Queue one pass on ctx->stream, up to and including the asynchronous copies of its read-back.  `stay`: the entries stay
// in the context's staging arrays (pt->o_a / o_b) whatever the flags say -- the second pipeline of a dual pass.
static int32_t extract_enqueue(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t n_clips, uint32_t fs,
                               double amp_min, uint32_t fan, uint32_t flags, bool want_hashes, const xparams& xp,
                               uint16_t* peak_f, uint32_t* peak_t, uint32_t* key32, uint32_t* t1, uint64_t cap, bool stay,
                               pass_tail* pt) {
  const bool out_dev = (flags & SHZ_OUT_DEVICE) != 0 && !stay;
  const mask_geom mg = xp.f32 ? mg_f32(p32_nw()) : MG_F64;
  // Two-stream pipeline (fp32 staging; OFF unless SHZ_OVERLAP_SPLIT >= 2): the batch is cut into that many sub-batches
  // and the STFT of sub-batch i+1 runs on a second stream beside peak picking and pair hashing of sub-batch i.
  // stft_psd is VALU/LDS-bound with HBM idle, peak_pick32 waits on memory with the VALU idle, and at two STFT workgroups
  // per CU both fit a CU together -- but measured on 1,000 x 30 s clips the step does not get shorter: 7.07 ms in
  // sequence, 7.10 with 2 sub-batches, 8.05 with 4, 7.58 with 8, 9.03 with 16.  peak_pick32 hides its memory latency
  // with many waves, and beside the STFT it gets one wave per SIMD (2.1 -> 4-6 ms while the STFT goes 4.4 -> 4.7).
  static const int ov_split = [] { const char* e = getenv("SHZ_OVERLAP_SPLIT"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : v; }();
  uint64_t frames_total = 0;
  for (uint32_t c = 0; c < n_clips; ++c) frames_total += frames_hop(clip_off[c + 1] - clip_off[c], ctx->hop);
  const bool want_overlap = xp.f32 && ov_split >= 2 && frames_total >= 65536;
  std::vector<sub_batch> subs;
  SHZ_TRY(plan_sub_batches(ctx, clip_off, n_clips, xp.f32 ? (uint64_t)P32_STRIDE * 4 : (uint64_t)DB_STRIDE * 8, subs,
                           want_overlap ? (frames_total + ov_split - 1) / ov_split : 0));
  const bool overlap = want_overlap && subs.size() >= 2;
  if (overlap && !ctx->stream2) {
    SHZ_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      SHZ_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_stft[i], hipEventDisableTiming));
      SHZ_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_free[i], hipEventDisableTiming));
    }
  }
  if (overlap) {  // the second stream starts behind everything queued on the first one (the caller's PCM may still be in the making)
    SHZ_HIP(ctx, hipEventRecord(ctx->ev_free[0], ctx->stream));
    SHZ_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_free[0], 0));
  }
  // whatever way this pass ends, the second stream is idle afterwards (workspace slots may be re-allocated by the next call)
  struct s2_guard { shz_ctx* c; bool on; ~s2_guard() { if (on && c->stream2) (void)hipStreamSynchronize(c->stream2); } } s2g{ctx, overlap};
  void *p_ctl, *p_offs;
  const uint64_t fb_words = ((uint64_t)n_clips + 31) / 32;   // bitmap of clips that need fp64 staging, behind the control block
  // A small pass with host outputs keeps everything that is read back in ONE device block, laid out like the mailbox:
  // control block | clip bitmap | per-clip offsets | entries a | entries b -- one copy at the end instead of four
  // (a copy is ~5 us of a 120 us call).
  const uint64_t b_a = want_hashes ? 4 : 2;
  const bool one_block = !out_dev && !stay && xp.stage_cap <= (1u << 15);
  const uint64_t ob_offs = 256 + ((fb_words * 4 + 255) & ~255ull);
  const uint64_t ob_a = ob_offs + (((uint64_t)(n_clips + 1) * 8 + 255) & ~255ull);
  const uint64_t ob_b = ob_a + ((xp.stage_cap * b_a + 255) & ~255ull);
  const uint64_t ob_bytes = ob_b + xp.stage_cap * 4;
  void* p_block = nullptr;
  if (one_block) {
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_KEY, ob_bytes + 64, &p_block));
    p_ctl = p_block;
    p_offs = (char*)p_block + ob_offs;
  } else {
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_CTL, 256 + fb_words * 4 + 64, &p_ctl));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_OFFS, (uint64_t)(n_clips + 1) * 8 + 64, &p_offs));
  }
  xctl* d_ctl = (xctl*)p_ctl;
  unsigned long long* d_offs = (unsigned long long*)p_offs;
  static_assert(sizeof(xctl) <= 256, "the clip bitmap starts 256 bytes behind the control block");
  uint32_t* d_fb = (uint32_t*)((char*)p_ctl + 256);
  SHZ_HIP(ctx, hipMemsetAsync(d_ctl, 0, 256 + ((fb_words * 4 + 63) & ~63ull), ctx->stream));   // (whole 64-byte lines: one fill, not a body and a tail)   // (offs[0] = 0 is written by xctl_offsets_kernel)
  // where the entries go: the caller's device arrays, or staging arrays that are copied out after the final sync
  void *o_a = want_hashes ? (void*)key32 : (void*)peak_f, *o_b = want_hashes ? (void*)t1 : (void*)peak_t;
  uint64_t o_cap = cap;
  if (one_block) {
    o_cap = xp.stage_cap;
    o_a = (char*)p_block + ob_a;
    o_b = (char*)p_block + ob_b;
  } else if (!out_dev) {
    o_cap = xp.stage_cap;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_KEY, o_cap * 4 + 64, &o_a));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_T1, o_cap * 4 + 64, &o_b));
  }
  double p_lo, p_hi;
  threshold_band(amp_min, &p_lo, &p_hi);
  std::vector<std::vector<uint64_t>> keep;
  // stage A of a sub-batch: PCM + tables on the device, STFT.  With the pipeline on it is issued on the second stream,
  // into the buffers of the sub-batch's parity, one sub-batch ahead of stage B.
  struct sub_state { sub_dev sd; const int16_t* d_pcm; void* d_pw; stft_args sa; };
  std::vector<sub_state> st(subs.size());
  const uint64_t pw_bytes_per_frame = xp.f32 ? (uint64_t)P32_STRIDE * 4 : (uint64_t)DB_STRIDE * 8;
  if (overlap) {  // every buffer of both parities at its final size before anything is in flight on two streams
    uint64_t fmax = 0, smax = 0, cmax = 0;
    for (const sub_batch& sb : subs) {
      fmax = std::max<uint64_t>(fmax, sb.frames);
      smax = std::max<uint64_t>(smax, clip_off[sb.c1] - clip_off[sb.c0]);
      cmax = std::max<uint64_t>(cmax, sb.c1 - sb.c0);
    }
    void* dummy;
    const uint64_t meta_bound = (3 * cmax + fmax / PK_SEG + cmax + 64) * 16;
    for (int par = 0; par < 2; ++par) {
      SHZ_TRY(shz_ws_reserve(ctx, par ? SHZ_WS_DB2 : SHZ_WS_DB, fmax * pw_bytes_per_frame, &dummy));
      SHZ_TRY(shz_ws_reserve(ctx, par ? SHZ_WS_META_B : SHZ_WS_META, meta_bound, &dummy));
      if (!(flags & SHZ_PCM_DEVICE)) SHZ_TRY(shz_ws_reserve(ctx, par ? SHZ_WS_PCM_B : SHZ_WS_PCM, smax * 2 + 64, &dummy));
    }
  }
  auto stage_a = [&](size_t i) -> int32_t {
    const sub_batch& sb = subs[i];
    const int par = overlap ? (int)(i & 1) : 0;
    hipStream_t main_stream = ctx->stream;
    struct swap_guard { shz_ctx* c; hipStream_t keep; ~swap_guard() { c->stream = keep; } } sg{ctx, main_stream};
    if (overlap) {
      ctx->stream = ctx->stream2;   // copies, profiling events and the launch below go to the second stream
      if (i >= 2) SHZ_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_free[par], 0));   // stage B of sub-batch i-2 is done with these buffers
    }
    sub_state& x = st[i];
    uint64_t base;
    SHZ_TRY(stage_pcm(ctx, pcm, clip_off, sb, flags, &x.d_pcm, &base, par ? SHZ_WS_PCM_B : SHZ_WS_PCM));
    SHZ_TRY(upload_meta(ctx, clip_off, sb, base, mg.n_slabs, xp.f32 ? 12 / mg.nw : 3, x.sd, keep, par ? SHZ_WS_META_B : SHZ_WS_META));
    SHZ_TRY(shz_ws_reserve(ctx, par ? SHZ_WS_DB2 : SHZ_WS_DB, (uint64_t)sb.frames * pw_bytes_per_frame, &x.d_pw));
    x.sa = make_stft_args(ctx, x.d_pcm, x.sd, sb.c1 - sb.c0, sb.frames, fs, x.d_pw);
    static const int ov_wgs = [] { const char* e = getenv("SHZ_OVERLAP_STFT_WGS"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 3 ? v : 2; }();
    if (xp.f32) SHZ_TRY(launch_stft<float>(ctx, x.sa, overlap ? ov_wgs : 0, xp.persistent_stft || overlap));
    else SHZ_TRY(launch_stft<double>(ctx, x.sa, 0, xp.persistent_stft));
    if (overlap) SHZ_HIP(ctx, hipEventRecord(ctx->ev_stft[par], ctx->stream2));
    return SHZ_OK;
  };
  SHZ_TRY(stage_a(0));
  for (size_t si = 0; si < subs.size(); ++si) {
    const sub_batch& sb = subs[si];
    if (overlap) {
      if (si + 1 < subs.size()) SHZ_TRY(stage_a(si + 1));
      SHZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_stft[si & 1], 0));
    } else if (si > 0) {
      SHZ_TRY(stage_a(si));
    }
    const uint32_t nc = sb.c1 - sb.c0;
    const sub_dev& sd = st[si].sd;
    void* d_pw = st[si].d_pw;
    const uint64_t n_words = (uint64_t)sb.frames * mg.n_slabs * mg.nw;
    if (n_words >= (1ull << 32)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "too many mask words in one sub-batch");
    const uint64_t cap_peaks64 = (uint64_t)sb.frames * xp.peaks_per_frame + 4096;
    if (cap_peaks64 * (fan > 1 ? fan - 1 : 1) >= (1ull << 32))
      SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "sub-batch of %u frames may yield 2^32 hashes; lower the workspace limit", sb.frames);
    const uint32_t cap_peaks = (uint32_t)cap_peaks64;
    void *d_mask, *d_woff, *d_und, *pf, *pt, *pc, *ft, *d_fcnt = nullptr;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MASK, n_words * 8, &d_mask));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_SCAN, n_words * 4, &d_woff));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_F, (uint64_t)cap_peaks * 2 + 64, &pf));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_T, (uint64_t)cap_peaks * 4 + 64, &pt));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_CLIP, (uint64_t)(nc + 1) * 4 + 64, &pc));
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC1, (uint64_t)sb.frames * 4 + 64, &ft));
    const stft_args& sa = st[si].sa;
    static const bool no_small_tail = [] { const char* e = getenv("SHZ_NO_SMALL_TAIL"); return e && atoi(e) != 0; }();
    const bool small_tail = want_hashes && !no_small_tail && n_words && n_words <= XT_MAX_WORDS && cap_peaks <= XT_MAX_PEAKS;
    if (xp.f32) {
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_UND, (uint64_t)UND_CAP * 8, &d_und));
      if (si) hipLaunchKernelGGL(xctl_begin_sub_kernel, dim3(1), dim3(1), 0, ctx->stream, d_ctl);   // (the block starts zeroed)
      SHZ_HIP(ctx, hipMemsetAsync(d_mask, 0, n_words * 8, ctx->stream));  // peak_pick32 writes non-zero words only
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC3, (uint64_t)sb.frames * 4 + 64, &d_fcnt));
      if (!small_tail) SHZ_HIP(ctx, hipMemsetAsync(d_fcnt, 0, (uint64_t)sb.frames * 4, ctx->stream));   // (the one-workgroup tail counts from the mask)
      {
        shz_prof_scope ps(ctx, 1);
        p32_args pa;
        pa.A = (const float*)d_pw;
        pa.segs = sd.d_segs;
        pa.n_slabs = mg.n_slabs;
        pa.n_segs = sd.n_segs;
        pa.p_lo = (float)p_lo;   // fp32(P) < fp32(p_lo) => P < p_lo; fp32(P) > fp32(p_hi) => P > p_hi (monotone rounding)
        pa.p_hi = (float)p_hi;
        pa.mask = (uint64_t*)d_mask;
        pa.und_list = (uint64_t*)d_und;
        pa.ctl = d_ctl;
        pa.und_cap = UND_CAP;
        pa.frame_cnt = (uint32_t*)d_fcnt;
        static const int p32_ahead = [] { const char* e = getenv("SHZ_PEAK_AHEAD"); return e ? atoi(e) : 1; }();
        if (p32_ahead == 2 && mg.nw == 2) {   // experiment: 14 rows in flight per lane (3 or 4 waves per SIMD by SHZ_PEAK_OCC)
          if (p32_occ() == 3) launch_pick32<2, 3, 2>(ctx, pa, sd.n_segs); else launch_pick32<2, 4, 2>(ctx, pa, sd.n_segs);
        } else
        switch (mg.nw * 10 + p32_occ()) {
          case 13: launch_pick32<1, 3>(ctx, pa, sd.n_segs); break;
          case 14: launch_pick32<1, 4>(ctx, pa, sd.n_segs); break;
          case 23: launch_pick32<2, 3>(ctx, pa, sd.n_segs); break;
          case 33: launch_pick32<3, 3>(ctx, pa, sd.n_segs); break;
          case 34: launch_pick32<3, 4>(ctx, pa, sd.n_segs); break;
          case 43: launch_pick32<4, 3>(ctx, pa, sd.n_segs); break;
          case 44: launch_pick32<4, 4>(ctx, pa, sd.n_segs); break;
          case 63: launch_pick32<6, 3>(ctx, pa, sd.n_segs); break;
          case 73: launch_pick32<7, 3>(ctx, pa, sd.n_segs); break;
          default: launch_pick32<2, 4>(ctx, pa, sd.n_segs); break;  // 128 VGPRs, 4 waves per SIMD: 2.12 ms vs 2.49 at 3
        }
        SHZ_HIP(ctx, hipGetLastError());
      }
      {
        shz_prof_scope ps(ctx, 4);
        verify_args va;
        va.st = sa;
        va.A = (const float*)d_pw;
        va.mask = (uint64_t*)d_mask;
        va.frame_cnt = (uint32_t*)d_fcnt;
        va.mg = mg;
        va.und_list = (const uint64_t*)d_und;
        va.ctl = d_ctl;
        va.und_cap = UND_CAP;
        va.p_hi32 = (float)p_hi;
        va.p_lo = p_lo;
        va.p_hi = p_hi;
        va.amp_min = amp_min;
        va.fb_clips = d_fb;
        va.clip0 = sb.c0;
        hipLaunchKernelGGL(peak_verify_kernel, dim3((unsigned)ctx->prop.multiProcessorCount), dim3(256), 0,
                           ctx->stream, va);
        SHZ_HIP(ctx, hipGetLastError());
      }
    } else {
      {
        shz_prof_scope ps(ctx, 1);
        hipLaunchKernelGGL(peak_pick_kernel<true>, dim3(mg.n_slabs, sd.n_segs), dim3(256), 0, ctx->stream,
                           (const double*)d_pw, (uint32_t)DB_STRIDE, (uint32_t)SHZ_NBINS, sd.d_segs, mg.n_slabs, amp_min,
                           p_lo, p_hi, (uint64_t*)d_mask);
        SHZ_HIP(ctx, hipGetLastError());
      }
      if (amp_min < 0.0 && n_words) {  // see peak_zero_plateau_kernel
        hipLaunchKernelGGL(peak_zero_plateau_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx->stream,
                           (uint64_t*)d_mask, n_words, mg, (const double*)d_pw, (uint32_t)DB_STRIDE, (uint32_t)SHZ_NBINS,
                           1.0, sd.d_foff, nc);
        SHZ_HIP(ctx, hipGetLastError());
      }
    }
    if (small_tail) {
      shz_prof_scope ps(ctx, 3);
      void* d_hoff;
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_HOFF, (uint64_t)cap_peaks * 4 + 64, &d_hoff));
      hipLaunchKernelGGL(extract_tail_small_kernel, dim3(1), dim3(XT_THREADS), 0, ctx->stream, (const uint64_t*)d_mask,
                         (uint32_t)n_words, mg, (const uint32_t*)sd.d_foff, nc, sb.c0, d_ctl, cap_peaks, (uint32_t)UND_CAP, fan,
                         (uint16_t*)pf, (uint32_t*)pt, (uint32_t*)pc, (uint32_t*)d_hoff, (uint32_t*)o_a, (uint32_t*)o_b, o_cap, d_offs);
      SHZ_HIP(ctx, hipGetLastError());
      if (overlap) SHZ_HIP(ctx, hipEventRecord(ctx->ev_free[si & 1], ctx->stream));
      continue;
    }
    {
      shz_prof_scope ps(ctx, 2);
      // offsets of the peaks: fp32 staging counted them per frame (d_woff = exclusive scan of the frames' counts),
      // the fp64 kernels leave the mask only (d_woff = exclusive scan of the words' popcounts)
      if (xp.f32) SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)d_fcnt, (uint32_t*)d_woff, sb.frames, (uint64_t*)&d_ctl->sub_peaks));
      else SHZ_TRY(shz_scan_popc64(ctx, (const uint64_t*)d_mask, (uint32_t*)d_woff, n_words, (uint64_t*)&d_ctl->sub_peaks));
      hipLaunchKernelGGL(xctl_after_peaks_kernel, dim3(1), dim3(1), 0, ctx->stream, d_ctl, cap_peaks, UND_CAP);
      if (n_words) {
        hipLaunchKernelGGL(frame_time_kernel, dim3((sb.frames + 255) / 256), dim3(256), 0, ctx->stream, sd.d_foff, nc,
                           sb.frames, (uint32_t*)ft);
        if (xp.f32)
          hipLaunchKernelGGL(peak_expand_frames_kernel, dim3((unsigned)((n_words + 256 * PXF_PER - 1) / (256 * PXF_PER))), dim3(256), 0, ctx->stream,
                             (const uint64_t*)d_mask, (const uint32_t*)d_woff, (uint32_t)n_words, mg, (const uint32_t*)ft,
                             (uint16_t*)pf, (uint32_t*)pt, cap_peaks);
        else
          hipLaunchKernelGGL(peak_expand_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, ctx->stream,
                             (const uint64_t*)d_mask, (const uint32_t*)d_woff, (uint32_t)n_words, mg, (const uint32_t*)ft,
                             (uint16_t*)pf, (uint32_t*)pt, cap_peaks);
      }
      hipLaunchKernelGGL(gather_offsets_kernel, dim3((nc + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                         (const uint32_t*)d_woff, (const uint64_t*)&d_ctl->sub_peaks, sd.d_foff,
                         xp.f32 ? (uint64_t)1 : (uint64_t)mg.n_slabs * mg.nw, xp.f32 ? (uint64_t)sb.frames : n_words, nc,
                         (uint32_t*)pc, cap_peaks);
      SHZ_HIP(ctx, hipGetLastError());
    }
    if (!want_hashes) {
      hipLaunchKernelGGL(xctl_offsets_kernel, dim3((nc + 255) / 256), dim3(256), 0, ctx->stream, d_ctl, false,
                         (const uint32_t*)pc, nc, sb.c0, d_offs);
      hipLaunchKernelGGL(peaks_out_kernel, dim3((cap_peaks + 255) / 256), dim3(256), 0, ctx->stream, d_ctl,
                         (const uint16_t*)pf, (const uint32_t*)pt, cap_peaks, (uint16_t*)o_a, (uint32_t*)o_b, o_cap);
      hipLaunchKernelGGL(xctl_advance_kernel, dim3(1), dim3(1), 0, ctx->stream, d_ctl, cap_peaks, false);
      SHZ_HIP(ctx, hipGetLastError());
      if (overlap) SHZ_HIP(ctx, hipEventRecord(ctx->ev_free[si & 1], ctx->stream));   // this parity's buffers are free again
      continue;
    }
    {
      shz_prof_scope ps(ctx, 3);
      void *d_cnt, *d_hoff, *d_choff;
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_HCNT, (uint64_t)cap_peaks * 4 + 64, &d_cnt));
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_HOFF, (uint64_t)cap_peaks * 4 + 64, &d_hoff));
      SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC2, (uint64_t)(nc + 1) * 4 + 64, &d_choff));
      const unsigned nb = (cap_peaks + 255) / 256;
      hipLaunchKernelGGL(pair_count_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const uint32_t*)pt, (const uint32_t*)pc,
                         nc, &d_ctl->sub_peaks, cap_peaks, fan, (uint32_t*)d_cnt);
      SHZ_TRY(shz_scan_u32(ctx, (const uint32_t*)d_cnt, (uint32_t*)d_hoff, cap_peaks, (uint64_t*)&d_ctl->sub_hashes));
      hipLaunchKernelGGL(pair_write_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const uint16_t*)pf, (const uint32_t*)pt,
                         (const uint32_t*)pc, nc, &d_ctl->sub_peaks, cap_peaks, fan, (const uint32_t*)d_hoff,
                         (uint32_t*)o_a, (uint32_t*)o_b, (uint64_t)0, &d_ctl->hash_base, o_cap);
      // per-clip hash offsets: hoff[pcoff[c]] (pcoff is clamped to cap_peaks; slot cap_peaks of the scan = the total)
      hipLaunchKernelGGL(gather_offsets_kernel, dim3((nc + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                         (const uint32_t*)d_hoff, (const uint64_t*)&d_ctl->sub_hashes, (const uint32_t*)pc, (uint64_t)1,
                         (uint64_t)cap_peaks, nc, (uint32_t*)d_choff, 0xFFFFFFFFu);
      hipLaunchKernelGGL(xctl_offsets_kernel, dim3((nc + 255) / 256), dim3(256), 0, ctx->stream, d_ctl, true,
                         (const uint32_t*)d_choff, nc, sb.c0, d_offs);
      hipLaunchKernelGGL(xctl_advance_kernel, dim3(1), dim3(1), 0, ctx->stream, d_ctl, cap_peaks, true);
      SHZ_HIP(ctx, hipGetLastError());
    }
    if (overlap) SHZ_HIP(ctx, hipEventRecord(ctx->ev_free[si & 1], ctx->stream));   // this parity's buffers are free again
  }
  // The one read-back of the pass, into pinned memory: control block | per-clip offsets | for small host outputs the
  // entries themselves (they ride along instead of costing a second round trip once the count is known).
  pt->b_a = b_a;
  pt->spec = one_block ? o_cap : 0;
  pt->off_offs = one_block ? ob_offs : 256;
  pt->off_a = one_block ? ob_a : pt->off_offs + (((uint64_t)(n_clips + 1) * 8 + 255) & ~255ull);
  pt->off_b = one_block ? ob_b : pt->off_a;
  void* mailp;
  SHZ_TRY(shz_mailbox(ctx, pt->off_b + pt->spec * 4, &mailp));
  pt->mp = (char*)mailp;
  pt->o_a = o_a;
  pt->o_b = o_b;
  pt->o_cap = o_cap;
  pt->n_clips = n_clips;
  pt->out_dev = out_dev;
  pt->want_hashes = want_hashes;
  pt->stay = stay;
  pt->d_fb = d_fb;
  if (one_block) {
    SHZ_HIP(ctx, hipMemcpyAsync(pt->mp, p_block, ob_bytes, hipMemcpyDeviceToHost, ctx->stream));
  } else {
    SHZ_HIP(ctx, hipMemcpyAsync(pt->mp, d_ctl, sizeof(xctl), hipMemcpyDeviceToHost, ctx->stream));
    SHZ_HIP(ctx, hipMemcpyAsync(pt->mp + pt->off_offs, d_offs, (uint64_t)(n_clips + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  s2g.on = false;   // the caller's extract_finish synchronises (both streams are joined by the events above)
  return SHZ_OK;
}

// wait for a queued pass; control block to *hctl, offsets to offs_out (n_clips + 1, may be null), host outputs copied
static int32_t extract_finish(shz_ctx* ctx, const pass_tail& pt, uint16_t* peak_f, uint32_t* peak_t, uint32_t* key32,
                              uint32_t* t1, uint64_t* offs_out, uint64_t cap, xctl* hctl) {
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->stream2) SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream2));
  memcpy(hctl, pt.mp, sizeof(xctl));
  if (offs_out) memcpy(offs_out, pt.mp + pt.off_offs, (uint64_t)(pt.n_clips + 1) * 8);
  const uint64_t total = pt.want_hashes ? hctl->hash_base : hctl->peak_base;
  const bool clean = !(hctl->flags & (XF_FALLBACK | XF_PEAK_CAP));   // (XF_FALLBACK_CLIP: the other clips' entries stand)
  if (clean && !pt.out_dev && !pt.stay && total <= cap && total <= pt.o_cap && total) {
    void* ha = pt.want_hashes ? (void*)key32 : (void*)peak_f;
    void* hb = pt.want_hashes ? (void*)t1 : (void*)peak_t;
    if (total <= pt.spec) {
      memcpy(ha, pt.mp + pt.off_a, total * pt.b_a);
      memcpy(hb, pt.mp + pt.off_b, total * 4);
    } else {
      SHZ_HIP(ctx, shz_memcpy(ctx, ha, pt.o_a, total * pt.b_a, hipMemcpyDeviceToHost));
      SHZ_HIP(ctx, shz_memcpy(ctx, hb, pt.o_b, total * 4, hipMemcpyDeviceToHost));
      SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  return SHZ_OK;
}

static int32_t extract_pass(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t n_clips, uint32_t fs,
                            double amp_min, uint32_t fan, uint32_t flags, bool want_hashes, const xparams& xp,
                            uint16_t* peak_f, uint32_t* peak_t, uint32_t* key32, uint32_t* t1, uint64_t* offs_out,
                            uint64_t cap, xctl* hctl, std::vector<uint32_t>* flagged_clips = nullptr) {
  pass_tail pt;
  SHZ_TRY(extract_enqueue(ctx, pcm, clip_off, n_clips, fs, amp_min, fan, flags, want_hashes, xp, peak_f, peak_t, key32, t1,
                          cap, false, &pt));
  SHZ_TRY(extract_finish(ctx, pt, peak_f, peak_t, key32, t1, offs_out, cap, hctl));
  if (flagged_clips) {
    flagged_clips->clear();
    if (hctl->flags & XF_FALLBACK_CLIP) {   // which clips: the bitmap behind the control block (a second, small read-back; rare)
      std::vector<uint32_t> bm(((uint64_t)n_clips + 31) / 32);
      SHZ_HIP(ctx, shz_memcpy(ctx, bm.data(), pt.d_fb, bm.size() * 4, hipMemcpyDeviceToHost));
      SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
      for (uint32_t c = 0; c < n_clips; ++c)
        if ((bm[c >> 5] >> (c & 31)) & 1u) flagged_clips->push_back(c);
    }
  }
  return SHZ_OK;
}

static int32_t extract_driver(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t n_clips, uint32_t fs,
                              double amp_min, uint32_t fan, uint32_t flags, bool want_hashes, uint16_t* peak_f, uint32_t* peak_t,
                              uint64_t* peak_off, uint32_t* key32, uint32_t* t1, uint64_t* hash_off, uint64_t cap, uint64_t* count);

// The clips fp32 staging could not settle (windows with more than PV_MAX_NEAR tied cells: a click per hop, a full-scale
// plateau) are fingerprinted again ONE BY ONE with fp64 staging, and their entries replace what the fp32 pass left for
// them -- the other 999 clips of a batch keep the entries they have.  A (2 or 4 bytes per entry) and B are the output
// arrays (device or host), offs the n_clips + 1 offsets of the fp32 pass; *total is updated.
static int32_t splice_f64_clips(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t fs, double amp_min,
                                uint32_t fan, uint32_t flags, bool want_hashes, const std::vector<uint32_t>& flagged, void* A,
                                void* B, uint64_t* offs, uint32_t n_clips, uint64_t cap, uint64_t* total) {
  const bool out_dev = (flags & SHZ_OUT_DEVICE) != 0;
  const uint64_t b_a = want_hashes ? 4 : 2;
  const bool saved = ctx->stage_f64;
  struct restore { shz_ctx* c; bool v; ~restore() { c->stage_f64 = v; } } rs{ctx, saved};
  struct redo { uint64_t cnt = 0, dev_at = 0; bool on_dev = false; std::vector<uint32_t> b, a32; std::vector<uint16_t> a16; };
  std::vector<redo> R(flagged.size());
  // 1) every flagged clip once more, fp64 staging.  Device outputs: the clip's new entries go straight behind the batch's
  // entries in the caller's arrays (the room is there or the call ends in SHZ_E_CAPACITY anyway) and are moved into place on
  // the device; host outputs: to host vectors.
  uint64_t new_total = *total, dev_end = *total;
  for (size_t fi = 0; fi < flagged.size(); ++fi) {
    const uint32_t c = flagged[fi];
    redo& r = R[fi];
    const uint64_t frames_c = frames_hop(clip_off[c + 1] - clip_off[c], ctx->hop);
    uint64_t cap_c = frames_c * 64 * (want_hashes ? (fan > 1 ? fan - 1 : 1) : 1) + 4096;
    uint64_t o2[2] = {0, 0};
    if (out_dev) {
      const uint64_t at = (dev_end + 63) & ~63ull;   // (16-byte aligned for 2- and 4-byte entries)
      if (at < cap) {
        ctx->stage_f64 = true;
        const int32_t rc = extract_driver(ctx, pcm, clip_off + c, 1, fs, amp_min, fan, flags, want_hashes,
                                          want_hashes ? nullptr : (uint16_t*)A + at, want_hashes ? nullptr : (uint32_t*)B + at,
                                          want_hashes ? nullptr : o2, want_hashes ? (uint32_t*)A + at : nullptr,
                                          want_hashes ? (uint32_t*)B + at : nullptr, want_hashes ? o2 : nullptr, cap - at, &r.cnt);
        ctx->stage_f64 = saved;
        if (rc == SHZ_OK) {
          r.on_dev = true;
          r.dev_at = at;
          dev_end = at + r.cnt;
          ++ctx->st_f64_clips;
          ctx->st_f64_frames += frames_c;
          new_total = new_total - (offs[c + 1] - offs[c]) + r.cnt;
          continue;
        }
        if (rc != SHZ_E_CAPACITY) return rc;
        // no room behind the batch: what the caller has to provide is at least this
        *total = std::max(new_total - (offs[c + 1] - offs[c]) + r.cnt, at + r.cnt);
        SHZ_FAIL(ctx, SHZ_E_CAPACITY, "output needs at least %llu entries, capacity %llu", (unsigned long long)*total, (unsigned long long)cap);
      }
      *total = cap + cap_c;
      SHZ_FAIL(ctx, SHZ_E_CAPACITY, "output needs more than %llu entries", (unsigned long long)cap);
    }
    for (int attempt = 0;; ++attempt) {
      r.b.resize(cap_c);
      if (want_hashes) r.a32.resize(cap_c); else r.a16.resize(cap_c);
      ctx->stage_f64 = true;
      const int32_t rc = extract_driver(ctx, pcm, clip_off + c, 1, fs, amp_min, fan, flags & ~SHZ_OUT_DEVICE, want_hashes,
                                        want_hashes ? nullptr : r.a16.data(), want_hashes ? nullptr : r.b.data(),
                                        want_hashes ? nullptr : o2, want_hashes ? r.a32.data() : nullptr,
                                        want_hashes ? r.b.data() : nullptr, want_hashes ? o2 : nullptr, cap_c, &r.cnt);
      ctx->stage_f64 = saved;
      if (rc == SHZ_E_CAPACITY && attempt == 0) { cap_c = r.cnt + 64; continue; }
      SHZ_TRY(rc);
      break;
    }
    ++ctx->st_f64_clips;
    ctx->st_f64_frames += frames_c;
    new_total = new_total - (offs[c + 1] - offs[c]) + r.cnt;
  }
  // 2) room for the result?  (*total = what the caller has to provide)
  if (new_total > cap || (out_dev && dev_end > cap)) {
    *total = new_total;
    SHZ_FAIL(ctx, SHZ_E_CAPACITY, "output needs %llu entries, capacity %llu", (unsigned long long)new_total, (unsigned long long)cap);
  }
  // 3) splice
  if (out_dev) {
    // the final arrays are put together in scratch -- the stretches between flagged clips from the batch's entries, a
    // flagged clip's entries from where they were parked -- and copied back: 2 |flagged| + 2 device copies per array
    void* tmp;
    SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC3, std::max<uint64_t>(new_total, 1) * 4 + 64, &tmp));
    for (int arr = 0; arr < 2; ++arr) {
      char* base = arr == 0 ? (char*)A : (char*)B;
      const uint64_t bs = arr == 0 ? b_a : 4;
      uint64_t src = 0, dst = 0;   // entries of the old arrays consumed / of the new arrays written
      for (size_t fi = 0; fi < flagged.size(); ++fi) {
        const uint32_t c = flagged[fi];
        const uint64_t keep = offs[c] - src;   // unflagged entries in front of the clip
        if (keep) SHZ_HIP(ctx, hipMemcpyAsync((char*)tmp + dst * bs, base + src * bs, keep * bs, hipMemcpyDeviceToDevice, ctx->stream));
        dst += keep;
        if (R[fi].cnt) SHZ_HIP(ctx, hipMemcpyAsync((char*)tmp + dst * bs, base + R[fi].dev_at * bs, R[fi].cnt * bs, hipMemcpyDeviceToDevice, ctx->stream));
        dst += R[fi].cnt;
        src = offs[c + 1];
      }
      const uint64_t rest = *total - src;
      if (rest) SHZ_HIP(ctx, hipMemcpyAsync((char*)tmp + dst * bs, base + src * bs, rest * bs, hipMemcpyDeviceToDevice, ctx->stream));
      dst += rest;
      if (dst) SHZ_HIP(ctx, hipMemcpyAsync(base, tmp, dst * bs, hipMemcpyDeviceToDevice, ctx->stream));
    }
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t fi = flagged.size(); fi-- > 0;) {
      const uint32_t c = flagged[fi];
      const uint64_t old = offs[c + 1] - offs[c];
      for (uint32_t i = c + 1; i <= n_clips; ++i) offs[i] = offs[i] - old + R[fi].cnt;
    }
    *total = new_total;
    return SHZ_OK;
  }
  // host arrays.  Spliced in place from the back (the offsets in front of a clip stay what they are) -- unless an
  // INTERMEDIATE total would outgrow the caller's arrays although the final one fits (a late clip grows by k, an early one
  // shrinks by k: the first move writes total + k entries): then the result is put together in scratch, front to
  // back, and copied over once.
  {
    uint64_t running = *total, peak = *total;
    for (size_t fi = flagged.size(); fi-- > 0;) {
      const uint32_t c = flagged[fi];
      running = running - (offs[c + 1] - offs[c]) + R[fi].cnt;
      peak = std::max(peak, running);
    }
    if (peak > cap) {
      std::vector<char> na(new_total * b_a), nb(new_total * 4);
      uint64_t src = 0, dst = 0;
      for (size_t fi = 0; fi < flagged.size(); ++fi) {
        const uint32_t c = flagged[fi];
        const redo& r = R[fi];
        const uint64_t keep = offs[c] - src;
        if (keep) { memcpy(na.data() + dst * b_a, (char*)A + src * b_a, keep * b_a); memcpy(nb.data() + dst * 4, (char*)B + src * 4, keep * 4); }
        dst += keep;
        if (r.cnt) {
          memcpy(na.data() + dst * b_a, want_hashes ? (const void*)r.a32.data() : (const void*)r.a16.data(), r.cnt * b_a);
          memcpy(nb.data() + dst * 4, r.b.data(), r.cnt * 4);
        }
        dst += r.cnt;
        src = offs[c + 1];
      }
      const uint64_t rest = *total - src;
      if (rest) { memcpy(na.data() + dst * b_a, (char*)A + src * b_a, rest * b_a); memcpy(nb.data() + dst * 4, (char*)B + src * 4, rest * 4); }
      dst += rest;
      if (dst) { memcpy(A, na.data(), dst * b_a); memcpy(B, nb.data(), dst * 4); }
      for (size_t fi = flagged.size(); fi-- > 0;) {
        const uint32_t c = flagged[fi];
        const uint64_t old = offs[c + 1] - offs[c];
        for (uint32_t i = c + 1; i <= n_clips; ++i) offs[i] = offs[i] - old + R[fi].cnt;
      }
      *total = new_total;
      return SHZ_OK;
    }
  }
  for (size_t fi = flagged.size(); fi-- > 0;) {
    const uint32_t c = flagged[fi];
    const redo& r = R[fi];
    const uint64_t cnt = r.cnt, old0 = offs[c], old1 = offs[c + 1], tail = *total - old1;
    const void* src_a = want_hashes ? (const void*)r.a32.data() : (const void*)r.a16.data();
    if (cnt != old1 - old0 && tail) {
      memmove((char*)A + (old0 + cnt) * b_a, (char*)A + old1 * b_a, tail * b_a);
      memmove((char*)B + (old0 + cnt) * 4, (char*)B + old1 * 4, tail * 4);
    }
    if (cnt) {
      memcpy((char*)A + old0 * b_a, src_a, cnt * b_a);
      memcpy((char*)B + old0 * 4, r.b.data(), cnt * 4);
    }
    for (uint32_t i = c + 1; i <= n_clips; ++i) offs[i] = offs[i] - (old1 - old0) + cnt;
    *total = *total - (old1 - old0) + cnt;
  }
  return SHZ_OK;
}

// shared driver for shz_peaks / shz_fingerprint_batch
static int32_t extract_driver(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t n_clips, uint32_t fs,
                              double amp_min, uint32_t fan, uint32_t flags, bool want_hashes,
                              // peaks outputs
                              uint16_t* peak_f, uint32_t* peak_t, uint64_t* peak_off,
                              // hash outputs
                              uint32_t* key32, uint32_t* t1, uint64_t* hash_off, uint64_t cap, uint64_t* count) {
  SHZ_TRY(check_common(ctx, pcm, clip_off, n_clips, fs));
  if (want_hashes && (fan < 1 || fan > 64)) SHZ_FAIL(ctx, SHZ_E_INVALID, "fan_value must be in [1,64]");
  uint64_t* offs = want_hashes ? hash_off : peak_off;
  if (offs) offs[0] = 0;
  if (count) *count = 0;
  if (n_clips == 0) return SHZ_OK;
  uint64_t frames = 0;
  for (uint32_t c = 0; c < n_clips; ++c) frames += frames_hop(clip_off[c + 1] - clip_off[c], ctx->hop);
  static const bool force_f64 = [] { const char* e = getenv("SHZ_STAGE_F64"); return e && atoi(e) != 0; }();
  xparams xp;
  // fp32 staging needs the threshold to be a positive normal fp32 power well inside the range; amp_min < 0 also needs
  // the zero-plateau rule (peak_zero_plateau_kernel), which reads the fp64 array
  xp.f32 = !force_f64 && !ctx->stage_f64 && amp_min >= 0.0 && amp_min <= 300.0;
  xp.peaks_per_frame = 12;                                      // 4.6 per frame on noise and music-like input
  const uint64_t per_frame_out = want_hashes ? (uint64_t)xp.peaks_per_frame * (fan > 1 ? fan - 1 : 0) : xp.peaks_per_frame;
  xp.stage_cap = std::min<uint64_t>(cap, frames * per_frame_out + 4096);
  xctl h;
  // Dual pass: the clips are cut in two halves by frames and each half runs as a pass of its own -- the first on this
  // context, the second on a twin context (own stream, own workspace) whose entries are appended behind the first
  // half's afterwards.  Two independent pipelines fill each other's stalls (STFT is VALU/LDS-bound, peak picking waits
  // on memory): 7.0 -> 6.48 ms on 1,000 x 30 s clips with the persistent STFT grid, which the stage-by-stage pipeline of
  // SHZ_OVERLAP_SPLIT does not reach.
  // Opt-in (SHZ_DUAL=1): with the STFT in short-lived workgroups one pipeline reaches 6.62 ms per 1,000 x 30 s clips and
  // two reach 6.48-6.50 -- 2 % for twice the workspace and kernel durations that no longer mean one kernel's own time.
  static const bool dual_on = [] { const char* e = getenv("SHZ_DUAL"); return e && atoi(e) != 0; }();
  if (dual_on && n_clips >= 2 && frames >= 131072) {
    uint32_t hc = 1;
    for (uint64_t f = 0; hc < n_clips - 1; ++hc) {
      f += frames_hop(clip_off[hc] - clip_off[hc - 1], ctx->hop);
      if (2 * f >= frames) break;
    }
    if (!ctx->twin) {
      SHZ_TRY(shz_ctx_create(ctx->device, &ctx->twin));
      SHZ_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_twin, hipEventDisableTiming));
      // the caller's numpy window (shz_set_numpy_window) goes with it
      SHZ_HIP(ctx, hipMemcpy(ctx->twin->d_np_window, ctx->d_np_window, sizeof(double) * SHZ_NFFT, hipMemcpyDeviceToDevice));
      ctx->twin->np_sumsq = ctx->np_sumsq;
      ctx->twin->np_unfused = ctx->np_unfused;
    }
    shz_ctx* tw = ctx->twin;
    tw->ws_limit = ctx->ws_limit;
    tw->profiling = ctx->profiling;
    tw->hop = ctx->hop;
    // the twin starts behind everything queued on this context's stream (the caller's PCM may still be in the making)
    SHZ_HIP(ctx, hipEventRecord(ctx->ev_twin, ctx->stream));
    SHZ_HIP(ctx, hipStreamWaitEvent(tw->stream, ctx->ev_twin, 0));
    xparams xa = xp, xb = xp;
    xa.persistent_stft = xb.persistent_stft = true;
    uint64_t fb = 0;
    for (uint32_t c = hc; c < n_clips; ++c) fb += frames_hop(clip_off[c + 1] - clip_off[c], ctx->hop);
    xa.stage_cap = std::min<uint64_t>(cap, (frames - fb) * per_frame_out + 4096);
    xb.stage_cap = fb * per_frame_out + 4096;
    pass_tail ta, tb;
    xctl ha, hb;
    std::vector<uint64_t> offs_b(n_clips - hc + 1);
    int32_t rc = extract_enqueue(ctx, pcm, clip_off, hc, fs, amp_min, fan, flags, want_hashes, xa, peak_f, peak_t, key32, t1,
                                 cap, false, &ta);
    int32_t rcb = rc == SHZ_OK ? extract_enqueue(tw, pcm, clip_off + hc, n_clips - hc, fs, amp_min, fan, flags, want_hashes, xb,
                                                 nullptr, nullptr, nullptr, nullptr, 0, true, &tb)
                               : SHZ_OK;
    if (rc == SHZ_OK) rc = extract_finish(ctx, ta, peak_f, peak_t, key32, t1, offs, cap, &ha);
    if (rcb == SHZ_OK && rc == SHZ_OK) rcb = extract_finish(tw, tb, nullptr, nullptr, nullptr, nullptr, offs_b.data(), 0, &hb);
    (void)hipStreamSynchronize(tw->stream);   // whatever happened, nothing of the twin is in flight past this point
    if (tw->stream2) (void)hipStreamSynchronize(tw->stream2);
    if (rcb != SHZ_OK && rc == SHZ_OK) { ctx->err = tw->err; rc = rcb; }
    SHZ_TRY(rc);
    ctx->st_und += ha.und_total + hb.und_total;
    ctx->st_und_f64 += ha.und_f64 + hb.und_f64;
    ctx->st_und_ffts += ha.und_ffts + hb.und_ffts;
    const uint64_t na = want_hashes ? ha.hash_base : ha.peak_base, nb = want_hashes ? hb.hash_base : hb.peak_base;
    const bool clean = !((ha.flags | hb.flags) & (XF_FALLBACK | XF_PEAK_CAP | XF_FALLBACK_CLIP)) && nb <= tb.o_cap &&
                       ((flags & SHZ_OUT_DEVICE) || na <= ta.o_cap || na > cap);
    if (clean) {   // (anything else -- fp64 fallback, a list too small -- is sorted out by the single pass below)
      if (offs)
        for (uint32_t c = hc; c < n_clips; ++c) offs[c + 1] = na + offs_b[c - hc + 1];
      if (count) *count = na + nb;
      if (na + nb > cap) SHZ_FAIL(ctx, SHZ_E_CAPACITY, "output needs %llu entries, capacity %llu", (unsigned long long)(na + nb),
                                  (unsigned long long)cap);
      if (nb) {   // the second half's entries go behind the first half's
        const hipMemcpyKind kd = (flags & SHZ_OUT_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        char* da = want_hashes ? (char*)key32 : (char*)peak_f;
        char* db = want_hashes ? (char*)t1 : (char*)peak_t;
        SHZ_HIP(ctx, shz_memcpy(ctx, da + na * tb.b_a, tb.o_a, nb * tb.b_a, kd));
        SHZ_HIP(ctx, shz_memcpy(ctx, db + na * 4, tb.o_b, nb * 4, kd));
        if (kd == hipMemcpyDeviceToHost) SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
      }
      return SHZ_OK;
    }
  }
  std::vector<uint64_t> offs_own;   // the per-clip fallback needs the offsets whether the caller wants them or not
  if (!offs) { offs_own.assign((size_t)n_clips + 1, 0); offs = offs_own.data(); }
  std::vector<uint32_t> flagged;
  for (int attempt = 0;; ++attempt) {
    static const bool xtrace = getenv("SHZ_TRACE_EXTRACT") != nullptr;
    if (xtrace) fprintf(stderr, "extract: pass attempt %d clips %u f32 %d\n", attempt, n_clips, (int)xp.f32);
    SHZ_TRY(extract_pass(ctx, pcm, clip_off, n_clips, fs, amp_min, fan, flags, want_hashes, xp, peak_f, peak_t, key32, t1,
                         offs, cap, &h, &flagged));
    if (xtrace) fprintf(stderr, "extract: pass done flags %u flagged %zu und %llu\n", h.flags, flagged.size(), (unsigned long long)h.und_total);
    ctx->st_und += h.und_total;
    ctx->st_und_f64 += h.und_f64;
    ctx->st_und_ffts += h.und_ffts;
    uint64_t total = want_hashes ? h.hash_base : h.peak_base;
    bool again = false;
    // many clips flagged: one fp64 pass over everything is cheaper than a pass per clip
    if (xp.f32 && !flagged.empty() && flagged.size() > std::max<size_t>(8, n_clips / 16)) h.flags |= XF_FALLBACK;
    if (xp.f32 && (h.flags & XF_FALLBACK)) {  // stationary / plateau material: decide on fp64 values
      xp.f32 = false;
      ++ctx->st_fallbacks;
      again = true;
    }
    if (h.flags & XF_PEAK_CAP) {
      // a sub-batch had more peaks than estimated: size the lists for the densest one seen (per frame of the smallest
      // sub-batch it could have been: all of them hold >= 64 frames or the whole call)
      const uint64_t f_min = std::min<uint64_t>(frames, 64);
      xp.peaks_per_frame = (uint32_t)std::min<uint64_t>(SHZ_NBINS, h.max_sub_peaks / f_min + 1);
      if (attempt >= 1) xp.peaks_per_frame = SHZ_NBINS;  // every cell a peak: the bound
      again = true;
    }
    if (!again && !(flags & SHZ_OUT_DEVICE) && total > xp.stage_cap && total <= cap) {
      xp.stage_cap = total;  // staging estimate too small, caller's arrays are not
      again = true;
    }
    if (again) {
      if (attempt >= 4) SHZ_FAIL(ctx, SHZ_E_STATE, "extract: no stable sizing after %d passes", attempt + 1);
      if (!(flags & SHZ_OUT_DEVICE))
        xp.stage_cap = std::max(xp.stage_cap, std::min<uint64_t>(cap, frames * (uint64_t)xp.peaks_per_frame *
                                                                          (want_hashes ? (fan > 1 ? fan - 1 : 0) : 1) + 4096));
      continue;
    }
    if (count) *count = total;
    if (total > cap) SHZ_FAIL(ctx, SHZ_E_CAPACITY, "output needs %llu entries, capacity %llu", (unsigned long long)total,
                              (unsigned long long)cap);
    if (xp.f32 && !flagged.empty()) {
      const int32_t rc = splice_f64_clips(ctx, pcm, clip_off, fs, amp_min, fan, flags, want_hashes, flagged,
                                          want_hashes ? (void*)key32 : (void*)peak_f, want_hashes ? (void*)t1 : (void*)peak_t, offs,
                                          n_clips, cap, &total);
      if (count) *count = total;   // (on SHZ_E_CAPACITY: what the caller has to provide)
      SHZ_TRY(rc);
    }
    return SHZ_OK;
  }
}

// ---- host-fed streaming --------------------------------------------------------------------------------------------
// Every caller of the reference hands fingerprint() samples that live in host memory (__init__.py:248-268: the decoded
// channels of a file; recognizer.py:377-382: the recorded buffer).  A large batch of host PCM is cut into chunks of whole
// clips (16 MB, then ~64 MB); a helper thread uploads chunk i + 1 on its own stream into the second of two device buffers while this
// thread runs the extraction pass of chunk i from the first -- the link and the kernels work side by side, and the call's
// rate is the link's (the kernels are ~8x faster than PCIe delivers: 4.7 M audio-s/s against ~0.6 M at 55 GB/s).  Pinned
// memory (shz_host_alloc) goes out by DMA as it is; pageable memory takes the runtime's staged copy.
#include <condition_variable>
#include <thread>
#define UP_CHUNK_BYTES (64ull << 20)
#define UP_FIRST_BYTES (16ull << 20)   // the first chunk is small: nothing runs beside its upload (1,058 MB at 57 GB/s: 128 MB chunks
                                       // reached 87 % of the link, the 2.2 ms of the first one and the pass of the last one exposed)
#define UP_MIN_BYTES (192ull << 20)   // smaller batches: one copy, one pass (the pipeline's two buffers and thread buy nothing)

static int32_t extract_streamed(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t n_clips, uint32_t fs,
                                double amp_min, uint32_t fan, uint32_t flags, bool want_hashes, uint16_t* peak_f, uint32_t* peak_t,
                                uint64_t* peak_off, uint32_t* key32, uint32_t* t1, uint64_t* hash_off, uint64_t cap, uint64_t* count) {
  SHZ_TRY(check_common(ctx, pcm, clip_off, n_clips, fs));
  uint64_t* offs = want_hashes ? hash_off : peak_off;
  // chunks of whole clips
  struct chunk { uint32_t c0, c1; };
  std::vector<chunk> chunks;
  uint64_t max_samples = 0;
  for (uint32_t c = 0; c < n_clips;) {
    uint32_t e = c + 1;
    const uint64_t lim = chunks.empty() ? UP_FIRST_BYTES : UP_CHUNK_BYTES;
    while (e < n_clips && (clip_off[e + 1] - clip_off[c]) * 2 <= lim) ++e;
    chunks.push_back(chunk{c, e});
    max_samples = std::max(max_samples, clip_off[e] - clip_off[c]);
    c = e;
  }
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->stream_up) SHZ_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_up, hipStreamNonBlocking));
  void* dbuf[2];
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PCM, max_samples * 2 + 64, &dbuf[0]));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PCM_B, max_samples * 2 + 64, &dbuf[1]));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));   // nothing queued earlier still reads the two buffers
  // (pinned and pageable sources take the same call: the runtime stages pageable memory itself, at 56 of the link's 57 GB/s here)
  struct shared {
    std::mutex mu;
    std::condition_variable cv;
    size_t ready = 0;        // chunks uploaded so far
    size_t consumed = 0;     // chunks whose pass is over
    bool stop = false;
    hipError_t err = hipSuccess;
    double copy_s = 0.0;
  } sh;
  const int device = ctx->device;
  hipStream_t us = ctx->stream_up;
  std::thread up([&]() {
    (void)hipSetDevice(device);
    for (size_t j = 0; j < chunks.size(); ++j) {
      {
        std::unique_lock<std::mutex> lk(sh.mu);
        sh.cv.wait(lk, [&] { return sh.stop || j < sh.consumed + 2; });   // buffer j & 1 is free once chunk j - 2 is consumed
        if (sh.stop) return;
      }
      const uint64_t s0 = clip_off[chunks[j].c0], bytes = (clip_off[chunks[j].c1] - s0) * 2;
      const double t0 = now_seconds();
      hipError_t e = bytes ? hipMemcpyAsync(dbuf[j & 1], pcm + s0, bytes, hipMemcpyHostToDevice, us) : hipSuccess;
      if (e == hipSuccess) e = hipStreamSynchronize(us);
      std::lock_guard<std::mutex> lk(sh.mu);
      sh.copy_s += now_seconds() - t0;
      if (e != hipSuccess) { sh.err = e; sh.stop = true; sh.cv.notify_all(); return; }
      sh.ready = j + 1;
      sh.cv.notify_all();
    }
  });
  struct joiner { std::thread& t; shared& s; ~joiner() { { std::lock_guard<std::mutex> lk(s.mu); s.stop = true; } s.cv.notify_all(); if (t.joinable()) t.join(); } } jn{up, sh};
  uint64_t total = 0;
  bool short_cap = false;
  std::vector<uint64_t> rel, coff;
  if (offs) offs[0] = 0;
  for (size_t i = 0; i < chunks.size(); ++i) {
    const double w0 = now_seconds();
    {
      std::unique_lock<std::mutex> lk(sh.mu);
      sh.cv.wait(lk, [&] { return sh.stop || sh.ready > i; });
      if (sh.err != hipSuccess) SHZ_FAIL(ctx, SHZ_E_HIP, "upload of PCM chunk %zu failed: %s", i, hipGetErrorString(sh.err));
    }
    ctx->st_up_wait_s += now_seconds() - w0;
    const uint32_t c0 = chunks[i].c0, nc = chunks[i].c1 - c0;
    rel.resize(nc + 1);
    for (uint32_t c = 0; c <= nc; ++c) rel[c] = clip_off[c0 + c] - clip_off[c0];
    coff.assign(nc + 1, 0);
    uint64_t cnt = 0;
    const uint64_t room = short_cap || total > cap ? 0 : cap - total;
    const int32_t rc = extract_driver(ctx, (const int16_t*)dbuf[i & 1], rel.data(), nc, fs, amp_min, fan, flags | SHZ_PCM_DEVICE, want_hashes,
                                      peak_f ? peak_f + total : nullptr, peak_t ? peak_t + total : nullptr, want_hashes ? nullptr : coff.data(),
                                      key32 ? key32 + total : nullptr, t1 ? t1 + total : nullptr, want_hashes ? coff.data() : nullptr, room, &cnt);
    if (rc == SHZ_E_CAPACITY) short_cap = true;   // the chunks that follow are only counted: the caller learns what to provide
    else if (rc != SHZ_OK) return rc;
    if (offs && !short_cap)
      for (uint32_t c = 1; c <= nc; ++c) offs[c0 + c] = total + coff[c];
    total += cnt;
    {
      std::lock_guard<std::mutex> lk(sh.mu);
      sh.consumed = i + 1;
    }
    sh.cv.notify_all();
    ++ctx->st_up_chunks;
  }
  ctx->st_up_bytes += (clip_off[n_clips] - clip_off[0]) * 2;
  { std::lock_guard<std::mutex> lk(sh.mu); ctx->st_up_copy_s += sh.copy_s; }
  if (count) *count = total;
  if (short_cap) SHZ_FAIL(ctx, SHZ_E_CAPACITY, "output needs %llu entries, capacity %llu", (unsigned long long)total, (unsigned long long)cap);
  return SHZ_OK;
}

// host PCM of a size worth pipelining -> extract_streamed; everything else -> one pass
static int32_t extract_any(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t n_clips, uint32_t fs,
                           double amp_min, uint32_t fan, uint32_t flags, bool want_hashes, uint16_t* peak_f, uint32_t* peak_t,
                           uint64_t* peak_off, uint32_t* key32, uint32_t* t1, uint64_t* hash_off, uint64_t cap, uint64_t* count) {
  static const bool off = [] { const char* e = getenv("SHZ_UPLOAD_PIPELINE"); return e && atoi(e) == 0; }();
  if (!off && ctx && pcm && clip_off && !(flags & SHZ_PCM_DEVICE) && n_clips >= 2) {
    bool ok = true;
    for (uint32_t c = 0; c < n_clips && ok; ++c) ok = clip_off[c + 1] >= clip_off[c];
    if (ok && (clip_off[n_clips] - clip_off[0]) * 2 >= UP_MIN_BYTES)
      return extract_streamed(ctx, pcm, clip_off, n_clips, fs, amp_min, fan, flags, want_hashes, peak_f, peak_t, peak_off, key32, t1,
                              hash_off, cap, count);
  }
  return extract_driver(ctx, pcm, clip_off, n_clips, fs, amp_min, fan, flags, want_hashes, peak_f, peak_t, peak_off, key32, t1, hash_off,
                        cap, count);
}

extern "C" int32_t shz_upload_stats(shz_ctx* ctx, uint64_t* chunks, uint64_t* bytes, double* copy_s, double* wait_s) {
  if (!ctx) return SHZ_E_INVALID;
  if (chunks) *chunks = ctx->st_up_chunks;
  if (bytes) *bytes = ctx->st_up_bytes;
  if (copy_s) *copy_s = ctx->st_up_copy_s;
  if (wait_s) *wait_s = ctx->st_up_wait_s;
  return SHZ_OK;
}

extern "C" int32_t shz_set_overlap(shz_ctx* ctx, uint32_t noverlap) {
  if (!ctx) return SHZ_E_INVALID;
  if (noverlap >= SHZ_NFFT) SHZ_FAIL(ctx, SHZ_E_INVALID, "noverlap must be less than NFFT (%u >= %d)", noverlap, SHZ_NFFT);   // mlab:242
  ctx->hop = SHZ_NFFT - noverlap;
  if (ctx->twin) ctx->twin->hop = ctx->hop;
  return SHZ_OK;
}

extern "C" int32_t shz_set_stage_f64(shz_ctx* ctx, int32_t enabled) {
  if (!ctx) return SHZ_E_INVALID;
  ctx->stage_f64 = enabled != 0;
  return SHZ_OK;
}

extern "C" int32_t shz_extract_stats(shz_ctx* ctx, uint64_t* undecided, uint64_t* decided_f64, uint64_t* frames_recomputed,
                                     uint64_t* f64_passes, uint64_t* f64_clips, uint64_t* f64_clip_frames) {
  if (!ctx) return SHZ_E_INVALID;
  if (undecided) *undecided = ctx->st_und;
  if (decided_f64) *decided_f64 = ctx->st_und_f64;
  if (frames_recomputed) *frames_recomputed = ctx->st_und_ffts;
  if (f64_passes) *f64_passes = ctx->st_fallbacks;
  if (f64_clips) *f64_clips = ctx->st_f64_clips;
  if (f64_clip_frames) *f64_clip_frames = ctx->st_f64_frames;
  return SHZ_OK;
}

extern "C" int32_t shz_peaks(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t n_clips, uint32_t fs,
                             double amp_min, uint32_t flags, uint16_t* peak_f, uint32_t* peak_t, uint64_t* peak_off,
                             uint64_t cap, uint64_t* count) {
  return extract_any(ctx, pcm, clip_off, n_clips, fs, amp_min, 0, flags, false, peak_f, peak_t, peak_off, nullptr,
                     nullptr, nullptr, cap, count);
}

extern "C" int32_t shz_fingerprint_batch(shz_ctx* ctx, const int16_t* pcm, const uint64_t* clip_off, uint32_t n_clips,
                                         uint32_t fs, double amp_min, uint32_t fan_value, uint32_t flags,
                                         uint32_t* key32, uint32_t* t1, uint64_t* hash_off, uint64_t cap,
                                         uint64_t* count) {
  return extract_any(ctx, pcm, clip_off, n_clips, fs, amp_min, fan_value, flags, true, nullptr, nullptr, nullptr,
                     key32, t1, hash_off, cap, count);
}

extern "C" int32_t shz_peaks_from_db(shz_ctx* ctx, const double* arr2d, uint32_t n_rows, uint32_t n_cols,
                                     double amp_min, uint32_t* out_f, uint32_t* out_t, uint64_t cap, uint64_t* count) {
  if (!ctx) return SHZ_E_INVALID;
  if (count) *count = 0;
  if (n_rows == 0 || n_cols == 0) return SHZ_OK;
  if (!arr2d) SHZ_FAIL(ctx, SHZ_E_INVALID, "arr2d is NULL");
  if (n_rows > 65535) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "at most 65535 rows (frequency bins)");
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t stride = (n_rows + 7) & ~7u;
  void *d_in, *d_db;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_MISC3, (uint64_t)n_rows * n_cols * 8, &d_in));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_DB, (uint64_t)n_cols * stride * 8, &d_db));
  SHZ_HIP(ctx, shz_memcpy(ctx, d_in, arr2d, (uint64_t)n_rows * n_cols * 8, hipMemcpyHostToDevice));
  dim3 grid((n_cols + 31) / 32, (n_rows + 31) / 32);
  hipLaunchKernelGGL(transpose_in_kernel, grid, dim3(32, 8), 0, ctx->stream, (const double*)d_in, n_rows, n_cols, stride,
                     (double*)d_db);
  SHZ_HIP(ctx, hipGetLastError());
  // one "clip" of n_cols frames
  sub_dev sd;
  std::vector<peak_seg> segs;
  for (uint32_t t0 = 0; t0 < n_cols; t0 += PK_SEG) segs.push_back(peak_seg{0, n_cols, t0, std::min(t0 + PK_SEG, n_cols)});
  void *p0, *p1;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_META, 64, &p0));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_META2, segs.size() * sizeof(peak_seg) + 64, &p1));
  uint32_t foff[2] = {0, n_cols};
  sd.d_foff = (uint32_t*)p0;
  sd.d_segs = (peak_seg*)p1;
  sd.n_segs = (uint32_t)segs.size();
  SHZ_HIP(ctx, shz_memcpy(ctx, sd.d_foff, foff, 8, hipMemcpyHostToDevice));
  SHZ_HIP(ctx, shz_memcpy(ctx, sd.d_segs, segs.data(), segs.size() * sizeof(peak_seg), hipMemcpyHostToDevice));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint16_t* d_pf;
  uint32_t *d_pt, *d_pcoff, n_peaks;
  SHZ_TRY(run_peaks(ctx, (const double*)d_db, stride, n_rows, sd, 1, n_cols, amp_min, false, &d_pf, &d_pt, &d_pcoff, &n_peaks));
  if (count) *count = n_peaks;
  if (n_peaks > cap) SHZ_FAIL(ctx, SHZ_E_CAPACITY, "need %u peaks", n_peaks);
  if (!n_peaks) return SHZ_OK;
  std::vector<uint16_t> pf(n_peaks);
  std::vector<uint32_t> pt(n_peaks), idx(n_peaks);
  SHZ_HIP(ctx, shz_memcpy(ctx, pf.data(), d_pf, (uint64_t)n_peaks * 2, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, shz_memcpy(ctx, pt.data(), d_pt, (uint64_t)n_peaks * 4, hipMemcpyDeviceToHost));
  SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // device order is (t asc, f asc); np.where order is (f asc, t asc): stable re-sort by f
  for (uint32_t i = 0; i < n_peaks; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return pf[a] < pf[b]; });
  for (uint32_t i = 0; i < n_peaks; ++i) {
    out_f[i] = pf[idx[i]];
    out_t[i] = pt[idx[i]];
  }
  return SHZ_OK;
}

extern "C" int32_t shz_pair_hash(shz_ctx* ctx, const uint16_t* peak_f, const uint32_t* peak_t, const uint64_t* peak_off,
                                 uint32_t n_clips, uint32_t fan_value, uint32_t* key32, uint32_t* t1, uint64_t* hash_off,
                                 uint64_t cap, uint64_t* count) {
  if (!ctx || !peak_off) return SHZ_E_INVALID;
  if (fan_value < 1 || fan_value > 64) SHZ_FAIL(ctx, SHZ_E_INVALID, "fan_value must be in [1,64]");
  SHZ_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t n = peak_off[n_clips] - peak_off[0];
  if (n >= (1ull << 30)) SHZ_FAIL(ctx, SHZ_E_UNSUPPORTED, "too many peaks in one call");
  if (hash_off) hash_off[0] = 0;
  if (count) *count = 0;
  if (n_clips == 0) return SHZ_OK;
  std::vector<uint32_t> pco(n_clips + 1);
  for (uint32_t c = 0; c <= n_clips; ++c) pco[c] = (uint32_t)(peak_off[c] - peak_off[0]);
  void *pf, *pt, *pc, *pk, *pt1;
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_F, n * 2 + 64, &pf));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_T, n * 4 + 64, &pt));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_PEAK_CLIP, (uint64_t)(n_clips + 1) * 4 + 64, &pc));
  const uint64_t upper = n * (fan_value - 1);
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_KEY, upper * 4 + 64, &pk));
  SHZ_TRY(shz_ws_reserve(ctx, SHZ_WS_T1, upper * 4 + 64, &pt1));
  if (n) {
    SHZ_HIP(ctx, shz_memcpy(ctx, pf, peak_f + peak_off[0], n * 2, hipMemcpyHostToDevice));
    SHZ_HIP(ctx, shz_memcpy(ctx, pt, peak_t + peak_off[0], n * 4, hipMemcpyHostToDevice));
  }
  SHZ_HIP(ctx, shz_memcpy(ctx, pc, pco.data(), (uint64_t)(n_clips + 1) * 4, hipMemcpyHostToDevice));
  uint64_t n_h = 0;
  std::vector<uint32_t> choff;
  SHZ_TRY(run_pairs(ctx, (const uint16_t*)pf, (const uint32_t*)pt, (const uint32_t*)pc, n_clips, (uint32_t)n, fan_value,
                    (uint32_t*)pk, (uint32_t*)pt1, 0, upper, &n_h, &choff));
  if (hash_off)
    for (uint32_t c = 0; c < n_clips; ++c) hash_off[c + 1] = choff[c + 1];
  if (count) *count = n_h;
  if (n_h > cap) SHZ_FAIL(ctx, SHZ_E_CAPACITY, "need %llu hashes", (unsigned long long)n_h);
  if (n_h) {
    SHZ_HIP(ctx, shz_memcpy(ctx, key32, pk, n_h * 4, hipMemcpyDeviceToHost));
    SHZ_HIP(ctx, shz_memcpy(ctx, t1, pt1, n_h * 4, hipMemcpyDeviceToHost));
    SHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return SHZ_OK;
}
