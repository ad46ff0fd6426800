// Correctly rounded log10 in double-double arithmetic (host + device), for the reference's dB-domain tests.
//
// The reference decides peaks on A = 10*np.log10(P) (__init__.py:241): `maximum_filter(A) == A` (:143) and
// `A > amp_min` (:161).  Several adjacent doubles of P share one double of A (8-37 of them for P in 1e4..1e16), so
// the tie set is a property of the logarithm's rounding and must be evaluated on dB values.  numpy's log10 is a
// vendor routine (SVML on AVX-512 hosts, libm elsewhere) that returns the correctly rounded value for 99.97 % / 95.7 %
// of inputs; the rounded value of the exact logarithm is the one figure every such routine approximates, so that is
// what the device computes.  Runs only for cells that share a window maximum to within 2^-40 or sit within 1e-9 of the
// amp_min threshold -- a few cells per million -- so ~150 fp64 operations per call do not matter.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define SHZ_HD __host__ __device__ inline
#else
#define SHZ_HD inline
#endif

struct shz_dd { double hi, lo; };

SHZ_HD shz_dd dd_two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return shz_dd{s, (a - (s - bb)) + (b - bb)};
}
SHZ_HD shz_dd dd_quick_two_sum(double a, double b) {  // |a| >= |b|
  const double s = a + b;
  return shz_dd{s, b - (s - a)};
}
SHZ_HD shz_dd dd_two_prod(double a, double b) {
  const double p = a * b;
  return shz_dd{p, fma(a, b, -p)};
}
SHZ_HD shz_dd dd_add(shz_dd a, shz_dd b) {
  shz_dd s = dd_two_sum(a.hi, b.hi);
  const shz_dd t = dd_two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = dd_quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return dd_quick_two_sum(s.hi, s.lo);
}
SHZ_HD shz_dd dd_mul(shz_dd a, shz_dd b) {
  shz_dd p = dd_two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return dd_quick_two_sum(p.hi, p.lo);
}
SHZ_HD shz_dd dd_mul_d(shz_dd a, double b) {
  shz_dd p = dd_two_prod(a.hi, b);
  p.lo += a.lo * b;
  return dd_quick_two_sum(p.hi, p.lo);
}
SHZ_HD shz_dd dd_div(shz_dd a, shz_dd b) {  // three quotient digits
  const double q1 = a.hi / b.hi;
  shz_dd r = dd_add(a, dd_mul_d(b, -q1));
  const double q2 = r.hi / b.hi;
  r = dd_add(r, dd_mul_d(b, -q2));
  const double q3 = r.hi / b.hi;
  const shz_dd q = dd_quick_two_sum(q1, q2);
  return dd_add(q, shz_dd{q3, 0.0});
}

// log10(x) for finite x > 0, rounded to nearest from a ~104-bit double-double value
SHZ_HD double shz_log10_cr(double x) {
  int e;
  double m = frexp(x, &e);                                   // x = m 2^e, m in [0.5, 1)
  if (m < 0.70710678118654752440) { m *= 2.0; --e; }         // m in [sqrt(1/2), sqrt(2))
  // ln m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716
  const shz_dd s = dd_div(shz_dd{m - 1.0, 0.0}, dd_two_sum(m, 1.0));   // m - 1 is exact (Sterbenz)
  const shz_dd s2 = dd_mul(s, s);
  shz_dd acc = dd_div(shz_dd{1.0, 0.0}, shz_dd{45.0, 0.0});            // sum_k s2^k / (2k+1), k = 22 .. 0
  for (int k = 21; k >= 0; --k)
    acc = dd_add(dd_mul(acc, s2), dd_div(shz_dd{1.0, 0.0}, shz_dd{(double)(2 * k + 1), 0.0}));
  shz_dd lnm = dd_mul(s, acc);
  lnm.hi *= 2.0;
  lnm.lo *= 2.0;
  const shz_dd log10_2{0x1.34413509f79ffp-2, -0x1.9dc1da994fd21p-59};
  const shz_dd log10_e{0x1.bcb7b1526e50ep-2, 0x1.95355baaafad3p-57};
  const shz_dd r = dd_add(dd_mul_d(log10_2, (double)e), dd_mul(lnm, log10_e));
  return r.hi + r.lo;
}

// the reference's dB value of a non-zero power: 10 * np.log10(P)
SHZ_HD double shz_db_of(double p) { return 10.0 * shz_log10_cr(p); }
