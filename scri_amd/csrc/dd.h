// Double-double arithmetic (unevaluated sum hi + lo of two fp64 numbers, ~32 significant digits) for the small tables
// whose rounding would otherwise show in the results: the spin-weighted spherical harmonics of the synthesis matrix and
// of the analysis tables, and the per-pixel supertranslation sums (wigner.h: SwshChain).  The reference's own analytic
// tests (tests/test_waveform_grid.py:17-158) sit at 2-3 ulp of the data; a plain fp64 l-recurrence costs 10-20 ulp at
// l = 8..16.  These tables are O(n_modes n_pix) per transformation -- nothing next to the O(N n_modes n_pix) contraction.
// Error-free transformations after Dekker / Knuth; products through fma.  Host and device.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#ifndef BMS_HD
#define BMS_HD __host__ __device__ __forceinline__
#endif

namespace bms {

struct dd {
  double hi, lo;
};

BMS_HD dd dd_from(double a) { return {a, 0.0}; }
BMS_HD double dd_to_double(dd a) { return a.hi + a.lo; }

BMS_HD dd quick_two_sum(double a, double b) {  // |a| >= |b|
#pragma clang fp contract(off)
  const double s = a + b;
  return {s, b - (s - a)};
}
BMS_HD dd two_sum(double a, double b) {
#pragma clang fp contract(off)
  const double s = a + b;
  const double bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
BMS_HD dd two_prod(double a, double b) {
#pragma clang fp contract(off)
  const double p = a * b;
  return {p, fma(a, b, -p)};
}

BMS_HD dd dd_neg(dd a) { return {-a.hi, -a.lo}; }
BMS_HD dd dd_add(dd a, dd b) {
#pragma clang fp contract(off)
  dd s = two_sum(a.hi, b.hi);
  const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return quick_two_sum(s.hi, s.lo);
}
BMS_HD dd dd_sub(dd a, dd b) { return dd_add(a, dd_neg(b)); }
BMS_HD dd dd_add_d(dd a, double b) {
#pragma clang fp contract(off)
  dd s = two_sum(a.hi, b);
  s.lo += a.lo;
  return quick_two_sum(s.hi, s.lo);
}
BMS_HD dd dd_mul(dd a, dd b) {
#pragma clang fp contract(off)
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return quick_two_sum(p.hi, p.lo);
}
BMS_HD dd dd_mul_d(dd a, double b) {
#pragma clang fp contract(off)
  dd p = two_prod(a.hi, b);
  p.lo += a.lo * b;
  return quick_two_sum(p.hi, p.lo);
}
BMS_HD dd dd_div(dd a, dd b) {
#pragma clang fp contract(off)
  const double q1 = a.hi / b.hi;
  dd r = dd_sub(a, dd_mul_d(b, q1));
  const double q2 = r.hi / b.hi;
  r = dd_sub(r, dd_mul_d(b, q2));
  const double q3 = r.hi / b.hi;
  dd q = quick_two_sum(q1, q2);
  return dd_add_d(q, q3);
}
BMS_HD dd dd_sqrt(dd a) {
#pragma clang fp contract(off)
  if (!(a.hi > 0.0)) return {0.0, 0.0};
  const double x = 1.0 / sqrt(a.hi);
  const double ax = a.hi * x;
  const dd e = dd_sub(a, two_prod(ax, ax));
  return dd_add_d(two_sum(ax, e.hi * (x * 0.5)), 0.0);
}
BMS_HD dd dd_ipow(dd x, int n) {
  dd r = {1.0, 0.0};
  while (n > 0) {
    if (n & 1) r = dd_mul(r, x);
    x = dd_mul(x, x);
    n >>= 1;
  }
  return r;
}

struct ddc {
  dd re, im;
};
BMS_HD ddc ddc_mul(ddc a, ddc b) {
  return {dd_sub(dd_mul(a.re, b.re), dd_mul(a.im, b.im)), dd_add(dd_mul(a.re, b.im), dd_mul(a.im, b.re))};
}
BMS_HD ddc ddc_pow_unit(ddc z, int n) {  // integer (possibly negative) power of a unit complex number
  if (n < 0) {
    z.im = dd_neg(z.im);
    n = -n;
  }
  ddc r = {{1.0, 0.0}, {0.0, 0.0}};
  while (n > 0) {
    if (n & 1) r = ddc_mul(r, z);
    z = ddc_mul(z, z);
    n >>= 1;
  }
  return r;
}

// 1 / (4 pi) to double-double
BMS_HD dd dd_inv_4pi() { return {0.07957747154594767, -4.9196691687956215e-18}; }

}  // namespace bms
