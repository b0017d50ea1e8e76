// dd.h — double-double arithmetic (an unevaluated sum hi + lo of two FP64 numbers, ~31 significant digits) for the
// interaction regions whose FP64 assembly loses the answer before the elimination starts: MPSA regions in which a stiff
// sub-cell sits between soft ones (mpsa.inc: mpsa_node_body<..., dd>).  The condensed system adds the stiff and the soft
// cells' tractions into the same entries; the stiff part alone is rank deficient (the symmetric gradient does not see the
// sub-cell's rigid rotation), so the rotation is fixed by the soft part -- eps * contrast below the entry it was added to.
// The reference keeps the two in separate columns of its gradient system (numerics/fv/mpsa.py:784-930) and loses nothing;
// here the flagged regions carry 106 bits through assembly, elimination and the response products instead.
//
// Error-free transformations (Dekker / Knuth; two-product by FMA): every product and sum below must be rounded on its own,
// hence `fp contract(off)` in every function body (hipcc contracts a * b + c by default; gcc's host build does not fuse
// without -mfma, and a fused form would only make the transformations more exact where it kept their algebra).
#pragma once

namespace pfv {

struct dd {
  double hi, lo;
  PFV_HD dd() : hi(0.0), lo(0.0) {}
  PFV_HD dd(double h) : hi(h), lo(0.0) {}
  PFV_HD dd(double h, double l) : hi(h), lo(l) {}
  PFV_HD explicit operator double() const { return hi; }  // (normalised: |lo| <= ulp(hi) / 2, hi is the rounded value)
};

PFV_FN dd dd_quick_two_sum(double a, double b) {  // |a| >= |b|
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const double s = a + b;
  return dd(s, b - (s - a));
}
PFV_FN dd dd_two_sum(double a, double b) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const double s = a + b;
  const double bb = s - a;
  return dd(s, (a - (s - bb)) + (b - bb));
}
PFV_FN dd dd_two_prod(double a, double b) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const double p = a * b;
  return dd(p, fma(a, b, -p));
}

PFV_FN dd operator-(const dd& a) { return dd(-a.hi, -a.lo); }
PFV_FN dd operator+(const dd& a, const dd& b) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  dd s = dd_two_sum(a.hi, b.hi);
  const dd t = dd_two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = dd_quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return dd_quick_two_sum(s.hi, s.lo);
}
PFV_FN dd operator-(const dd& a, const dd& b) { return a + (-b); }
PFV_FN dd operator*(const dd& a, const dd& b) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  dd p = dd_two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return dd_quick_two_sum(p.hi, p.lo);
}
PFV_FN dd operator/(const dd& a, const dd& b) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const double q1 = a.hi / b.hi;
  dd r = a - b * dd(q1);
  const double q2 = r.hi / b.hi;
  r = r - b * dd(q2);
  const double q3 = r.hi / b.hi;
  const dd q = dd_quick_two_sum(q1, q2);
  return q + dd(q3);
}
PFV_FN dd& operator+=(dd& a, const dd& b) { a = a + b; return a; }
PFV_FN dd& operator-=(dd& a, const dd& b) { a = a - b; return a; }
PFV_FN dd& operator*=(dd& a, const dd& b) { a = a * b; return a; }
PFV_FN dd& operator/=(dd& a, const dd& b) { a = a / b; return a; }
PFV_FN bool operator==(const dd& a, const dd& b) { return a.hi == b.hi && a.lo == b.lo; }
PFV_FN bool operator!=(const dd& a, const dd& b) { return !(a == b); }
PFV_FN bool operator<(const dd& a, const dd& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
PFV_FN bool operator>(const dd& a, const dd& b) { return b < a; }
PFV_FN bool operator<=(const dd& a, const dd& b) { return !(b < a) && a.hi == a.hi && b.hi == b.hi; }
PFV_FN bool operator>=(const dd& a, const dd& b) { return b <= a; }

// scalar helpers the bodies templated on the arithmetic use instead of fabs / casts (a pfv::fabs(dd) would hide ::fabs
// for every unqualified call in the namespace and quietly route FP64 code through a conversion)
PFV_FN double t_abs(double x) { return fabs(x); }
PFV_FN dd t_abs(const dd& x) { return (x.hi < 0.0 || (x.hi == 0.0 && x.lo < 0.0)) ? -x : x; }
PFV_FN double t_dbl(double x) { return x; }
PFV_FN double t_dbl(const dd& x) { return x.hi; }

}  // namespace pfv
