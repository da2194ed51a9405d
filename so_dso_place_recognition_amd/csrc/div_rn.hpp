// div_rn.hpp — x / s correctly rounded from the reciprocal r = RN(1 / s) (Markstein's division: q = RN(x r),
// e = x - q s exactly (fma), result = RN(q + e r)).  With r the correctly rounded reciprocal the result IS the IEEE
// quotient for finite operands whose quotient neither overflows nor underflows (Markstein 1990; Cornea-Harrison-Tang,
// "Scientific computing on Itanium", thm 8.4 - the one excluded divisor, significand all ones, cannot be a standard
// deviation of fp32 data summed in fp64).  tests/native/div_rn_check.cpp fuzzes it against the hardware division.
#pragma once
#if defined(__HIPCC__)
#define PR_DIV_HD __host__ __device__
#else
#define PR_DIV_HD
#endif

namespace pr {
PR_DIV_HD inline double div_rn(double x, double s, double r) {
  const double q = x * r;
  return __builtin_fma(__builtin_fma(-q, s, x), r, q);
}
}  // namespace pr
