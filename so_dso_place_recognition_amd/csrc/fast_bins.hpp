// fast_bins.hpp — polar bin indices of the generators without fp64 transcendentals.
//
// The reference computes  si = floor((atan2(y, x) + pi) * S)  and  ri = floor(sqrt(x^2 + y^2) * R)  in fp64
// (SC.cpp:37-38, M2DP.cpp:59-62).  Only the integer parts are used, so both are first evaluated in fp32
// (atan2f / sqrtf on the rounded coordinates, error < 1e-5 bins) and accepted when the fractional part is at least
// MARGIN = 1e-3 bins away from an integer; otherwise (about 0.2 % of the points, plus every exact-zero / signed-zero /
// NaN case, which all land on an integer in fp32) the fp64 expression of the reference is evaluated verbatim.
// The result is therefore identical to the all-fp64 evaluation; only the cost changes (fp32 VALU instead of fp64).
#pragma once
#include <hip/hip_runtime.h>

namespace pr {

__host__ __device__ __forceinline__ int polar_sector(double y, double x, double S_res_inv, float S_f) {
  const float xf = (float)x, yf = (float)y;
  const float t = (atan2f(yf, xf) + 3.14159274f) * S_f;
  const float fl = floorf(t), fr = t - fl;
  const float mx = fmaxf(fabsf(xf), fabsf(yf));          // the casts must neither overflow nor lose bits to denormals
  if (fr > 1e-3f && fr < 0.999f && mx > 1e-30f && mx < 3.0e38f) return (int)fl;
  return (int)floor((atan2(y, x) + M_PI) * S_res_inv);
}

// 16 sectors (M2DP.cpp:59, S = 16 / 2pi): the sector is decided by the signs of x, y, by |y| > |x| and by
// min(|x|,|y|) > max(|x|,|y|) tan(pi/8) - no arctangent.  Accepted when the point is at least 1e-3 (relative to
// max(|x|,|y|), i.e. >= 1e-3 rad) away from all three kinds of boundary (axis, 22.5 deg, diagonal); everything else -
// boundaries, zeros of either sign, NaN, float overflow / underflow - takes the reference's fp64 expression.
__host__ __device__ __forceinline__ int polar_sector16(double y, double x, double S_res_inv) {
  const float xf = (float)x, yf = (float)y;
  const float a = fabsf(xf), b = fabsf(yf);
  const float mn = fminf(a, b), mx = fmaxf(a, b);
  const float e = 1e-3f * mx, t = 0.414213568f * mx;                       // tan(pi/8)
  if (mn > e && fabsf(mn - t) > e && (mx - mn) > e && mx > 1e-30f && mx < 3.0e38f) {
    // quadrant of theta + pi: (x<0,y<0) 0, (x>0,y<0) 1, (x>0,y>0) 2, (x<0,y>0) 3; inside it the angle grows from the
    // negative x axis (even quadrants: tan = |y|/|x|) or from the y axis (odd quadrants: tan = |x|/|y|)
    const int q = (yf < 0.f) ? (xf < 0.f ? 0 : 1) : (xf < 0.f ? 3 : 2);
    const bool big = (q & 1) ? (a > b) : (b > a);                         // tangent above 1
    const bool h = mn > t;                                                // more than 22.5 deg from the nearer axis
    return q * 4 + (big ? (h ? 2 : 3) : (h ? 1 : 0));
  }
  return (int)floor((atan2(y, x) + M_PI) * S_res_inv);
}

__host__ __device__ __forceinline__ int polar_ring(double x, double y, double R_res_inv, float R_f) {
  const float xf = (float)x, yf = (float)y;
  const float t = sqrtf(xf * xf + yf * yf) * R_f;
  const float fl = floorf(t), fr = t - fl;
  if (fr > 1e-3f && fr < 0.999f && t < 1e6f) return (int)fl;
  return (int)floor(sqrt(x * x + y * y) * R_res_inv);
}

// Both bin indices of one M2DP projection (M2DP.cpp:59-62) with ONE accept test and no control flow in the fast path:
// hipcc turns the short-circuit conditions of the two functions above into a chain of exec-mask branches (~100 issue
// slots per projection, most of them not arithmetic).  Sector: four sign / order bits -> a 16-entry nibble table; ring:
// the raw v_sqrt_f32 (1 ulp; sqrtf() is a 16-instruction correctly rounded sequence) - the 1e-3-bin margins cover both.
// Everything the fast path cannot vouch for falls back to the reference's fp64 expressions, as above.
__host__ __device__ __forceinline__ void polar_bins16(double y, double x, double S_res_inv, double R_res_inv, float R_f,
                                                     int& si, int& ri) {
  const float xf = (float)x, yf = (float)y;
  const float a = fabsf(xf), b = fabsf(yf);
  const float mn = fminf(a, b), mx = fmaxf(a, b);
  const float e = 1e-3f * mx, t = 0.414213568f * mx;                       // tan(pi/8)
#if defined(__HIP_DEVICE_COMPILE__)
  const float r = __builtin_amdgcn_sqrtf(xf * xf + yf * yf);
#else
  const float r = sqrtf(xf * xf + yf * yf);
#endif
  const float tt = r * R_f, fl = floorf(tt), fr = tt - fl;
  const bool ok = (mn > e) & (fabsf(mn - t) > e) & ((mx - mn) > e) & (mx > 1e-30f) & (mx < 3.0e38f) &
                  (fr > 1e-3f) & (fr < 0.999f) & (tt < 1e6f);
  if (ok) {
    // index = 8 (y < 0) + 4 (x < 0) + 2 (|y| > |x|) + (min > max tan(pi/8)); entries: quadrant of theta + pi (see polar_sector16)
    // q0 (y<0,x<0): tan = |y|/|x|; q1 (y<0,x>0): tan = |x|/|y|; q2 (y>0,x>0): |y|/|x|; q3 (y>0,x<0): |x|/|y|
    const unsigned idx = ((yf < 0.f) ? 8u : 0u) | ((xf < 0.f) ? 4u : 0u) | ((b > a) ? 2u : 0u) | ((mn > t) ? 1u : 0u);
    // y>=0,x>=0 (q2: 8..11): 8, 9, 11, 10 | y>=0,x<0 (q3: 12..15; big = |x|>|y| = !(b>a)): 15, 14, 12, 13
    // y<0,x>=0 (q1: 4..7; big = !(b>a)): 7, 6, 4, 5 | y<0,x<0 (q0: 0..3): 0, 1, 3, 2
    constexpr unsigned long long TAB = 0x2310ull << 48 | 0x5467ull << 32 | 0xDCEFull << 16 | 0xAB98ull;
    si = (int)((TAB >> (4 * idx)) & 15ull);
    ri = (int)fl;
    return;
  }
  si = (int)floor((atan2(y, x) + M_PI) * S_res_inv);
  ri = (int)floor(sqrt(x * x + y * y) * R_res_inv);
}

// ---------------------------------------------------------------------------------------------------------------- M2DP projections in fp32
// m2dp_bin_kernel evaluates 16 plane projections per point (M2DP.cpp:56-62).  The reference's fp64 dots xp = xProj.pt, yp = yProj.pt
// only decide two integers, so the fast path takes them in fp32 as well - from the point and the plane vectors rounded to float - and the
// accept test carries the dot's error on top of the classifier's own:
//   |xpf - xp| <= 5 * 2^-24 * sum |p_i q_i|  (two operand roundings, three roundings of the fma chain)  <= 3e-7 * (|q0| + |q1| + |q2|)
// because the plane vectors have components <= 1 (pr_api.cpp: xa - (xa.n) n and n x that).  With delta = 4e-7 * (|q0f| + |q1f| + |q2f|):
//   sector: accepted when min(|x|,|y|), | min - tan(pi/8) max | and max - min are all above  2e-5 max + 2 delta + 1e-30
//           (2 delta bounds the error of each of the three quantities; 2e-5 max covers the fp32 arithmetic of the test itself, ~2e-7 max);
//   ring:   accepted when frac(r R) is within 0.5 - (2e-5 + 1.5 delta R) of 0.5  (|r_f - r| <= sqrt(2) delta + 1 ulp of v_sqrt_f32 and the
//           products: < 2e-6 bins for r R < 16; larger r R only has to stay >= 8, the bins that are dropped).
// Anything else - boundaries, zeros of either sign, NaN, overflow, denormals - evaluates the reference's fp64 expressions on the fp64 dot.
// About 0.03 % of the projections take that path (0.7 % with the 1e-3 margins of polar_bins16, i.e. a third of all WAVES).
struct PointF {
  float q0, q1, q2;
  float dq2;   // 2 delta + 1e-30
  float hw;    // accepted half-width of frac(r R) around 0.5
};
__host__ __device__ __forceinline__ PointF make_pointf(double q0, double q1, double q2, float R_f) {
  PointF p;
  p.q0 = (float)q0; p.q1 = (float)q1; p.q2 = (float)q2;
  const float delta = 4e-7f * ((fabsf(p.q0) + fabsf(p.q1)) + fabsf(p.q2));
  p.dq2 = 2.f * delta + 1e-30f;
  p.hw = 0.5f - (2e-5f + 1.5f * delta * R_f);
  return p;
}
// pf: the plane's xProj[3], yProj[3] rounded to float.  Branch-free (the caller runs several projections side by side so that their
// dependent chains overlap); returns false when the fast path cannot vouch for the bins - si, ri are then meaningless.
__host__ __device__ __forceinline__ bool proj_bins16_fast(const PointF& pt, const float* pf, float R_f, int& si, int& ri) {
  const float xf = fmaf(pf[0], pt.q0, fmaf(pf[1], pt.q1, pf[2] * pt.q2));
  const float yf = fmaf(pf[3], pt.q0, fmaf(pf[4], pt.q1, pf[5] * pt.q2));
  const float a = fabsf(xf), b = fabsf(yf);
  const float mn = fminf(a, b), mx = fmaxf(a, b);
  const float e = fmaf(2e-5f, mx, pt.dq2), t = 0.414213568f * mx;            // tan(pi/8)
  const float d3 = fminf(fminf(mn, fabsf(mn - t)), mx - mn);
#if defined(__HIP_DEVICE_COMPILE__)
  const float r = __builtin_amdgcn_sqrtf(fmaf(xf, xf, yf * yf));
#else
  const float r = sqrtf(fmaf(xf, xf, yf * yf));
#endif
  const float tt = r * R_f, fl = floorf(tt), fr = tt - fl;
  // (no range test on tt: from 2^23 on floor(tt) = tt and fr = 0, +Inf gives NaN - both fail the ring test; below that a huge ring index is
  // merely one of the dropped ones)
  const bool ok = (d3 > e) & (fabsf(fr - 0.5f) < pt.hw);
  // Sector: sign bits instead of comparisons (an accepted point has neither coordinate, nor a - b, nor mn - t at zero):
  //   shift = 16 [x < 0] + 8 [|y| > |x|] + 4 [min < max tan(pi/8)], table by the sign of y - the nibble table of polar_bins16 with the
  //   entries of each pair exchanged (its last index bit is min > max tan(pi/8))
  const unsigned sx = __builtin_bit_cast(unsigned, xf) >> 31, sab = __builtin_bit_cast(unsigned, a - b) >> 31,
                 smt = __builtin_bit_cast(unsigned, mn - t) >> 31;
  const unsigned sh = (((sx << 1 | sab) << 1) | smt) << 2;
  constexpr unsigned TAB_YPOS = 0xCDFEBA89u, TAB_YNEG = 0x32014576u;
  si = (int)((((int)__builtin_bit_cast(unsigned, yf) < 0 ? TAB_YNEG : TAB_YPOS) >> sh) & 15u);
#if defined(__HIP_DEVICE_COMPILE__)
  ri = (int)fl;                                  // (v_cvt_i32_f32 saturates; a rejected projection's ri is never used)
#else
  ri = ok ? (int)fl : 0;
#endif
  return ok;
}
// the reference's evaluation of one projection (M2DP.cpp:56-62; no zero seed: SURVEY.md H4 / N5).  Compile without fp contraction.
__host__ __device__ __forceinline__ void proj_bins16_exact(double q0, double q1, double q2, const double* pl, double S_res_inv,
                                                          double R_res_inv, int& si, int& ri) {
  const double xp = pl[0] * q0 + (pl[1] * q1 + pl[2] * q2);
  const double yp = pl[3] * q0 + (pl[4] * q1 + pl[5] * q2);
  si = (int)floor((atan2(yp, xp) + M_PI) * S_res_inv);
  ri = (int)floor(sqrt(xp * xp + yp * yp) * R_res_inv);
}

}  // namespace pr
