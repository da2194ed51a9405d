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

__device__ __forceinline__ int polar_sector(double y, double x, double S_res_inv, float S_f) {
  const float t = (atan2f((float)y, (float)x) + 3.14159274f) * S_f;
  const float fl = floorf(t), fr = t - fl;
  if (fr > 1e-3f && fr < 0.999f) return (int)fl;
  return (int)floor((atan2(y, x) + M_PI) * S_res_inv);
}

__device__ __forceinline__ int polar_ring(double x, double y, double R_res_inv, float R_f) {
  const float xf = (float)x, yf = (float)y;
  const float t = sqrtf(xf * xf + yf * yf) * R_f;
  const float fl = floorf(t), fr = t - fl;
  if (fr > 1e-3f && fr < 0.999f && t < 1e6f) return (int)fl;
  return (int)floor(sqrt(x * x + y * y) * R_res_inv);
}

}  // namespace pr
