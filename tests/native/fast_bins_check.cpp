// Host-side fuzz of the bin classifiers of csrc/fast_bins.hpp against the reference's fp64 expressions
// (M2DP.cpp:59-62, SC.cpp:37-38): uniform points, points on and next to every sector / ring boundary, zeros of both signs,
// denormals, huge values, NaN.  Built and run by tests/test_fast_bins.py (hipcc, host code only).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>

#include "../../so_dso_place_recognition_amd/csrc/fast_bins.hpp"

static long long bad = 0, total = 0;

static void check(double y, double x) {
  const double S16 = 16 / (2.0 * M_PI), S60 = 60 / (2.0 * M_PI), R = 8 / 45.0;
  const int r16 = (int)std::floor((std::atan2(y, x) + M_PI) * S16);
  const int r60 = (int)std::floor((std::atan2(y, x) + M_PI) * S60);
  const int rr = (int)std::floor(std::sqrt(x * x + y * y) * R);
  const int f16 = pr::polar_sector16(y, x, S16);
  const int f60 = pr::polar_sector(y, x, S60, (float)S60);
  const int fr = pr::polar_ring(x, y, R, (float)R);
  int c16, cr;
  pr::polar_bins16(y, x, S16, R, (float)R, c16, cr);   // the combined classifier of m2dp_bin_kernel
  total++;
  if (c16 != r16 || cr != rr) {
    if (bad < 10) printf("MISMATCH (combined) y=%a x=%a: sector16 %d/%d ring %d/%d\n", y, x, c16, r16, cr, rr);
    bad++;
  }
  if (f16 != r16 || f60 != r60 || fr != rr) {
    if (bad < 10) printf("MISMATCH y=%a x=%a: sector16 %d/%d sector60 %d/%d ring %d/%d\n", y, x, f16, r16, f60, r60, fr, rr);
    bad++;
  }
}

int main() {
  std::mt19937_64 g(12345);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (int i = 0; i < 4000000; i++) check(u(g) * 50, u(g) * 50);
  const double scales[] = {1e-300, 1e-40, 1e-6, 1.0, 37.0, 1e6, 1e30, 1e200};
  for (int k = 0; k <= 64; k++) {             // every multiple of pi/32: all 16- and some 60-sector boundaries
    for (int j = 0; j <= 60; j++) {
      const double th = (k ? k * M_PI / 32 : j * 2 * M_PI / 60) - M_PI;
      for (double s : scales)
        for (int d = -40; d <= 40; d++) {
          const double t2 = th + d * 1e-17 * (1 + std::fabs(th)) * (d % 2 ? 1e3 : 1) * (d % 5 ? 1 : 1e9);
          check(s * std::sin(t2), s * std::cos(t2));
          check(std::nextafter(s * std::sin(th), d) , s * std::cos(th));
        }
      if (k) break;
    }
  }
  const double sp[] = {0.0, -0.0, 5e-324, -5e-324, 1e-310, 1.0, -1.0, 1e308, -1e308, INFINITY, -INFINITY, NAN, 3.4e38, 3.5e38, 1e-45, 1e-46};
  for (double a : sp) for (double b : sp) check(a, b);
  for (int r = 0; r <= 9; r++)                  // ring boundaries r * 45 / 8
    for (int d = -30; d <= 30; d++) {
      const double rad = r * 45.0 / 8 + d * 1e-15 * (d % 3 ? 1 : 1e9);
      for (int a = 0; a < 16; a++) check(rad * std::sin(a * 0.4 + 0.1), rad * std::cos(a * 0.4 + 0.1));
    }
  printf("checked %lld points, %lld mismatches\n", total, bad);
  return bad ? 1 : 0;
}
