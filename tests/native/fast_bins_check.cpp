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

// the fp32 projection classifier of m2dp_bin_kernel (proj_bins16_fast) against the reference's fp64 evaluation of the same projection
static long long fast_taken = 0, fast_total = 0;
static void check_proj(const double q[3], const double pl[6]) {
  const double S16 = 16 / (2.0 * M_PI), R = 8 / 45.0;
  int rs, rr;
  pr::proj_bins16_exact(q[0], q[1], q[2], pl, S16, R, rs, rr);
  float pf[6];
  for (int i = 0; i < 6; i++) pf[i] = (float)pl[i];
  const pr::PointF pt = pr::make_pointf(q[0], q[1], q[2], (float)R);
  int fs = -1, fr = -1;
  total++; fast_total++;
  if (!pr::proj_bins16_fast(pt, pf, (float)R, fs, fr)) return;
  fast_taken++;
  if (fs != rs || fr != rr) {
    if (bad < 10) printf("MISMATCH (projection) q=%a %a %a: sector %d/%d ring %d/%d\n", q[0], q[1], q[2], fs, rs, fr, rr);
    bad++;
  }
}
static void fuzz_projections(std::mt19937_64& g) {
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (int it = 0; it < 3000; it++) {
    // a plane as pr_api.cpp builds them: unit normal n, xProj = xa - (xa.n) n, yProj = n x xProj (components <= 1); every 50th degenerate
    double n[3] = {u(g), u(g), u(g)};
    const double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (double& v : n) v /= nn;
    double pl[6];
    const double d = n[0];
    pl[0] = 1.0 - d * n[0]; pl[1] = -d * n[1]; pl[2] = -d * n[2];
    pl[3] = n[1] * pl[2] - n[2] * pl[1]; pl[4] = n[2] * pl[0] - n[0] * pl[2]; pl[5] = n[0] * pl[1] - n[1] * pl[0];
    if (it % 50 == 49) for (double& v : pl) v = 0.0;
    for (int j = 0; j < 400; j++) {           // uniform points of a 45 m cloud
      const double q[3] = {u(g) * 45, u(g) * 45, u(g) * 12};
      check_proj(q, pl);
    }
    // points whose projection sits on / next to a sector or ring boundary: q = alpha xProj' + beta yProj' + gamma n with (alpha, beta)
    // on the boundary (xProj', yProj': the normalised in-plane axes, so that the projection is ~ (alpha |xProj|, beta |yProj|))
    const double lx = std::sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]), ly = std::sqrt(pl[3] * pl[3] + pl[4] * pl[4] + pl[5] * pl[5]);
    if (lx < 1e-3 || ly < 1e-3) continue;
    for (int k = 0; k < 32; k++)
      for (int rb = 0; rb < 10; rb++)
        for (int dd = -6; dd <= 6; dd++) {
          const double th = k * M_PI / 16 - M_PI + (rb % 2 ? dd * 8e-6 : 0.37 + 0.01 * dd);
          const double rad = (rb % 2 ? 3.3 + rb : rb * 45.0 / 8 + dd * 2e-5 * (rb + 1));
          const double al = rad * std::cos(th) / (lx * lx), be = rad * std::sin(th) / (ly * ly), ga = u(g) * 20;
          const double q[3] = {al * pl[0] + be * pl[3] + ga * n[0], al * pl[1] + be * pl[4] + ga * n[1], al * pl[2] + be * pl[5] + ga * n[2]};
          check_proj(q, pl);
        }
  }
  const double sp[] = {0.0, -0.0, 5e-324, 1e-40, -1e-40, 1e-20, 1.0, -37.5, 1e30, 3.5e38, 1e300, INFINITY, NAN};
  const double pls[2][6] = {{1, 0, 0, 0, 1, 0}, {0.6, -0.48, 0.64, 0, 0.8, 0.6}};
  for (double a : sp) for (double b : sp) for (double c : sp) for (auto& pl : pls) { const double q[3] = {a, b, c}; check_proj(q, pl); }
}

int main() {
  std::mt19937_64 g(12345);
  fuzz_projections(g);
  printf("projection classifier: fast path taken for %lld of %lld (%.3f %% fall back)\n", fast_taken, fast_total,
         100.0 * (fast_total - fast_taken) / fast_total);
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
