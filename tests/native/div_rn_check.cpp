// Host fuzz of csrc/div_rn.hpp against the IEEE division (x86 fma and division are correctly rounded, as on gfx950):
// the operand ranges of fuse_select (x = d - mean in [-2, 2], s = std in [1e-9, 10]) plus wide-range and edge operands.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../so_dso_place_recognition_amd/csrc/div_rn.hpp"

int main() {
  std::mt19937_64 g(99);
  long long bad = 0, total = 0;
  auto check = [&](double x, double s) {
    const double a = x / s, b = pr::div_rn(x, s, 1.0 / s);
    total++;
    if (std::memcmp(&a, &b, 8) != 0 && !(a != a && b != b)) { if (bad < 10) printf("MISMATCH %a / %a: %a vs %a\n", x, s, a, b); bad++; }
  };
  std::uniform_real_distribution<double> ux(-2.0, 2.0), us(-9.0, 1.0), ue(-200.0, 200.0);
  for (int i = 0; i < 20000000; i++) check(ux(g), std::pow(10.0, us(g)));
  for (int i = 0; i < 5000000; i++) check(ux(g) * std::pow(2.0, ue(g)), (1.0 + std::fabs(ux(g))) * std::pow(2.0, ue(g)));
  for (int i = 0; i < 2000000; i++) {           // fp32 distances about an fp64 mean, over an fp64 std: the real operands
    const float d = (float)std::fabs(ux(g)) * 0.25f;
    check((double)d - 0.31234567890123, 0.01 + std::fabs(ux(g)) * 0.1);
    check(0.0, 0.01 + std::fabs(ux(g)));
    check((double)d, 1.0); check((double)d, 0.5); check((double)d, 3.0);
  }
  printf("checked %lld quotients, %lld mismatches\n", total, bad);
  return bad ? 1 : 0;
}
