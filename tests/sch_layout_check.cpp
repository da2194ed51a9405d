// Host-only check of the split-f16 operand-pair contract of csrc/kernels.hpp (sch_qlo_byte, sch_a1_byte ... sch_b2_byte) against the image
// layout sc_pack.hip writes: walking the 2 x 4 lane groups x 8 K-slots of the two stage-1 MFMAs of a frequency, every one of the 60 products
// q_hi d_hi, q_hi d_lo, q_lo d_hi (20 rings each) must appear exactly once and nothing else may meet a non-zero DB value.
// Built and run by tests/test_abi.py::test_split_f16_operand_pairs_cover_the_three_products_once (no GPU).
#include "kernels.hpp"
#include <cstdio>
#include <map>
#include <string>
using namespace pr;
int main() {
  std::map<std::string, int> seen;
  int bad = 0;
  for (int pair = 1; pair <= 2; pair++)
    for (int lane = 0; lane < 64; lane += 16)                // entry 0 of each lane group (the entry index only selects the tile lane)
      for (int slot = 0; slot < 8; slot++) {
        const int kg = lane >> 4;
        const int ab = (pair == 1 ? sch_a1_byte(kg) : sch_a2_byte(kg)) + 2 * slot;              // byte of the 80-byte query row
        std::string a;
        if (ab < 40) a = "Qh" + std::to_string(ab / 2);
        else { int ring = -1; for (int r = 0; r < 20; r++) if (sch_qlo_byte(r) == ab) ring = r; if (ring < 0) { bad++; continue; } a = "Ql" + std::to_string(ring); }
        const int bb = (pair == 1 ? sch_b1_byte(lane) : sch_b2_byte(lane)) + 2 * slot;           // byte from the hi tile's first byte
        const bool lo = bb >= SCH_DTILE;
        const int tb = lo ? bb - SCH_DTILE : bb, tl = tb / 16, ts = (tb % 16) / 2, ring = 8 * (tl >> 4) + ts;
        if (tl >= 48) { bad++; continue; }
        std::string b;
        if (ring < 20) b = (lo ? "Dl" : "Dh") + std::to_string(ring);
        else if (lo && ring < 24) b = "Dh" + std::to_string(ring - 4);       // the pack's copy of D hi 16..19 behind D lo 16..19
        else b = "0";
        if (b == "0") continue;
        seen[a + "*" + b]++;
      }
  for (int r = 0; r < 20; r++) {
    const std::string R = std::to_string(r);
    for (const char* t : {"Qh%s*Dh%s", "Qh%s*Dl%s", "Ql%s*Dh%s"}) {
      char k[64]; snprintf(k, sizeof k, t, R.c_str(), R.c_str());
      if (seen[k] != 1) { printf("missing or repeated %s: %d\n", k, seen[k]); bad++; }
      seen.erase(k);
    }
  }
  for (auto& kv : seen) if (kv.second) { printf("stray product %s x %d\n", kv.first.c_str(), kv.second); bad++; }
  printf(bad ? "FAILED\n" : "ok\n");
  return bad != 0;
}
