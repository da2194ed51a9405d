// m2dp_match_h.hip — all-pairs M2DP distance on the gfx950 f16 matrix cores with split-f16 operands (processM2DP.m:12-22).
//
// Same GEMM and fused 4x4 block-min epilogue as m2dp_match.hip ([4m x 192] . [192 x 4n] per channel), same arithmetic
// idea as sc_match_h.hip: every fp32 entry x (scaled by 2^8) is carried as hi = f16(x), lo = f16(x - hi), every product as
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation (16x the fp32 MFMA rate for 3x the products).
// Both operands are inputs, so the split happens once at pack time; the kernel is a plain tiled GEMM.
//
// Layout (m2dp_pack_h): per channel, per tile of 32 rows (8 signatures x 4 variants): [K-step 0..11][hi | lo][lane][8 f16],
// lane l holds row l & 31 at k = 16*step + 8*(l >> 5) + 0..7: exactly the A (or B) operand registers of one 32x32x16 MFMA,
// one 16-byte load per operand and K-step; 24 576 B per tile and channel (the size of the fp32 tile).
// A workgroup (4 waves) owns 4 query tiles (32 queries) resident in LDS (96 KB) and sweeps the DB 8 tiles at a time; each
// wave computes all 4 query tiles x its own 2 DB tiles per step (24 MFMAs per K-step and 4 DB operand loads: half the
// L1 traffic per MFMA of a 2x2 wave tile), DB operands straight from L2/L1 into VGPRs.
#include <cstdlib>

#include "kernels.hpp"

namespace pr {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TB = M2_TILE * 4;            // bytes per (channel, tile)
constexpr int TV = TB / 16;                // 16-byte vectors per tile: [step 12][hi|lo][64]

// One thread per 8 consecutive values of a variant row: they are 8 consecutive halves of the image (16 bytes of the hi tile, 16 of the lo
// tile), so the image leaves as two 16-byte stores per thread (round 5; the first form stored every half by itself - 2-byte scattered
// stores, 0.94 ms per 50 000 signatures at ~1 TB/s).  The arithmetic per value is unchanged: the same image, bit for bit.
template <typename T>
__global__ __launch_bounds__(192) void m2dp_pack_h_kernel(const T* __restrict__ sig, int sigs, unsigned short* __restrict__ packed,
                                                           int tiles, int sg0 /* place of sig's first signature in the image (an append) */) {
  const int sg = sg0 + blockIdx.x, tid = threadIdx.x;      // one workgroup per signature (4 variant rows x 384 = 192 runs of 8)
  const int tile = sg >> 3, e = sg & 7;
  const int var = tid / 48, c = (tid - var * 48) * 8, ch = c / 192, k = c - ch * 192;
  const T* src = sig + ((size_t)blockIdx.x * 4 + var) * 384 + c;
  T x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = src[i];
  unsigned short h[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const double v = (double)x[i] * 256.0;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (double)hi);
    h[i] = __builtin_bit_cast(unsigned short, hi);
    l[i] = __builtin_bit_cast(unsigned short, lo);
  }
  const int row = e * 4 + var, step = k >> 4, lane = (((k >> 3) & 1) << 5) | row;
  const size_t base = ((size_t)ch * tiles + tile) * (TB / 2) + ((size_t)(step * 2) * 64 + lane) * 8;      // (k & 7 = 0: a 16-byte boundary)
  u32x4 hv, lv;
#pragma unroll
  for (int i = 0; i < 4; i++) { hv[i] = h[2 * i] | ((unsigned)h[2 * i + 1] << 16); lv[i] = l[2 * i] | ((unsigned)l[2 * i + 1] << 16); }
  *reinterpret_cast<u32x4*>(packed + base) = hv;
  *reinterpret_cast<u32x4*>(packed + base + 64 * 8) = lv;
}

struct HL { u32x4 h, l; };

// QTB = query tiles per workgroup: 4 (96 KB of LDS: one workgroup per CU) or 3 (72 KB: two, their epilogues and request stalls overlap)
// LO = false: the single-product arithmetic (PR_SC_ARITH_F16): the hi halves of the same tiles only, one MFMA per product
template <int QTB, bool LO>
__global__ __launch_bounds__(256, 2) void m2dp_match_h_kernel(const u32x4* __restrict__ qpk, const u32x4* __restrict__ dpk,
                                                              float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                              int m, int n, int QT, int DT, int nsplit, int DTS /* channel stride of the DB image in tiles */) {
  extern __shared__ __attribute__((aligned(16))) u32x4 ldsv[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int split = b % nsplit;
  b /= nsplit;
  const int ch = b & 1, qt4 = b >> 1;               // QTB query tiles per workgroup
  if (qt4 * QTB * 8 >= m) return;                   // padding tiles only (the query tile count is rounded up to a multiple of 12)
  const int DT8 = (DT + 7) / 8;                     // DB swept in steps of 8 tiles (2 per wave)
  const int s0 = (int)((long long)DT8 * split / nsplit), s1 = (int)((long long)DT8 * (split + 1) / nsplit);
  {
    const u32x4* src = qpk + ((size_t)ch * QT + (size_t)qt4 * QTB) * TV;
    for (int i = tid; i < QTB * TV; i += 256) ldsv[i] = src[i];
  }
  __syncthreads();
  const u32x4* la = ldsv + lane;
  float* dist = ch ? dist_i : dist_p;
  // distances of this workgroup's 32 query rows: invalid rows fall outside the descriptor's range
  const int qrow0 = qt4 * QTB * 8;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      dist + (size_t)qrow0 * n, 0, (qrow0 < m ? (m - qrow0 < QTB * 8 ? m - qrow0 : QTB * 8) : 0) * n * 4, 0x00020000);
  const unsigned st_lane = (unsigned)((2 * (lane & 3) + (lane >> 5)) * n + ((lane & 31) >> 2)) * 4u;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (s0 >= s1) return;
  // DB operand stream of this wave: tiles (8 s + 2 w) and (8 s + 2 w + 1), requested two K-steps ahead (three register sets);
  // the query tiles come from LDS one tile at a time, each reloaded for the next K-step right behind its own six MFMAs.
  // launch_bounds(256, 2) caps the wave at 256 unified registers, which keeps the 128 accumulators in ArchVGPRs (no
  // v_accvgpr_read in the epilogue).  The packed buffer has a readable tail for the requests past the last sweep step.
  const u32x4* pb = dpk + ((size_t)ch * DTS + (size_t)s0 * 8 + w * 2) * TV + lane;
  struct BSet { HL b0, b1; };
  BSet bs[3];               // K-steps st, st + 1, st + 2 (period 3 divides the 12 K-steps of a sweep step)
  HL a[QTB];                // the 4 query tiles of the current K-step; tile t is reloaded for the next K-step behind its own MFMAs
#define LDB(dst, p, st) { dst.b0.h = (p)[(st) * 128]; dst.b1.h = (p)[TV + (st) * 128]; if constexpr (LO) { dst.b0.l = (p)[(st) * 128 + 64]; dst.b1.l = (p)[TV + (st) * 128 + 64]; } }
#define LDA1(t, st) { a[t].h = la[(t) * TV + (st) * 128]; if constexpr (LO) a[t].l = la[(t) * TV + (st) * 128 + 64]; }
#define MF(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)
  LDB(bs[0], pb, 0)
  LDB(bs[1], pb, 1)
#pragma unroll
  for (int t = 0; t < QTB; t++) LDA1(t, 0)
  for (int s = s0; s < s1; s++) {
    const int dt0 = s * 8 + w * 2;
    const u32x4* pn = pb + 8 * TV;           // same wave column, next sweep step
    f32x16 acc[QTB][2];
#define SBAR() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int st = 0; st < 12; st++) {
      // Every request sits alone in the gap behind one MFMA (a 32x32x16 MFMA occupies the pipe for 32 cycles, a request
      // issues in 5-20): batched at the top of the K-step the 12 requests left the matrix pipe idle for ~160 of 930 cycles.
      // DB operands two K-steps ahead, one 16-byte piece behind the first MFMA of each query tile; query tile t is
      // reloaded for the next K-step behind the last MFMAs that read it.
      const BSet& c = bs[st % 3];
      BSet& nx = bs[(st + 2) % 3];
      const u32x4* pq = (st + 2 < 12) ? pb + (st + 2) * 128 : pn + (st + 2 - 12) * 128;
      const bool first = st == 0;
#pragma unroll
      for (int t = 0; t < QTB; t++) {
        SBAR();
        acc[t][0] = MF(a[t].h, c.b0.h, first ? zero : acc[t][0]);
        SBAR();
        if constexpr (LO) {
          if (t == 0) nx.b0.h = pq[0]; else if (t == 1) nx.b0.l = pq[64]; else if (t == 2) nx.b1.h = pq[TV]; else nx.b1.l = pq[TV + 64];
          if (QTB == 3 && t == 2) { SBAR(); nx.b1.l = pq[TV + 64]; }
        } else {
          if (t == 0) nx.b0.h = pq[0]; else if (t == 1) nx.b1.h = pq[TV];
        }
        SBAR();
        acc[t][1] = MF(a[t].h, c.b1.h, first ? zero : acc[t][1]);
        if constexpr (LO) {
          acc[t][0] = MF(a[t].h, c.b0.l, acc[t][0]);
          acc[t][1] = MF(a[t].h, c.b1.l, acc[t][1]);
        }
        SBAR();
        if constexpr (LO) {
          a[t].h = la[t * TV + ((st + 1) % 12) * 128];              // the query tiles do not depend on the sweep step
          SBAR();
          acc[t][0] = MF(a[t].l, c.b0.h, acc[t][0]);
          acc[t][1] = MF(a[t].l, c.b1.h, acc[t][1]);
          SBAR();
          a[t].l = la[t * TV + ((st + 1) % 12) * 128 + 64];
        } else {
          a[t].h = la[t * TV + ((st + 1) % 12) * 128];
        }
        SBAR();
      }
    }
#undef SBAR
    // 12 K-steps advance the period-3 rotation by 0: bs[0], bs[1] already hold K-steps 0, 1 of the next sweep step
    pb = pn;
    // epilogue: C layout col = lane&31 -> (entry = col>>2, variant = col&3); row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    // -> (query = 2*(reg>>2) + (lane>>5), variant = reg&3).  d = min (1-dot)/2 = 0.5 - 0.5 * 2^-16 * max dot.
    // Branch-free: quad reductions as DPP-source v_max, one buffer store per (query tile, DB tile, query pair) whose
    // invalid lanes are out of range.
    // After the quad reduction all four lanes of a quad hold the block maximum, so lane (lane & 3) = gq keeps the value
    // of query pair gq: ONE store per (query tile, DB tile) with all 64 lanes active instead of four with 16.
#pragma unroll
    for (int i = 0; i < QTB; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        float sel = 0.f;
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
          float mx = fmaxf(fmaxf(acc[i][j][gq * 4], acc[i][j][gq * 4 + 1]),
                           fmaxf(acc[i][j][gq * 4 + 2], acc[i][j][gq * 4 + 3]));
          mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xf, 0xf, true)));   // quad_perm [1,0,3,2]
          mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x4E, 0xf, 0xf, true)));   // quad_perm [2,3,0,1]
          sel = ((lane & 3) == gq) ? mx : sel;
        }
        // query row (local to the workgroup's 32) = 8 i + 2 (lane & 3) + (lane >> 5), entry = 8 (dt0 + j) + ((lane & 31) >> 2)
        const int drow = (dt0 + j) * 8 + ((lane & 31) >> 2);
        const unsigned off = (drow < n) ? st_lane + (unsigned)((8 * i) * n + (dt0 + j) * 8) * 4u : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(sel, -0x1p-17f, 0.5f)), rd, (int)off, 0, 0);   // processM2DP.m:15,19
      }
  }
#undef LDB
#undef LDA1
#undef MF
}


// A/B variant (PR_M2_WAVES=8, not the default): EIGHT waves per workgroup (two per SIMD), each 4 query tiles x ONE DB tile (64
// accumulators, 12 MFMAs and 2 DB operand loads per K-step: the same loads per MFMA as the 4-wave kernel above).  The 4-wave
// kernel runs one wave per SIMD (96 KB of query tiles leave room for one workgroup per CU), so nothing covers its epilogue (8
// tiles x ~40 instructions per sweep step = 14 % of the step) nor its operand-request stalls: the matrix pipe is 76 % busy.
// Here the second wave of every SIMD fills those holes - and the launch is 2 % SLOWER (5.64 against 5.50 ms at 4096 x 50k,
// same box, same run): the chip runs this kernel at ~1.7 GHz against its power limit, so a busier matrix pipe is paid back
// in clock (MI355X_MICROARCH.md "DVFS give-back").  Same LDS image, same packed layouts, same result bit for bit.
template <bool LO>
__global__ __launch_bounds__(512, 2) void m2dp_match_h8_kernel(const u32x4* __restrict__ qpk, const u32x4* __restrict__ dpk,
                                                               float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                               int m, int n, int QT, int DT, int nsplit, int DTS /* channel stride of the DB image in tiles */) {
  constexpr int QTB = 4;
  extern __shared__ __attribute__((aligned(16))) u32x4 ldsv[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..7
  int b = blockIdx.x;
  const int split = b % nsplit;
  b /= nsplit;
  const int ch = b & 1, qt4 = b >> 1;
  if (qt4 * QTB * 8 >= m) return;
  const int DT8 = (DT + 7) / 8;                     // DB swept in steps of 8 tiles (one per wave)
  const int s0 = (int)((long long)DT8 * split / nsplit), s1 = (int)((long long)DT8 * (split + 1) / nsplit);
  {
    const u32x4* src = qpk + ((size_t)ch * QT + (size_t)qt4 * QTB) * TV;
    for (int i = tid; i < QTB * TV; i += 512) ldsv[i] = src[i];
  }
  __syncthreads();
  const u32x4* la = ldsv + lane;
  float* dist = ch ? dist_i : dist_p;
  const int qrow0 = qt4 * QTB * 8;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      dist + (size_t)qrow0 * n, 0, (qrow0 < m ? (m - qrow0 < QTB * 8 ? m - qrow0 : QTB * 8) : 0) * n * 4, 0x00020000);
  const unsigned st_lane = (unsigned)((2 * (lane & 3) + (lane >> 5)) * n + ((lane & 31) >> 2)) * 4u;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (s0 >= s1) return;
  const u32x4* pb = dpk + ((size_t)ch * DTS + (size_t)s0 * 8 + w) * TV + lane;
  // DB operands of K-steps st .. st + RD - 1.  Split-f16: a K-step is 12 MFMAs per wave (~770 cycles with the SIMD's other wave), two ahead cover an L2
  // hit.  Single product: 4 MFMAs (~260 cycles) - two ahead were 500 cycles, LESS than the L2's latency under load, and the matrix pipe sat 37 %
  // busy waiting for operands (round 6 counters); eleven ahead (RD = 12: the whole K loop of the next DB tile is in flight; 12 K-steps advance the rotation by 0 for RD in {3, 4, 6, 12}).  4096 x 50k: RD 3: 2.93 ms, 4: 2.55, 6: 2.34, 12: 2.27 = 0.34 -> 0.445 of 2.5 PF (alternating runs, tools/exp_m2dp_single.py)
#ifndef M2_RD1
#define M2_RD1 12
#endif
  constexpr int RD = LO ? 3 : M2_RD1;
  HL bs[RD];
  HL a[QTB];
#define MF(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)
#define SBAR() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
  for (int r = 0; r < RD - 1; r++) { bs[r].h = pb[r * 128]; if constexpr (LO) bs[r].l = pb[r * 128 + 64]; }
#pragma unroll
  for (int t = 0; t < QTB; t++) { a[t].h = la[t * TV]; if constexpr (LO) a[t].l = la[t * TV + 64]; }
  for (int s = s0; s < s1; s++) {
    const int dt0 = s * 8 + w;
    const u32x4* pn = pb + 8 * TV;
    f32x16 acc[QTB];
#pragma unroll
    for (int st = 0; st < 12; st++) {
      const HL& c = bs[st % RD];
      HL& nx = bs[(st + RD - 1) % RD];
      const u32x4* pq = (st + RD - 1 < 12) ? pb + (st + RD - 1) * 128 : pn + (st + RD - 1 - 12) * 128;
      const bool first = st == 0;
#pragma unroll
      for (int t = 0; t < QTB; t++) {
        SBAR();
        acc[t] = MF(a[t].h, c.h, first ? zero : acc[t]);
        SBAR();
        if (t == 0) nx.h = pq[0]; else if (LO && t == 1) nx.l = pq[64];
        SBAR();
        if constexpr (LO) {
          acc[t] = MF(a[t].h, c.l, acc[t]);
          SBAR();
          a[t].h = la[t * TV + ((st + 1) % 12) * 128];
          SBAR();
          acc[t] = MF(a[t].l, c.h, acc[t]);
          SBAR();
          a[t].l = la[t * TV + ((st + 1) % 12) * 128 + 64];
        } else {
          a[t].h = la[t * TV + ((st + 1) % 12) * 128];
        }
        SBAR();
      }
    }
    pb = pn;
#pragma unroll
    for (int i = 0; i < QTB; i++) {
      float sel = 0.f;
#pragma unroll
      for (int gq = 0; gq < 4; gq++) {
        float mx = fmaxf(fmaxf(acc[i][gq * 4], acc[i][gq * 4 + 1]), fmaxf(acc[i][gq * 4 + 2], acc[i][gq * 4 + 3]));
        mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xf, 0xf, true)));   // quad_perm [1,0,3,2]
        mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x4E, 0xf, 0xf, true)));   // quad_perm [2,3,0,1]
        sel = ((lane & 3) == gq) ? mx : sel;
      }
      const int drow = dt0 * 8 + ((lane & 31) >> 2);
      const unsigned off = (drow < n) ? st_lane + (unsigned)((8 * i) * n + dt0 * 8) * 4u : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(sel, -0x1p-17f, 0.5f)), rd, (int)off, 0, 0);   // processM2DP.m:15,19
    }
  }
#undef SBAR
#undef MF
}

}  // namespace

void launch_m2dp_pack_h(hipStream_t st, const void* sig, int dtype, int sigs, void* packed, int tiles, int sg0) {
  if (sigs <= 0) return;
  if (dtype == 0)
    hipLaunchKernelGGL(m2dp_pack_h_kernel<double>, dim3(sigs), dim3(192), 0, st, (const double*)sig, sigs, (unsigned short*)packed, tiles, sg0);
  else
    hipLaunchKernelGGL(m2dp_pack_h_kernel<float>, dim3(sigs), dim3(192), 0, st, (const float*)sig, sigs, (unsigned short*)packed, tiles, sg0);
}

void launch_m2dp_match_h(hipStream_t st, const void* qpk, int m, const void* dpk, int n, float* d_p, float* d_i, int single, int dts) {
  if (m <= 0 || n <= 0) return;
  const int QT = m2_qtiles(m), DT = m2_tiles(n), DTS = dts > 0 ? dts : DT;
  // 4 query tiles per workgroup; PR_M2_QTB=3 selects the two-workgroups-per-CU variant for A/B runs (measured 6.4 ms against
  // 5.55 ms at 4096 x 50k: the overlap of two workgroups does not pay for a third more DB operand traffic per MFMA)
  static const int qtb = (getenv("PR_M2_QTB") && atoi(getenv("PR_M2_QTB")) == 3) ? 3 : 4;
  static const bool eight = getenv("PR_M2_WAVES") && atoi(getenv("PR_M2_WAVES")) == 8;   // PR_M2_WAVES=8: the two-waves-per-SIMD kernel, for A/B runs
  const int base = (QT / qtb) * 2, DT8 = (DT + 7) / 8;
  // DB ranges per query block: enough workgroups for ~4 rounds, and among the next few counts the one that wastes the
  // least of its last round (the workgroups that hold padding tiles only return at once and do not count)
  const int real = ((m + qtb * 8 - 1) / (qtb * 8)) * 2, slots = 256 * (qtb == 3 ? 2 : 1);
  int ns0 = (1024 + real - 1) / real, nsplit = ns0;
  double best = 0.0;
  for (int ns = ns0; ns < ns0 + 12; ns++) {
    const long long blocks = (long long)real * ns, rounds = (blocks + slots - 1) / slots;
    const double util = (double)blocks / (double)(rounds * slots);
    if (util > best + 1e-9) { best = util; nsplit = ns; }
  }
  if (nsplit > DT8 / 4) nsplit = DT8 / 4;
  if (nsplit < 1) nsplit = 1;
  const size_t lds = (size_t)qtb * TB;
  if (single) {
    // one product per term: a wave of the 4-wave kernel is request-bound (8 MFMAs per 6 operand requests), so the single-product form runs
    // EIGHT waves per workgroup (two per SIMD, 4 query tiles x one DB tile each); PR_M2_WAVES=4 selects the 4-wave form for A/B runs
    static const bool four = getenv("PR_M2_WAVES") && atoi(getenv("PR_M2_WAVES")) == 4;
    const int base4 = (QT / 4) * 2;
    if (four) {
      auto* k1 = m2dp_match_h_kernel<4, false>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)4 * TB));
      hipLaunchKernelGGL(k1, dim3(base4 * nsplit), dim3(256), (size_t)4 * TB, st, static_cast<const u32x4*>(qpk),
                         static_cast<const u32x4*>(dpk), d_p, d_i, m, n, QT, DT, nsplit, DTS);
    } else {
      auto* k8 = m2dp_match_h8_kernel<false>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)4 * TB));
      hipLaunchKernelGGL(k8, dim3(base4 * nsplit), dim3(512), (size_t)4 * TB, st, static_cast<const u32x4*>(qpk),
                         static_cast<const u32x4*>(dpk), d_p, d_i, m, n, QT, DT, nsplit, DTS);
    }
    return;
  }
  if (eight && qtb == 4) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(m2dp_match_h8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(m2dp_match_h8_kernel<true>, dim3(base * nsplit), dim3(512), lds, st, static_cast<const u32x4*>(qpk),
                       static_cast<const u32x4*>(dpk), d_p, d_i, m, n, QT, DT, nsplit, DTS);
    return;
  }
  auto* k = qtb == 4 ? m2dp_match_h_kernel<4, true> : m2dp_match_h_kernel<3, true>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(base * nsplit), dim3(256), lds, st, static_cast<const u32x4*>(qpk),
                     static_cast<const u32x4*>(dpk), d_p, d_i, m, n, QT, DT, nsplit, DTS);
}

}  // namespace pr
