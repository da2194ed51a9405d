// sc_match_f.hip — EXPERIMENT, not built into the library: the SC matcher (processSC.m:22-33) of the single-product arithmetic
// (PR_SC_ARITH_F16), eight waves per workgroup, WITHOUT permlane swaps.  Result (MI355X, 4096 x 100k, bit-identical distance error
// 1.285e-4 against the fp64 oracle): 21.2 ms per launch against 18.5 ms of sc_match_e<single product> - the 240 VALU instructions a unit
// saves do not pay for its 32 additional 32x32x16 MFMAs; two waves of a SIMD overlap their MFMA and VALU streams far less than the
// instruction counts suggest (either order of the stage-2 chains, one or two tile sets: the same time).  To try it: add it to the Makefile,
// declare launch_sc_match_f in kernels.hpp, build the [+|-][4][64][8] f16 constant table (lanes 0-31: w cos, lanes 32-63: -/+ w sin,
// x 2^10, f = 8 block + e) and call it for sc_mode == PR_SC_ARITH_F16.
//
// Same mathematics and the same packed images as sc_match_e.hip<single product> (SCF_* layout: query rows [Qr;Qi] of 8 queries per group
// and frequency, DB column tiles Re | Im per frequency); what changes is the way from the stage-1 tiles to the stage-2 operands.  The
// single-product form of sc_match_e is bound by its VALU work (swaps, combinations, conversions, the E + |O| reduction: ~630
// instructions per unit against 94 MFMAs), so this kernel trades VALU for matrix work:
//
//  * stage 1 (v_mfma_f32_16x16x32_f16, K = 20 rings) per frequency f:  T1 = [Qr;Qi] . Dr^T   and   T2 = [Qi;Qr] . Di^T  (the query tile
//    read a second time with row ^ 8, as sc_match_h.hip does), so that with s = +1 in lanes 0-31 and -1 in lanes 32-63
//        F = T1 + s T2 = (Re S | Im S)      M = T1 - s T2 = (Re P | Im P)                S = Q conj(D) (forward), P = Q D (mirror)
//    already lie (Re | Im) by lane half: two packed fmas per register pair, NO permlane swap.
//  * stage 2 (v_mfma_f32_32x32x16_f16) takes K = (Re of 8 consecutive frequencies | Im of the same 8) - exactly what the two lane halves
//    of pack2(F_f, F_f+1) hold - against the constants A+ = (w cos | -w sin) and A- = (w cos | +w sin): the tile of the shifts +k and the
//    tile of the shifts -k (k = 0..30), 2 x 4 MFMAs per (register, direction) instead of 4, and the reduction is a plain max3 over
//    both tiles (no E + |O|): 16 instructions instead of 24.
//
// Per unit (8 queries x 16 entries): 62 + 64 MFMAs, ~390 VALU (sc_match_e<single>: 62 + 32 and ~630), two LDS reads + two vector loads
// per frequency.  Two waves per SIMD; 64 queries (8 groups) per workgroup, or ONE group shared by the eight waves (m <= 8, online).
#include "kernels.hpp"
#ifndef F_BD
#define F_BD 5          // depth of the DB operand ring (walk positions)
#endif
#ifndef F_AD
#define F_AD 3          // depth of the query operand ring
#endif

namespace pr {
namespace {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a8 __attribute__((aligned(8)));
typedef const u32x4_a8 __attribute__((address_space(3))) * lds_tile_p;

struct AOps { u32x4 h, x; };        // query rows [Qr;Qi] and the exchanged rows [Qi;Qr]
struct BOps { u32x4 re, im; };      // DB column tiles

#define MF0(d, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b))
#define M32Z(d, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b))
#define M32A(d, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b))
#define DRAIN() asm volatile("s_nop 9")
#define SB() __builtin_amdgcn_sched_barrier(0)
#define PIN(x) asm volatile("" : "+v"(x))

__device__ __forceinline__ unsigned pack2(float x, float y) {
  const f32x2 v = {x, y};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// (t1, t2) -> (F, M) in place, registers r0, r0 + 1
__device__ __forceinline__ void cmb_sigma(f32x4& t1, f32x4& t2, f32x2 sg, int r0) {
  const f32x2 a = {t1[r0], t1[r0 + 1]}, b = {t2[r0], t2[r0 + 1]};
  const f32x2 f = b * sg + a, m = a - b * sg;
  t1[r0] = f[0]; t1[r0 + 1] = f[1]; t2[r0] = m[0]; t2[r0 + 1] = m[1];
}
// 2 queries x 16 entries (lanes 0..31): d = (1 - max)/2 with the 2^-25 operand scaling folded in (processSC.m:30)
__device__ __forceinline__ void ep_store(float mx, __amdgpu_buffer_rsrc_t rd, int st_off) {
  const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
  mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(mx, -0x1p-26f, 0.5f)), rd, st_off, 0, 0);
}

// stage-2 operands of a unit: [8-frequency block][stage-1 register]; element e = frequencies (8 b + 2 e, 8 b + 2 e + 1), Re | Im by lane half
struct Ops { u32x4 F[4][4], M[4][4]; };
struct Consts { u32x4 p[4], m[4]; };        // A+ and A- of the four blocks

// The 32 walk positions (= frequencies; 31 is a ghost) form 16 pairs; pair J writes T[J & 1] = t1a, t2a, t1b, t2b.  The VALU work of
// pair J - 1 sits in the four gaps of pair J: combination of a, combination of b (two MFMAs behind the last one that wrote them),
// the four F conversions, the four M conversions.
template <int J, int G>
__device__ __forceinline__ void valu_slot(f32x4 (&T)[2][4], Ops& ops, f32x2 sg) {
  if constexpr (J >= 1) {
    f32x4 (&t)[4] = T[(J - 1) & 1];
    constexpr int B = (J - 1) >> 2, E = (J - 1) & 3;
    if constexpr (G == 0) { cmb_sigma(t[0], t[1], sg, 0); cmb_sigma(t[0], t[1], sg, 2); }
    if constexpr (G == 1) { cmb_sigma(t[2], t[3], sg, 0); cmb_sigma(t[2], t[3], sg, 2); }
    if constexpr (G == 2) {
#pragma unroll
      for (int r = 0; r < 4; r++) { unsigned h = pack2(t[0][r], t[2][r]); PIN(h); ops.F[B][r][E] = h; }
    }
    if constexpr (G == 3) {
#pragma unroll
      for (int r = 0; r < 4; r++) { unsigned h = pack2(t[1][r], t[3][r]); PIN(h); ops.M[B][r][E] = h; }
    }
  }
}

template <int NQG>
__global__ __launch_bounds__(512, 2) void sc_match_f_kernel(const char* __restrict__ qpk,   // [2][QG8][31][648 B]
                                                            const char* __restrict__ dpk,   // [2][DG][31][2][768 B] + zero groups
                                                            const u32x4* __restrict__ cst,  // [+|-][4 blocks][64] x 16 B
                                                            float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                            int m, int n, int QG8, int DG, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NW = 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware mapping as in sc_match_d.hip: all workgroups of an XCD work on ONE channel and the same quarter of the DB ranges
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ch = xcd & 1;
  const int range = (xcd >> 1) + 4 * (idx % nsplit), qgw = idx / nsplit;
  const int nrange = 4 * nsplit;
  const int g0 = (int)((long long)DG * range / nrange), g1 = (int)((long long)DG * (range + 1) / nrange);
  constexpr int QBLK = SCF_QBLK, QIMG = SCF_QIMG, DIMG = SCF_DIMG, QROW = 40;
  {  // the query groups of this workgroup -> LDS (linear copy; the packed image IS the LDS image) + zeroed tail
    const u32x4* src = reinterpret_cast<const u32x4*>(qpk + ((size_t)ch * QG8 + (size_t)qgw * NQG) * QIMG);
    u32x4* dst = reinterpret_cast<u32x4*>(lds);
    constexpr int NV = NQG * QIMG / 16;
    for (int i = tid; i < NV + 4; i += 64 * NW) dst[i] = (i < NV) ? src[i] : u32x4{0u, 0u, 0u, 0u};
  }
  __syncthreads();

  const int row = lane & 15, kg = lane >> 4, xrow = row ^ 8;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
  const int wq = w & (NQG - 1), dpar = (NW > NQG) ? (w / NQG) : 0;
  constexpr int GSTEP = NW / NQG;
  const int gcnt = g1 - g0;
  constexpr int BD = F_BD, AD = F_AD;
  const unsigned nat0 = lds0 + wq * QIMG + row * QROW + (row >= 8 ? 8 : 0) + kg * 16;
  const unsigned rot0 = lds0 + wq * QIMG + xrow * QROW + (xrow >= 8 ? 8 : 0) + kg * 16;
  const int voff = (lane < 48) ? lane * 16 : (int)0x80000000;     // lanes 48-63: out of range -> zeros (K = 24..31)
  float* dist = ch ? dist_i : dist_p;
  const char* dbase = dpk + ((size_t)ch * DG) * DIMG;
  const int qrow0 = qgw * (8 * NQG) + wq * 8;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      dist + (size_t)qrow0 * n, 0, (qrow0 < m ? (m - qrow0 < 8 ? m - qrow0 : 8) : 0) * n * 4, 0x00020000);
  const int pf_slot = (qgw & 31) * NW + w;
  constexpr int PFL = 2;                                           // cache lines per wave: 32 NW waves cover GSTEP x DIMG / 128 lines
  unsigned pf_sink = 0;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(cst), 0, 8192, 0x00020000);
  if (dpar >= gcnt) return;
  float sgs = lane < 32 ? 1.f : -1.f;
  asm volatile("" : "+v"(sgs));
  const f32x2 sg = {sgs, sgs};

  AOps At[AD];
  BOps Bt[BD];
  __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)(g0 + dpar) * DIMG), 0, DIMG, 0x00020000);
#define LD_B(P, IM, RSRC) { const u32x4 _v = __builtin_amdgcn_raw_buffer_load_b128(RSRC, voff, (P) * SCF_DFREQ + (IM) * SCH_DTILE, 0); \
                            if (IM) Bt[(P) % BD].im = _v; else Bt[(P) % BD].re = _v; }
#define LD_A(P, X) { const u32x4 _v = *reinterpret_cast<lds_tile_p>(((X) ? rot0 : nat0) + (P) * QBLK);       \
                     if (X) At[(P) % AD].x = _v; else At[(P) % AD].h = _v; }
// request K of the first ones of a unit: the DB tiles of walk positions 0 .. BD - 2, then the query tiles of positions 0 .. AD - 2
  constexpr int NREQ = 2 * (BD - 1) + 2 * (AD - 1);
#define FIRST_REQ(K, RSRC)                                                                       \
  { if constexpr ((K) < 2 * (BD - 1)) { LD_B(((K) < 2 * (BD - 1) ? (K) : 0) >> 1, (K) & 1, RSRC) }                        \
    else if constexpr ((K) < NREQ) { constexpr int _k = (K) < 2 * (BD - 1) ? 0 : (K) - 2 * (BD - 1); LD_A(_k >> 1, _k & 1) } }
  FIRST_REQ(0, rs) FIRST_REQ(1, rs) FIRST_REQ(2, rs) FIRST_REQ(3, rs) FIRST_REQ(4, rs) FIRST_REQ(5, rs) FIRST_REQ(6, rs) FIRST_REQ(7, rs)
  FIRST_REQ(8, rs) FIRST_REQ(9, rs) FIRST_REQ(10, rs) FIRST_REQ(11, rs) FIRST_REQ(12, rs) FIRST_REQ(13, rs) FIRST_REQ(14, rs) FIRST_REQ(15, rs)
  static_assert(NREQ <= 16, "FIRST_REQ list too short");

  for (int g = g0 + dpar; g < g1; g += GSTEP) {
    const int gn = g + GSTEP;                       // (the image ends with zero groups: the requests past the last group are harmless)
    const __amdgpu_buffer_rsrc_t rsn =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)gn * DIMG), 0, DIMG, 0x00020000);
    Ops ops;
    Consts cs;
    f32x4 T[2][4];
#define LDB(Q, IM) { if constexpr ((Q) < SC_NF) LD_B((Q) < SC_NF ? (Q) : 0, IM, rs) }
#define LDA(Q, X) { if constexpr ((Q) < SC_NF) LD_A((Q) < SC_NF ? (Q) : 0, X) }
#define VS(J, G) valu_slot<J, G>(T, ops, sg)
// one pair of walk positions: 4 MFMAs; behind each the requests for positions P + AD - 1 (query) and P + BD - 1 (DB) and 4 VALU
#define POS(J, H)                                                                                                      \
  { constexpr int _p = 2 * (J) + (H);                                                                                  \
    if constexpr (_p < SC_NF) {                                                                                        \
      SB(); MF0(T[(J) & 1][2 * (H)], At[_p % AD].h, Bt[_p % BD].re); SB(); LDA(_p + AD - 1, 0); LDB(_p + BD - 1, 0); VS(J, 2 * (H));         \
      SB(); MF0(T[(J) & 1][2 * (H) + 1], At[_p % AD].x, Bt[_p % BD].im); SB(); LDA(_p + AD - 1, 1); LDB(_p + BD - 1, 1); VS(J, 2 * (H) + 1); \
    } else {                                                                                                           \
      T[(J) & 1][2 * (H)] = f32x4{0.f, 0.f, 0.f, 0.f}; T[(J) & 1][2 * (H) + 1] = f32x4{0.f, 0.f, 0.f, 0.f};             \
      SB(); VS(J, 2 * (H)); SB(); VS(J, 2 * (H) + 1);                                                                  \
    } }
#define PAIR(J) POS(J, 0) POS(J, 1)
    PAIR(0) PAIR(1) PAIR(2) PAIR(3) PAIR(4) PAIR(5) PAIR(6) PAIR(7) PAIR(8) PAIR(9) PAIR(10) PAIR(11) PAIR(12) PAIR(13)
    // the last pairs also request the stage-2 constants (the operand rings are draining)
    POS(14, 0)
#pragma unroll
    for (int b = 0; b < 4; b++) cs.p[b] = __builtin_amdgcn_raw_buffer_load_b128(rc, lane * 16, b * 1024, 0);
    POS(14, 1)
#pragma unroll
    for (int b = 0; b < 4; b++) cs.m[b] = __builtin_amdgcn_raw_buffer_load_b128(rc, lane * 16, (4 + b) * 1024, 0);
    PAIR(15)
    SB(); DRAIN(); SB();
    VS(16, 0); VS(16, 1); VS(16, 2); VS(16, 3);
    SB();
    {  // L2 prefetch for the whole XCD (as in sc_match_e.hip): the group(s) of the iteration after next, 2 cache lines per wave
      asm volatile("" : : "v"(pf_sink));
      const int gp = (g - dpar) + 2 * GSTEP;
      const int pf_bytes = (gp + GSTEP <= DG) ? GSTEP * DIMG : (gp < DG ? (DG - gp) * DIMG : 0);
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)gp * DIMG), 0, pf_bytes, 0x00020000);
      int lp = lane;
      asm volatile("" : "+v"(lp));
      const int pf_off = (lp < PFL) ? (pf_slot * PFL + lp) * 128 : (int)0x80000000;
      pf_sink = __builtin_amdgcn_raw_buffer_load_b32(rp, pf_off, 0, 0);
    }
    // ---------------------------------------------------------------- stage 2: group S = 2 R + V (register R, forward | mirror V):
    // four chained MFMAs into the + tile, four into the - tile; the + tile is reduced behind the last - MFMAs, the - tile behind
    // the + MFMAs of the next group; the first requests of the next unit sit in the other gaps
    f32x16 tp[2], tm[2];
    float mx = -__builtin_inff();
    int le = lane;
    asm volatile("" : "+v"(le));
    const int st_lane = ((le & 16) ? 4 * n : 0) * 4 + (le & 15) * 4;
    const int st_base = (le < 32 && g * 16 + (le & 15) < n) ? st_lane : (int)0x80000000;
#define OP(S, B) (((S) & 1) ? ops.M[B][(S) >> 1] : ops.F[B][(S) >> 1])
#define RP4(t, i) { mx = fmaxf(fmaxf(mx, t[4 * (i)]), t[4 * (i) + 1]); mx = fmaxf(fmaxf(mx, t[4 * (i) + 2]), t[4 * (i) + 3]); }
#define ST(S) { ep_store(mx, rd, st_base + ((S) >> 1) * 4 * n + g * 64); mx = -__builtin_inff(); }
#define NX(K) FIRST_REQ(K, rsn)
#define NONE ((void)0)
// the + and - chains of a group alternate; the tiles of group S - 1 (the other set) are reduced behind the last five MFMAs of group S
#define S2G(S, W0, W1, W2, PREV)                                                                                      \
  { f32x16& _p = tp[(S) & 1]; f32x16& _m = tm[(S) & 1]; f32x16& _pp = tp[((S) & 1) ^ 1]; f32x16& _pm = tm[((S) & 1) ^ 1]; \
    SB(); M32Z(_p, cs.p[0], OP(S, 0)); SB(); W0; SB(); M32Z(_m, cs.m[0], OP(S, 0)); SB(); W1;                         \
    SB(); M32A(_p, cs.p[1], OP(S, 1)); SB(); W2;                                                                      \
    SB(); M32A(_m, cs.m[1], OP(S, 1)); SB(); if constexpr ((S) > 0) { RP4(_pp, 0); RP4(_pp, 1); }                     \
    SB(); M32A(_p, cs.p[2], OP(S, 2)); SB(); if constexpr ((S) > 0) { RP4(_pp, 2); RP4(_pp, 3); }                     \
    SB(); M32A(_m, cs.m[2], OP(S, 2)); SB(); if constexpr ((S) > 0) { RP4(_pm, 0); RP4(_pm, 1); }                     \
    SB(); M32A(_p, cs.p[3], OP(S, 3)); SB(); if constexpr ((S) > 0) { RP4(_pm, 2); RP4(_pm, 3); PREV; }               \
    SB(); M32A(_m, cs.m[3], OP(S, 3)); SB(); }
    S2G(0, NX(0), NX(1), NX(2), NONE)
    S2G(1, NX(3), NX(4), NX(5), NONE)
    S2G(2, NX(6), NX(7), NX(8), ST(1))
    S2G(3, NX(9), NX(10), NX(11), NONE)
    S2G(4, NX(12), NX(13), NX(14), ST(3))
    S2G(5, NX(15), NONE, NONE, NONE)
    S2G(6, NONE, NONE, NONE, ST(5))
    S2G(7, NONE, NONE, NONE, NONE)
    asm volatile("s_nop 15\n\ts_nop 15");          // the last tiles are read next: nothing pads an asm MFMA
    SB();
    RP4(tp[1], 0); RP4(tp[1], 1); RP4(tp[1], 2); RP4(tp[1], 3); RP4(tm[1], 0); RP4(tm[1], 1); RP4(tm[1], 2); RP4(tm[1], 3);
    ST(7)
    rs = rsn;
  }
}

}  // namespace

size_t sc_match_f_lds_bytes(int nqg) { return (size_t)nqg * SCF_QIMG + 64; }

void launch_sc_match_f(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p, float* d_i,
                       int nsplit_override) {
  if (m <= 0 || n <= 0) return;
  const int QG8 = sc_qgroups8_f16(m), DG = sc_dgroups(n);
  const int nqg = m <= 8 ? 1 : 8;
  const int QGW = m <= 8 ? 1 : QG8 / nqg;               // workgroups along the queries (8 nqg queries each)
  int nsplit = (128 + QGW - 1) / QGW;
  if (nsplit > DG / 32) nsplit = DG / 32;
  if (nsplit < 1) nsplit = 1;
  if (nqg == 1) nsplit = DG / 128 > 0 ? DG / 128 : 1;   // an online call: ~32 DB groups per workgroup, 4 per wave; ~400 workgroups at n = 100k
  if (nsplit_override > 0) nsplit = nsplit_override * 4 <= DG ? nsplit_override : (DG >= 4 ? DG / 4 : 1);
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc_match_f_lds_bytes(nqg));
    hipLaunchKernelGGL(kern, dim3(8 * QGW * nsplit), dim3(512), sc_match_f_lds_bytes(nqg), st, static_cast<const char*>(qpk),
                       static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i, m, n, QG8, DG, nsplit);
  };
  if (m <= 8) go(sc_match_f_kernel<1>);
  else go(sc_match_f_kernel<8>);
}

}  // namespace pr
