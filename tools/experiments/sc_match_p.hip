// sc_match_p.hip — EXPERIMENT, not the default (PR_SC_KERNEL=p selects it for m > 8): the split-f16 SC matcher
// (processSC.m:22-33, same mathematics and packed images as sc_match_h.hip) with TWO waves per SIMD.
//
// sc_match_h.hip keeps 256 stage-2 accumulators per wave, i.e. one wave per SIMD, whose in-order stream serialises stage 1,
// stage 2 and the epilogue (matrix pipe 37 % busy).  Here a (8 queries x 16 entries) unit belongs to a PAIR of waves:
//   * stage 1 is split by FREQUENCY: role 0 computes frequencies 0..15, role 1 frequencies 16..30 - nothing is computed
//     twice, every wave issues half of the unit's operand requests;
//   * stage 2 is split by stage-1 register r (= query pair): role 0 owns r = 0, 1, role 1 owns r = 2, 3 - 128 accumulators
//     each, in ArchVGPRs (no v_accvgpr_read in the epilogue), and each wave stores its own query rows;
//   * what a wave computes in stage 1 for the OTHER wave's registers goes through LDS, already split into hi / lo f16 pairs:
//     8 KB per quarter (8 frequencies) and direction, two barriers per quarter;
//   * stage 2 works by quarter with the hi and lo halves of the intermediate concatenated along K:
//       B = (S_hi | S_lo) by lane half - one v_permlane32_swap of a packed register's hi with its lo gives the Re and the
//       Im operand -, A = (C_hi | C_hi) for hi*hi + hi*lo and A = (C_lo | 0) for lo*hi: 8 MFMAs per register and quarter
//       (128 per unit instead of the 96 of sc_match_h.hip: the zero half of the lo*hi product is the price).
// A workgroup = 4 waves = 2 units = (1 query group) x (2 DB groups), 80 928 B of LDS (query image 39 952 + exchange 32 768 +
// constants 8 208): TWO workgroups per CU, so a SIMD hosts one wave of each and their barriers do not couple them.  Plain HIP
// with MFMA intrinsics: <= 256 registers per wave, all MFMAs in VGPR form, hipcc schedules and pads.
//
// Measured (MI355X, 4096 x 100k, tools/experiments/README.md): parity-green, 51.6 ms per launch against 46.1 ms of
// sc_match_h.hip.  Each SIMD has two waves resident for the whole launch; a wave is issuing 33 % of its cycles, waiting on
// counters 30 %, waiting to issue 36 %; the matrix pipe is 43 % busy with 17 % more matrix work than sc_match_h.hip.  The
// 256-register budget is the limit: accumulators 128 + own quarter 32 + operands of two frequencies 64 leave no room to request
// the DB operands more than one frequency ahead (a ring of three sets spills: 59 - 86 ms), and LDS has no room for an
// LDS-DMA operand ring next to the exchange buffers.
#include "kernels.hpp"

namespace pr {
namespace {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a8 __attribute__((aligned(8)));
typedef const u32x4_a8 __attribute__((address_space(3))) * lds_tile_p;
typedef u32x4 __attribute__((address_space(3))) * lds_v4_p;

constexpr int P_XQ = 8192;                          // exchange: one quarter, one direction: 4 pairs x 2 registers x 64 lanes x 16 B
constexpr int P_CST = 8192;                         // the stage-2 constants: [quarter][E | O][hi | lo][32 rows] x 16 B, + 16 B of zeros
// the query image + 24 zero bytes (the operand reads of lanes 48-63 = K 24..31 run up to 24 bytes past a frequency block: finite
// values there, the DB operand of those lanes is zero), rounded to 16: the exchange buffers are read and written 16 bytes at a time
constexpr int P_QIMG = (SCH_QIMG + 24 + 15) & ~15;
constexpr int P_LDS = P_QIMG + 2 * 2 * P_XQ + P_CST + 16;   // 39 952 + 32 768 + 8 208 = 80 928: two workgroups per CU

struct AOps { u32x4 h, l, rh, rl; };          // query row operands: Q hi, Q lo, and the same with Re / Im rows exchanged
struct BOps { u32x4 reh, rel, imh, iml; };    // DB column operands: Re hi, Re lo, Im hi, Im lo

__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// (hi, lo) split of two fp32 values into packed f16 pairs (as sc_match_h.hip)
__device__ __forceinline__ void split2(float x, float y, unsigned& hi, unsigned& lo) {
  const f32x2 v = {x, y};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
  f32x2 r;
  asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(r[0]), "=&v"(r[1])
      : "v"(hi), "v"(x), "v"(y));
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {   // lanes 32-63 of a <-> lanes 0-31 of b
  const u32x2 v = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = v[0];
  b = v[1];
}

// packed results of one quarter (4 frequency pairs) for ONE register: element j = pair j
struct QReg { u32x4 hF, lF, hM, lM; };
struct Cst { u32x4 ehh, elh, ohh, olh; };

// constants of a quarter from the LDS-resident table.  A = (C_hi | C_hi) along K for hi*hi + hi*lo: both lane halves read
// the same row; A = (C_lo | 0) for lo*hi: lanes 32-63 read the 16 zero bytes behind the table
__device__ __forceinline__ Cst load_cst(unsigned cst_lds, int qd, int lane) {
  const unsigned c = cst_lds + qd * 2048 + (lane & 31) * 16;
  const unsigned z = cst_lds + P_CST;
  Cst k;
  k.ehh = *reinterpret_cast<lds_v4_p>(c);
  k.elh = *reinterpret_cast<lds_v4_p>(lane < 32 ? c + 512 : z);
  k.ohh = *reinterpret_cast<lds_v4_p>(c + 1024);
  k.olh = *reinterpret_cast<lds_v4_p>(lane < 32 ? c + 1536 : z);
  return k;
}

// stage 2 of one quarter's data for one register of the wave
template <bool FIRST>   // FIRST: the unit's first contribution - the accumulators start from an inline zero, no register clearing
__device__ __forceinline__ void stage2(QReg q, const Cst& k, f32x16 (&accE)[2], f32x16 (&accO)[2]) {
#pragma unroll
  for (int j = 0; j < 4; j++) {      // (Re_hi | Im_hi), (Re_lo | Im_lo) -> (Re_hi | Re_lo), (Im_hi | Im_lo)
    unsigned a = q.hF[j], b = q.lF[j];
    swap32(a, b);
    q.hF[j] = a; q.lF[j] = b;
    a = q.hM[j]; b = q.lM[j];
    swap32(a, b);
    q.hM[j] = a; q.lM[j] = b;
  }
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; e++) z[e] = 0.f;
#ifdef P_PRIO
  __builtin_amdgcn_s_setprio(1);
#endif
  accE[0] = mfma32(k.ehh, q.hF, FIRST ? z : accE[0]);
  accO[0] = mfma32(k.ohh, q.lF, FIRST ? z : accO[0]);
  accE[1] = mfma32(k.ehh, q.hM, FIRST ? z : accE[1]);
  accO[1] = mfma32(k.ohh, q.lM, FIRST ? z : accO[1]);
  accE[0] = mfma32(k.elh, q.hF, accE[0]);
  accO[0] = mfma32(k.olh, q.lF, accO[0]);
  accE[1] = mfma32(k.elh, q.hM, accE[1]);
  accO[1] = mfma32(k.olh, q.lM, accO[1]);
#ifdef P_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

template <int ROLE>
__device__ __forceinline__ void unit_body(unsigned lds0, int lane, int u, const char* dgrp, const char* dnext,
                                          BOps (&B)[3], f32x16 (&accE)[2][2], f32x16 (&accO)[2][2]) {
  const int row = lane & 15, kg = lane >> 4;
  const unsigned nat = lds0 + row * 80 + (row >= 8 ? 8 : 0) + kg * 16;
  const unsigned rot = lds0 + (row ^ 8) * 80 + (row >= 8 ? 0 : 8) + kg * 16;
  const int voff = (lane < 48) ? lane * 16 : (int)0x80000000;     // lanes 48-63: out of range -> zeros (K = 24..31)
  const float sg = (lane < 32) ? 1.0f : -1.0f;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dgrp), 0, SCH_DIMG, 0x00020000);
  const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dnext), 0, SCH_DIMG, 0x00020000);
  const unsigned cst_lds = lds0 + P_QIMG + 2 * 2 * P_XQ;
  // exchange buffers of this unit: [direction = producing role][pair 0..3][register 0..1][lane] x 16 B {hF, lF, hM, lM}
  const unsigned xout = lds0 + P_QIMG + (u * 2 + ROLE) * P_XQ + lane * 16;
  const unsigned xin = lds0 + P_QIMG + (u * 2 + (1 - ROLE)) * P_XQ + lane * 16;
  constexpr int F0 = 16 * ROLE;                                   // this wave's frequencies F0 .. F0 + 15 (31 = padding)
  constexpr int OWN = 2 * ROLE, OTH = 2 - 2 * ROLE;               // registers owned / handed to the partner

  // DB operands come from L2 (latency ~700 cycles): requested TWO frequencies ahead into a ring of three register sets; the query
  // operands come from LDS (~100 cycles) and are read right before use - the partner wave's instructions fill that gap
  auto load_a = [&](int f, AOps& a) {
    const unsigned an = nat + f * SCH_QBLK, ar = rot + f * SCH_QBLK;
    a.h = *reinterpret_cast<lds_tile_p>(an);
    a.l = *reinterpret_cast<lds_tile_p>(an + 40);
    a.rh = *reinterpret_cast<lds_tile_p>(ar);
    a.rl = *reinterpret_cast<lds_tile_p>(ar + 40);
  };
  auto load_b = [&](const __amdgpu_buffer_rsrc_t& r, int f, BOps& b) {
    const int so = f * SCH_DFREQ;
    b.reh = __builtin_amdgcn_raw_buffer_load_b128(r, voff, so, 0);
    b.rel = __builtin_amdgcn_raw_buffer_load_b128(r, voff + SCH_DTILE, so, 0);
    b.imh = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 2 * SCH_DTILE, so, 0);
    b.iml = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 3 * SCH_DTILE, so, 0);
  };
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  auto stage1 = [&](const AOps& a, const BOps& b, f32x4& F, f32x4& M) {
    f32x4 t1 = mfma16(a.h, b.reh, z4);
    f32x4 t2 = mfma16(a.rh, b.imh, z4);
    t1 = mfma16(a.l, b.reh, t1);
    t2 = mfma16(a.rl, b.imh, t2);
    t1 = mfma16(a.h, b.rel, t1);
    t2 = mfma16(a.rh, b.iml, t2);
#pragma unroll
    for (int r = 0; r < 4; r++) { F[r] = __builtin_fmaf(t2[r], sg, t1[r]); M[r] = __builtin_fmaf(t2[r], -sg, t1[r]); }
  };

  // B[i % 3] holds the wave's i-th frequency; on entry B[0], B[1] hold the first two (requested by the previous unit / the prologue),
  // on exit they hold the next unit's
  constexpr int NFQ = (ROLE == 0) ? 16 : 15;          // real frequencies of this wave
#ifndef P_BD
#define P_BD 2
#endif
#ifndef P_AP
#define P_AP 1
#endif
  constexpr int BD = P_BD, AH = BD - 1;               // ring depth, frequencies requested ahead
  BOps nx[2];                                         // the next unit's first frequencies
  AOps A[2];
  if (P_AP) load_a(F0, A[0]);
  auto step = [&](int i, f32x4& F, f32x4& M) {
    if (i + AH < NFQ) load_b(rs, F0 + i + AH, B[(i + AH) % BD]); else if (i + AH - NFQ < AH) load_b(rn, F0 + i + AH - NFQ, nx[i + AH - NFQ]);
    if (P_AP) { if (i + 1 < NFQ) load_a(F0 + i + 1, A[(i + 1) & 1]); } else load_a(F0 + i, A[i & 1]);
#ifdef P_PIN
    __builtin_amdgcn_sched_barrier(0);
#endif
    stage1(A[i & 1], B[i % BD], F, M);
  };
#pragma unroll
  for (int qs = 0; qs < 2; qs++) {                    // the wave's two quarters
    QReg own[2];
#pragma unroll
    for (int j = 0; j < 4; j++) {                     // frequency pair (fa, fb) of the quarter
      const int ia = 8 * qs + 2 * j, ib = ia + 1;     // indices inside the wave's range
      f32x4 Fa, Ma, Fb, Mb;
      step(ia, Fa, Ma);
      if (ib < NFQ) step(ib, Fb, Mb); else { Fb = z4; Mb = z4; }
      if (j == 0) __syncthreads();                    // the partner has read the previous quarter: the exchange buffer is free
#pragma unroll
      for (int ri = 0; ri < 2; ri++) {
        unsigned h0, l0, h1, l1;
        split2(Fa[OWN + ri], Fb[OWN + ri], h0, l0);
        split2(Ma[OWN + ri], Mb[OWN + ri], h1, l1);
        own[ri].hF[j] = h0; own[ri].lF[j] = l0; own[ri].hM[j] = h1; own[ri].lM[j] = l1;
        split2(Fa[OTH + ri], Fb[OTH + ri], h0, l0);
        split2(Ma[OTH + ri], Mb[OTH + ri], h1, l1);
        const u32x4 o = {h0, l0, h1, l1};
        *reinterpret_cast<lds_v4_p>(xout + (j * 2 + ri) * 1024) = o;
      }
    }
#ifndef P_ORDER
#define P_ORDER 0
#endif
    if (P_ORDER == 0) {
      {   // the wave's own half needs no exchange: its matrix work runs while the partner finishes the quarter
        const Cst k = load_cst(cst_lds, 2 * ROLE + qs, lane);
        if (qs == 0) { stage2<true>(own[0], k, accE[0], accO[0]); stage2<true>(own[1], k, accE[1], accO[1]); }
        else { stage2<false>(own[0], k, accE[0], accO[0]); stage2<false>(own[1], k, accE[1], accO[1]); }
      }
      __syncthreads();                                  // both directions of the quarter are in LDS
      const Cst k = load_cst(cst_lds, 2 * (1 - ROLE) + qs, lane);
#pragma unroll
      for (int ri = 0; ri < 2; ri++) {
        QReg o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const u32x4 v = *reinterpret_cast<lds_v4_p>(xin + (j * 2 + ri) * 1024);
          o.hF[j] = v[0]; o.lF[j] = v[1]; o.hM[j] = v[2]; o.lM[j] = v[3];
        }
        stage2<false>(o, k, accE[ri], accO[ri]);
      }
    } else {   // the partner's half is requested first and arrives under the matrix work on the wave's own half
      __syncthreads();                                  // both directions of the quarter are in LDS
      const Cst ko = load_cst(cst_lds, 2 * ROLE + qs, lane);
      const Cst kp = load_cst(cst_lds, 2 * (1 - ROLE) + qs, lane);
#pragma unroll
      for (int ri = 0; ri < 2; ri++) {
        QReg o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const u32x4 v = *reinterpret_cast<lds_v4_p>(xin + (j * 2 + ri) * 1024);
          o.hF[j] = v[0]; o.lF[j] = v[1]; o.hM[j] = v[2]; o.lM[j] = v[3];
        }
        if (qs == 0) stage2<true>(own[ri], ko, accE[ri], accO[ri]); else stage2<false>(own[ri], ko, accE[ri], accO[ri]);
        stage2<false>(o, kp, accE[ri], accO[ri]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < AH; t++) B[t] = nx[t];
}

__global__ __launch_bounds__(256, 2) void sc_match_p_kernel(const char* __restrict__ qpk,   // [2][QG8][31][1288 B]
                                                            const char* __restrict__ dpk,   // [2][DG][31][4][768 B] + zero groups
                                                            const u32x4* __restrict__ cst,  // [4 quarters][4 tiles][64] x 16 B
                                                            float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                            int m, int n, int QG8, int DG, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u = w & 1, role = w >> 1, eg = u;
  // XCD-aware mapping as sc_match_h.hip: an XCD works on one channel and one quarter of the DB ranges
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ch = xcd & 1;
  const int range = (xcd >> 1) + 4 * (idx % nsplit), qg8 = idx / nsplit;
  const int nrange = 4 * nsplit;
  const int g0 = (int)((long long)DG * range / nrange), g1 = (int)((long long)DG * (range + 1) / nrange);
  {  // the query group of this workgroup -> LDS (the packed image IS the LDS image); the constants: rows 0-31 of the hh and lh tiles
    const u32x2* src = reinterpret_cast<const u32x2*>(qpk + ((size_t)ch * QG8 + (size_t)qg8) * SCH_QIMG);   // images are 8-byte multiples
    u32x2* dst = reinterpret_cast<u32x2*>(lds);
    constexpr int NV = SCH_QIMG / 8;
    for (int i = tid; i < NV; i += 256) dst[i] = src[i];
    if (tid < (P_QIMG - SCH_QIMG) / 8) dst[NV + tid] = u32x2{0u, 0u};
    u32x4* cd = reinterpret_cast<u32x4*>(lds + P_QIMG + 2 * 2 * P_XQ);
    for (int i = tid; i < P_CST / 16; i += 256) cd[i] = cst[(i >> 5) * 64 + (i & 31)];
    if (tid == 0) cd[P_CST / 16] = u32x4{0u, 0u, 0u, 0u};
  }
  __syncthreads();
  if (g0 >= g1) return;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
  float* dist = ch ? dist_i : dist_p;
  const char* dbase = dpk + ((size_t)ch * DG) * SCH_DIMG;
  const int qrow0 = qg8 * 8;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      dist + (size_t)qrow0 * n, 0, (qrow0 < m ? (m - qrow0 < 8 ? m - qrow0 : 8) : 0) * n * 4, 0x00020000);
  const int st_lane = ((lane & 16) ? 4 * n : 0) * 4 + (lane & 15) * 4;
  BOps B[3];
  {   // prologue: the first two frequencies of this wave's first unit
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)(g0 + eg) * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
    const int voff = (lane < 48) ? lane * 16 : (int)0x80000000;
#pragma unroll
    for (int t = 0; t < P_BD - 1; t++) {
      const int so = (16 * role + t) * SCH_DFREQ;
      B[t].reh = __builtin_amdgcn_raw_buffer_load_b128(r0, voff, so, 0);
      B[t].rel = __builtin_amdgcn_raw_buffer_load_b128(r0, voff + SCH_DTILE, so, 0);
      B[t].imh = __builtin_amdgcn_raw_buffer_load_b128(r0, voff + 2 * SCH_DTILE, so, 0);
      B[t].iml = __builtin_amdgcn_raw_buffer_load_b128(r0, voff + 3 * SCH_DTILE, so, 0);
    }
  }
  for (int gp = g0; gp < g1; gp += 2) {
    const int g = gp + eg;                           // this unit's DB group (past g1: computed on the zero / next groups, not stored)
    f32x16 accE[2][2], accO[2][2];                   // started by the first stage-2 products of the unit
    const char* dgrp = dbase + (size_t)g * SCH_DIMG;
    const char* dnext = dbase + (size_t)(g + 2) * SCH_DIMG;   // the next unit's group: its first two frequencies are requested by this unit
    if (role == 0) unit_body<0>(lds0, lane, u, dgrp, dnext, B, accE, accO);
    else unit_body<1>(lds0, lane, u, dgrp, dnext, B, accE, accO);
    // epilogue: shift rows e of the wave's registers -> max over E + |O| of forward and mirror  (processSC.m:30-31)
#pragma unroll
    for (int ri = 0; ri < 2; ri++) {
      const int R = 2 * role + ri;
      float mx = -__builtin_inff();
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const float vf = accE[ri][0][e] + __builtin_fabsf(accO[ri][0][e]);
        const float vm = accE[ri][1][e] + __builtin_fabsf(accO[ri][1][e]);
        mx = fmaxf(fmaxf(mx, vf), vm);
      }
      const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      const int st_base = (lane < 32 && g < g1 && g * 16 + (lane & 15) < n) ? st_lane : (int)0x80000000;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(mx, -0x1p-26f, 0.5f)), rd, st_base + R * 4 * n + g * 64, 0, 0);
    }
  }
}

}  // namespace

size_t sc_match_p_lds_bytes() { return (size_t)P_LDS; }

void launch_sc_match_p(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p, float* d_i,
                       int nsplit_override) {
  if (m <= 0 || n <= 0) return;
  const int QG8 = sc_qgroups8(m), DG = sc_dgroups(n);
  int nsplit = (128 + QG8 - 1) / QG8;               // >= 1024 workgroups: two resident per CU
  if (nsplit > DG / 32) nsplit = DG / 32;
  if (nsplit < 1) nsplit = 1;
  if (nsplit_override > 0) nsplit = nsplit_override * 4 <= DG ? nsplit_override : (DG >= 4 ? DG / 4 : 1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_match_p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
  hipLaunchKernelGGL(sc_match_p_kernel, dim3(8 * QG8 * nsplit), dim3(256), P_LDS, st, static_cast<const char*>(qpk),
                     static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i, m, n, QG8, DG, nsplit);
}

}  // namespace pr
