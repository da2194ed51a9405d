// sc_match_e.hip — the default SC matcher (processSC.m:22-33) in both f16 arithmetics: split-f16 (PR_SC_ARITH_F16X2: every fp32 factor as
// f16 hi + lo, three f16 products per product: in stage 1 as two MFMAs over 60 of 64 K-slots - the operand pairs of kernels.hpp -, in stage 2
// as three MFMAs) and single product (PR_SC_ARITH_F16: hi only).
//
// Same mathematics, packed split images and stage-2 constants as sc_match_h.hip / sc_match_d.hip (read those headers first).  What changes:
//
//  * no row-exchanged query operand.  sc_match_h computes T2 = [Qi;Qr].Di^T from the LDS image read a second time with row ^ 8, so that
//    F = T1 + s T2 is lane-aligned.  Here T2' = [Qr;Qi].Di^T comes from the SAME query tiles as T1 (two LDS reads per frequency
//    instead of four, 32 operand registers fewer) and the exchange of the Re and Im row halves is folded into the permlane swap that
//    the stage-2 operand layout needs anyway: the frequencies are walked as pairs (f, f + 8), and
//        swap32(T1_f, T1_f+8) = (QrDr_f | QrDr_f+8) =: X , (QiDr_f | QiDr_f+8) =: Y        swap32(T2'_f, T2'_f+8) = U (QrDi), V (QiDi)
//        Re S = X + V   Im S = Y - U   Re P = X - V   Im P = Y + U           each already (f | f + 8) by lane half
//    which is, after the split, exactly the operand layout sc_match_d reaches with 128 swaps of packed registers.
//  * the schedule is a function of (quad, slot) - valu_slot<> below - instead of hand-written macro rows: 4 or 8 VALU instructions behind every
//    stage-1 MFMA, the swaps / combination / split of a quad under the MFMAs of the next; DB tiles three walk positions ahead.
//  * split-f16 (LO = true): one wave per SIMD, 4 query groups per workgroup; the packed operands of the whole unit are 256 registers, so
//    the first half's are parked in AccVGPRs (park_half) and read from there by the stage-2 MFMAs.  ~3 % faster than sc_match_d; the
//    cost model behind that (tools/ubench/mfma16_fillers.hip and the ablation table of DESIGN.md): one in-order wave adds up MFMA time (4 x 17.8 cycles per walk
//    position; 6 x until round 5, when the three zero-padded split products became two full operand pairs), VALU beyond the two an MFMA hides (~4.4 cycles each; 8.5 for a permlane swap or an AccVGPR write), and ~22 cycles per
//    operand request, vector or LDS alike - nothing overlaps much, whatever the placement.
//  * single product (LO = false): the unit fits 256 registers, so the workgroup has EIGHT waves - two per SIMD, which do overlap - over
//    8 query groups of the compact image (SCF_*: 64 queries in 160 KB of LDS); for m <= 8 (an online call) all eight waves share ONE query
//    group and split the DB groups (NQG = 1: 20 KB of LDS, several workgroups per CU, 0.11 ms per 100k-entry DB = 5.3 TB/s).
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"
// Split-f16 form, register placement (round 3): one wave per SIMD owns 256 ArchVGPRs + 256 AccVGPRs, and everything an MFMA only READS can
// live in the AccVGPR half without ever passing through a VALU instruction: vector-memory and LDS loads write AccVGPRs directly and MFMAs
// take A / B operands from there.  So the DB operand ring (E_BACC), the query operand ring (E_AACC) and the stage-2 constants are loaded
// straight into AccVGPRs (hipcc does this by itself once every reader of the loaded value is an asm operand with an "a" constraint) - 136
// ArchVGPRs free, which (a) lets the rings run deeper (5 / 3 walk positions instead of 4 / 2), (b) keeps the first half's hi operand
// tuples and the lo tuples of register 0 in ArchVGPRs - 48 v_accvgpr_write per unit instead of 128 - and (c) leaves hipcc no reason to
// shuffle values through AccVGPRs on its own (it had parked 16 registers and restored them right in front of asm MFMAs).  Measured
// (alternating runs, three boxes): 40.3 -> 39.1, 41.2 -> 39.6, 41.5 -> 39.9 ms per 4096 x 100k launch (-3.9 %); -DE_NO_ACC_OPERANDS builds the
// previous placement.
#ifndef E_NO_WAIT1
#define E_WAIT1         // one s_waitcnt per walk position (TOUCH_OPS below): 166 -> 42 s_waitcnt per unit, -2.3 % per launch
#endif
#ifndef E_NO_ACC_OPERANDS
#define E_BACC
#define E_AACC
#define E_SPLIT3
#ifndef E_BD
#define E_BD 5
#endif
#ifndef E_AD
#define E_AD 3
#endif
#ifndef E_PARKR
#define E_PARKR 1
#endif
#endif
#ifndef E_BD
#define E_BD 4          // depth of the DB operand ring, split-f16 form
#endif
#ifndef E_BD1
#define E_BD1 8         // the same, single-product form (a position is only two MFMAs long); round 6: 6 -> 8, binary launch 9.06 -> 8.90 ms (-1.8 %, alternating
                        // runs); 252 of the 256 registers a wave has at two per SIMD - 9 spills
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

struct AOps { u32x4 h, l; };                  // query row operands [Qr;Qi]: hi, lo
struct BOps { u32x4 reh, rel, imh, iml; };    // DB column operands: Re hi, Re lo, Im hi, Im lo
enum { A_H = 0, A_L = 1 };
enum { B_REH = 0, B_REL = 1, B_IMH = 2, B_IML = 3 };

// walk order of the frequencies: position P -> f.  Pairs (f, f + 8) inside each half of 16; P = 31 is the ghost frequency 31
__host__ __device__ constexpr int seqf(int P) { return 16 * (P >> 4) + ((P & 15) >> 1) + 8 * (P & 1); }

typedef const u32x4_a8 __attribute__((address_space(3))) * lds_tile_p;
template <int T>
__device__ __forceinline__ void load_a(AOps& a, unsigned addr) {   // addr = this lane's 16 B of the frequency's rows: operand pair 1 (h) | pair 2 (l), kernels.hpp
  const u32x4 v = *reinterpret_cast<lds_tile_p>(addr);
  if (T == A_H) a.h = v; else a.l = v;
}
template <bool LO, int F, int T, bool SV = false>
__device__ __forceinline__ void load_b(BOps& b, __amdgpu_buffer_rsrc_t rs, int voff) {
  // split-f16 image: [f][Re hi | Re lo | Im hi | Im lo]; single-product image: [f][Re | Im]
  // split-f16 form: the frequency in the scalar offset (one SGPR per walk position, kept), the Re | Im tile pair added to the lane offset - 64
  // s_mov per unit less, -0.6 %; voff = the lane's place in the pair's hi + lo tiles for operand pair 1 (reh, imh) | pair 2 (rel, iml),
  // kernels.hpp: sch_b1_byte / sch_b2_byte.  The single-product form (registers are scarcer there) keeps everything in the scalar offset
  // SV: the single-product form reading the hi tiles of a split-f16 image
  constexpr int soff = LO ? F * SCH_DFREQ : SV ? F * SCH_DFREQ + (T == B_IMH ? 2 : 0) * SCH_DTILE : F * SCF_DFREQ + (T == B_IMH ? 1 : 0) * SCH_DTILE,
                ioff = LO ? ((T == B_IMH || T == B_IML) ? 2 * SCH_DTILE : 0) : 0;
#ifdef E_ABL_ONE_B        // ablation (wrong results): the single-product form without its Im tile requests
  if (!LO && T == B_IMH) { b.imh = b.reh; return; }
#endif
#ifdef E_ABL_B_HOT        // ablation (wrong results): every DB tile request reads the first tile pair of the group - the same requests, all of them L1 hits
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + ioff, 0, 0);
#else
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + ioff, soff, 0);
#endif
  if (T == B_REH) b.reh = v; else if (T == B_REL) b.rel = v; else if (T == B_IMH) b.imh = v; else b.iml = v;
}

// MFMAs as asm statements (VGPR form; hipcc pads nothing around asm, cdna_hip_programming.md §5.7): every VALU reader of a result sits
// behind a DRAIN() or at least two further MFMAs + their fillers
#define MF0_(d, a, b, AC, BC) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(d) : AC(a), BC(b))
#define MFA_(d, a, b, AC, BC) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : AC(a), BC(b))
#ifdef E_BACC      // split-f16 form: DB operands loaded straight into AccVGPRs (one wave per SIMD: the AccVGPR half is otherwise idle)
#ifdef E_AACC      // ... and the query operands
#define E_AC "a"
#else
#define E_AC "v"
#endif
#define MF0(d, a, b) { if constexpr (LO) MF0_(d, a, b, E_AC, "a"); else MF0_(d, a, b, "v", "v"); }
#define MFA(d, a, b) { if constexpr (LO) MFA_(d, a, b, E_AC, "a"); else MFA_(d, a, b, "v", "v"); }
#else
#define MF0(d, a, b) MF0_(d, a, b, "v", "v")
#define MFA(d, a, b) MFA_(d, a, b, "v", "v")
#endif
#ifndef E_AD
#define E_AD 2
#endif
#ifndef E_PARKR
#define E_PARKR 0
#endif
#ifdef E_BACC      // (split-f16 form) the stage-2 constants are loaded into AccVGPRs as well
#define E_CC "a"
#else
#define E_CC "v"
#endif
#define M32Z_(d, a, b, AC, BC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : AC(a), BC(b))
#define M32A_(d, a, b, AC, BC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : AC(a), BC(b))
#define M32Z(d, a, b, BC) { if constexpr (LO) M32Z_(d, a, b, E_CC, BC); else M32Z_(d, a, b, "v", BC); }
#define M32A(d, a, b, BC) { if constexpr (LO) M32A_(d, a, b, E_CC, BC); else M32A_(d, a, b, "v", BC); }
#define DRAIN() asm volatile("s_nop 9")
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void split2(float x, float y, unsigned& hi, unsigned& lo) {   // see sc_match_h.hip
  const f32x2 v = {x, y};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
#ifdef E_SPLIT3   // the conversion of the residuals inside the asm: hipcc pads an s_nop between an asm result and its first reader (110 per unit)
  float r0, r1;
  asm volatile("v_fma_mix_f32 %1, %3, -1.0, %4 op_sel_hi:[1,0,0]\n\t"      // volatile: stays where the schedule puts it (no PIN(lo) behind it)
      "v_fma_mix_f32 %2, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_cvt_pk_f16_f32 %0, %1, %2"
      : "=v"(lo), "=&v"(r0), "=&v"(r1)
      : "v"(hi), "v"(x), "v"(y));
#else
  f32x2 r;
  asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(r[0]), "=&v"(r[1])
      : "v"(hi), "v"(x), "v"(y));
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
#endif
}
__device__ __forceinline__ unsigned pack2(float x, float y) {
  const f32x2 v = {x, y};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

// Packed results of 16 frequencies (one half) for the 4 stage-1 registers r: element e of [kind][r] = frequencies (2e, 2e+1) in lanes
// 0-31 and (2e+8, 2e+9) in lanes 32-63: the B operand of a 32x32x16 MFMA over the half's 16 frequencies
template <bool LO>
struct Half {
  u32x4 reFh[4], imFh[4], reMh[4], imMh[4];
  u32x4 reFl[LO ? 4 : 1], imFl[LO ? 4 : 1], reMl[LO ? 4 : 1], imMl[LO ? 4 : 1];
};

__device__ __forceinline__ void swap32f(f32x4& a, f32x4& b, int e) {   // lanes 32-63 of a[e] <-> lanes 0-31 of b[e]
#ifdef E_ABL_NO_SWAP      // ablation (wrong results): what the stage-1 swaps cost
  return;
#endif
  const u32x2 v = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[e]), __float_as_uint(b[e]), false, false);
  a[e] = __uint_as_float(v[0]);
  b[e] = __uint_as_float(v[1]);
}

struct Consts { u32x4 ch, cl, sh, sl; };   // A operands of one half: cos hi/lo, -sin hi/lo
template <int HALF, bool LO>
__device__ __forceinline__ void load_consts(Consts& c, __amdgpu_buffer_rsrc_t rc, int lane16) {   // [E|O][half][hi|lo][64] x 16 B
  c.ch = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, ((0 * 2 + HALF) * 2 + 0) * 1024, 0);
  c.sh = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, ((1 * 2 + HALF) * 2 + 0) * 1024, 0);
  if constexpr (LO) {
    c.cl = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, ((0 * 2 + HALF) * 2 + 1) * 1024, 0);
    c.sl = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, ((1 * 2 + HALF) * 2 + 1) * 1024, 0);
  }
}

// MFMA I of group (register R, forward | mirror V).  LO: I = 0..11, E tile (even I) from the Re operands and O tile (odd I) from the Im
// operands, I >> 1 = 0..2 the three split products of the first half, 3..5 those of the second half.  !LO: I = 0..3, one product per half.
template <bool LO, bool PARK, int R, int V, int I>
__device__ __forceinline__ void s2_one(const Half<LO>& h0, const Half<LO>& h1, const Consts& c0, const Consts& c1, f32x16& e, f32x16& o) {
  constexpr int PART = I & 1, T = LO ? (I >> 1) % 3 : 0, HF = LO ? (I >> 1) / 3 : (I >> 1);
  const Half<LO>& h = HF ? h1 : h0;
  const Consts& c = HF ? c1 : c0;
  const u32x4& ca = PART ? (T == 1 ? c.sl : c.sh) : (T == 1 ? c.cl : c.ch);
  const u32x4& op = PART ? (V ? (T == 2 ? h.imMl[LO ? R : 0] : h.imMh[R]) : (T == 2 ? h.imFl[LO ? R : 0] : h.imFh[R]))
                         : (V ? (T == 2 ? h.reMl[LO ? R : 0] : h.reMh[R]) : (T == 2 ? h.reFl[LO ? R : 0] : h.reFh[R]));
  f32x16& d = PART ? o : e;
  // PARK: the first half's operands live in AccVGPRs (parked there by park_half, read by the MFMA directly)
#ifdef E_BACC
  constexpr bool in_acc = PARK && HF == 0 && T == 2 && R >= E_PARKR;   // only the lo tuples of the first half (registers E_PARKR..3) are parked
#else
  constexpr bool in_acc = PARK && HF == 0;
#endif
  if constexpr (in_acc) { if constexpr (I < 2) { M32Z(d, ca, op, "a"); } else { M32A(d, ca, op, "a"); } }
  else { if constexpr (I < 2) { M32Z(d, ca, op, "v"); } else { M32A(d, ca, op, "v"); } }
}
// parks the operand tuples of a finished half in AccVGPRs (an empty asm whose operand must be an AccVGPR tuple: one v_accvgpr_write per
// register, placed by hipcc right here)
template <bool LO>
__device__ __forceinline__ void park_half(Half<LO>& h) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
#ifndef E_BACC
    asm volatile("" : "+a"(h.reFh[r]), "+a"(h.imFh[r]), "+a"(h.reMh[r]), "+a"(h.imMh[r]));
    if constexpr (LO) asm volatile("" : "+a"(h.reFl[r]), "+a"(h.imFl[r]), "+a"(h.reMl[r]), "+a"(h.imMl[r]));
#else
    if (LO && r >= E_PARKR) asm volatile("" : "+a"(h.reFl[r]), "+a"(h.imFl[r]), "+a"(h.reMl[r]), "+a"(h.imMl[r]));
#endif
  }
}
// one step of the reduction of a finished (E, O) tile pair: shift rows 2 i, 2 i + 1
__device__ __forceinline__ float red_piece(float mx, const f32x16& e, const f32x16& o, int i) {
  return fmaxf(fmaxf(mx, e[2 * i] + __builtin_fabsf(o[2 * i])), e[2 * i + 1] + __builtin_fabsf(o[2 * i + 1]));
}
// 2 queries x 16 entries (lanes 0..31): d = (1 - max)/2 with the 2^-25 operand scaling folded in (processSC.m:30); branch-free
__device__ __forceinline__ void ep_store(float mx, __amdgpu_buffer_rsrc_t rd, int st_off) {
  const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
  mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));     // max over the two lane halves (shift rows +0..3 | +4..7)
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(mx, -0x1p-26f, 0.5f)), rd, st_off, 0, 0);
}
// the same for a binary channel (kernels.hpp: ScBin): x = max x 2^-25 x sqrt(ones_q ones_d) is the integer count of the best variant up to the
// pass's error, which is at most b = bnd x sqrt(ones_q ones_d) for EVERY variant of the pair (sc_bin_bound).  If |x - rint(x)| + b < 1, then
// rint(x) is the largest count: the best variant's own count is the only integer within b of x, and no other variant's count (at most its
// value + b <= x + b < rint(x) + 1) exceeds it.  d = (1 - count / sqrt(ones_q ones_d)) / 2 from that integer (processSC.m:30); a pair that
// fails the test raises `viol` and the whole channel is redone by the split-f16 kernel behind this one.
// qi = {sqrt(ones_q) 2^-25, 1/sqrt(ones_q)} of this lane's query, di = {sqrt(ones_d), 1/sqrt(ones_d)} of its entry, bnd = bound x 2^25
__device__ __forceinline__ void ep_store_round(float mx, f32x2 qi, f32x2 di, float bnd, int& viol, __amdgpu_buffer_rsrc_t rd, int st_off) {
  const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
  mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  const float nn = qi[0] * di[0], x = mx * nn, cnt = __builtin_rintf(x);
  viol |= (st_off >= 0 && !(__builtin_fmaf(bnd, nn, __builtin_fabsf(x - cnt)) < 0.98f)) ? 1 : 0;      // (lanes without a store: st_off < 0)
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(cnt * (qi[1] * di[1]), -0.5f, 0.5f)), rd, st_off, 0, 0);
}
// ---------------------------------------------------------------------------------------------------------------- stage-1 schedule
// The 32 walk positions of a unit form 8 quads (4 per half); quad QD computes T1 / T2' of frequencies (2E, 2E+8, 2E+1, 2E+9) of its half
// into register set QD & 1:  T[set][0..7] = t1a, t2a, t1b, t2b (first pair), t1c, t2c, t1d, t2d (second pair).  A position has 6 slots of
// VALU work, a quad 24 (split-f16: behind its 4 MFMAs as 1, 2, 1, 2 slots; one product per term: the first two behind its 2 MFMAs, the last four
// follow each other directly).  4 instructions per slot:
//   slot 21..23 of quad QD and slot 0 of quad QD+1: swap + combination of quad QD's FIRST pair  (in place: t1a = Re S, t1b = Im S, t2b = Re P, t2a = Im P)
//   slot 1..4 of quad QD+1:                         swap + combination of quad QD's SECOND pair (t1c, t1d, t2d, t2c)
//   slot 5..20 of quad QD+1:                        the 16 splits of quad QD -> element E of the half's 16 (32) operand tuples
// so every reader of an MFMA result sits two or more MFMAs + eight VALU behind it, and set QD & 1 is free again when quad QD+2 starts.
template <bool LO>
__device__ __forceinline__ void swp2(f32x4& x, f32x4& y, f32x4& u, f32x4& v, int e) { swap32f(x, y, e); swap32f(u, v, e); }
// X, Y, U, V -> Re S = X + V (in x), Re P = X - V (in v), Im S = Y - U (in y), Im P = Y + U (in u), registers r0, r0 + 1
__device__ __forceinline__ void cmb2(f32x4& x, f32x4& y, f32x4& u, f32x4& v, int r0) {
#ifdef E_ABL_NO_CMB       // ablation (wrong results): what the combination costs
  return;
#endif
#ifdef E_PLAIN_CMB   // A/B: eight plain adds (asm: hipcc would pack them again) instead of four v_pk_add_f32
#pragma unroll
  for (int e = r0; e < r0 + 2; e++) {
    float sr, pr, si, pi;
    asm volatile("v_add_f32 %0, %4, %7\n\tv_sub_f32 %1, %4, %7\n\tv_sub_f32 %2, %5, %6\n\tv_add_f32 %3, %5, %6"
                 : "=&v"(sr), "=&v"(pr), "=&v"(si), "=&v"(pi) : "v"(x[e]), "v"(y[e]), "v"(u[e]), "v"(v[e]));
    x[e] = sr; v[e] = pr; y[e] = si; u[e] = pi;
  }
  return;
#endif
  const f32x2 _x = {x[r0], x[r0 + 1]}, _y = {y[r0], y[r0 + 1]}, _u = {u[r0], u[r0 + 1]}, _v = {v[r0], v[r0 + 1]};
  const f32x2 _sr = _x + _v, _pr = _x - _v, _si = _y - _u, _pi = _y + _u;
  x[r0] = _sr[0]; x[r0 + 1] = _sr[1]; v[r0] = _pr[0]; v[r0 + 1] = _pr[1];
  y[r0] = _si[0]; y[r0 + 1] = _si[1]; u[r0] = _pi[0]; u[r0 + 1] = _pi[1];
}
// (the empty volatile asm pins the packed value HERE: hipcc otherwise sinks the whole pure combine / convert chain down to its stage-2
// consumer and keeps the fp32 values live instead - twice the registers)
#define PIN(x) asm volatile("" : "+v"(x))
template <bool LO, int E, int R, int KIND>
__device__ __forceinline__ void split_piece(Half<LO>& hb, const f32x4 (&t)[8]) {
  constexpr int ia = KIND == 0 ? 0 : KIND == 1 ? 2 : KIND == 2 ? 3 : 1;     // Re S: t1a|t1c, Im S: t1b|t1d, Re P: t2b|t2d, Im P: t2a|t2c
  unsigned h, l = 0;
#ifdef E_SPLIT3
  if constexpr (LO) { split2(t[ia][R], t[ia + 4][R], h, l); }
#else
  if constexpr (LO) { split2(t[ia][R], t[ia + 4][R], h, l); PIN(h); PIN(l); }
#endif
  else { h = pack2(t[ia][R], t[ia + 4][R]); PIN(h); }
  if constexpr (KIND == 0) { hb.reFh[R][E] = h; if constexpr (LO) hb.reFl[R][E] = l; }
  if constexpr (KIND == 1) { hb.imFh[R][E] = h; if constexpr (LO) hb.imFl[R][E] = l; }
  if constexpr (KIND == 2) { hb.reMh[R][E] = h; if constexpr (LO) hb.reMl[R][E] = l; }
  if constexpr (KIND == 3) { hb.imMh[R][E] = h; if constexpr (LO) hb.imMl[R][E] = l; }
}
// the VALU work of slot G (0..23) of quad QD (0..8; 8 = the drain behind the last quad)
template <bool LO, int QD, int G>
__device__ __forceinline__ void valu_slot(f32x4 (&T)[2][8], Half<LO> (&hbs)[2]) {
  constexpr int cur = QD & 1, prv = cur ^ 1;
  if constexpr (QD >= 1) {            // quad QD - 1, in set prv
    constexpr int PH = (QD - 1) >> 2, PE = (QD - 1) & 3;
    f32x4 (&t)[8] = T[prv];
    if constexpr (G == 0) cmb2(t[0], t[2], t[1], t[3], 2);
    if constexpr (G == 1) { swp2<LO>(t[4], t[6], t[5], t[7], 0); swp2<LO>(t[4], t[6], t[5], t[7], 1); }
    if constexpr (G == 2) cmb2(t[4], t[6], t[5], t[7], 0);
    if constexpr (G == 3) { swp2<LO>(t[4], t[6], t[5], t[7], 2); swp2<LO>(t[4], t[6], t[5], t[7], 3); }
    if constexpr (G == 4) cmb2(t[4], t[6], t[5], t[7], 2);
    if constexpr (G >= 5 && G <= 20) split_piece<LO, PE, ((G - 5) >> 2), ((G - 5) & 3)>(hbs[PH], t);
  }
  if constexpr (QD <= 7) {            // this quad's first pair
    f32x4 (&t)[8] = T[cur];
    if constexpr (G == 21) { swp2<LO>(t[0], t[2], t[1], t[3], 0); swp2<LO>(t[0], t[2], t[1], t[3], 1); }
    if constexpr (G == 22) cmb2(t[0], t[2], t[1], t[3], 0);
    if constexpr (G == 23) { swp2<LO>(t[0], t[2], t[1], t[3], 2); swp2<LO>(t[0], t[2], t[1], t[3], 3); }
  }
}

// NQG = query groups (of 8) a workgroup holds in LDS: 4 (split-f16), 8 (single product), or 1 (single product, m <= 8: an online call - all
// eight waves share the one group and split the DB groups, 20 KB of LDS, several workgroups per CU)
// SV (single-product form only): the kernel reads the hi halves of SPLIT-f16 images (the query groups are gathered into the compact LDS
// image, the DB loads address the hi tiles) and rounds the result to the integer count of a binary channel (ep_store_round)
template <bool LO, int NW, int NQG, bool SV = false>
__global__ __launch_bounds__(64 * NW, NW / 4) void sc_match_e_kernel(const char* __restrict__ qpk,   // [2][QG32][4][31][1288 B]
                                                            const char* __restrict__ dpk,   // [2][DG][31][4][768 B] + zero groups
                                                            const u32x4* __restrict__ cst,  // [2][2][2][64] x 16 B
                                                            float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                            int m, int n, int QG8, int DG, int nsplit, ScBin bin,
                                                            int DGS /* channel stride of the DB image in groups: DG, or the capacity's of an appendable set */) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  static_assert(!SV || !LO, "SV: single-product form");
  float bbound = 0.f;
  // the channel-0 launch of a binary-channel call clears the violation word the two channel-1 launches behind it speak through: `gen` is a
  // kernel argument, i.e. frozen inside a captured hipGraph - without this a replay that once failed a pair would keep finding its own
  // value there and redo channel 1 in split-f16 for ever
  if (bin.gate == 0 && bin.chsel == 0 && bin.viol && blockIdx.x == 0 && threadIdx.x == 0) *bin.viol = 0;
  if (bin.gate) {      // (workgroup-uniform: scalar loads of six numbers per set)
    const bool pred = sc_bin_bound(bin, bbound);
    if (bin.gate == 1 ? !pred : (pred && *bin.viol != bin.gen)) return;      // gate 2: the split-f16 kernel behind the single-product one
  }
  [[maybe_unused]] const float bnd25 = bbound * 0x1p25f * bin.pair_scale;
  [[maybe_unused]] int viol = 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware mapping as in sc_match_d.hip: all workgroups of an XCD work on ONE channel and the same quarter of the DB ranges;
  // chsel >= 0: one channel on all eight XCDs, an eighth of the ranges each
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ch = bin.chsel < 0 ? (xcd & 1) : bin.chsel;
  const int range = (bin.chsel < 0 ? (xcd >> 1) + 4 * (idx % nsplit) : xcd + 8 * (idx % nsplit)), qg32 = idx / nsplit;
  const int nrange = (bin.chsel < 0 ? 4 : 8) * nsplit;
  const int g0 = (int)((long long)DG * range / nrange), g1 = (int)((long long)DG * (range + 1) / nrange);

  // image geometry: split-f16 (4 query groups per workgroup) | single product (hi halves only: 8 query groups, kernels.hpp SCF_*)
  constexpr int QBLK = LO ? SCH_QBLK : SCF_QBLK, QIMG = LO ? SCH_QIMG : SCF_QIMG, DIMG = (LO || SV) ? SCH_DIMG : SCF_DIMG;
  constexpr int QROW = LO ? 80 : 40;
  static_assert(NW == 4 || !LO, "two waves per SIMD: single-product form only (a split-f16 unit needs all 512 registers)");
  if constexpr (SV) {  // the hi halves of this workgroup's query groups of the split image -> the compact LDS image, 8 bytes at a time:
    // compact block (group, f) = 80 words: 16 rows of 5 words; split block = 1288 B, row = 80 B (hi | lo), rows 8..15 shifted by 8 B
    constexpr int WPG = SC_NF * 80;
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(lds);
    for (int i = tid; i < NQG * WPG + 8; i += 64 * NW) {
      const int grp = i / WPG, rem = i - grp * WPG, f = rem / 80, j = rem - f * 80;
      const int gq = qg32 * NQG + grp;
      const int rr = j / 5, c = j - rr * 5;
      unsigned long long v = 0ull;
      if (i < NQG * WPG && gq < QG8)
        v = *reinterpret_cast<const unsigned long long*>(qpk + ((size_t)ch * QG8 + gq) * SCH_QIMG + (size_t)f * SCH_QBLK + rr * 80 + (rr >= 8 ? 8 : 0) + c * 8);
      dst[i] = v;
    }
    // ... and behind the image {sqrt(ones) 2^-25, 1/sqrt(ones)} of the workgroup's 8 NQG queries (ep_store_round)
    float* qi = reinterpret_cast<float*>(lds + NQG * QIMG + 64);
    for (int i = tid; i < 8 * NQG; i += 64 * NW) {
      const int r = qg32 * (8 * NQG) + i;
      qi[2 * i] = r < m ? bin.qinfo[2 * r] * 0x1p-25f : 0.f;
      qi[2 * i + 1] = r < m ? bin.qinfo[2 * r + 1] : 0.f;
    }
  } else {  // the query groups of this workgroup -> LDS (linear copy; the packed image IS the LDS image) + zeroed tail
    const u32x4* src = reinterpret_cast<const u32x4*>(qpk + ((size_t)ch * QG8 + (size_t)qg32 * NQG) * QIMG);
    u32x4* dst = reinterpret_cast<u32x4*>(lds);
    constexpr int NV = NQG * QIMG / 16;
    for (int i = tid; i < NV + 4; i += 64 * NW) dst[i] = (i < NV) ? src[i] : u32x4{0u, 0u, 0u, 0u};
  }
  __syncthreads();

  const int row = lane & 15, kg = lane >> 4;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
  // query group inside the image; with more waves than query groups, wave w takes the DB groups g0 + w / NQG, + GSTEP, ...
  const int wq = w & (NQG - 1), dpar = (NW > NQG) ? (w / NQG) : 0;
  constexpr int GSTEP = NW / NQG;
  const int gcnt = g1 - g0;
  constexpr int BD = LO ? E_BD : E_BD1;            // depth of the DB operand ring: the tiles of BD - 1 walk positions are in flight
  constexpr int AD = LO ? E_AD : 3;                // the same for the query tiles
  // split-f16 form: the two operand pairs of a frequency (kernels.hpp) - this lane's 16 bytes of its query row, and of the DB tiles below
  const unsigned natr = lds0 + wq * QIMG + row * QROW + ((LO && row >= 8) ? 8 : 0);
  const unsigned nat0 = natr + (LO ? sch_a1_byte(kg) : kg * 16);
  [[maybe_unused]] const unsigned nat1 = natr + sch_a2_byte(kg);
  const int voff = LO ? sch_b1_byte(lane) : ((lane < 48) ? lane * 16 : (int)0x80000000);     // single product, lanes 48-63: out of range -> zeros (K = 24..31)
  [[maybe_unused]] const int voff2 = sch_b2_byte(lane);
  float* dist = ch ? dist_i : dist_p;
  const char* dbase = dpk + ((size_t)ch * DGS) * DIMG;
  const int qrow0 = qg32 * (8 * NQG) + wq * 8;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      dist + (size_t)qrow0 * n, 0, (qrow0 < m ? (m - qrow0 < 8 ? m - qrow0 : 8) : 0) * n * 4, 0x00020000);
  // this wave's share of the group(s) it helps to prefetch: the 32 workgroups resident on an XCD are consecutive idx, i.e. rsd = 32 / nsplit
  // consecutive query blocks per range.  Split-f16 form (nsplit = 1 on the metric workload): 32 NW waves x 6 lines >= the 744 lines of a
  // group, as tuned.  Single-product form with several query groups per workgroup: rsd NW waves share the GSTEP x 372 lines (hi tiles only
  // in the SV view) - with the fixed two lines per wave of round 3, nsplit = 2 left half of every group to demand misses.
  constexpr int GLINES = SC_NF * 12;                                         // 128-byte lines of the Re / Im (hi) tiles of one group
  const int rsd = (LO || NQG == 1) ? 32 : (nsplit >= 32 ? 1 : 32 / nsplit);
  const int pf_slot = (LO || NQG == 1) ? (qg32 & 31) * NW + w : (qg32 % rsd) * NW + w;
  constexpr int PFL = LO ? 6 : 2;
  const int pfl = (LO || NQG == 1) ? PFL : ((NW / NQG) * GLINES + rsd * NW - 1) / (rsd * NW);
  unsigned pf_sink = 0;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(cst), 0, 8192, 0x00020000);
  if (dpar >= gcnt) return;
  f32x2 qinf[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};      // SV: of this lane's query of store R: row R (lanes 0-15) | 4 + R (16-31)
  if constexpr (SV) {
    const f32x2* qi = reinterpret_cast<const f32x2*>(lds + NQG * QIMG + 64);
#pragma unroll
    for (int r = 0; r < 4; r++) qinf[r] = qi[wq * 8 + r + ((lane & 16) ? 4 : 0)];
  }

  AOps At[AD];
  BOps Bt[BD];
  __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)(g0 + dpar) * DIMG), 0, DIMG, 0x00020000);
// request K of the first ones of a unit (issued before the loop for the first unit, in the stage-2 gaps of the previous unit otherwise):
// the DB tiles of walk positions 0 .. BD - 2 (TPB tiles each), then the query tiles of positions 0 .. AD - 2
  constexpr int TPB = LO ? 4 : 2, TPA = LO ? 2 : 1, NREQ = (BD - 1) * TPB + (AD - 1) * TPA;
#define FIRST_REQ(K, RSRC)                                                                                         \
  { if constexpr ((K) < (BD - 1) * TPB) {                                                                          \
      constexpr int _p = (K) / TPB, _t = LO ? (K) % TPB : 2 * ((K) % TPB);      /* order Re hi, (Re lo), Im hi, (Im lo) */ \
      constexpr int _tt = LO ? (_t == 1 ? B_IMH : _t == 2 ? B_REL : _t == 3 ? B_IML : B_REH) : _t;                 \
      load_b<LO, seqf(_p), _tt, SV>(Bt[_p % BD], RSRC, (LO && (_tt == B_REL || _tt == B_IML)) ? voff2 : voff);        \
    } else if constexpr ((K) < NREQ) {                                                                             \
      constexpr int _k = (K) - (BD - 1) * TPB, _p = _k / TPA, _t = _k % TPA;                                       \
      load_a<_t>(At[_p % AD], (_t == A_L ? nat1 : nat0) + seqf(_p) * QBLK);                                        \
    } }
  FIRST_REQ(0, rs) FIRST_REQ(1, rs) FIRST_REQ(2, rs) FIRST_REQ(3, rs) FIRST_REQ(4, rs) FIRST_REQ(5, rs) FIRST_REQ(6, rs) FIRST_REQ(7, rs)
  FIRST_REQ(8, rs) FIRST_REQ(9, rs) FIRST_REQ(10, rs) FIRST_REQ(11, rs) FIRST_REQ(12, rs) FIRST_REQ(13, rs) FIRST_REQ(14, rs) FIRST_REQ(15, rs)
  FIRST_REQ(16, rs) FIRST_REQ(17, rs) FIRST_REQ(18, rs) FIRST_REQ(19, rs) FIRST_REQ(20, rs) FIRST_REQ(21, rs) FIRST_REQ(22, rs) FIRST_REQ(23, rs)
  static_assert(NREQ <= 24, "FIRST_REQ list too short");

  for (int g = g0 + dpar; g < g1; g += GSTEP) {
    const int gn = g + GSTEP;                       // (the image ends with zero groups: the requests past the last group are harmless)
    const __amdgpu_buffer_rsrc_t rsn =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)gn * DIMG), 0, DIMG, 0x00020000);
    Half<LO> hbs[2];
    Consts c0, c1;
    f32x4 T[2][8];

// request tile T of walk position Q of this unit (Q >= 31: nothing)
#define LDB(Q, TT) { if constexpr ((Q) < 32 && seqf((Q) < 32 ? (Q) : 0) < SC_NF && (LO || (TT == B_REH || TT == B_IMH))) load_b<LO, seqf((Q) < 32 ? (Q) : 0), TT, SV>(Bt[(Q) % BD], rs, (LO && (TT == B_REL || TT == B_IML)) ? voff2 : voff); }
#define LDA(Q, TT) { if constexpr ((Q) < 32 && seqf((Q) < 32 ? (Q) : 0) < SC_NF && (LO || TT == A_H)) load_a<TT>(At[(Q) % AD], (TT == A_L ? nat1 : nat0) + seqf((Q) < 32 ? (Q) : 0) * QBLK); }
#define VS(P, G) valu_slot<LO, ((P) >> 2), (((P) & 3) * 6 + (G))>(T, hbs)
// one walk position: its 6 (LO) or 2 MFMAs, the requests for positions P + AD - 1 (query tiles) and P + BD - 1 (DB tiles), the quad's VALU
// work of its six slots
// E_WAIT1: an empty asm that names all six operand tiles of the position in front of its first MFMA - hipcc then waits for them ONCE
// (the rings are several positions deep: they have long arrived) instead of in front of every MFMA
#ifdef E_WAIT1
#ifdef E_WAIT1F
#define TOUCH_OPS_F(P) asm volatile("" : : "v"(At[(P) % AD].h), "v"(Bt[(P) % BD].reh), "v"(Bt[(P) % BD].imh))
#else
#define TOUCH_OPS_F(P)
#endif
#define TOUCH_OPS(P) { if constexpr (LO) asm volatile("" : : "a"(At[(P) % AD].h), "a"(At[(P) % AD].l), "a"(Bt[(P) % BD].reh), "a"(Bt[(P) % BD].imh), "a"(Bt[(P) % BD].rel), "a"(Bt[(P) % BD].iml)); else TOUCH_OPS_F(P); }
#else
#define TOUCH_OPS(P)
#endif
// (split-f16 form) what goes behind the four MFMAs of a position: the six requests and the six VALU slots
#ifndef E_LDD
#define E_LDD 0
#endif
#ifndef E_VSD
#define E_VSD 0
#endif
#if E_LDD == 0
#define G0_REQ(P) LDA((P) + AD - 1, A_H)
#define G1_REQ(P) LDA((P) + AD - 1, A_L)
#define G2_REQ(P) { LDB((P) + BD - 1, B_REH); LDB((P) + BD - 1, B_IMH); }
#define G3_REQ(P) { LDB((P) + BD - 1, B_REL); LDB((P) + BD - 1, B_IML); }
#else
#define G0_REQ(P) { LDA((P) + AD - 1, A_H); LDB((P) + BD - 1, B_REH); }
#define G1_REQ(P) { LDA((P) + AD - 1, A_L); LDB((P) + BD - 1, B_IMH); }
#define G2_REQ(P) LDB((P) + BD - 1, B_REL)
#define G3_REQ(P) LDB((P) + BD - 1, B_IML)
#endif
#if E_VSD == 0
#define G0_VS(P) VS(P, 0)
#define G1_VS(P) { VS(P, 1); VS(P, 2); }
#define G2_VS(P) VS(P, 3)
#define G3_VS(P) { VS(P, 4); VS(P, 5); }
#elif E_VSD == 1
#define G0_VS(P) { VS(P, 0); VS(P, 1); }
#define G1_VS(P) VS(P, 2)
#define G2_VS(P) { VS(P, 3); VS(P, 4); }
#define G3_VS(P) VS(P, 5)
#elif E_VSD == 2
#define G0_VS(P) VS(P, 0)
#define G1_VS(P) VS(P, 1)
#define G2_VS(P) { VS(P, 2); VS(P, 3); }
#define G3_VS(P) { VS(P, 4); VS(P, 5); }
#else
#define G0_VS(P) { VS(P, 0); VS(P, 1); }
#define G1_VS(P) { VS(P, 2); VS(P, 3); }
#define G2_VS(P) VS(P, 4)
#define G3_VS(P) VS(P, 5)
#endif
#define FREQ(P)                                                                                                  \
  {                                                                                                              \
    f32x4& t1 = T[((P) >> 2) & 1][2 * ((P) & 3)];                                                                \
    f32x4& t2 = T[((P) >> 2) & 1][2 * ((P) & 3) + 1];                                                            \
    if constexpr (seqf(P) < SC_NF && LO) {            /* operand pair 1, then pair 2, into T1 (Re tiles) and T2' (Im tiles) */ \
      SB(); TOUCH_OPS(P); MF0(t1, At[(P) % AD].h, Bt[(P) % BD].reh); SB(); G0_REQ(P); G0_VS(P);                  \
      SB(); MF0(t2, At[(P) % AD].h, Bt[(P) % BD].imh); SB(); G1_REQ(P); G1_VS(P);                                \
      SB(); MFA(t1, At[(P) % AD].l, Bt[(P) % BD].rel); SB(); G2_REQ(P); G2_VS(P);                                \
      SB(); MFA(t2, At[(P) % AD].l, Bt[(P) % BD].iml); SB(); G3_REQ(P); G3_VS(P);                                \
      SB();                                                                                                      \
    } else if constexpr (seqf(P) < SC_NF) {                                                                      \
      SB(); TOUCH_OPS(P); MF0(t1, At[(P) % AD].h, Bt[(P) % BD].reh); SB(); LDA((P) + AD - 1, A_H); VS(P, 0);      \
      SB(); MF0(t2, At[(P) % AD].h, Bt[(P) % BD].imh); SB(); VS(P, 1);                                           \
      SB(); LDB((P) + BD - 1, B_REH); VS(P, 2);                                                                  \
      SB(); LDB((P) + BD - 1, B_IMH); VS(P, 3);                                                                  \
      SB(); VS(P, 4);                                                                                            \
      SB(); VS(P, 5);                                                                                            \
      SB();                                                                                                      \
    } else {                                                                                                     \
      t1 = f32x4{0.f, 0.f, 0.f, 0.f}; t2 = t1;          /* the ghost frequency 31 */                              \
      SB(); VS(P, 0); SB(); VS(P, 1); SB(); VS(P, 2); SB(); VS(P, 3); SB(); VS(P, 4); SB(); VS(P, 5); SB();       \
    }                                                                                                            \
  }
    FREQ(0) FREQ(1) FREQ(2) FREQ(3) FREQ(4) FREQ(5) FREQ(6) FREQ(7) FREQ(8) FREQ(9) FREQ(10) FREQ(11) FREQ(12) FREQ(13) FREQ(14) FREQ(15)
    FREQ(16) FREQ(17) FREQ(18) FREQ(19)
    // (quad 4 has just finished the splits of quad 3: the first half's operands are complete)
    if constexpr (NW == 4) { SB(); park_half<LO>(hbs[0]); SB(); }      // one wave per SIMD: 512 registers, half of them AccVGPRs
    FREQ(20) FREQ(21) FREQ(22) FREQ(23) FREQ(24) FREQ(25) FREQ(26) FREQ(27)
    // the last quad also requests the stage-2 constants (the operand rings are draining)
    FREQ(28) FREQ(29)
    load_consts<0, LO>(c0, rc, lane * 16);
    FREQ(30)
    load_consts<1, LO>(c1, rc, lane * 16);
    FREQ(31)
    SB(); DRAIN(); SB();
#define VD(G) valu_slot<LO, 8, G>(T, hbs);
    VD(0) VD(1) VD(2) VD(3) VD(4) VD(5) VD(6) VD(7) VD(8) VD(9) VD(10) VD(11) VD(12) VD(13) VD(14) VD(15) VD(16) VD(17) VD(18) VD(19) VD(20)
    SB();
    {  // L2 prefetch for the whole XCD: the 32 NW waves that sweep this range on this XCD cover the group(s) of the iteration after next
       // with 6 cache lines each, one dword per line into a register nobody reads before the same point of the next unit
      asm volatile("" : : "v"(pf_sink));
      const int gp = (g - dpar) + 2 * GSTEP;
      const int pf_bytes = (gp + GSTEP <= DG) ? GSTEP * DIMG : (gp < DG ? (DG - gp) * DIMG : 0);
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)gp * DIMG), 0, pf_bytes, 0x00020000);
      int lp = lane;
      asm volatile("" : "+v"(lp));
      int pf_off = (lp < pfl) ? (pf_slot * pfl + lp) * 128 : (int)0x80000000;   // waves x lines >= the lines of the step's group(s); past the end: out of range
      if constexpr (SV) {                      // line L of the 31 x 2 hi tiles (6 lines each) of the step's groups -> its place in the split image
        const int L0 = pf_slot * pfl + lp, gi = L0 / GLINES, L = L0 - gi * GLINES, f = L / 12, r = L - f * 12;
        pf_off = (lp < pfl && gi < GSTEP) ? gi * SCH_DIMG + f * SCH_DFREQ + (r >= 6 ? 2 * SCH_DTILE : 0) + (r >= 6 ? r - 6 : r) * 128 : (int)0x80000000;
      }
      pf_sink = __builtin_amdgcn_raw_buffer_load_b32(rp, pf_off, 0, 0);
    }
    // ---------------------------------------------------------------- stage 2: group S = 2 R + V, its MFMAs into a fresh (E, O) tile pair;
    // group S + 1 is issued before group S is reduced; the first requests of the next unit sit between the groups
    f32x16 tE[2], tO[2];
    float mx = -__builtin_inff();
    int le = lane;
    asm volatile("" : "+v"(le));
    const int st_lane = ((le & 16) ? 4 * n : 0) * 4 + (le & 15) * 4;
    const int st_base = (le < 32 && g * 16 + (le & 15) < n) ? st_lane : (int)0x80000000;
    f32x2 dinf = {0.f, 0.f};
    if constexpr (SV) {                        // {sqrt(ones), 1/sqrt(ones)} of this lane's entry (entries past n: 0 -> count 0, store out of range)
      const int e = g * 16 + (le & 15);
      const float* dp = bin.dinfo + 2 * (size_t)(e < n ? e : 0);
      dinf = f32x2{dp[0], dp[1]};
    }
#define S2I(S, I) s2_one<LO, (NW == 4), ((S) >> 1), ((S) & 1), I>(hbs[0], hbs[1], c0, c1, tE[(S) & 1], tO[(S) & 1])
#define S2G(S, W0, W1, W2, W3, W4, W5, W6, W7, W8, W9, W10, W11)                                   \
  { if constexpr (LO) {                                                                            \
    SB(); S2I(S, 0); SB(); W0;  SB(); S2I(S, 1); SB(); W1;  SB(); S2I(S, 2); SB(); W2;             \
    SB(); S2I(S, 3); SB(); W3;  SB(); S2I(S, 4); SB(); W4;  SB(); S2I(S, 5); SB(); W5;             \
    SB(); S2I(S, 6); SB(); W6;  SB(); S2I(S, 7); SB(); W7;  SB(); S2I(S, 8); SB(); W8;             \
    SB(); S2I(S, 9); SB(); W9;  SB(); S2I(S, 10); SB(); W10; SB(); S2I(S, 11); SB(); W11; SB();    \
  } else {                                                                                         \
    SB(); S2I(S, 0); SB(); W0; W1; W2; SB(); S2I(S, 1); SB(); S2I(S, 2); SB(); W3; W4; W5;         \
    W6; W7; W8; SB(); S2I(S, 3); SB(); W9; W10; W11; SB(); } }
// (single-product form: the first piece of a tile pair's reduction carries an empty volatile asm that names the tiles - volatile statements
// keep their order, so the reduction stays behind the three MFMAs of the next group that the schedule puts in front of it; hipcc otherwise
// hoisted the pure reduction up to the tile's last MFMA in the SV form and reused the registers for the next group: a read 5 wait states
// behind an asm MFMA that nothing pads - tools/audit_asm_hazards.py)
#define RP(S, i) { if constexpr (!LO && (i) == 0) asm volatile("" : "+v"(tE[(S) & 1]), "+v"(tO[(S) & 1])); mx = red_piece(mx, tE[(S) & 1], tO[(S) & 1], i); }
#define ST(S) { if constexpr (SV) ep_store_round(mx, qinf[(S) >> 1], dinf, bnd25, viol, rd, st_base + ((S) >> 1) * 4 * n + g * 64);   \
                else ep_store(mx, rd, st_base + ((S) >> 1) * 4 * n + g * 64);                                                  \
                mx = -__builtin_inff(); }
#define NX(K) FIRST_REQ(K, rsn)
#define NONE ((void)0)
    S2G(0, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE)
    S2G(1, NX(16), NX(17), NX(18), RP(0, 0), RP(0, 1), RP(0, 2), RP(0, 3), RP(0, 4), RP(0, 5), RP(0, 6), RP(0, 7), NX(19))
    S2G(2, NX(0), NX(1), NX(20), RP(1, 0), RP(1, 1), RP(1, 2), RP(1, 3), RP(1, 4), RP(1, 5), RP(1, 6), RP(1, 7), ST(1))
    S2G(3, NX(2), NX(3), NX(4), RP(2, 0), RP(2, 1), RP(2, 2), RP(2, 3), RP(2, 4), RP(2, 5), RP(2, 6), RP(2, 7), NX(22))
    S2G(4, NX(5), NX(6), NX(21), RP(3, 0), RP(3, 1), RP(3, 2), RP(3, 3), RP(3, 4), RP(3, 5), RP(3, 6), RP(3, 7), ST(3))
    S2G(5, NX(7), NX(8), NX(9), RP(4, 0), RP(4, 1), RP(4, 2), RP(4, 3), RP(4, 4), RP(4, 5), RP(4, 6), RP(4, 7), NX(23))
    S2G(6, NX(10), NX(11), NX(12), RP(5, 0), RP(5, 1), RP(5, 2), RP(5, 3), RP(5, 4), RP(5, 5), RP(5, 6), RP(5, 7), ST(5))
    S2G(7, NX(13), NX(14), NX(15), RP(6, 0), RP(6, 1), RP(6, 2), RP(6, 3), RP(6, 4), RP(6, 5), RP(6, 6), RP(6, 7), NONE)
    // the last tiles are read next and nothing pads an asm MFMA; the tiles are operands of the statement, so their readers stay behind it
    // (without them hipcc hoisted the reduction above the nops in the SV form: stale tiles for the unit's last two queries)
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(tE[1]), "+v"(tO[1]));
    SB();
    RP(7, 0); RP(7, 1); RP(7, 2); RP(7, 3); RP(7, 4); RP(7, 5); RP(7, 6); RP(7, 7);
    ST(7)
    rs = rsn;
  }
  if constexpr (SV) { if (__any(viol)) { if (lane == 0) atomicExch(bin.viol, bin.gen); } }
}

}  // namespace

size_t sc_match_e_lds_bytes(int single, int nqg) { return (single ? (size_t)nqg * SCF_QIMG : (size_t)nqg * SCH_QIMG) + 64 + (single ? 64 * nqg : 0); }

void launch_sc_match_e(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p, float* d_i,
                       int nsplit_override, int single, int dgs) {
  if (m <= 0 || n <= 0) return;
  const int QG8 = single ? sc_qgroups8_f16(m) : sc_qgroups8(m), DG = sc_dgroups(n), DGS = dgs > 0 ? dgs : DG;
  // query groups per workgroup: as many as hold queries (an online batch of 9 .. 16 | .. 32 keyframes: the waves a group would leave to the
  // padding of the image take other DB groups instead)
  const int QGr = (m + 7) / 8;
  const int nqg = single ? (m <= 8 ? 1 : m <= 16 ? 2 : m <= 32 ? 4 : 8) : (m <= 16 && m > 8 ? 2 : 4);
  const int QGW = (single && m <= 8) ? 1 : ((QGr < QG8 ? QGr : QG8) + nqg - 1) / nqg;   // workgroups along the queries (8 nqg queries each)
  int nsplit = (128 + QGW - 1) / QGW;
  if (QGW <= 2) nsplit = 32 / QGW;                      // a few query groups (an online batch of 9 .. 64 keyframes): exactly one workgroup per CU, see launch_sc_match_e_bin
  if (nsplit > DG / 32) nsplit = DG / 32;
  if (nsplit < 1) nsplit = 1;
  if (nqg == 1) nsplit = DG / 128 > 0 ? DG / 128 : 1;   // an online call: ~32 DB groups per workgroup, 4 per wave; ~400 workgroups at n = 100k
  if (nsplit_override > 0) nsplit = nsplit_override * 4 <= DG ? nsplit_override : (DG >= 4 ? DG / 4 : 1);
  const ScBin none = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 1.f, 0, -1};
  auto go = [&](auto kern, int nw) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc_match_e_lds_bytes(single, nqg));
    hipLaunchKernelGGL(kern, dim3(8 * QGW * nsplit), dim3(64 * nw), sc_match_e_lds_bytes(single, nqg), st, static_cast<const char*>(qpk),
                       static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i, m, n, QG8, DG, nsplit, none, DGS);
  };
  if (single && m <= 8) go(sc_match_e_kernel<false, 8, 1>, 8);
  else if (single && nqg == 2) go(sc_match_e_kernel<false, 8, 2>, 8);
  else if (single && nqg == 4) go(sc_match_e_kernel<false, 8, 4>, 8);
  else if (single) go(sc_match_e_kernel<false, 8, 8>, 8);
  else if (m <= 8) {   // an online call in split-f16, both channels: one query group per workgroup (see launch_sc_match_e_bin); 8 x nsplit = one workgroup per CU
    int ns = DG / 32 < 32 ? (DG / 32 > 0 ? DG / 32 : 1) : 32;
    if (nsplit_override > 0) ns = nsplit_override * 4 <= DG ? nsplit_override : (DG >= 4 ? DG / 4 : 1);
    auto kern = sc_match_e_kernel<true, 4, 1>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc_match_e_lds_bytes(0, 1));
    hipLaunchKernelGGL(kern, dim3(8 * ns), dim3(256), sc_match_e_lds_bytes(0, 1), st, static_cast<const char*>(qpk), static_cast<const char*>(dpk),
                       static_cast<const u32x4*>(cst), d_p, d_i, m, n, QG8, DG, ns, none, DGS);
  }
  else if (nqg == 2) go(sc_match_e_kernel<true, 4, 2>, 4);
  else go(sc_match_e_kernel<true, 4, 4>, 4);
}

// (kernels.hpp) split-f16 images, m > 8: channel 0 in split-f16 on all XCDs; channel 1 twice -
// the single-product kernel on the hi halves with integer rounding (leaves at once when the bound does not hold), then the split-f16 kernel
// (leaves at once when that pass ran and every pair passed its rounding test)
void launch_sc_match_e_bin(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p, float* d_i,
                           int nsplit_override, ScBin bin, hipEvent_t* ev, int online_h, int dgs) {
  if (m <= 0 || n <= 0) return;
  const int QG8 = sc_qgroups8(m), DG = sc_dgroups(n), DGS = dgs > 0 ? dgs : DG;
  auto grid = [&](int QGW) {     // ranges per XCD: >= ~4 workgroups per CU in total, >= 8 DB groups per workgroup (an eighth of the ranges per XCD)
    int nsplit = (128 + QGW - 1) / QGW;
    // one or two workgroups along the queries (m <= 64: an online batch): 8 QGW nsplit = 256 workgroups = ONE per CU (a workgroup fills its
    // CU's LDS), each walking 1 / 256 of the DB groups - the throughput rule's 776 workgroups ran in 3.03 rounds, i.e. four, with four
    // prologues: 0.60 -> 0.48 ms per call at m = 32, 0.92 -> 0.82 at m = 64 (tools/latency_probe.py)
    if (QGW <= 2) nsplit = 32 / QGW;
    if (nsplit > DG / 64) nsplit = DG / 64;
    if (nsplit < 1) nsplit = 1;
    if (nsplit_override > 0) nsplit = nsplit_override * 8 <= DG ? nsplit_override : (DG >= 8 ? DG / 8 : 1);
    return nsplit;
  };
  auto go = [&](auto kern, int nw, int nqg, int single, int chsel, int gate) {
    const int QGr = (m + 7) / 8;                                       // query groups that hold queries (the image is padded to fours)
    const int QGW = ((QGr < QG8 ? QGr : QG8) + nqg - 1) / nqg, nsplit = grid(QGW);
    ScBin b = bin;
    b.chsel = chsel; b.gate = gate;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc_match_e_lds_bytes(single, nqg));
    hipLaunchKernelGGL(kern, dim3(8 * QGW * nsplit), dim3(64 * nw), sc_match_e_lds_bytes(single, nqg), st, static_cast<const char*>(qpk),
                       static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i, m, n, QG8, DG, nsplit, b, DGS);
  };
  if (m <= 8) {      // an online call: sc_match_h.hip's one-group form for the split-f16 launches, all eight waves on one query group here
    ScBin b = bin;
    // the split-f16 launches of an online call: this file's kernel with ONE query group per workgroup - its four waves split the DB groups,
    // the DB operand ring five positions deep in AccVGPRs (0.303 -> 0.278 ms per call at m = 1, 0.317 -> 0.300 at m = 8 against
    // sc_match_h.hip's one-group form, which PR_SC_ONLINE=h brings back)
    const bool online_e = !online_h;
    auto split1 = [&](int chsel, int gate) {
      b.chsel = chsel; b.gate = gate;
      if (!online_e) { launch_sc_match_h(st, qpk, m, dpk, n, cst, d_p, d_i, nsplit_override, &b); return; }
      int nsplit = DG / 64 < 32 ? (DG / 64 > 0 ? DG / 64 : 1) : 32;          // 8 x nsplit workgroups = one per CU, >= 8 units per wave
      if (nsplit_override > 0) nsplit = nsplit_override * 8 <= DG ? nsplit_override : (DG >= 8 ? DG / 8 : 1);
      auto kern = sc_match_e_kernel<true, 4, 1>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc_match_e_lds_bytes(0, 1));
      hipLaunchKernelGGL(kern, dim3(8 * nsplit), dim3(256), sc_match_e_lds_bytes(0, 1), st, static_cast<const char*>(qpk),
                         static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i, m, n, QG8, DG, nsplit, b, DGS);
    };
    if (ev) (void)hipEventRecord(ev[0], st);
    split1(0, 0);
    if (ev) (void)hipEventRecord(ev[1], st);
    {
      int nsplit = DG / 256 > 0 ? DG / 256 : 1;                      // ~32 DB groups per workgroup, 4 per wave (launch_sc_match_e: DG / 128 with four XCDs per channel)
      if (nsplit_override > 0) nsplit = nsplit_override * 8 <= DG ? nsplit_override : (DG >= 8 ? DG / 8 : 1);
      b.chsel = 1; b.gate = 1;
      auto kern = sc_match_e_kernel<false, 8, 1, true>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc_match_e_lds_bytes(1, 1));
      hipLaunchKernelGGL(kern, dim3(8 * nsplit), dim3(512), sc_match_e_lds_bytes(1, 1), st, static_cast<const char*>(qpk),
                         static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i, m, n, QG8, DG, nsplit, b, DGS);
    }
    if (ev) (void)hipEventRecord(ev[2], st);
    split1(1, 2);
    if (ev) (void)hipEventRecord(ev[3], st);
    return;
  }
  if (m <= 16) {     // 9 .. 16 queries: two query groups per workgroup, the other two (four) wave pairs take other DB groups instead of multiplying padding
    if (ev) (void)hipEventRecord(ev[0], st);
    go(sc_match_e_kernel<true, 4, 2>, 4, 2, 0, 0, 0);
    if (ev) (void)hipEventRecord(ev[1], st);
    go(sc_match_e_kernel<false, 8, 2, true>, 8, 2, 1, 1, 1);
    if (ev) (void)hipEventRecord(ev[2], st);
    go(sc_match_e_kernel<true, 4, 2>, 4, 2, 0, 1, 2);
    if (ev) (void)hipEventRecord(ev[3], st);
    return;
  }
  if (ev) (void)hipEventRecord(ev[0], st);
  go(sc_match_e_kernel<true, 4, 4>, 4, 4, 0, 0, 0);                 // channel 0
  if (ev) (void)hipEventRecord(ev[1], st);
  // channel 1, one product per term + rounding: runs when the bound predicts success (up to 32 queries: four query groups per workgroup,
  // two DB groups at a time - with eight, half of the waves would multiply padding)
  if (QG8 <= 4) go(sc_match_e_kernel<false, 8, 4, true>, 8, 4, 1, 1, 1);
  else go(sc_match_e_kernel<false, 8, 8, true>, 8, 8, 1, 1, 1);
  if (ev) (void)hipEventRecord(ev[2], st);
  go(sc_match_e_kernel<true, 4, 4>, 4, 4, 0, 1, 2);                 // channel 1 in split-f16: runs when it did not run, or a pair failed its test
  if (ev) (void)hipEventRecord(ev[3], st);
}

}  // namespace pr
