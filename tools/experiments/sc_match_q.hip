// sc_match_q.hip — all-pairs Scan-Context distance on the gfx950 f16 matrix cores with split-f16 ("hi + lo") operands
// (processSC.m:22-33), quarter-pipelined.  Mathematics as in sc_match.hip: per-ring sector spectra Q_r[f], D_r[f],
//   S_f = sum_r Q_r[f] conj(D_r[f])  (forward)      P_f = sum_r Q_r[f] D_r[f]  (mirror)      f = 0..30
//   corr_fwd[+k] = sum_f w_f (Re S_f cos(2 pi f k/60) - Im S_f sin(2 pi f k/60)),  corr_fwd[-k] = the same with + sin
// and max over the 60 + 60 shifts = max_{k=0..30} over {fwd,mirror} x {+k,-k}.
//
// Arithmetic: every fp32 factor x is carried as x = hi + lo, hi = f16(x), lo = f16(x - hi) (22 significand bits) and
// every product as the three f16 MFMAs a_hi*b_hi + a_hi*b_lo + a_lo*b_hi accumulated in fp32.  On MI355X the f16 MFMA rate
// is 16x the fp32 MFMA rate and, unlike v_mfma_f32_*_f32 (which occupies the SIMD's fp32 VALU datapath), the f16 MFMAs
// run beside the wave's own VALU work.  Spectra are pre-scaled (queries 2^8, DB 2^7, constants 2^10) so that hi and lo stay
// in the normal f16 range; the correlation is rescaled by 2^-25 at the end.
//
// One wave = 8 queries x 16 DB entries of one channel.
//   stage 1  v_mfma_f32_16x16x32_f16, K = 20 rings (+12 zero), per frequency: rows = {Re,Im} x 8 queries, cols = 16 entries
//            T1 = [Qr;Qi].Dr^T = (QrDr | QiDr)      T2 = [Qi;Qr].Di^T = (QiDi | QrDi)        lanes <32 | >=32
//            (T2's row operand = the same LDS image read with row ^ 8)
//            F = T1 + s T2 = (Re S_f | Im S_f)      M = T1 - s T2 = (Re P_f | Im P_f)        s = +1 | -1
//   split    F, M of two consecutive frequencies -> v_cvt_pk_f16_f32 (hi), residual -> v_cvt_pk_f16_f32 (lo): one register =
//            (Re | Im by lane half) x (2 frequencies); FOUR such registers (8 frequencies = one quarter) are, lane for lane,
//            the B operand of a 32x32x16 MFMA whose K index = 8*(lane>>5) + 0..7 = (Re | Im) x 8 frequencies
//   stage 2  v_mfma_f32_32x32x16_f16 per quarter:  D1[r][F|M] += [cos | -sin] . operand,  D2[r][F|M] += [cos | +sin] . operand
//            (A operand = constant [shift 0..31][Re x 8 | Im x 8] tile, hi and lo; 3 MFMAs per chain and quarter, 48 per
//            quarter; 256 accumulators) - no cross-lane movement at all between the two stages
//   epilogue max over D1, D2 (shifts +k, -k), forward and mirror -> d = 0.5 - 0.5 * 2^-25 * max       (processSC.m:30)
//
// Schedule.  The wave issues in order, one wave per SIMD (256 accumulators + ~250 VGPRs), so everything is placed by hand
// and pinned with sched_barrier.  A "gap" is the space behind one stage-1 MFMA; 6 gaps per frequency, 48 per quarter.
// Every gap carries ONE stage-2 MFMA of an earlier quarter (lag 12 gaps: quarter q's 48 stage-2 MFMAs run under the stage-1
// work of quarter q+1, gaps 12..47, and q+2, gaps 0..11), ~3 VALU instructions (F/M combination, split/pack of the
// previous pair) and 1-2 operand requests (DB tiles 8-12 slots ahead, query tiles 6-8 slots ahead).  The packed operands
// are double-buffered by quarter parity.  The epilogue of group g-1 runs between gaps 11 and 12 of quarter 1 of group g
// (no MFMA in flight: v_accvgpr_read beside running MFMAs costs ~25 cycles each); the loop runs one extra iteration to
// drain the last group.  A workgroup (4 waves) keeps the split spectra of 32 queries of one channel in LDS (159 712 B) and
// sweeps a range of the DB; DB operands stream L2 -> L1 -> VGPR with raw buffer loads (lanes 48-63 are out of range and
// read zeros = K padding 24..31; K = 20..23 is stored as zeros).  Workgroups are mapped XCD-aware and every wave
// prefetches a 1/128 share of the group two ahead into the XCD's L2 (see the kernel).
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

struct AOps { u32x4 h, l, rh, rl; };          // query row operands: Q hi, Q lo, and the same with Re/Im rows exchanged
struct BOps { u32x4 reh, rel, imh, iml; };    // DB column operands: Re hi, Re lo, Im hi, Im lo
struct Quarter { u32x4 Fh[4], Fl[4], Mh[4], Ml[4]; };   // packed stage-2 operands of 8 frequencies, per stage-1 register r
struct Consts { u32x4 d1h, d1l, d2h, d2l; };            // stage-2 A operands of one quarter: [cos|-sin] hi/lo, [cos|+sin] hi/lo

enum { A_H = 0, A_L = 1, A_RH = 2, A_RL = 3 };
enum { B_REH = 0, B_REL = 1, B_IMH = 2, B_IML = 3 };

struct State {          // everything a wave carries; fully scalarised by the compiler (all indices are compile-time)
  AOps At[2];           // query tiles of position p in At[p & 1]
  BOps Bt[2];           // DB tiles of position p in Bt[p & 1]   (32 positions per group: 31 frequencies + one ghost)
  Quarter hb[2];        // packed operands by quarter parity
  Consts c[4];          // constants by quarter
  f32x16 acc[4][2][2];  // [r][F|M][D1|D2]
  f32x4 Fa, Ma, Fb, Mb, t1a, t2a, t1b, t2b;
  unsigned ncur, rcur, nnxt, rnxt;   // LDS addresses of this lane's tiles: current pair, next pair (natural / row ^ 8)
};

#define SB() __builtin_amdgcn_sched_barrier(0)
// Stage-1 MFMAs in VGPR form, ONE instruction per asm statement (the 256 stage-2 accumulators own the AccVGPR half and
// hipcc picks one MFMA register form per function).  hipcc pads nothing around asm (cdna_hip_programming.md §5.7): an
// accumulate chain on the same vDst needs no wait states; every VALU reader of t1/t2 sits at least two MFMAs behind
// the last write.
#define MF0(d, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b))
#define MFA(d, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b))

// ------------------------------------------------------------------------------------------------ operand requests
// Position X (0..30 this group, 32.. = frequency X-32 of the next group; 31 = ghost, nothing).  The query image is the
// same for every group; its LDS addresses are kept per PAIR of frequencies (ncur / nnxt), the odd frequency and the lo
// tile are immediate offsets of the ds_read2_b64.
template <int P, int X, int T>
__device__ __forceinline__ void req_a(State& s, unsigned nat0, unsigned rot0) {
  if constexpr (X != 31 && X < 34) {
    constexpr int F = X < 31 ? X : X - 32;
    constexpr bool same_pair = (X >> 1) == (P >> 1);
    const unsigned base = (T & 2) ? (same_pair ? s.rcur : s.rnxt) : (same_pair ? s.ncur : s.nnxt);
    const u32x4 v = *reinterpret_cast<lds_tile_p>(base + (F & 1) * SCH_QBLK + (T & 1) * 40);
    AOps& a = s.At[X & 1];
    if (T == A_H) a.h = v; else if (T == A_L) a.l = v; else if (T == A_RH) a.rh = v; else a.rl = v;
  }
}
template <int X, int T>
__device__ __forceinline__ void req_b(State& s, __amdgpu_buffer_rsrc_t rs, __amdgpu_buffer_rsrc_t rsn, int voff) {
  if constexpr (X != 31 && X < 34) {
    constexpr int F = X < 31 ? X : X - 32;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(X < 31 ? rs : rsn, voff + T * SCH_DTILE, F * SCH_DFREQ, 0);
    BOps& b = s.Bt[X & 1];
    if (T == B_REH) b.reh = v; else if (T == B_REL) b.rel = v; else if (T == B_IMH) b.imh = v; else b.iml = v;
  }
}
template <int Q>
__device__ __forceinline__ void req_consts(State& s, __amdgpu_buffer_rsrc_t rc, int lane16, int t) {   // [quarter][D1|D2][hi|lo][64] x 16 B
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, (Q * 4 + t) * 1024, 0);
  Consts& c = s.c[Q];
  if (t == 0) c.d1h = v; else if (t == 1) c.d1l = v; else if (t == 2) c.d2h = v; else c.d2l = v;
}

// ------------------------------------------------------------------------------------------------ VALU pieces
// F = T1 + s T2, M = T1 - s T2 for registers r0, r0+1: two v_pk_fma_f32
__device__ __forceinline__ void fm2(f32x4& F, f32x4& M, const f32x4& t1, const f32x4& t2, f32x2 sg2, int r0) {
  const f32x2 a = {t1[r0], t1[r0 + 1]}, b = {t2[r0], t2[r0 + 1]};
  const f32x2 f = __builtin_elementwise_fma(b, sg2, a), m = __builtin_elementwise_fma(b, -sg2, a);
  F[r0] = f[0]; F[r0 + 1] = f[1]; M[r0] = m[0]; M[r0 + 1] = m[1];
}
// (hi, lo) split of two fp32 values into packed f16 pairs: hi = f16(x) (v_cvt_pk_f16_f32), lo = f16(x - hi), the residual
// formed exactly in fp32 by v_fma_mix_f32 (f16 operand x -1 + f32 operand)
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
// pair JG (0..15 in the group; frequencies 2JG, 2JG+1) -> element JG & 3 of the operands of quarter JG >> 2
template <int JG, int R, bool MIRROR>
__device__ __forceinline__ void pack(State& s) {
  Quarter& q = s.hb[(JG >> 2) & 1];
  unsigned h, l;
  if (MIRROR) { split2(s.Ma[R], s.Mb[R], h, l); q.Mh[R][JG & 3] = h; q.Ml[R][JG & 3] = l; }
  else        { split2(s.Fa[R], s.Fb[R], h, l); q.Fh[R][JG & 3] = h; q.Fl[R][JG & 3] = l; }
}

// The VALU work of gap I of position P (0..30; 31 = the ghost position after frequency 30).  Even P = first frequency of
// pair JG = P/2: F/M of the previous pair's second frequency (its last MFMA is >= 2 MFMAs back), then pack pieces of the
// previous pair; odd P: the remaining pack pieces, in the last gap F/M of this pair's first frequency (after the last
// reader of the old Fa/Ma).  Pair 15 has a single frequency (30): its Fb/Mb are zero.
template <int P, int I>
__device__ __forceinline__ void valu_gap(State& s, f32x2 sg2) {
  constexpr int JG = P >> 1, JP = (JG + 15) & 15;   // this pair, previous pair (15 = of the previous group)
  constexpr bool prev_has_b = JP != 15;
  if constexpr ((P & 1) == 0) {
    if constexpr (I == 1) { if (prev_has_b) fm2(s.Fb, s.Mb, s.t1b, s.t2b, sg2, 0); }
    if constexpr (I == 2) { if (prev_has_b) fm2(s.Fb, s.Mb, s.t1b, s.t2b, sg2, 2); }
    if constexpr (I == 3) pack<JP, 0, false>(s);
    if constexpr (I == 4) pack<JP, 0, true>(s);
    if constexpr (I == 5) pack<JP, 1, false>(s);
  } else {
    if constexpr (I == 0) pack<JP, 1, true>(s);
    if constexpr (I == 1) pack<JP, 2, false>(s);
    if constexpr (I == 2) pack<JP, 2, true>(s);
    if constexpr (I == 3) pack<JP, 3, false>(s);
    if constexpr (I == 4) pack<JP, 3, true>(s);
    if constexpr (I == 5) {
      if constexpr (P == 31) { s.Fb = f32x4{0.f, 0.f, 0.f, 0.f}; s.Mb = s.Fb; }
      fm2(s.Fa, s.Ma, s.t1a, s.t2a, sg2, 0);
      fm2(s.Fa, s.Ma, s.t1a, s.t2a, sg2, 2);
    }
  }
}

// ------------------------------------------------------------------------------------------------ stage 2
// MFMA number S (0..47) of source quarter QS: r = S/12; within r: product T = (S%12)/4 (hi x hi, lo(const) x hi, hi x lo),
// V = forward | mirror, D = D1 | D2.  The four chains of a register are 4 gaps apart.
template <int QS, int S>
__device__ __forceinline__ void stage2_mfma(State& s, const f32x16& zero) {
  constexpr int R = S / 12, I = S % 12, T = I >> 2, V = (I >> 1) & 1, D = I & 1;
  const Consts& c = s.c[QS];
  const Quarter& q = s.hb[QS & 1];
  const u32x4& ca = D ? (T == 1 ? c.d2l : c.d2h) : (T == 1 ? c.d1l : c.d1h);
  const u32x4& op = V ? (T == 2 ? q.Ml[R] : q.Mh[R]) : (T == 2 ? q.Fl[R] : q.Fh[R]);
  f32x16& acc = s.acc[R][V][D];
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ca), __builtin_bit_cast(f16x8, op),
                                                (QS == 0 && T == 0) ? zero : acc, 0, 0, 0);
}
// the stage-2 MFMA carried by gap GAM (0..47) of quarter Q: gaps 0..11 finish quarter Q-2, gaps 12..47 run quarter Q-1
template <int Q, int GAM>
__device__ __forceinline__ void stage2_gap(State& s, const f32x16& zero) {
  if constexpr (GAM < 12) stage2_mfma<(Q + 2) & 3, 36 + GAM>(s, zero);
  else stage2_mfma<(Q + 3) & 3, GAM - 12>(s, zero);
}

// ------------------------------------------------------------------------------------------------ one position
// The six gaps of position P: stage-1 MFMA (P < 31), its stage-2 MFMA, operand requests, VALU pieces.
template <int P>
__device__ __forceinline__ void position(State& s, f32x2 sg2, const f32x16& zero, unsigned nat0, unsigned rot0,
                                         __amdgpu_buffer_rsrc_t rs, __amdgpu_buffer_rsrc_t rsn, int voff,
                                         __amdgpu_buffer_rsrc_t rc, int lane16) {
  constexpr int Q = P >> 3, G0 = (P & 7) * 6;
  constexpr bool REAL = P < 31;
  AOps& a = s.At[P & 1];
  BOps& b = s.Bt[P & 1];
  f32x4& t1 = (P & 1) ? s.t1b : s.t1a;
  f32x4& t2 = (P & 1) ? s.t2b : s.t2a;
  // constants of the quarter whose stage 2 starts at gap 12 of this quarter: requested just in time (the hi tiles of the
  // previous set are used up to gap 11, its lo tiles up to gap 7) in gaps 6..9 = the second position of the quarter
  constexpr bool CL = (P & 7) == 1;
  constexpr int QC = (Q + 3) & 3;
#define GAP(I, MF, REQS)                                                    \
  SB(); if constexpr (REAL) { MF; }                                         \
  SB(); stage2_gap<Q, G0 + I>(s, zero);                                     \
  SB(); REQS; valu_gap<P, I>(s, sg2);
  // operand requests: DB tiles of the next position 5-6 slots ahead, query tiles 3 slots ahead (every slot now also holds
  // a 32-cycle stage-2 MFMA, so a slot is >= 48 cycles)
  GAP(0, MF0(t1, a.h, b.reh),  (req_b<P + 1, B_REH>(s, rs, rsn, voff), req_a<P, P, A_RL>(s, nat0, rot0), CL ? req_consts<QC>(s, rc, lane16, 0) : (void)0))
  GAP(1, MF0(t2, a.rh, b.imh), (req_b<P + 1, B_IMH>(s, rs, rsn, voff), CL ? req_consts<QC>(s, rc, lane16, 2) : (void)0))
  GAP(2, MFA(t1, a.l, b.reh),  (CL ? req_consts<QC>(s, rc, lane16, 1) : (void)0))
  GAP(3, MFA(t2, a.rl, b.imh), (req_a<P, P + 1, A_H>(s, nat0, rot0), CL ? req_consts<QC>(s, rc, lane16, 3) : (void)0))
  GAP(4, MFA(t1, a.h, b.rel),  (req_b<P + 1, B_REL>(s, rs, rsn, voff), req_a<P, P + 1, A_RH>(s, nat0, rot0)))
  GAP(5, MFA(t2, a.rh, b.iml), (req_b<P + 1, B_IML>(s, rs, rsn, voff), req_a<P, P + 1, A_L>(s, nat0, rot0)))
#undef GAP
  SB();
  if constexpr ((P & 1) == 1) {   // end of a pair: LDS addresses of the pair after the next one (wraps to the next group)
    constexpr int JN = ((P >> 1) + 2) & 15;
    s.ncur = s.nnxt;
    s.rcur = s.rnxt;
    s.nnxt = nat0 + JN * 2 * SCH_QBLK;
    s.rnxt = rot0 + JN * 2 * SCH_QBLK;
    asm("" : "+v"(s.nnxt));
    asm("" : "+v"(s.rnxt));
  }
}

// epilogue piece: shift rows e of register R -> running max over D1, D2 of forward and mirror
template <int R>
__device__ __forceinline__ void ep_elem(float& mx, const State& s, int e) {
  mx = fmaxf(fmaxf(mx, s.acc[R][0][0][e]), s.acc[R][0][1][e]);
  mx = fmaxf(fmaxf(mx, s.acc[R][1][0][e]), s.acc[R][1][1][e]);
}
// 2 queries x 16 entries (lanes 0..31): d = (1 - max)/2 with the 2^-25 operand scaling folded in   (processSC.m:30).
// Branch-free (a buffer store whose invalid lanes are out of range), so that the whole group body stays ONE basic block
// and the hand-placed order survives the compiler's sinking passes.
template <int R>
__device__ __forceinline__ void epilogue(const State& s, __amdgpu_buffer_rsrc_t rd, int st_off) {
  float mx = -__builtin_inff();
#pragma unroll
  for (int e = 0; e < 16; e++) ep_elem<R>(mx, s, e);
  const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
  mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));     // max over the two lane halves (shift rows +0..3 | +4..7)
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(mx, -0x1p-26f, 0.5f)), rd, st_off, 0, 0);
}

__global__ __launch_bounds__(256, 1) void sc_match_q_kernel(const char* __restrict__ qpk,   // [2][QG32][4][31][1288 B]
                                                            const char* __restrict__ dpk,   // [2][DG][31][4][768 B] + 2 zero groups
                                                            const u32x4* __restrict__ cst,  // [4][2][2][64] x 16 B
                                                            float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                            int m, int n, int QG8, int DG, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware mapping (workgroups go round-robin to the 8 XCDs, each with its own L2): all workgroups of one XCD work
  // on ONE channel and on the same quarter of the DB ranges, consecutive workgroups of an XCD on consecutive 32-query
  // blocks - so the ~32 resident workgroups of an XCD sweep the same DB range together and share it through that L2.
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ch = xcd & 1;
  const int range = (xcd >> 1) + 4 * (idx % nsplit), qg32 = idx / nsplit;      // nsplit = ranges per XCD slice
  const int nrange = 4 * nsplit;
  const int g0 = (int)((long long)DG * range / nrange), g1 = (int)((long long)DG * (range + 1) / nrange);

  {  // the 4 query groups of this workgroup -> LDS (linear copy; the packed image IS the LDS image) + zeroed tail
    const u32x4* src = reinterpret_cast<const u32x4*>(qpk + ((size_t)ch * QG8 + (size_t)qg32 * 4) * SCH_QIMG);
    u32x4* dst = reinterpret_cast<u32x4*>(lds);
    constexpr int NV = 4 * SCH_QIMG / 16;
    for (int i = tid; i < NV + 4; i += 256) dst[i] = (i < NV) ? src[i] : u32x4{0u, 0u, 0u, 0u};
  }
  __syncthreads();
  if (g0 >= g1) return;

  const int row = lane & 15, kg = lane >> 4;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
  const unsigned nat0 = lds0 + w * SCH_QIMG + row * 80 + (row >= 8 ? 8 : 0) + kg * 16;
  const unsigned rot0 = lds0 + w * SCH_QIMG + (row ^ 8) * 80 + (row >= 8 ? 0 : 8) + kg * 16;
  const int voff = (lane < 48) ? lane * 16 : (int)0x80000000;     // lanes 48-63: out of range -> zeros (K = 24..31)
  const float sg = (lane < 32) ? 1.0f : -1.0f;
  const f32x2 sg2 = {sg, sg};
  float* dist = ch ? dist_i : dist_p;
  const char* dbase = dpk + ((size_t)ch * DG) * SCH_DIMG;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int qrow0 = qg32 * 32 + w * 8;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(cst), 0, 16384, 0x00020000);
  // distances of this wave's 8 query rows: byte offset = ((local row) * n + entry) * 4; local row = R (lanes 0-15) or 4 + R (16-31)
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      dist + (size_t)qrow0 * n, 0, (qrow0 < m ? (m - qrow0 < 8 ? m - qrow0 : 8) : 0) * n * 4, 0x00020000);
  const int st_lane = ((lane & 16) ? 4 * n : 0) * 4 + (lane & 15) * 4;
  const int pf_slot = (qg32 & 31) * 4 + w;                                  // 0..127
  const int pf_off = (lane < 6) ? (pf_slot * 6 + lane) * 128 : (int)0x80000000;   // lines past the group are out of range
  unsigned pf_sink = 0;
  const int lane16 = lane * 16;

  State s;
  {  // defined (zero) contents for everything the first iteration consumes on behalf of the non-existent previous group
    const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) { s.hb[b].Fh[r] = z4; s.hb[b].Fl[r] = z4; s.hb[b].Mh[r] = z4; s.hb[b].Ml[r] = z4; }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int v = 0; v < 2; v++) { s.acc[r][v][0] = zero; s.acc[r][v][1] = zero; }
    s.Fa = f32x4{0.f, 0.f, 0.f, 0.f}; s.Ma = s.Fa; s.Fb = s.Fa; s.Mb = s.Fa;
    s.c[2].d1h = z4; s.c[2].d1l = z4; s.c[2].d2h = z4; s.c[2].d2l = z4;   // "quarter 2 of the previous group"
  }
  __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)g0 * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
  // requests that positions "-2" and "-1" would have issued for positions 0 and 1
  s.ncur = nat0; s.rcur = rot0; s.nnxt = nat0; s.rnxt = rot0;
  req_b<0, B_REH>(s, rs, rs, voff); req_b<0, B_IMH>(s, rs, rs, voff); req_b<0, B_REL>(s, rs, rs, voff); req_b<0, B_IML>(s, rs, rs, voff);
  req_a<0, 0, A_H>(s, nat0, rot0); req_a<0, 0, A_RH>(s, nat0, rot0); req_a<0, 0, A_L>(s, nat0, rot0);
  s.nnxt = nat0 + 2 * SCH_QBLK; s.rnxt = rot0 + 2 * SCH_QBLK;
  req_consts<3>(s, rc, lane16, 0); req_consts<3>(s, rc, lane16, 1); req_consts<3>(s, rc, lane16, 2); req_consts<3>(s, rc, lane16, 3);

  for (int g = g0; g <= g1; g++) {   // iteration g: stage 1 of group g, tail of stage 2 + epilogue of group g-1
    const __amdgpu_buffer_rsrc_t rsn =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)(g + 1) * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
    {  // L2 prefetch of group g + 2 for the whole XCD: this wave's 6 of its 744 cache lines, one dword per line into a
       // register nobody reads before the same point of the next group.  The ~128 waves that sweep this range on this
       // XCD cover the group between them, so the demand loads two groups later hit the L2 instead of paying HBM latency
       // in the middle of the in-order load queue.
      asm volatile("" : : "v"(pf_sink));
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(dbase + (size_t)(g + 2) * SCH_DIMG), 0, (g + 2 < DG) ? SCH_DIMG : 0, 0x00020000);
      pf_sink = __builtin_amdgcn_raw_buffer_load_b32(rp, pf_off, 0, 0);
    }
#define POS(P) position<P>(s, sg2, zero, nat0, rot0, rs, rsn, voff, rc, lane16);
    POS(0) POS(1) POS(2) POS(3) POS(4) POS(5) POS(6) POS(7) POS(8) POS(9)
    {  // all 192 stage-2 MFMAs of group g-1 have been issued: its epilogue, before quarter 0 of group g restarts the chains
      const bool ok = g > g0 && lane < 32 && (g - 1) * 16 + (lane & 15) < n;
      const unsigned st_base = ok ? (unsigned)(st_lane + (g - 1) * 64) : 0x80000000u;   // invalid lanes: out of range
      SB();
      epilogue<0>(s, rd, (int)st_base);
      epilogue<1>(s, rd, (int)(st_base + 4u * (unsigned)n));
      epilogue<2>(s, rd, (int)(st_base + 8u * (unsigned)n));
      epilogue<3>(s, rd, (int)(st_base + 12u * (unsigned)n));
      SB();
    }
    POS(10) POS(11) POS(12) POS(13) POS(14) POS(15)
    POS(16) POS(17) POS(18) POS(19) POS(20) POS(21) POS(22) POS(23)
    POS(24) POS(25) POS(26) POS(27) POS(28) POS(29) POS(30) POS(31)
#undef POS
    rs = rsn;
  }
}

}  // namespace

size_t sc_match_q_lds_bytes() { return (size_t)4 * SCH_QIMG + 64; }

void launch_sc_match_q(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p,
                       float* d_i, int nsplit_override) {
  if (m <= 0 || n <= 0) return;
  const int QG8 = sc_qgroups8(m), DG = sc_dgroups(n);
  const int QG32 = QG8 / 4;
  // grid = 8 XCD slices (channel x quarter of the ranges) x QG32 query blocks x nsplit ranges per slice
  int nsplit = (128 + QG32 - 1) / QG32;            // >= ~4 workgroups per CU in total, for tail balance
  if (nsplit > DG / 32) nsplit = DG / 32;          // keep >= 8 DB groups (128 entries) per workgroup
  if (nsplit < 1) nsplit = 1;
  if (nsplit_override > 0) nsplit = nsplit_override * 4 <= DG ? nsplit_override : (DG >= 4 ? DG / 4 : 1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_match_q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sc_match_q_lds_bytes());
  hipLaunchKernelGGL(sc_match_q_kernel, dim3(8 * QG32 * nsplit), dim3(256), sc_match_q_lds_bytes(), st,
                     static_cast<const char*>(qpk), static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i,
                     m, n, QG8, DG, nsplit);
}

}  // namespace pr
