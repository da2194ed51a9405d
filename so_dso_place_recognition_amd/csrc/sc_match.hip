// sc_match.hip — all-pairs Scan-Context distance on gfx950 fp32 MFMA (processSC.m:22-33).
//
// Reference arithmetic: d(i,j) = min over the 120 variants (60 column rotations + 60 mirrored rotations,
// permute_sc processSC.m:37-45) of (1 - <variant_k(q_i), d_j>)/2 on L2-normalised 1200-vectors, per channel.
// The 120 inner products are the circular cross-correlation (forward) and circular convolution (mirror) of
// the 60 sectors summed over the 20 rings, so with the per-ring sector spectra Q_r[f], D_r[f] (sc_pack.hip)
//   S_f = sum_r Q_r[f] conj(D_r[f])   (forward)        P_f = sum_r Q_r[f] D_r[f]   (mirror)          f = 0..30
//   dot_fwd[k] = E[k] + O[k],  dot_fwd[60-k] = E[k] - O[k],   k = 0..30
//   E[k] = sum_f w_f Re S_f cos(2 pi f k/60)  (even in k)     O[k] = -sum_f w_f Im S_f sin(2 pi f k/60)  (odd in k)
// hence  max over the 60 shifts = max_{k=0..30} ( E[k] + |O[k]| ), and the same with P_f for the 60 mirrored shifts
// (w_0 = w_30 = 1, else 2; the 1/60 is folded into the packed spectra).  23 856 FLOP per (query, entry) pair
// instead of the dense 576 000.
//
// Mapping to the matrix cores (one wave = 8 queries x 16 DB entries; the 31 frequencies are processed as 16 slots
// of two frequencies: (0,30), (1,2), ..., (27,28), (29,-)):
//   stage 1  v_mfma_f32_16x16x4_f32, K = 20 rings, per frequency:  rows = {Re,Im} x 8 queries, cols = 16 entries
//            T1 = [Qre;Qim] . Dre^T = (A | C)      T2 = [Qim;-Qre] . Dim^T = (B | -E)
//            (T2's row operand = T1's through DPP row_ror:8 with a sign flip - no second LDS image)
//            F = T1 + T2 = (Re S_f | Im S_f)       M = T1 - T2 = (Re P_f | Im P_f)              lanes <32 | >=32
//   swap     v_permlane32_swap(F_fa[r], F_fb[r]) -> (Re S_fa | Re S_fb), (Im S_fa | Im S_fb): each register is now, lane
//            for lane, the B operand (k = lane>>5, pair = lane&31) of a 32x32x2 MFMA over K = {fa, fb}
//   stage 2  v_mfma_f32_32x32x2_f32:  E[r][fwd|mir] += Ccos[slot] . Re-tile,  O[r][fwd|mir] += Csin[slot] . Im-tile
//            (A operand = constant [shift 0..31][fa|fb] tile; 16 MFMAs per slot; 256 accumulator registers)
//   epilogue max over shifts of E + |O| (in-lane over 16 registers + one cross-half exchange) -> d = 0.5 - 0.5*max.
// Stage 1 feeds stage 2 through registers only: no LDS or HBM round trip of the m x n x 124 intermediate.
// A workgroup (4 waves, one per SIMD, <=512 VGPR+AGPR each) keeps the spectra of 32 queries of one channel
// resident in LDS (158 720 B) and sweeps a range of the DB; the DB operand stream goes HBM/L2 -> VGPR directly
// (all 4 waves read the same 5 KB per slot, so 3 of 4 hit L1).  Bound: MFMA fp32 (157.3 TFLOP/s).
#include "kernels.hpp"

namespace pr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct FreqOps {    // operands of one frequency
  float a[5];       // query rows, K-steps s = 0..4                    (LDS)
  f32x4 bre, bim;   // DB Re / Im, K-steps 0..3                        (global)
  float2 b4;        // DB (Re, Im) of K-step 4                         (global)
};
struct SlotOps {    // operands of one slot = two frequencies + the two stage-2 constant tiles
  FreqOps f[2];
  float ce, co;
};
struct Tiles {      // stage-2 B operands of one slot: {forward, mirror} x {Re, Im} x 4 registers
  f32x4 fre, fim, mre, mim;
};

constexpr int SS = 2 * SC_DSTEP;   // floats per slot in the packed DB stream

__device__ __forceinline__ float ror8(float x) {  // DPP row_ror:8 inside each 16-lane row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));
}

// Slot kinds: slot 0 = (0,30): both spectra are real (no Im loads, no T2 chain, no O accumulation);
// slot 15 = (29,-): single frequency; every other slot is a full pair.
template <int SLOT, int H>
__device__ __forceinline__ void load_freq_lds(SlotOps& o, const float* __restrict__ la) {
  if (SLOT == SC_NSLOT - 1 && H == 1) return;
  const float* a = la + (2 * SLOT + H) * 320;
#pragma unroll
  for (int s = 0; s < 5; s++) o.f[H].a[s] = a[s * 64];
}
template <int SLOT, int H>
__device__ __forceinline__ void load_freq_db(SlotOps& o, const float* __restrict__ db, int lane) {
  if (SLOT == SC_NSLOT - 1 && H == 1) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(db + H * SC_DSTEP);
  o.f[H].bre = p[lane];
  if (SLOT != 0) {
    o.f[H].bim = p[64 + lane];
    o.f[H].b4 = reinterpret_cast<const float2*>(db + H * SC_DSTEP + 512)[lane];
  } else {
    o.f[H].b4.x = db[H * SC_DSTEP + 512 + 2 * lane];
  }
}
template <int SLOT, int H>
__device__ __forceinline__ void load_freq(SlotOps& o, const float* __restrict__ la, const float* __restrict__ db, int lane) {
  load_freq_lds<SLOT, H>(o, la);
  load_freq_db<SLOT, H>(o, db, lane);
}
template <int SLOT>
__device__ __forceinline__ void load_consts(SlotOps& o, const float* __restrict__ cst, int lane) {
  o.ce = cst[SLOT * 128 + lane];
  if (SLOT != 0) o.co = cst[SLOT * 128 + 64 + lane];
}
template <int SLOT>
__device__ __forceinline__ void load_slot(SlotOps& o, const float* __restrict__ la, const float* __restrict__ db,
                                          const float* __restrict__ cst, int lane) {
  load_freq<SLOT, 0>(o, la, db, lane);
  load_freq<SLOT, 1>(o, la, db, lane);
  load_consts<SLOT>(o, cst, lane);
}

// Stage 1 of one frequency in VGPR form.  The 256 accumulators of stage 2 fill the whole AccVGPR half of the register
// file and hipcc selects one MFMA form per function (AGPR C/D here), so the small stage-1 accumulators are kept in
// ArchVGPRs by hand: ONE asm statement with the MFMAs and their wait states inside (cdna_hip_programming.md §5.7):
// s_nop 1 covers the VALU-written A operands (the DPP products), the trailing s_nop 10 covers the 8-pass MFMA D -> VALU
// read of the add/sub/swap that consumes t1/t2 (placed after a block of stage-2 MFMAs, but never rely on that);
// back-to-back SrcC == vDst chains need none.
struct Rot { float r[5]; };
__device__ __forceinline__ void make_rot(const FreqOps& o, float sgn, Rot& q) {   // [Qim; -Qre] row operand of T2
#pragma unroll
  for (int k = 0; k < 5; k++) q.r[k] = ror8(o.a[k]) * sgn;
}
__device__ __forceinline__ void stage1_full(const FreqOps& o, const Rot& q, f32x4& t1, f32x4& t2) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %2, %12, 0\n\t"
      "v_mfma_f32_16x16x4_f32 %1, %7, %17, 0\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %3, %13, %0\n\t"
      "v_mfma_f32_16x16x4_f32 %1, %8, %18, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %4, %14, %0\n\t"
      "v_mfma_f32_16x16x4_f32 %1, %9, %19, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %5, %15, %0\n\t"
      "v_mfma_f32_16x16x4_f32 %1, %10, %20, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %6, %16, %0\n\t"
      "v_mfma_f32_16x16x4_f32 %1, %11, %21, %1\n\t"
      "s_nop 10"
      : "=&v"(t1), "=&v"(t2)
      : "v"(o.a[0]), "v"(o.a[1]), "v"(o.a[2]), "v"(o.a[3]), "v"(o.a[4]),          // %2..%6   query rows
        "v"(q.r[0]), "v"(q.r[1]), "v"(q.r[2]), "v"(q.r[3]), "v"(q.r[4]),          // %7..%11  rotated, sign-flipped rows
        "v"(o.bre[0]), "v"(o.bre[1]), "v"(o.bre[2]), "v"(o.bre[3]), "v"(o.b4.x),  // %12..%16 DB Re
        "v"(o.bim[0]), "v"(o.bim[1]), "v"(o.bim[2]), "v"(o.bim[3]), "v"(o.b4.y)); // %17..%21 DB Im
}
// Real spectra (f = 0, 30): Im parts are zero, so T2 = 0 and F = M = T1.  Two independent half-chains (K-steps
// 0,2,4 and 1,3) keep the 40-cycle dependent latency of the 16x16x4 MFMA off the critical path.
__device__ __forceinline__ void stage1_real(const FreqOps& o, f32x4& t1) {
  f32x4 u;
  asm volatile(
      "v_mfma_f32_16x16x4_f32 %0, %2, %7, 0\n\t"
      "v_mfma_f32_16x16x4_f32 %1, %3, %8, 0\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %4, %9, %0\n\t"
      "v_mfma_f32_16x16x4_f32 %1, %5, %10, %1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %6, %11, %0\n\t"
      "s_nop 10"
      : "=&v"(t1), "=&v"(u)
      : "v"(o.a[0]), "v"(o.a[1]), "v"(o.a[2]), "v"(o.a[3]), "v"(o.a[4]),
        "v"(o.bre[0]), "v"(o.bre[1]), "v"(o.bre[2]), "v"(o.bre[3]), "v"(o.b4.x));
  t1 = t1 + u;
}

__device__ __forceinline__ void swap_halves(const f32x4& xa, const f32x4& xb, f32x4& re, f32x4& im) {
#pragma unroll
  for (int r = 0; r < 4; r++) {   // lanes 32-63 of xa <-> lanes 0-31 of xb
    const u32x2 v = __builtin_amdgcn_permlane32_swap(__float_as_uint(xa[r]), __float_as_uint(xb[r]), false, false);
    re[r] = __uint_as_float(v[0]);
    im[r] = __uint_as_float(v[1]);
  }
}

#define SB() __builtin_amdgcn_sched_barrier(0)


// ONE stage-2 MFMA of slot S: accumulator (r, v = forward|mirror, part = E|O)
template <int S, int V, int PART>
__device__ __forceinline__ void stage2_one(int r, const Tiles& t, float ce, float co, f32x16 (&accE)[4][2],
                                           f32x16 (&accO)[4][2], const f32x16& zero) {
  if (S == 0) {                    // real pair: forward and mirror get the same E contribution, O stays empty
    if (PART == 0) accE[r][V] = __builtin_amdgcn_mfma_f32_32x32x2f32(ce, t.fre[r], zero, 0, 0, 0);
    return;
  }
  // slot 15 (single frequency): unswapped (Re|Im) tiles against (cos|0) and (0|-sin); slot 1 starts O from zero
  const f32x4& re = V ? t.mre : t.fre;
  const f32x4& im = (S == SC_NSLOT - 1) ? (V ? t.mre : t.fre) : (V ? t.mim : t.fim);
  if (PART == 0) accE[r][V] = __builtin_amdgcn_mfma_f32_32x32x2f32(ce, re[r], accE[r][V], 0, 0, 0);
  else accO[r][V] = __builtin_amdgcn_mfma_f32_32x32x2f32(co, im[r], (S == 1) ? zero : accO[r][V], 0, 0, 0);
}
template <int S>
__device__ __forceinline__ void stage2_r(int r, const Tiles& t, float ce, float co, f32x16 (&accE)[4][2],
                                         f32x16 (&accO)[4][2], const f32x16& zero) {
  stage2_one<S, 0, 0>(r, t, ce, co, accE, accO, zero);
  stage2_one<S, 0, 1>(r, t, ce, co, accE, accO, zero);
  stage2_one<S, 1, 0>(r, t, ce, co, accE, accO, zero);
  stage2_one<S, 1, 1>(r, t, ce, co, accE, accO, zero);
}

// One pipeline step: stage 2 of slot S (tiles tc), stage 1 + swap of slot S+1 (operands X -> tiles for the next
// step), request of the operands of slot S+3 (into Z).  The wave issues in order, and an MFMA only blocks issue
// until the matrix pipe accepts it, so the pipe stays busy as long as no more than a handful of other instructions
// sit between two MFMAs.  The schedule is therefore written out by hand and pinned with sched_barrier: every chunk
// of loads / VALU (<= 8 instructions) is followed by an MFMA that was independent of it.
template <int S>
__device__ __forceinline__ void slot_step(SlotOps& X, const SlotOps& Y, SlotOps& Z, Tiles& tc, float& ce, float& co,
                                          f32x16 (&accE)[4][2], f32x16 (&accO)[4][2], const f32x16& zero,
                                          const float* __restrict__ la, const float*& dbn,
                                          const float* __restrict__ cst, int lane, float sgn, Rot& rotA) {
  constexpr int N1 = (S + 1) & (SC_NSLOT - 1);   // slot whose stage 1 runs in this step
  constexpr int N2 = (S + 2) & (SC_NSLOT - 1);
  constexpr int N3 = (S + 3) & (SC_NSLOT - 1);   // slot whose operands are requested in this step
  constexpr bool REAL = (N1 == 0), SINGLE = (N1 == SC_NSLOT - 1);
  Tiles tn;
  f32x4 t1a, t2a, t1b, t2b, Fa, Ma, Fb, Mb;
  Rot rotB;
  SB();
  if (REAL) stage1_real(X.f[0], t1a); else stage1_full(X.f[0], rotA, t1a, t2a);      // 10 MFMAs (VGPR form)
  SB();
  // r = 0: four MFMAs, the VALU consumers of stage 1 (first frequency) in their shadow
  stage2_one<S, 0, 0>(0, tc, ce, co, accE, accO, zero);
  SB();
  if (REAL) Fa = t1a; else Fa = t1a + t2a;
  SB();
  stage2_one<S, 0, 1>(0, tc, ce, co, accE, accO, zero);
  SB();
  if (!REAL) Ma = t1a - t2a;
  SB();
  stage2_one<S, 1, 0>(0, tc, ce, co, accE, accO, zero);
  SB();
  if (!REAL && !SINGLE) make_rot(X.f[1], sgn, rotB);
  SB();
  stage2_one<S, 1, 1>(0, tc, ce, co, accE, accO, zero);
  SB();
  if (!SINGLE) { if (REAL) stage1_real(X.f[1], t1b); else stage1_full(X.f[1], rotB, t1b, t2b); }
  SB();
  // r = 1: add/sub and permlane swaps of the second frequency
  stage2_one<S, 0, 0>(1, tc, ce, co, accE, accO, zero);
  SB();
  if (REAL) Fb = t1b; else if (!SINGLE) Fb = t1b + t2b;
  SB();
  stage2_one<S, 0, 1>(1, tc, ce, co, accE, accO, zero);
  SB();
  if (!REAL && !SINGLE) Mb = t1b - t2b;
  SB();
  stage2_one<S, 1, 0>(1, tc, ce, co, accE, accO, zero);
  SB();
  if (SINGLE) tn.fre = Fa; else swap_halves(Fa, Fb, tn.fre, tn.fim);   // real pair: (Re_0 | Re_30), Im tile unused
  SB();
  stage2_one<S, 1, 1>(1, tc, ce, co, accE, accO, zero);
  SB();
  if (SINGLE) tn.mre = Ma; else if (!REAL) swap_halves(Ma, Mb, tn.mre, tn.mim);
  SB();
  // r = 2, 3: eight MFMAs; the operand requests of slot S+3 (<= 5 memory instructions per gap) and the row
  // operand of the next step's first stage 1 go in between
  stage2_one<S, 0, 0>(2, tc, ce, co, accE, accO, zero);
  SB();
  load_freq_db<N3, 0>(Z, dbn, lane);
  SB();
  stage2_one<S, 0, 1>(2, tc, ce, co, accE, accO, zero);
  SB();
  load_freq_lds<N3, 0>(Z, la);
  SB();
  stage2_one<S, 1, 0>(2, tc, ce, co, accE, accO, zero);
  SB();
  load_freq_db<N3, 1>(Z, dbn, lane);
  SB();
  stage2_one<S, 1, 1>(2, tc, ce, co, accE, accO, zero);
  SB();
  load_freq_lds<N3, 1>(Z, la);
  SB();
  stage2_one<S, 0, 0>(3, tc, ce, co, accE, accO, zero);
  SB();
  load_consts<N3>(Z, cst, lane);
  dbn += SS;
  SB();
  stage2_one<S, 0, 1>(3, tc, ce, co, accE, accO, zero);
  SB();
  if (N2 != 0) make_rot(Y.f[0], sgn, rotA);   // row operand of the NEXT step's first stage 1 (Y = operands of slot S+2)
  SB();
  stage2_one<S, 1, 0>(3, tc, ce, co, accE, accO, zero);
  stage2_one<S, 1, 1>(3, tc, ce, co, accE, accO, zero);
  SB();
  tc = tn;
  ce = X.ce;
  co = X.co;
}

// Epilogue of one stage-1 register r of a finished DB group: max over the 120 variants = max_k E + |O| over forward
// and mirror, then d = (1 - max)/2 for 2 queries x 16 entries (lanes 0..31).
__device__ __forceinline__ void epilogue_r(int r, const f32x16 (&accE)[4][2], const f32x16 (&accO)[4][2],
                                           float* __restrict__ dist, int qrow0, int drow0, int m, int n, int lane) {
  f32x16 v = __builtin_elementwise_max(accE[r][0] + __builtin_elementwise_abs(accO[r][0]),
                                       accE[r][1] + __builtin_elementwise_abs(accO[r][1]));
  float mx = v[0];
#pragma unroll
  for (int e = 1; e < 16; e++) mx = fmaxf(mx, v[e]);
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  if (lane < 32) {
    const int qrow = qrow0 + ((lane < 16) ? r : 4 + r);
    const int drow = drow0 + (lane & 15);
    if (qrow < m && drow < n) {
      // The store is issued through inline asm on purpose.  On gfx9-class targets loads and stores share vmcnt but
      // may retire out of order with respect to each other, so once hipcc sees a store in flight it turns every
      // later operand wait into s_waitcnt vmcnt(0) and may drain the 3-slot prefetch pipeline.  Hidden from the waitcnt pass, the counted waits stay: they remain sufficient because
      // loads still return in order among themselves (at worst a wait also covers a few of these stores).  The data
      // is a single dword, so no store-data hazard; the stores complete before the wave ends.
      float* ptr = dist + ((size_t)qrow * n + drow);
      const float val = 0.5f - 0.5f * mx;                                          // processSC.m:30
      asm volatile("global_store_dword %0, %1, off" : : "v"(ptr), "v"(val) : "memory");
    }
  }
}

// Group boundary: the epilogue of the finished group (VALU: 256 accumulator reads + ~260 ops, no matrix work of its
// own) is interleaved with pipeline step 0 of the NEXT group, whose MFMAs do not depend on it: stage 1 of slot 1 runs
// beside epilogue_r(0), and the stage-2 MFMAs of slot 0 that restart accE[r] from zero are issued right after
// epilogue_r(r) has read it.  After the last group the same step runs once on the zero-padded tail of the DB stream.
__device__ __forceinline__ void boundary_step(SlotOps& X, const SlotOps& Y, SlotOps& Z, Tiles& tc, float& ce, float& co,
                                              f32x16 (&accE)[4][2], f32x16 (&accO)[4][2], const f32x16& zero,
                                              const float* __restrict__ la, const float*& dbn,
                                              const float* __restrict__ cst, int lane, float sgn,
                                              float* __restrict__ dist, int qrow0, int drow0, int m, int n,
                                              Rot& rotA) {
  SB();
  load_slot<3>(Z, la, dbn, cst, lane);
  dbn += SS;
  Tiles tn;
  f32x4 t1a, t2a, t1b, t2b;
  stage1_full(X.f[0], rotA, t1a, t2a);
  SB();
  epilogue_r(0, accE, accO, dist, qrow0, drow0, m, n, lane);
  SB();
  stage2_r<0>(0, tc, ce, co, accE, accO, zero);
  const f32x4 Fa = t1a + t2a, Ma = t1a - t2a;
  Rot rotB;
  make_rot(X.f[1], sgn, rotB);
  stage1_full(X.f[1], rotB, t1b, t2b);
  SB();
  epilogue_r(1, accE, accO, dist, qrow0, drow0, m, n, lane);
  SB();
  stage2_r<0>(1, tc, ce, co, accE, accO, zero);
  const f32x4 Fb = t1b + t2b, Mb = t1b - t2b;
  swap_halves(Fa, Fb, tn.fre, tn.fim);
  swap_halves(Ma, Mb, tn.mre, tn.mim);
  SB();
  epilogue_r(2, accE, accO, dist, qrow0, drow0, m, n, lane);
  SB();
  stage2_r<0>(2, tc, ce, co, accE, accO, zero);
  SB();
  epilogue_r(3, accE, accO, dist, qrow0, drow0, m, n, lane);
  SB();
  stage2_r<0>(3, tc, ce, co, accE, accO, zero);
  SB();
  make_rot(Y.f[0], sgn, rotA);
  SB();
  tc = tn;
  ce = X.ce;
  co = X.co;
}

__global__ __launch_bounds__(256, 1) void sc_match_kernel(const float* __restrict__ qpk,  // [2][QG8][31][5][64]
                                                          const float* __restrict__ dpk,  // [2][DG][32][640]
                                                          const float* __restrict__ cst,  // [16][2][64]
                                                          float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                          int m, int n, int QG8, int DG, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int split = b % nsplit;
  b /= nsplit;
  const int ch = b & 1, qg32 = b >> 1;
  const int g0 = (int)((long long)DG * split / nsplit), g1 = (int)((long long)DG * (split + 1) / nsplit);

  {  // the 4 query groups of this workgroup -> LDS (linear copy; the packed image IS the LDS image)
    const f32x4* src = reinterpret_cast<const f32x4*>(qpk + ((size_t)ch * QG8 + (size_t)qg32 * 4) * SC_QIMG);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    for (int i = tid; i < SC_QIMG; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  if (g0 >= g1) return;

  const float* la = lds + w * SC_QIMG + lane;
  const float* db = dpk + ((size_t)ch * DG + g0) * SC_DIMG;   // slot stream: SS floats per slot, contiguous over groups
  float* dist = ch ? dist_i : dist_p;
  const float sgn = ((lane & 15) >= 8) ? -1.0f : 1.0f;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // Software pipeline over the flat slot sequence (16 slots per DB group).  While stage 2 of slot i runs: operands of
  // slot i+3 are being requested, those of i+2 are in flight, stage 1 + swap of slot i+1 are issued.  Three operand
  // buffers rotate with period 3; the 16 slots of a group are fully unrolled (compile-time slot kinds) and the only
  // register copies are two buffer renames per group.  The packed DB buffer has a readable tail of >= 3 slots.
  SlotOps oA, oB, oC;   // at the top of a group: oA = operands(slot 1), oB = operands(slot 2) in flight, oC free
  Tiles tc;             // tiles of the current slot
  float ce, co = 0.f;   // constants of the current slot
  {
    f32x4 t1a, t1b;
    load_slot<0>(oC, la, db, cst, lane);
    load_slot<1>(oA, la, db + SS, cst, lane);
    load_slot<2>(oB, la, db + 2 * SS, cst, lane);
    stage1_real(oC.f[0], t1a);
    stage1_real(oC.f[1], t1b);
    swap_halves(t1a, t1b, tc.fre, tc.fim);
    ce = oC.ce;
  }
  const float* dbn = db + 3 * SS;                              // next slot to request

  f32x16 accE[4][2], accO[4][2];
#define STEP(S, X, Y, Z) slot_step<S>(X, Y, Z, tc, ce, co, accE, accO, zero, la, dbn, cst, lane, sgn, rotA)
  Rot rotA;
  make_rot(oA.f[0], sgn, rotA);
  STEP(0, oA, oB, oC);             // first group: nothing to overlap with
  for (int g = g0; g < g1; g++) {
                          STEP(1, oB, oC, oA);  STEP(2, oC, oA, oB);
    STEP(3, oA, oB, oC);  STEP(4, oB, oC, oA);  STEP(5, oC, oA, oB);
    STEP(6, oA, oB, oC);  STEP(7, oB, oC, oA);  STEP(8, oC, oA, oB);
    STEP(9, oA, oB, oC);  STEP(10, oB, oC, oA); STEP(11, oC, oA, oB);
    STEP(12, oA, oB, oC); STEP(13, oB, oC, oA); STEP(14, oC, oA, oB);
    STEP(15, oA, oB, oC);
    // 16 slots advance the period-3 rotation by one: rename so that the next group starts in the same roles
    oA = oB;   // operands(slot 1 of the next group)
    oB = oC;   // operands(slot 2 of the next group), possibly still in flight
    // epilogue of group g fused with step 0 of group g+1
    boundary_step(oA, oB, oC, tc, ce, co, accE, accO, zero, la, dbn, cst, lane, sgn, dist, qg32 * 32 + w * 8, g * 16, m, n,
                  rotA);
  }
#undef STEP
}

}  // namespace

size_t sc_match_lds_bytes() { return (size_t)4 * SC_QIMG * sizeof(float); }

void launch_sc_match(hipStream_t st, const float* qpk, int m, const float* dpk, int n, const float* cst,
                     float* d_p, float* d_i, int nsplit_override) {
  if (m <= 0 || n <= 0) return;
  const int QG8 = sc_qgroups8(m), DG = sc_dgroups(n);
  const int base = (QG8 / 4) * 2;
  int nsplit = (1024 + base - 1) / base;           // >= ~4 workgroups per CU in total, for tail balance
  if (nsplit > DG / 8) nsplit = DG / 8;            // keep >= 8 DB groups (128 entries) per workgroup
  if (nsplit < 1) nsplit = 1;
  if (nsplit_override > 0) nsplit = nsplit_override < DG ? nsplit_override : DG;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_match_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc_match_lds_bytes());
  hipLaunchKernelGGL(sc_match_kernel, dim3(base * nsplit), dim3(256), sc_match_lds_bytes(), st, qpk, dpk, cst, d_p,
                     d_i, m, n, QG8, DG, nsplit);
}

}  // namespace pr
