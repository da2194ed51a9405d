// sc_match_r.hip — EXPERIMENT, not the default (PR_SC_KERNEL=r selects it for m > 8).  The split-f16 SC matcher (processSC.m:22-33; mathematics, packed images and stage 1 of sc_match_h.hip)
// as ONE ROLLING software pipeline per (8 queries x 16 entries) unit instead of the phases stage 1 / stage 2 / epilogue.
//
// sc_match_h.hip runs, per half of the frequencies, a VALU-bound phase (stage 1: 16x16x32 MFMAs that cover a third of the
// split / pack work issued beside them) and then an MFMA-bound phase (stage 2: 48 back-to-back 32x32x16 MFMAs with hardly any
// VALU to put beside them): the two times add up (PMC: VALU issue 7.2 k + matrix pipe 6.0 k of the 16.5 k cycles per unit).
// Here stage 2 works by QUARTER (8 frequencies = 4 pairs) with the operand form of sc_match_p.hip - the hi and lo halves of
// the intermediate concatenated along K:  B = (S_hi | S_lo) by lane half (one v_permlane32_swap of a packed register's hi with
// its lo gives the Re and the Im operand),  A = (C_hi | C_hi) for hi*hi + hi*lo and A = (C_lo | 0) for lo*hi - so the packed
// operands of a quarter are 64 registers, two quarters fit, and the 32 stage-2 MFMAs of quarter q-1 are issued BETWEEN the
// stage-1 MFMAs of quarter q: pair P of a unit (two frequencies, 12 stage-1 MFMAs) hosts the 8 stage-2 MFMAs of stage-1
// register (P - 5) & 3 of quarter (P - 5) >> 2, the permlane swaps of the next register, the combination of its own
// frequencies and the split / pack of pair P - 1.  Every pair is then ~450 matrix cycles with ~60 VALU issues beside them.
// 128 stage-2 MFMAs per unit instead of 96 (the zero half of the lo*hi product) is the price of the 8-frequency granularity.
//
// Measured (MI355X, 4096 x 100k, tools/experiments/README.md): parity-green, 47.3 ms per launch = the 46.1 ms of sc_match_h.hip.
// The interleaving buys nothing: leaving out 7 of every 8 stage-2 MFMAs saves 9.5 ms = exactly their 32 cycles each - on this
// in-order wave the 32x32x16 MFMAs cost their full pipe time wherever they stand; with the VALU pieces AFTER each of them
// (in its shadow) instead of before, the launch takes 50.1 ms.
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

struct AOps { u32x4 h, l, rh, rl; };          // query row operands: Q hi, Q lo, and the same with Re/Im rows exchanged
struct BOps { u32x4 reh, rel, imh, iml; };    // DB column operands: Re hi, Re lo, Im hi, Im lo

// one operand tile (4 registers) per call, so that every request can be placed in its own MFMA gap
enum { A_H = 0, A_L = 1, A_RH = 2, A_RL = 3 };
enum { B_REH = 0, B_REL = 1, B_IMH = 2, B_IML = 3 };
// nat / rot: 32-bit LDS byte addresses of this lane's 16 B in the block of frequency (P & ~1); the odd frequency of the
// pair and the lo tile are immediate offsets of the ds_read2_b64 (8-bit, in units of 8 B: 1288 + 40 + 8 < 2048)
typedef const u32x4_a8 __attribute__((address_space(3))) * lds_tile_p;
template <int P, int T>
__device__ __forceinline__ void load_a(AOps& a, unsigned nat, unsigned rot) {
  const unsigned addr = ((T & 2) ? rot : nat) + (P & 1) * SCH_QBLK + (T & 1) * 40;
  const u32x4 v = *reinterpret_cast<lds_tile_p>(addr);
  if (T == A_H) a.h = v; else if (T == A_L) a.l = v; else if (T == A_RH) a.rh = v; else a.rl = v;
}
template <int P, int T>
__device__ __forceinline__ void load_b(BOps& b, __amdgpu_buffer_rsrc_t rs, int voff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, P * SCH_DFREQ + T * SCH_DTILE, 0);   // frequency and tile in the scalar offset: ONE lane-offset register
  if (T == B_REH) b.reh = v; else if (T == B_REL) b.rel = v; else if (T == B_IMH) b.imh = v; else b.iml = v;
}

// Stage-1 MFMAs in VGPR form, ONE instruction per asm statement so that VALU work can be placed between them by hand
// (the wave issues in order: back-to-back MFMAs would block it).  The 256 stage-2 accumulators own the AccVGPR half and
// hipcc picks one MFMA register form per function, hence asm.  hipcc pads nothing around asm (cdna_hip_programming.md
// §5.7): an accumulate chain on the same vDst needs no wait states; every VALU reader of t1/t2 below sits at least two
// MFMAs + their fillers behind the last write, except the one after DRAIN().
#define MF0(d, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b))
#define MFA(d, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b))
#define DRAIN() asm volatile("s_nop 9")
#define SB() __builtin_amdgcn_sched_barrier(0)
#ifdef PR_SCH_TIMING
#define TICK(i) { SB(); unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)); tacc[i] += _t - tprev; tprev = _t; SB(); }
#else
#define TICK(i)
#endif

// (hi, lo) split of two fp32 values into packed f16 pairs: hi = f16(x) (v_cvt_pk_f16_f32), lo = f16(x - hi) with the
// residual formed exactly in fp32 by v_fma_mix_f32 (f16 operand x -1 + f32 operand) - the mixlo/mixhi forms that write a
// 16-bit half directly cost ~2x the issue time of a full-register VALU op on gfx950 (tools/ubench/valu_rate.hip).
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

// Packed results of one quarter (4 pairs = 8 frequencies) for the 4 stage-1 registers r: element j of [r] holds pair j as
// (Re | Im) by lane half until the swap, then hX = (Re_hi | Re_lo), lX = (Im_hi | Im_lo): the B operands of stage 2.
struct Qtr { u32x4 hF[4], lF[4], hM[4], lM[4]; };

__device__ __forceinline__ void swap32(u32x4& a, u32x4& b, int e) {   // lanes 32-63 of a[e] <-> lanes 0-31 of b[e]
  const u32x2 v = __builtin_amdgcn_permlane32_swap(a[e], b[e], false, false);
  a[e] = v[0];
  b[e] = v[1];
}

// pair J of the quarter, register R: split (hi, lo) of the two frequencies' values -> element J
template <int J, int R>
__device__ __forceinline__ void pack_F(Qtr& s, const f32x4& Fa, const f32x4& Fb) {
  unsigned h, l;
  split2(Fa[R], Fb[R], h, l);
  s.hF[R][J] = h; s.lF[R][J] = l;
}
template <int J, int R>
__device__ __forceinline__ void pack_M(Qtr& s, const f32x4& Ma, const f32x4& Mb) {
  unsigned h, l;
  split2(Ma[R], Mb[R], h, l);
  s.hM[R][J] = h; s.lM[R][J] = l;
}
template <int R>
__device__ __forceinline__ void swap_r(Qtr& s, int e0, int e1) {   // elements e0..e1-1 of the 4 operands of register R
  for (int e = e0; e < e1; e++) {
    swap32(s.hF[R], s.lF[R], e);
    swap32(s.hM[R], s.lM[R], e);
  }
}

__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// A operands of one quarter: [E | O][hh+hl | lh] tiles of the sc_match_p.hip table ([quarter][4 tiles][64 lanes] x 16 B)
struct Cst { u32x4 ehh, elh, ohh, olh; };
template <int QD>
__device__ __forceinline__ void load_cst(Cst& c, __amdgpu_buffer_rsrc_t rc, int lane16) {
  c.ehh = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, (QD * 4 + 0) * 1024, 0);
  c.elh = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, (QD * 4 + 1) * 1024, 0);
  c.ohh = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, (QD * 4 + 2) * 1024, 0);
  c.olh = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, (QD * 4 + 3) * 1024, 0);
}

// One of the 8 stage-2 MFMAs of stage-1 register R for one quarter: I = 0..3 (C_hi | C_hi) tiles: E forward, O forward,
// E mirror, O mirror (they start the chains in the first quarter); I = 4..7 the same with the (C_lo | 0) tiles.
template <bool FIRST, int R, int I>
__device__ __forceinline__ void stage2_one(const Qtr& s, const Cst& c, f32x16 (&accE)[4][2], f32x16 (&accO)[4][2],
                                           const f32x16& zero) {
  constexpr int V = (I >> 1) & 1, PART = I & 1, T = I >> 2;
  const u32x4& ca = PART ? (T ? c.olh : c.ohh) : (T ? c.elh : c.ehh);
  const u32x4& op = PART ? (V ? s.lM[R] : s.lF[R]) : (V ? s.hM[R] : s.hF[R]);
  f32x16& acc = PART ? accO[R][V] : accE[R][V];
  acc = mfma32(ca, op, (FIRST && T == 0) ? zero : acc);
}

// epilogue piece: shift rows e of register R -> running max over E + |O| of forward and mirror
template <int R>
__device__ __forceinline__ void ep_elem(float& mx, const f32x16 (&accE)[4][2], const f32x16 (&accO)[4][2], int e) {
  const float vf = accE[R][0][e] + __builtin_fabsf(accO[R][0][e]);
  const float vm = accE[R][1][e] + __builtin_fabsf(accO[R][1][e]);
  mx = fmaxf(fmaxf(mx, vf), vm);
}
// 2 queries x 16 entries (lanes 0..31): d = (1 - max)/2 with the 2^-25 operand scaling folded in   (processSC.m:30).
// Branch-free (a buffer store whose invalid lanes are out of range), so that the whole group body stays ONE basic block
// and the hand-placed order survives the compiler's sinking passes.
template <int R>
__device__ __forceinline__ void ep_store(float mx, __amdgpu_buffer_rsrc_t rd, int st_off) {
  const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
  mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));     // max over the two lane halves (shift rows +0..3 | +4..7)
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(mx, -0x1p-26f, 0.5f)), rd, st_off, 0, 0);   // plain store: the nt hint cost 0.7 % and 28 % more HBM write traffic (partial lines bypass the L2 merge)
}

// F = T1 + s T2, M = T1 - s T2 for registers r0, r0+1: two v_pk_fma_f32
#define FM2(F, M, t1, t2, r0)                                                        \
  {                                                                                  \
    const f32x2 _a = {t1[r0], t1[r0 + 1]}, _b = {t2[r0], t2[r0 + 1]};                \
    const f32x2 _f = __builtin_elementwise_fma(_b, sg2, _a), _m = __builtin_elementwise_fma(_b, -sg2, _a); \
    F[r0] = _f[0]; F[r0 + 1] = _f[1]; M[r0] = _m[0]; M[r0 + 1] = _m[1];             \
  }

__global__ __launch_bounds__(256, 1) void sc_match_r_kernel(const char* __restrict__ qpk,   // [2][QG32][4][31][1288 B]
                                                            const char* __restrict__ dpk,   // [2][DG][31][4][768 B] + one zero group
                                                            const u32x4* __restrict__ cst,  // [4][4][64] x 16 B
                                                            float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                            int m, int n, int QG8, int DG, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware mapping as sc_match_h.hip
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ch = xcd & 1;
  const int range = (xcd >> 1) + 4 * (idx % nsplit), qg32 = idx / nsplit;
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
  constexpr int GS = 1;
  const int wq = w;                                                 // this wave's query group inside the workgroup's image
  const unsigned nat0 = lds0 + wq * SCH_QIMG + row * 80 + (row >= 8 ? 8 : 0) + kg * 16;
  const unsigned rot0 = lds0 + wq * SCH_QIMG + (row ^ 8) * 80 + (row >= 8 ? 0 : 8) + kg * 16;
  const int voff = (lane < 48) ? lane * 16 : (int)0x80000000;     // lanes 48-63: out of range -> zeros (K = 24..31)
  const float sg = (lane < 32) ? 1.0f : -1.0f;
  const f32x2 sg2 = {sg, sg};
  float* dist = ch ? dist_i : dist_p;
  const char* dbase = dpk + ((size_t)ch * DG) * SCH_DIMG;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int qrow0 = qg32 * 32 + wq * 8;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      dist + (size_t)qrow0 * n, 0, (qrow0 < m ? (m - qrow0 < 8 ? m - qrow0 : 8) : 0) * n * 4, 0x00020000);
  const int pf_slot = (qg32 & 31) * 4 + w;                                  // 0..127
  unsigned pf_sink = 0;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(cst), 0, 16384, 0x00020000);

  // Stage 1 as in sc_match_h.hip: per operand TILE software pipeline (DB tiles requested 8-12 MFMA slots ahead, query tiles 4),
  // ONE instruction per asm statement / sched_barrier group so that the order below is the issue order of the in-order wave.
  // Per pair P (frequencies 2P, 2P + 1; slots a0..a5, b0..b5 = the gaps after its 12 stage-1 MFMAs):
  //   a1, a2   F / M combination of the previous pair's second frequency
  //   a3 .. b4 split / pack of pair P - 1, registers 0..3, into element (P - 1) & 3 of operand set ((P - 1) >> 2) & 1
  //   b5       F / M combination of this pair's first frequency
  //   a0 a1 a3 a4 b0 b1 b3 b4   the 8 stage-2 MFMAs of item P - 5  (item s = quarter s >> 2, register s & 3)
  //   a5 b2 b5                  the permlane swaps of item P - 4 (its last pack is the one in a3 / a4 of this pair at the latest)
  // The constants of quarter k are requested before pair 4k + 2 and first used in pair 4k + 5; the items 11..15 follow the last
  // pair (tail), with the first requests of the next unit in their gaps, then the epilogue.
  AOps At[4];
  BOps Bt[4];
  __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)g0 * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
  load_b<0, B_REH>(Bt[0], rs, voff); load_b<0, B_IMH>(Bt[0], rs, voff); load_b<0, B_REL>(Bt[0], rs, voff); load_b<0, B_IML>(Bt[0], rs, voff);
  load_a<0, A_H>(At[0], nat0, rot0); load_a<0, A_RH>(At[0], nat0, rot0); load_a<0, A_L>(At[0], nat0, rot0); load_a<0, A_RL>(At[0], nat0, rot0);
  load_b<1, B_REH>(Bt[1], rs, voff); load_b<1, B_IMH>(Bt[1], rs, voff); load_b<1, B_REL>(Bt[1], rs, voff);
  load_a<1, A_H>(At[1], nat0, rot0); load_a<1, A_RH>(At[1], nat0, rot0);
#ifdef PR_SCH_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev));
#endif
  unsigned nbase = nat0, rbase = rot0;       // carried through the running tile addresses: nothing address-like stays live (= spilled) across a unit
  for (int g = g0; g < g1; g += GS) {
    const __amdgpu_buffer_rsrc_t rsn =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)(g + GS) * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
    f32x16 accE[4][2], accO[4][2];
    Qtr Sx[2];
    Cst Cx[2];
    f32x4 Fa, Ma, Fb, Mb, t1a, t2a, t1b, t2b;
    unsigned ncur, rcur, nnxt = nbase, rnxt = rbase;
    TICK(7)
#define LDB(Q, T) { if ((Q) < SC_NF) load_b<((Q) < SC_NF ? (Q) : 0), T>(Bt[(Q) & 3], rs, voff); }
#define LDA(P, Q, T) { if ((Q) < SC_NF) { if (((Q) >> 1) == ((P) >> 1)) load_a<((Q) < SC_NF ? (Q) : 0), T>(At[(Q) & 3], ncur, rcur); \
                                        else load_a<((Q) < SC_NF ? (Q) : 0), T>(At[(Q) & 3], nnxt, rnxt); } }
#define FREQ(P, t1, t2, W0, W1, W2, W3, W4, W5)                                                   \
  {                                                                                               \
    SB(); MF0(t1, At[(P) & 3].h, Bt[(P) & 3].reh);  SB(); LDB((P) + 1, B_IML); LDA(P, (P) + 1, A_L); W0;   \
    SB(); MF0(t2, At[(P) & 3].rh, Bt[(P) & 3].imh); SB(); LDB((P) + 2, B_REH); W1;                \
    SB(); MFA(t1, At[(P) & 3].l, Bt[(P) & 3].reh);  SB(); LDB((P) + 2, B_IMH); LDA(P, (P) + 1, A_RL); W2;  \
    SB(); MFA(t2, At[(P) & 3].rl, Bt[(P) & 3].imh); SB(); LDA(P, (P) + 2, A_H); W3;               \
    SB(); MFA(t1, At[(P) & 3].h, Bt[(P) & 3].rel);  SB(); LDB((P) + 2, B_REL); W4;                \
    SB(); MFA(t2, At[(P) & 3].rh, Bt[(P) & 3].iml); SB(); LDA(P, (P) + 2, A_RH); W5;              \
    SB();                                                                                         \
  }
// pack of pair PP (0..15), register R
#define PKF(PP, R) pack_F<(PP) & 3, R>(Sx[((PP) >> 2) & 1], Fa, Fb)
#define PKM(PP, R) pack_M<(PP) & 3, R>(Sx[((PP) >> 2) & 1], Ma, Mb)
#define PK(PP, R) { PKF(PP, R); PKM(PP, R); }
// stage-2 MFMA I of item S, swaps of item S (elements e0..e1-1); nothing for items outside 0..15
#define X(S, I) { if ((S) >= 0) { SB(); stage2_one<(((S) >> 2) == 0), (S) & 3, I>(Sx[((S) >> 2) & 1], Cx[((S) >> 2) & 1], accE, accO, zero); SB(); } }
#define SWP(S, e0, e1) { if ((S) >= 0 && (S) < 16) swap_r<(S) & 3>(Sx[((S) >> 2) & 1], e0, e1); }
#define LDC(K) { SB(); load_cst<K>(Cx[(K) & 1], rc, lane * 16); SB(); }
#define NONE ((void)0)
#define ADV() { ncur = nnxt; rcur = rnxt; nnxt = ncur + 2 * SCH_QBLK; rnxt = rcur + 2 * SCH_QBLK; asm("" : "+v"(nnxt)); asm("" : "+v"(rnxt)); }
#define FMA_ALL(F, M, t1, t2) { FM2(F, M, t1, t2, 0); FM2(F, M, t1, t2, 2); }
#define PAIR0()                                                                                   \
  ADV()                                                                                           \
  FREQ(0, t1a, t2a, NONE, NONE, NONE, NONE, NONE, NONE)                                           \
  FREQ(1, t1b, t2b, NONE, NONE, NONE, NONE, NONE, FMA_ALL(Fa, Ma, t1a, t2a))
#define PAIRR(P)                                                                                  \
  ADV()                                                                                           \
  FREQ(2 * (P), t1a, t2a,                                                                         \
       X((P) - 5, 0),                                                                             \
       { FM2(Fb, Mb, t1b, t2b, 0); X((P) - 5, 1); },                                              \
       FM2(Fb, Mb, t1b, t2b, 2),                                                                  \
       { PKF((P) - 1, 0); X((P) - 5, 2); },                                                       \
       { PKM((P) - 1, 0); X((P) - 5, 3); },                                                       \
       { PKF((P) - 1, 1); SWP((P) - 4, 0, 1); })                                                  \
  FREQ(2 * (P) + 1, t1b, t2b,                                                                     \
       { PKM((P) - 1, 1); X((P) - 5, 4); },                                                       \
       { PKF((P) - 1, 2); X((P) - 5, 5); },                                                       \
       { PKM((P) - 1, 2); SWP((P) - 4, 1, 2); },                                                  \
       { PKF((P) - 1, 3); X((P) - 5, 6); },                                                       \
       { PKM((P) - 1, 3); X((P) - 5, 7); },                                                       \
       { FMA_ALL(Fa, Ma, t1a, t2a); SWP((P) - 4, 2, 4); })

    PAIR0() PAIRR(1) LDC(0) PAIRR(2) PAIRR(3)
    TICK(0)
    PAIRR(4) PAIRR(5) LDC(1) PAIRR(6) PAIRR(7)
    TICK(1)
    PAIRR(8) PAIRR(9) LDC(2) PAIRR(10) PAIRR(11)
    TICK(2)
    PAIRR(12) PAIRR(13) LDC(3) PAIRR(14)
    // pair 15 = frequency 30 alone (31 is the zero pad): item 10, swaps of item 11, pack of pair 14
    ADV()
    FREQ(30, t1a, t2a,
         X(10, 0),
         { FM2(Fb, Mb, t1b, t2b, 0); X(10, 1); },
         { FM2(Fb, Mb, t1b, t2b, 2); X(10, 2); },
         { PK(14, 0); X(10, 3); },
         { PK(14, 1); X(10, 4); },
         { PK(14, 2); X(10, 5); })
    PK(14, 3)
    X(10, 6)
    SWP(11, 0, 2)
    X(10, 7)
    SWP(11, 2, 4)
    TICK(3)
    DRAIN();
    SB();
    FM2(Fa, Ma, t1a, t2a, 0); FM2(Fa, Ma, t1a, t2a, 2);
    Fb = f32x4{0.f, 0.f, 0.f, 0.f}; Mb = Fb;
    {  // L2 prefetch of group g + 2 for the whole XCD (as sc_match_h.hip)
      asm volatile("" : : "v"(pf_sink));
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(dbase + (size_t)(g + 2 * GS) * SCH_DIMG), 0, (g + 2 * GS < DG) ? SCH_DIMG : 0, 0x00020000);
      int lp = lane;
      asm volatile("" : "+v"(lp));                 // recomputed here, not kept (or spilled) across the unit
      const int pf_off = (lp < 6) ? (pf_slot * 6 + lp) * 128 : (int)0x80000000;   // lines past the group are out of range
      pf_sink = __builtin_amdgcn_raw_buffer_load_b32(rp, pf_off, 0, 0);
    }
    SB();
    nbase = nnxt - 16 * 2 * SCH_QBLK; rbase = rnxt - 16 * 2 * SCH_QBLK;   // 16 pair advances back: the image's first block
    // ---------------------------------------------------------------- tail: items 11..15, pack of pair 15, next unit's first requests
#define NB(P, T) load_b<P, T>(Bt[P], rsn, voff)
#define NA(P, T) load_a<P, T>(At[P], nbase, rbase)
    X(11, 0) PK(15, 0) X(11, 1) PK(15, 1) X(11, 2) PK(15, 2) X(11, 3) PK(15, 3)
    X(11, 4) SWP(12, 0, 2) X(11, 5) SWP(12, 2, 4) X(11, 6) X(11, 7)
    X(12, 0) SWP(13, 0, 1) X(12, 1) SWP(13, 1, 2) X(12, 2) SWP(13, 2, 3) X(12, 3) SWP(13, 3, 4) X(12, 4) X(12, 5) X(12, 6) X(12, 7)
    X(13, 0) SWP(14, 0, 1) X(13, 1) SWP(14, 1, 2) X(13, 2) SWP(14, 2, 3) X(13, 3) SWP(14, 3, 4) X(13, 4) X(13, 5) X(13, 6) X(13, 7)
    X(14, 0) SWP(15, 0, 1) X(14, 1) SWP(15, 1, 2) X(14, 2) SWP(15, 2, 3) X(14, 3) SWP(15, 3, 4)
    X(14, 4) NB(0, B_REH); X(14, 5) NB(0, B_IMH); X(14, 6) NB(0, B_REL); X(14, 7) NB(0, B_IML);
    X(15, 0) NB(1, B_REH); X(15, 1) NB(1, B_IMH); X(15, 2) NB(1, B_REL); X(15, 3) NA(0, A_H);
    X(15, 4) NA(0, A_RH); X(15, 5) NA(0, A_L); X(15, 6) { NA(0, A_RL); NA(1, A_H); } X(15, 7) NA(1, A_RH);
    SB();
    TICK(4)
#define EPILOGUE(R)                                                                               \
  { const int st_base = (le < 32 && g * 16 + (le & 15) < n) ? st_lane : (int)0x80000000;          \
    float mx = -__builtin_inff();                                                                 \
    _Pragma("unroll") for (int e = 0; e < 16; e++) ep_elem<R>(mx, accE, accO, e);                 \
    ep_store<R>(mx, rd, st_base + (R) * 4 * n + g * 64); }
    int le = lane;
    asm volatile("" : "+v"(le));                   // the store addresses are recomputed per unit, not kept (or spilled) across it
    const int st_lane = ((le & 16) ? 4 * n : 0) * 4 + (le & 15) * 4;
    EPILOGUE(0) EPILOGUE(1) EPILOGUE(2) EPILOGUE(3)
    TICK(5)
    rs = rsn;
  }
#ifdef PR_SCH_TIMING
  if (blockIdx.x == 8 * 40 && tid == 0)   // one wave somewhere in the middle of the grid; written over the first distances
    for (int i = 0; i < 8; i++) reinterpret_cast<unsigned long long*>(dist_p)[i] = tacc[i] / (unsigned long long)((g1 - g0) / GS);
#endif
}

}  // namespace

size_t sc_match_r_lds_bytes() { return (size_t)4 * SCH_QIMG + 64; }

void launch_sc_match_r(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p,
                       float* d_i, int nsplit_override) {
  if (m <= 0 || n <= 0) return;
  const int QG8 = sc_qgroups8(m), DG = sc_dgroups(n);
  const int QG32 = QG8 / 4;
  int nsplit = (128 + QG32 - 1) / QG32;
  if (nsplit > DG / 32) nsplit = DG / 32;
  if (nsplit < 1) nsplit = 1;
  if (nsplit_override > 0) nsplit = nsplit_override * 4 <= DG ? nsplit_override : (DG >= 4 ? DG / 4 : 1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_match_r_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sc_match_r_lds_bytes());
  hipLaunchKernelGGL(sc_match_r_kernel, dim3(8 * QG32 * nsplit), dim3(256), sc_match_r_lds_bytes(), st,
                     static_cast<const char*>(qpk), static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i,
                     m, n, QG8, DG, nsplit);
}

}  // namespace pr
