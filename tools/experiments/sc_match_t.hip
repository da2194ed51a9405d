// sc_match_t.hip — EXPERIMENT (PR_SC_KERNEL=t selects it for m > 8): the split-f16 SC matcher (processSC.m:22-33; mathematics, packed
// images and the hand-placed stage 1 of sc_match_h.hip) with TRANSIENT stage-2 accumulators.
//
// sc_match_h.hip accumulates 32-shift x 32-pair tiles over two halves of the frequencies: 256 accumulators live for the whole unit, an
// MFMA-only stage-2 phase and a VALU-only epilogue that has to pull every accumulator out of the AccVGPRs.  Here stage 2 uses
// 16x16x32 MFMAs whose K = 32 slots are ALL frequencies (31 + pad): a (16 shifts x 16 pairs) tile is complete after the three MFMAs of
// the split products, is reduced (E + |O|, max) and dropped.  What has to live through stage 1 instead are the packed intermediates of
// the whole unit (256 registers), as MFMA B operands: lane group G of a B operand holds frequencies 8G..8G+7, so the packed values of
// the four frequency octets are transposed over the four 16-lane groups (v_permlane32_swap + v_permlane16_swap, four swaps per four
// registers).  Stage 1 therefore walks the frequencies octet-interleaved - pairs (0,1) (8,9) (16,17) (24,25) (2,3) ... - so that a
// transpose has its four partners after every four pairs.
#include "kernels.hpp"

#ifndef T_PARTS
#define T_PARTS 2
#endif
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
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, P * SCH_DFREQ + T * SCH_DTILE, 0);   // frequency and tile in the scalar offset: ONE lane-offset register (voff + T * 768 cost three more, spilled)
  if (T == B_REH) b.reh = v; else if (T == B_REL) b.rel = v; else if (T == B_IMH) b.imh = v; else b.iml = v;
}

// Stage-1 MFMAs in VGPR form, ONE instruction per asm statement so that VALU work can be placed between them by hand
// (the wave issues in order: back-to-back MFMAs would block it).  The 256 stage-2 accumulators own the AccVGPR half and
// hipcc picks one MFMA register form per function, hence asm.  hipcc pads nothing around asm (cdna_hip_programming.md
// §5.7): an accumulate chain on the same vDst needs no wait states; every VALU reader of t1/t2 below sits at least two
// MFMAs + their fillers behind the last write, except the one after DRAIN().
// (s_nop 1 in front: in this kernel hipcc parks stage-1 operands in AccVGPRs and may restore one right before the asm statement, which it
//  does not pad - tools/audit_asm_hazards.py)
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


// sequence index 0..31 -> frequency: step j = i >> 3 (register j of the B operands), octet o = (i & 7) >> 1, parity i & 1
__host__ __device__ constexpr int seqf(int i) { return 8 * ((i & 7) >> 1) + 2 * (i >> 3) + (i & 1); }

// packed stage-2 B operands of the unit: [forward | mirror][stage-1 register r][lane-group block][register j] - before the transpose of a
// step the third index is the frequency octet (and the lane group the row type), after it the row type (and the lane group the octet)
struct Packed { u32x4 h[2][4][4], l[2][4][4]; };

__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {   // lanes 32-63 of a <-> lanes 0-31 of b
  const u32x2 v = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = v[0]; b = v[1];
}
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {   // 16-lane blocks 1, 3 of a <-> blocks 0, 2 of b
  const u32x2 v = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = v[0]; b = v[1];
}
// 4 x 4 transpose of 16-lane blocks over four registers: z[o] block g  ->  z[g] block o
__device__ __forceinline__ void transpose4(unsigned& z0, unsigned& z1, unsigned& z2, unsigned& z3) {
  swap32(z0, z2); swap32(z1, z3);
  swap16(z0, z1); swap16(z2, z3);
}
// the packed values of one step (four pairs) before their transpose: [forward | mirror][register r][octet]
struct Step { unsigned h[2][4][4], l[2][4][4]; };
template <int X, int R>
__device__ __forceinline__ void transpose_xr(Packed& p, Step& z, int j) {
  transpose4(z.h[X][R][0], z.h[X][R][1], z.h[X][R][2], z.h[X][R][3]);
  p.h[X][R][0][j] = z.h[X][R][0]; p.h[X][R][1][j] = z.h[X][R][1]; p.h[X][R][2][j] = z.h[X][R][2]; p.h[X][R][3][j] = z.h[X][R][3];
  transpose4(z.l[X][R][0], z.l[X][R][1], z.l[X][R][2], z.l[X][R][3]);
  p.l[X][R][0][j] = z.l[X][R][0]; p.l[X][R][1][j] = z.l[X][R][1]; p.l[X][R][2][j] = z.l[X][R][2]; p.l[X][R][3][j] = z.l[X][R][3];
}

// stage-2 MFMA: constants in ArchVGPRs, the packed B operand in AccVGPRs (the constraint pins the finished operands there: hipcc would
// otherwise shuttle them between the register files), result tile in ArchVGPRs
__device__ __forceinline__ void mfma16z(f32x4& d, const u32x4& a, const u32x4& b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma16a(f32x4& d, const u32x4& a, const u32x4& b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}

// A operands of stage 2: [E | O][shift rows 0..15 | 16..31][hi | lo] tiles, lane = 16 (K group) + row
struct CstT { u32x4 t[2][2][2]; };

// stage-2 group k = (R * 2 + H) * 2 + X: the four (16 shift rows x 16 pairs) tiles E / O x shift rows 0-15 / 16-31 of (forward | mirror) X,
// stage-1 register R, query half H (rows q = 4 H + R): twelve MFMAs (three split products per tile); every B operand serves both row blocks
struct Tiles { f32x4 e0, o0, e1, o1; };
template <int K>
__device__ __forceinline__ void s2_issue(const Packed& p, const CstT& c, Tiles& t) {
  constexpr int X = K & 1, H = (K >> 1) & 1, R = K >> 2;
  mfma16z(t.e0, c.t[0][0][0], p.h[X][R][H]);
  mfma16z(t.e1, c.t[0][1][0], p.h[X][R][H]);
  mfma16z(t.o0, c.t[1][0][0], p.h[X][R][H + 2]);
  mfma16z(t.o1, c.t[1][1][0], p.h[X][R][H + 2]);
  mfma16a(t.e0, c.t[0][0][1], p.h[X][R][H]);
  mfma16a(t.e1, c.t[0][1][1], p.h[X][R][H]);
  mfma16a(t.o0, c.t[1][0][1], p.h[X][R][H + 2]);
  mfma16a(t.o1, c.t[1][1][1], p.h[X][R][H + 2]);
  mfma16a(t.e0, c.t[0][0][0], p.l[X][R][H]);
  mfma16a(t.e1, c.t[0][1][0], p.l[X][R][H]);
  mfma16a(t.o0, c.t[1][0][0], p.l[X][R][H + 2]);
  mfma16a(t.o1, c.t[1][1][0], p.l[X][R][H + 2]);
}
__device__ __forceinline__ float s2_reduce(float mx, const Tiles& t) {
#pragma unroll
  for (int i = 0; i < 4; i++) mx = fmaxf(fmaxf(mx, t.e0[i] + __builtin_fabsf(t.o0[i])), t.e1[i] + __builtin_fabsf(t.o1[i]));
  return mx;
}

// F = T1 + s T2, M = T1 - s T2 for registers r0, r0+1: two v_pk_fma_f32
#define FM2(F, M, t1, t2, r0)                                                        \
  {                                                                                  \
    const f32x2 _a = {t1[r0], t1[r0 + 1]}, _b = {t2[r0], t2[r0 + 1]};                \
    const f32x2 _f = __builtin_elementwise_fma(_b, sg2, _a), _m = __builtin_elementwise_fma(_b, -sg2, _a); \
    F[r0] = _f[0]; F[r0 + 1] = _f[1]; M[r0] = _m[0]; M[r0 + 1] = _m[1];             \
  }

__global__ __launch_bounds__(256, 1) void sc_match_t_kernel(const char* __restrict__ qpk,   // [2][QG32][4][31][1288 B]
                                                            const char* __restrict__ dpk,   // [2][DG][31][4][768 B] + zero groups
                                                            const u32x4* __restrict__ cst,  // [E|O][rows 0-15|16-31][hi|lo][64] x 16 B
                                                            float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                            int m, int n, int QG8, int DG, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const unsigned nat0 = lds0 + w * SCH_QIMG + row * 80 + (row >= 8 ? 8 : 0) + kg * 16;
  const unsigned rot0 = lds0 + w * SCH_QIMG + (row ^ 8) * 80 + (row >= 8 ? 0 : 8) + kg * 16;
  const int voff = (lane < 48) ? lane * 16 : (int)0x80000000;     // lanes 48-63: out of range -> zeros (K = 24..31)
  const float sg = (lane < 32) ? 1.0f : -1.0f;
  const f32x2 sg2 = {sg, sg};
  float* dist = ch ? dist_i : dist_p;
  const char* dbase = dpk + ((size_t)ch * DG) * SCH_DIMG;
  const int qrow0 = qg32 * 32 + w * 8;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      dist + (size_t)qrow0 * n, 0, (qrow0 < m ? (m - qrow0 < 8 ? m - qrow0 : 8) : 0) * n * 4, 0x00020000);
  const int pf_slot = (qg32 & 31) * 4 + w;                                  // 0..127
  unsigned pf_sink = 0;
  CstT c;                                               // the stage-2 constants stay in registers for the whole kernel
#pragma unroll
  for (int eo = 0; eo < 2; eo++)
#pragma unroll
    for (int part = 0; part < 2; part++)
#pragma unroll
      for (int hl = 0; hl < 2; hl++) c.t[eo][part][hl] = cst[((eo * 2 + part) * 2 + hl) * 64 + lane];

  // Stage 1: the per-tile software pipeline of sc_match_h.hip over the SEQUENCE index i (frequency seqf(i)): operand slots rotate with
  // i & 3, the DB tiles of i + 2 / the query tiles of i + 1 are requested in the gaps of i.
  AOps At[4];
  BOps Bt[4];
  __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)g0 * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
#define QOFF(I) (seqf((I) & ~1) * SCH_QBLK)            /* block of the pair's even frequency */
  {
    const unsigned n0 = nat0 + QOFF(0), r0 = rot0 + QOFF(0);
    load_b<seqf(0), B_REH>(Bt[0], rs, voff); load_b<seqf(0), B_IMH>(Bt[0], rs, voff); load_b<seqf(0), B_REL>(Bt[0], rs, voff); load_b<seqf(0), B_IML>(Bt[0], rs, voff);
    load_a<0, A_H>(At[0], n0, r0); load_a<0, A_RH>(At[0], n0, r0); load_a<0, A_L>(At[0], n0, r0); load_a<0, A_RL>(At[0], n0, r0);
    load_b<seqf(1), B_REH>(Bt[1], rs, voff); load_b<seqf(1), B_IMH>(Bt[1], rs, voff); load_b<seqf(1), B_REL>(Bt[1], rs, voff);
    load_a<1, A_H>(At[1], n0, r0); load_a<1, A_RH>(At[1], n0, r0);
  }
#ifdef PR_SCH_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev));
#endif
  for (int g = g0; g < g1; g++) {
    const __amdgpu_buffer_rsrc_t rsn =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)(g + 1) * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
    Packed pk;
    Step zs;
    f32x4 Fa, Ma, Fb, Mb, t1a, t2a, t1b, t2b;
    unsigned ncur, rcur, nnxt = nat0 + QOFF(0), rnxt = rot0 + QOFF(0);
// request tile T of sequence index Q (Q >= 31: nothing)
#define LDB(Q, T) { if ((Q) < 31) load_b<seqf((Q) < 31 ? (Q) : 0), T>(Bt[(Q) & 3], rs, voff); }
#define LDA(I, Q, T) { if ((Q) < 31) { if (((Q) >> 1) == ((I) >> 1)) load_a<(Q) & 1, T>(At[(Q) & 3], ncur, rcur); \
                                       else load_a<(Q) & 1, T>(At[(Q) & 3], nnxt, rnxt); } }
#define FREQ(I, t1, t2, W0, W1, W2, W3, W4, W5)                                                   \
  {                                                                                               \
    SB(); MF0(t1, At[(I) & 3].h, Bt[(I) & 3].reh);  SB(); LDB((I) + 1, B_IML); LDA(I, (I) + 1, A_L); W0;   \
    SB(); MF0(t2, At[(I) & 3].rh, Bt[(I) & 3].imh); SB(); LDB((I) + 2, B_REH); W1;                \
    SB(); MFA(t1, At[(I) & 3].l, Bt[(I) & 3].reh);  SB(); LDB((I) + 2, B_IMH); LDA(I, (I) + 1, A_RL); W2;  \
    SB(); MFA(t2, At[(I) & 3].rl, Bt[(I) & 3].imh); SB(); LDA(I, (I) + 2, A_H); W3;               \
    SB(); MFA(t1, At[(I) & 3].h, Bt[(I) & 3].rel);  SB(); LDB((I) + 2, B_REL); W4;                \
    SB(); MFA(t2, At[(I) & 3].rh, Bt[(I) & 3].iml); SB(); LDA(I, (I) + 2, A_RH); W5;              \
    SB();                                                                                         \
  }
// pair P (sequence indices 2P, 2P+1): step j = P >> 2 (register j), octet o = P & 3 (block o until the transpose)
#define PKF(P, R) { split2(Fa[R], Fb[R], zs.h[0][R][(P) & 3], zs.l[0][R][(P) & 3]); }
#define PKM(P, R) { split2(Ma[R], Mb[R], zs.h[1][R][(P) & 3], zs.l[1][R][(P) & 3]); }
#define NONE ((void)0)
// LDS bases of this lane's tiles: current pair and next pair P + 1 (its block offset is a constant of the sequence)
#define ADV(P) { ncur = nnxt; rcur = rnxt; nnxt = nat0 + QOFF(2 * (P) + 2 < 32 ? 2 * (P) + 2 : 0); rnxt = rot0 + QOFF(2 * (P) + 2 < 32 ? 2 * (P) + 2 : 0); \
                 asm("" : "+v"(nnxt)); asm("" : "+v"(rnxt)); }
#define FMA_ALL(F, M, t1, t2) { FM2(F, M, t1, t2, 0); FM2(F, M, t1, t2, 2); }
#define PAIR0()                                                                                   \
  ADV(0)                                                                                          \
  FREQ(0, t1a, t2a, NONE, NONE, NONE, NONE, NONE, NONE)                                           \
  FREQ(1, t1b, t2b, NONE, NONE, NONE, NONE, NONE, FMA_ALL(Fa, Ma, t1a, t2a))
#define PAIR(P)                                                                                   \
  ADV(P)                                                                                          \
  FREQ(2 * (P), t1a, t2a, NONE, FM2(Fb, Mb, t1b, t2b, 0), FM2(Fb, Mb, t1b, t2b, 2), PKF((P) - 1, 0), PKM((P) - 1, 0), PKF((P) - 1, 1)) \
  FREQ(2 * (P) + 1, t1b, t2b, PKM((P) - 1, 1), PKF((P) - 1, 2), PKM((P) - 1, 2), PKF((P) - 1, 3), PKM((P) - 1, 3), FMA_ALL(Fa, Ma, t1a, t2a))
// the pair that packs the LAST pair of step J = ((P) - 1) >> 2 also transposes the step, register by register, right behind the packs
#define TX(X, R, J) transpose_xr<X, R>(pk, zs, J)
#define PAIRT(P)                                                                                  \
  ADV(P)                                                                                          \
  FREQ(2 * (P), t1a, t2a, NONE, FM2(Fb, Mb, t1b, t2b, 0), FM2(Fb, Mb, t1b, t2b, 2), PKF((P) - 1, 0),                                   \
       { PKM((P) - 1, 0); TX(0, 0, ((P) - 1) >> 2); }, { PKF((P) - 1, 1); TX(1, 0, ((P) - 1) >> 2); })                                   \
  FREQ(2 * (P) + 1, t1b, t2b, { PKM((P) - 1, 1); TX(0, 1, ((P) - 1) >> 2); }, { PKF((P) - 1, 2); TX(1, 1, ((P) - 1) >> 2); },           \
       { PKM((P) - 1, 2); TX(0, 2, ((P) - 1) >> 2); }, { PKF((P) - 1, 3); TX(1, 2, ((P) - 1) >> 2); },                                   \
       { PKM((P) - 1, 3); TX(0, 3, ((P) - 1) >> 2); }, { FMA_ALL(Fa, Ma, t1a, t2a); TX(1, 3, ((P) - 1) >> 2); })
// the transposes of step J (pairs 4J..4J+3 packed): 64 swaps
#define TRANSPOSE(J) { SB(); transpose_xr<0, 0>(pk, zs, J); transpose_xr<0, 1>(pk, zs, J); transpose_xr<0, 2>(pk, zs, J); transpose_xr<0, 3>(pk, zs, J); \
                       transpose_xr<1, 0>(pk, zs, J); transpose_xr<1, 1>(pk, zs, J); transpose_xr<1, 2>(pk, zs, J); transpose_xr<1, 3>(pk, zs, J); SB(); }

    PAIR0() PAIR(1) PAIR(2) PAIR(3)
    TICK(0)
    PAIRT(4) PAIR(5) PAIR(6) PAIR(7)
    TICK(1)
    PAIRT(8) PAIR(9) PAIR(10) PAIR(11)
    TICK(2)
    PAIRT(12) PAIR(13) PAIR(14)
    ADV(15)
    FREQ(30, t1a, t2a, NONE, FM2(Fb, Mb, t1b, t2b, 0), FM2(Fb, Mb, t1b, t2b, 2), { PKF(14, 0); PKM(14, 0); }, { PKF(14, 1); PKM(14, 1); },
         { PKF(14, 2); PKM(14, 2); })
    PKF(14, 3) PKM(14, 3)
    DRAIN();
    SB();
    FM2(Fa, Ma, t1a, t2a, 0); FM2(Fa, Ma, t1a, t2a, 2);
    Fb = f32x4{0.f, 0.f, 0.f, 0.f}; Mb = Fb;            // sequence index 31 = frequency 31: the zero pad of the K = 32 slots
    PKF(15, 0) PKM(15, 0) PKF(15, 1) PKM(15, 1) PKF(15, 2) PKM(15, 2) PKF(15, 3) PKM(15, 3)
    TICK(3)
    TRANSPOSE(3)
    TICK(4)
    {  // L2 prefetch of group g + 2 for the whole XCD (as sc_match_h.hip): this wave's 6 of its 744 cache lines
      asm volatile("" : : "v"(pf_sink));
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(dbase + (size_t)(g + 2) * SCH_DIMG), 0, (g + 2 < DG) ? SCH_DIMG : 0, 0x00020000);
      int lp = lane;
      asm volatile("" : "+v"(lp));
      const int pf_off = (lp < 6) ? (pf_slot * 6 + lp) * 128 : (int)0x80000000;
      pf_sink = __builtin_amdgcn_raw_buffer_load_b32(rp, pf_off, 0, 0);
    }
    // first requests of the next group
    {
      const unsigned n0 = nat0 + QOFF(0), r0 = rot0 + QOFF(0);
#define NB(I, T) load_b<seqf(I), T>(Bt[I], rsn, voff)
#define NA(I, T) load_a<(I) & 1, T>(At[I], n0, r0)
      NB(0, B_REH); NB(0, B_IMH); NB(0, B_REL); NB(0, B_IML); NB(1, B_REH); NB(1, B_IMH); NB(1, B_REL);
      NA(0, A_H); NA(0, A_RH); NA(0, A_L); NA(0, A_RL); NA(1, A_H); NA(1, A_RH);
    }
    SB();
    // ---------------------------------------------------------------- stage 2 + epilogue: 16 (register, query half) results, each the max
    // over forward / mirror and the 32 shift rows; rows of a tile sit in the four 16-lane blocks: max over the blocks, lanes 0-15 store
    int le = lane;
    asm volatile("" : "+v"(le));
    const bool live = le < 16 && g * 16 + le < n;
    // software pipeline over the 16 groups: the twelve MFMAs of group k + 1 are issued before group k's tiles are reduced (their 48+ wait
    // states are what an asm MFMA result needs before a VALU read: nothing is padded around asm), 2 groups = one (R, H) result
    TICK(5)
    Tiles tl[2];
    float mxr = -__builtin_inff();
#define FINISH(RH)                                                                                \
  { unsigned a = __float_as_uint(mxr), b = a;                                                     \
    swap32(a, b); mxr = fmaxf(__uint_as_float(a), __uint_as_float(b));                            \
    a = __float_as_uint(mxr); b = a;                                                              \
    swap16(a, b); mxr = fmaxf(__uint_as_float(a), __uint_as_float(b));                            \
    const int st = live ? ((4 * ((RH) & 1) + ((RH) >> 1)) * n + g * 16 + le) * 4 : (int)0x80000000; \
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(mxr, -0x1p-26f, 0.5f)), rd, st, 0, 0); \
    mxr = -__builtin_inff(); }
#define STEP2(K)                                                                                  \
  { SB(); s2_issue<K>(pk, c, tl[(K) & 1]); SB();                                                  \
    mxr = s2_reduce(mxr, tl[((K) - 1) & 1]);                                                      \
    if ((((K) - 1) & 1) == 1) FINISH(((K) - 1) >> 1)                                              \
    SB(); }
    SB(); s2_issue<0>(pk, c, tl[0]); SB();
    STEP2(1) STEP2(2) STEP2(3) STEP2(4) STEP2(5) STEP2(6) STEP2(7) STEP2(8) STEP2(9) STEP2(10) STEP2(11) STEP2(12) STEP2(13) STEP2(14) STEP2(15)
    DRAIN(); SB();
    mxr = s2_reduce(mxr, tl[1]);
    FINISH(7)
    TICK(6)
    rs = rsn;
  }
#ifdef PR_SCH_TIMING
  if (blockIdx.x == 8 * 40 && tid == 0)
    for (int i = 0; i < 8; i++) reinterpret_cast<unsigned long long*>(dist_p)[i] = tacc[i] / (unsigned long long)(g1 - g0);
#endif
}

}  // namespace

size_t sc_match_t_lds_bytes() { return (size_t)4 * SCH_QIMG + 64; }

void launch_sc_match_t(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p, float* d_i,
                       int nsplit_override) {
  if (m <= 0 || n <= 0) return;
  const int QG8 = sc_qgroups8(m), DG = sc_dgroups(n);
  const int QG32 = QG8 / 4;
  int nsplit = (128 + QG32 - 1) / QG32;
  if (nsplit > DG / 32) nsplit = DG / 32;
  if (nsplit < 1) nsplit = 1;
  if (nsplit_override > 0) nsplit = nsplit_override * 4 <= DG ? nsplit_override : (DG >= 4 ? DG / 4 : 1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_match_t_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sc_match_t_lds_bytes());
  hipLaunchKernelGGL(sc_match_t_kernel, dim3(8 * QG32 * nsplit), dim3(256), sc_match_t_lds_bytes(), st,
                     static_cast<const char*>(qpk), static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i,
                     m, n, QG8, DG, nsplit);
}

}  // namespace pr
