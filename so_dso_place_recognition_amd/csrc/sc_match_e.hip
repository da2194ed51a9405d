// sc_match_e.hip — the split-f16 SC matcher with TWO waves per SIMD (processSC.m:22-33).
//
// Same mathematics, packed images and stage-2 constants as sc_match_h.hip / sc_match_d.hip (read those headers first).  What changes:
//
//  * a wave's whole (8 queries x 16 entries x 31 frequencies) unit lives in 256 registers, so a 512-thread workgroup puts two waves on
//    every SIMD: wave w works on query group (w & 3) of the 32-query LDS image and on the DB groups of parity (w >> 2).  One in-order
//    wave issues an instruction every ~4.8 cycles and cannot overlap its VALU work (F / M combination, hi / lo split, reduction) with
//    its own MFMAs unless every instruction is hand-placed; two waves interleave in hardware.
//  * no row-exchanged query operand.  sc_match_h computes T2 = [Qi;Qr].Di^T from the LDS image read a second time with row ^ 8, so that
//    F = T1 + s T2 is lane-aligned.  Here T2' = [Qr;Qi].Di^T comes from the SAME query tiles as T1 (two LDS reads per frequency
//    instead of four, 32 operand registers fewer) and the exchange of the Re and Im row halves is folded into the permlane swap that
//    the stage-2 operand layout needs anyway: the frequencies are walked as pairs (f, f + 8), and
//        swap32(T1_f, T1_f+8) = (QrDr_f | QrDr_f+8) =: X , (QiDr_f | QiDr_f+8) =: Y        swap32(T2'_f, T2'_f+8) = U (QrDi), V (QiDi)
//        Re S = X + V   Im S = Y - U   Re P = X - V   Im P = Y + U           each already (f | f + 8) by lane half
//    which is, after the split, exactly the operand layout sc_match_d reaches with 128 swaps of packed registers.
//  * all packed operands stay in ArchVGPRs (nothing is parked in AccVGPRs, no v_accvgpr_write); the stage-2 constants are requested
//    once per unit while the last stage-1 quad runs.
//  * LO = false is the single-product arithmetic (PR_SC_ARITH_F16): operands hi only, one MFMA per product.
#include "kernels.hpp"
#ifndef E_PP
#define E_PP 2
#endif
#ifndef E_BD
#define E_BD 3          // depth of the DB operand ring (tiles of E_BD - 1 frequencies in flight)
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
__device__ __forceinline__ void load_a(AOps& a, unsigned addr) {   // addr = this lane's 16 B of the frequency's hi tile; lo tile at + 40
  const u32x4 v = *reinterpret_cast<lds_tile_p>(addr + T * 40);
  if (T == A_H) a.h = v; else a.l = v;
}
template <int F, int T>
__device__ __forceinline__ void load_b(BOps& b, __amdgpu_buffer_rsrc_t rs, int voff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, F * SCH_DFREQ + T * SCH_DTILE, 0);
  if (T == B_REH) b.reh = v; else if (T == B_REL) b.rel = v; else if (T == B_IMH) b.imh = v; else b.iml = v;
}

template <bool LO, int F, int T>
__device__ __forceinline__ void req_b(BOps& b, __amdgpu_buffer_rsrc_t rs, int voff) { if constexpr (LO || T == B_REH || T == B_IMH) load_b<F, T>(b, rs, voff); }
template <bool LO, int T>
__device__ __forceinline__ void req_a(AOps& a, unsigned addr) { if constexpr (LO || T == A_H) load_a<T>(a, addr); }

// MFMAs as asm statements (VGPR form; hipcc pads nothing around asm, cdna_hip_programming.md §5.7): every VALU reader of a result sits
// behind a DRAIN() or at least two further MFMAs + their fillers
#define MF0(d, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b))
#define MFA(d, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b))
#define M32Z(d, a, b, BC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), BC(b))
#define M32A(d, a, b, BC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), BC(b))
#define DRAIN() asm volatile("s_nop 9")
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void split2(float x, float y, unsigned& hi, unsigned& lo) {   // see sc_match_h.hip
  const f32x2 v = {x, y};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
  f32x2 r;
  asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(r[0]), "=&v"(r[1])
      : "v"(hi), "v"(x), "v"(y));
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
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
  if constexpr (PARK && HF == 0) { if (I < 2) M32Z(d, ca, op, "a"); else M32A(d, ca, op, "a"); }
  else { if (I < 2) M32Z(d, ca, op, "v"); else M32A(d, ca, op, "v"); }
}
// parks the operand tuples of a finished half in AccVGPRs (an empty asm whose operand must be an AccVGPR tuple: one v_accvgpr_write per
// register, placed by hipcc right here)
template <bool LO>
__device__ __forceinline__ void park_half(Half<LO>& h) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    asm volatile("" : "+a"(h.reFh[r]), "+a"(h.imFh[r]), "+a"(h.reMh[r]), "+a"(h.imMh[r]));
    if constexpr (LO) asm volatile("" : "+a"(h.reFl[r]), "+a"(h.imFl[r]), "+a"(h.reMl[r]), "+a"(h.imMl[r]));
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

template <bool LO, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void sc_match_e_kernel(const char* __restrict__ qpk,   // [2][QG32][4][31][1288 B]
                                                            const char* __restrict__ dpk,   // [2][DG][31][4][768 B] + zero groups
                                                            const u32x4* __restrict__ cst,  // [2][2][2][64] x 16 B
                                                            float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                            int m, int n, int QG8, int DG, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware mapping as in sc_match_d.hip: all workgroups of an XCD work on ONE channel and the same quarter of the DB ranges
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ch = xcd & 1;
  const int range = (xcd >> 1) + 4 * (idx % nsplit), qg32 = idx / nsplit;
  const int nrange = 4 * nsplit;
  const int g0 = (int)((long long)DG * range / nrange), g1 = (int)((long long)DG * (range + 1) / nrange);

  {  // the 4 query groups of this workgroup -> LDS (linear copy; the packed image IS the LDS image) + zeroed tail
    const u32x4* src = reinterpret_cast<const u32x4*>(qpk + ((size_t)ch * QG8 + (size_t)qg32 * 4) * SCH_QIMG);
    u32x4* dst = reinterpret_cast<u32x4*>(lds);
    constexpr int NV = 4 * SCH_QIMG / 16;
    for (int i = tid; i < NV + 4; i += 64 * NW) dst[i] = (i < NV) ? src[i] : u32x4{0u, 0u, 0u, 0u};
  }
  __syncthreads();

  const int row = lane & 15, kg = lane >> 4;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
  const int wq = w & 3, dpar = w >> 2;                    // (NW = 4: every wave takes every group)
  constexpr int GSTEP = NW / 4;                    // query group inside the image; parity of this wave's DB groups
  const unsigned nat0 = lds0 + wq * SCH_QIMG + row * 80 + (row >= 8 ? 8 : 0) + kg * 16;
  const int voff = (lane < 48) ? lane * 16 : (int)0x80000000;     // lanes 48-63: out of range -> zeros (K = 24..31)
  float* dist = ch ? dist_i : dist_p;
  const char* dbase = dpk + ((size_t)ch * DG) * SCH_DIMG;
  const int qrow0 = qg32 * 32 + wq * 8;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      dist + (size_t)qrow0 * n, 0, (qrow0 < m ? (m - qrow0 < 8 ? m - qrow0 : 8) : 0) * n * 4, 0x00020000);
  const int pf_slot = (qg32 & 31) * NW + w;                                 // 0..32 NW - 1
  unsigned pf_sink = 0;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(cst), 0, 8192, 0x00020000);
  if (g0 + dpar >= g1) return;

  AOps At[2];
  BOps Bt[E_BD];
  __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)(g0 + dpar) * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
#define NB(P, T) (req_b<LO, seqf(P), T>(Bt[P], RS, voff))
#define NA(T) (req_a<LO, T>(At[0], nat0))
#define NB1(T) (req_b<LO && E_BD == 3 || (!LO && E_BD == 3 && (T == B_REH || T == B_IMH)), seqf(1), T>(Bt[1 % E_BD], RS, voff))
#define RS rs
  NB(0, B_REH); NB(0, B_IMH); NB(0, B_REL); NB(0, B_IML);
  NA(A_H); NA(A_L);
  if constexpr (E_BD == 3) { NB(1, B_REH); NB(1, B_IMH); NB(1, B_REL); NB(1, B_IML); }
#undef RS

  for (int g = g0 + dpar; g < g1; g += GSTEP) {
    const __amdgpu_buffer_rsrc_t rsn =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)(g + GSTEP) * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
    Half<LO> hbs[2];
    Consts c0, c1;
    f32x4 t1a, t2a, t1b, t2b, t1c, t2c, t1d, t2d;
    unsigned na, nb, nnxt = nat0;          // LDS addresses of the pair's two hi tiles (f, f + 8) and of the next pair's first

// request tile T of walk position Q of this group (Q >= 31: nothing - the next group's first requests are issued in stage 2)
#define LDB(Q, T) { if constexpr (seqf(Q) < SC_NF && (Q) < 32 && (LO || (T == B_REH || T == B_IMH))) load_b<(seqf(Q) < SC_NF ? seqf(Q) : 0), T>(Bt[(Q) % E_BD], rs, voff); }
// query tiles of walk position Q = P + 1, requested during position P: the partner of the pair (odd Q) or the next pair's first (even Q)
#define LDA(Q, T) { if constexpr (seqf(Q) < SC_NF && (Q) < 32 && (LO || T == A_H)) { if constexpr (((Q) & 1) != 0) load_a<T>(At[(Q) & 1], nb); else load_a<T>(At[(Q) & 1], nnxt); } }
// start of the pair at walk position P (even): this pair's addresses, and the next pair's (the half boundary jumps 9 blocks)
#define ADV(P) { na = nnxt; nb = na + 8 * SCH_QBLK; nnxt = na + ((((P) & 15) == 14) ? 9 : 1) * SCH_QBLK; asm("" : "+v"(nb)); asm("" : "+v"(nnxt)); }
#define NONE ((void)0)
// one frequency (walk position P): its 6 (LO) or 2 MFMAs, the requests for positions P + 1 (query tiles) and P + 2 (DB tiles), and
// VALU pieces W0..W5 in the gaps
#define FREQ(P, t1, t2, W0, W1, W2, W3, W4, W5)                                                                  \
  {                                                                                                              \
    SB(); MF0(t1, At[(P) & 1].h, Bt[(P) % E_BD].reh); SB(); LDA((P) + 1, A_H); W0;                                  \
    SB(); MF0(t2, At[(P) & 1].h, Bt[(P) % E_BD].imh); SB(); LDA((P) + 1, A_L); W1;                                  \
    SB(); if constexpr (LO) MFA(t1, At[(P) & 1].l, Bt[(P) % E_BD].reh); SB(); LDB((P) + E_BD - 1, B_REH); W2;                        \
    SB(); if constexpr (LO) MFA(t2, At[(P) & 1].l, Bt[(P) % E_BD].imh); SB(); LDB((P) + E_BD - 1, B_IMH); W3;                        \
    SB(); if constexpr (LO) MFA(t1, At[(P) & 1].h, Bt[(P) % E_BD].rel); SB(); LDB((P) + E_BD - 1, B_REL); W4;                        \
    SB(); if constexpr (LO) MFA(t2, At[(P) & 1].h, Bt[(P) % E_BD].iml); SB(); LDB((P) + E_BD - 1, B_IML); W5;                        \
    SB();                                                                                                        \
  }
// (T1_f, T1_f+8, T2'_f, T2'_f+8) -> X, Y, U, V (in place), then Re S, Im S, Re P, Im P (in place: x <- X + V, v <- X - V, y <- Y - U, u <- Y + U)
#define SWP(x, y, u, v, e) { swap32f(x, y, e); swap32f(u, v, e); }
#define CMB(x, y, u, v, r0)                                                                       \
  {                                                                                               \
    const f32x2 _x = {x[r0], x[r0 + 1]}, _y = {y[r0], y[r0 + 1]}, _u = {u[r0], u[r0 + 1]}, _v = {v[r0], v[r0 + 1]}; \
    const f32x2 _sr = _x + _v, _pr = _x - _v, _si = _y - _u, _pi = _y + _u;                       \
    x[r0] = _sr[0]; x[r0 + 1] = _sr[1]; v[r0] = _pr[0]; v[r0 + 1] = _pr[1];                       \
    y[r0] = _si[0]; y[r0 + 1] = _si[1]; u[r0] = _pi[0]; u[r0 + 1] = _pi[1];                       \
  }
// after CMB: t1a = Re S, t1b = Im S, t2b = Re P, t2a = Im P of the quad's first pair; t1c, t1d, t2d, t2c of its second pair.
// element E of the half's operands for register R
// (the empty volatile asm pins the packed value HERE: hipcc otherwise sinks the whole pure combine / convert chain down to its stage-2
// consumer and keeps the fp32 values live instead - twice the registers)
#define PIN(x) asm volatile("" : "+v"(x))
#define PKQ(E, R)                                                                                 \
  {                                                                                               \
    if constexpr (LO) {                                                                           \
      unsigned _h, _l;                                                                            \
      split2(t1a[R], t1c[R], _h, _l); PIN(_h); PIN(_l); hb.reFh[R][E] = _h; hb.reFl[LO ? R : 0][E] = _l; \
      split2(t1b[R], t1d[R], _h, _l); PIN(_h); PIN(_l); hb.imFh[R][E] = _h; hb.imFl[LO ? R : 0][E] = _l; \
      split2(t2b[R], t2d[R], _h, _l); PIN(_h); PIN(_l); hb.reMh[R][E] = _h; hb.reMl[LO ? R : 0][E] = _l; \
      split2(t2a[R], t2c[R], _h, _l); PIN(_h); PIN(_l); hb.imMh[R][E] = _h; hb.imMl[LO ? R : 0][E] = _l; \
    } else {                                                                                      \
      unsigned _a = pack2(t1a[R], t1c[R]), _b = pack2(t1b[R], t1d[R]), _c = pack2(t2b[R], t2d[R]), _d = pack2(t2a[R], t2c[R]); \
      PIN(_a); PIN(_b); PIN(_c); PIN(_d);                                                         \
      hb.reFh[R][E] = _a; hb.imFh[R][E] = _b; hb.reMh[R][E] = _c; hb.imMh[R][E] = _d;             \
    }                                                                                             \
  }
// quad E of half H: walk positions 16 H + 4 E .. + 3 = frequencies (2E, 2E+8, 2E+1, 2E+9) of the half
#define QUAD(H, E, X0, X1)                                                                        \
  ADV(16 * (H) + 4 * (E))                                                                         \
  FREQ(16 * (H) + 4 * (E), t1a, t2a, NONE, NONE, NONE, NONE, NONE, NONE)                          \
  FREQ(16 * (H) + 4 * (E) + 1, t1b, t2b, NONE, NONE, NONE, NONE, NONE, NONE)                      \
  ADV(16 * (H) + 4 * (E) + 2)                                                                     \
  FREQ(16 * (H) + 4 * (E) + 2, t1c, t2c, NONE, NONE, SWP(t1a, t1b, t2a, t2b, 0), SWP(t1a, t1b, t2a, t2b, 1), SWP(t1a, t1b, t2a, t2b, 2), SWP(t1a, t1b, t2a, t2b, 3)) \
  if constexpr (seqf(16 * (H) + 4 * (E) + 3) < SC_NF) {                                                 \
    FREQ(16 * (H) + 4 * (E) + 3, t1d, t2d, NONE, CMB(t1a, t1b, t2a, t2b, 0), NONE, CMB(t1a, t1b, t2a, t2b, 2), X0, X1) \
  } else {                                                                                        \
    t1d = f32x4{0.f, 0.f, 0.f, 0.f}; t2d = t1d;                                                   \
    CMB(t1a, t1b, t2a, t2b, 0) CMB(t1a, t1b, t2a, t2b, 2) X0; X1;                                 \
  }                                                                                               \
  SB(); DRAIN(); SB();                                                                            \
  SWP(t1c, t1d, t2c, t2d, 0) SWP(t1c, t1d, t2c, t2d, 1) SWP(t1c, t1d, t2c, t2d, 2) SWP(t1c, t1d, t2c, t2d, 3) \
  CMB(t1c, t1d, t2c, t2d, 0) CMB(t1c, t1d, t2c, t2d, 2)                                           \
  PKQ(E, 0) PKQ(E, 1) PKQ(E, 2) PKQ(E, 3)                                                         \
  SB();

#define hb hbs[0]
    QUAD(0, 0, NONE, NONE) QUAD(0, 1, NONE, NONE) QUAD(0, 2, NONE, NONE) QUAD(0, 3, NONE, NONE)
#undef hb
    if constexpr (NW == 4) { SB(); park_half<LO>(hbs[0]); SB(); }      // one wave per SIMD: 512 registers, half of them AccVGPRs
#define hb hbs[1]
    QUAD(1, 0, NONE, NONE) QUAD(1, 1, NONE, NONE) QUAD(1, 2, NONE, NONE)
    // the last quad also requests the stage-2 constants (the operand rings are draining)
    QUAD(1, 3, (load_consts<0, LO>(c0, rc, lane * 16)), (load_consts<1, LO>(c1, rc, lane * 16)))
#undef hb
    {  // L2 prefetch for the whole XCD: the 256 waves that sweep this range on this XCD cover the two groups of the iteration after next
       // (1488 cache lines) with 6 lines each, one dword per line into a register nobody reads before the same point of the next unit
      asm volatile("" : : "v"(pf_sink));
      const int gp = (g - dpar) + 2 * GSTEP;
      const int pf_bytes = (gp + GSTEP <= DG) ? GSTEP * SCH_DIMG : (gp < DG ? (DG - gp) * SCH_DIMG : 0);
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)gp * SCH_DIMG), 0, pf_bytes, 0x00020000);
      int lp = lane;
      asm volatile("" : "+v"(lp));
      const int pf_off = (lp < 6) ? (pf_slot * 6 + lp) * 128 : (int)0x80000000;   // 32 NW waves x 6 lines >= GSTEP x 744 lines
      pf_sink = __builtin_amdgcn_raw_buffer_load_b32(rp, pf_off, 0, 0);
    }
    // ---------------------------------------------------------------- stage 2: group S = 2 R + V, its MFMAs into a fresh (E, O) tile pair;
    // group S + 1 is issued before group S is reduced; the first requests of the next unit sit between the groups
    constexpr int PP = E_PP;
    f32x16 tE[PP], tO[PP];
    float mx = -__builtin_inff();
    int le = lane;
    asm volatile("" : "+v"(le));
    const int st_lane = ((le & 16) ? 4 * n : 0) * 4 + (le & 15) * 4;
    const int st_base = (le < 32 && g * 16 + (le & 15) < n) ? st_lane : (int)0x80000000;
#define S2I(S, I) s2_one<LO, (NW == 4), ((S) >> 1), ((S) & 1), I>(hbs[0], hbs[1], c0, c1, tE[(S) & (PP - 1)], tO[(S) & (PP - 1)])
#define S2G(S, W0, W1, W2, W3, W4, W5, W6, W7, W8, W9, W10, W11)                                   \
  { if constexpr (LO) {                                                                            \
    SB(); S2I(S, 0); SB(); W0;  SB(); S2I(S, 1); SB(); W1;  SB(); S2I(S, 2); SB(); W2;             \
    SB(); S2I(S, 3); SB(); W3;  SB(); S2I(S, 4); SB(); W4;  SB(); S2I(S, 5); SB(); W5;             \
    SB(); S2I(S, 6); SB(); W6;  SB(); S2I(S, 7); SB(); W7;  SB(); S2I(S, 8); SB(); W8;             \
    SB(); S2I(S, 9); SB(); W9;  SB(); S2I(S, 10); SB(); W10; SB(); S2I(S, 11); SB(); W11; SB();    \
  } else {                                                                                         \
    SB(); S2I(S, 0); SB(); W0; W1; W2; SB(); S2I(S, 1); SB(); S2I(S, 2); SB(); W3; W4; W5;         \
    W6; W7; W8; SB(); S2I(S, 3); SB(); W9; W10; W11; SB(); } }
#define RP(S, i) mx = red_piece(mx, tE[(S) & (PP - 1)], tO[(S) & (PP - 1)], i)
#define ST(S) { ep_store(mx, rd, st_base + ((S) >> 1) * 4 * n + g * 64); mx = -__builtin_inff(); }
#define RS rsn
    if constexpr (PP == 2) {
    S2G(0, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE)
    S2G(1, NONE, NONE, NONE, RP(0, 0), RP(0, 1), RP(0, 2), RP(0, 3), RP(0, 4), RP(0, 5), RP(0, 6), RP(0, 7), NONE)
    S2G(2, NONE, NONE, NONE, RP(1, 0), RP(1, 1), RP(1, 2), RP(1, 3), RP(1, 4), RP(1, 5), RP(1, 6), RP(1, 7), ST(1))
    S2G(3, NB(0, B_REH), NB(0, B_IMH), NB(0, B_REL), RP(2, 0), RP(2, 1), RP(2, 2), RP(2, 3), RP(2, 4), RP(2, 5), RP(2, 6), RP(2, 7), NB(0, B_IML))
    S2G(4, NA(A_H), NA(A_L), NONE, RP(3, 0), RP(3, 1), RP(3, 2), RP(3, 3), RP(3, 4), RP(3, 5), RP(3, 6), RP(3, 7), ST(3))
    S2G(5, NB1(B_REH), NB1(B_IMH), NB1(B_REL), RP(4, 0), RP(4, 1), RP(4, 2), RP(4, 3), RP(4, 4), RP(4, 5), RP(4, 6), RP(4, 7), NB1(B_IML))
    S2G(6, NONE, NONE, NONE, RP(5, 0), RP(5, 1), RP(5, 2), RP(5, 3), RP(5, 4), RP(5, 5), RP(5, 6), RP(5, 7), ST(5))
    S2G(7, NONE, NONE, NONE, RP(6, 0), RP(6, 1), RP(6, 2), RP(6, 3), RP(6, 4), RP(6, 5), RP(6, 6), RP(6, 7), NONE)
    asm volatile("s_nop 15\n\ts_nop 15");          // the last tiles are read next: nothing pads an asm MFMA
    SB();
    RP(7, 0); RP(7, 1); RP(7, 2); RP(7, 3); RP(7, 4); RP(7, 5); RP(7, 6); RP(7, 7);
    ST(7)
    } else {
// one tile pair: the group's MFMAs, then (behind the wait states nothing else provides) its reduction; the partner wave has the pipe meanwhile
#define S2R(S, W0, W1, W2, W3)                                                                   \
    S2G(S, W0, W1, W2, W3, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE)                        \
    asm volatile("s_nop 15\n\ts_nop 15"); SB();                                                   \
    RP(S, 0); RP(S, 1); RP(S, 2); RP(S, 3); RP(S, 4); RP(S, 5); RP(S, 6); RP(S, 7); SB();
    S2R(0, NONE, NONE, NONE, NONE)
    S2R(1, NONE, NONE, NONE, NONE) ST(1)
    S2R(2, NB(0, B_REH), NB(0, B_IMH), NB(0, B_REL), NB(0, B_IML))
    S2R(3, NA(A_H), NA(A_L), NONE, NONE) ST(3)
    S2R(4, NB1(B_REH), NB1(B_IMH), NB1(B_REL), NB1(B_IML))
    S2R(5, NONE, NONE, NONE, NONE) ST(5)
    S2R(6, NONE, NONE, NONE, NONE)
    S2R(7, NONE, NONE, NONE, NONE) ST(7)
    }
#undef RS
    rs = rsn;
  }
}

}  // namespace

size_t sc_match_e_lds_bytes() { return (size_t)4 * SCH_QIMG + 64; }

void launch_sc_match_e(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p, float* d_i,
                       int nsplit_override, int single) {
  if (m <= 0 || n <= 0) return;
  const int QG8 = sc_qgroups8(m), DG = sc_dgroups(n);
  const int QG32 = QG8 / 4;
  int nsplit = (128 + QG32 - 1) / QG32;
  if (nsplit > DG / 32) nsplit = DG / 32;
  if (nsplit < 1) nsplit = 1;
  if (nsplit_override > 0) nsplit = nsplit_override * 4 <= DG ? nsplit_override : (DG >= 4 ? DG / 4 : 1);
  auto go = [&](auto kern, int nw) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc_match_e_lds_bytes());
    hipLaunchKernelGGL(kern, dim3(8 * QG32 * nsplit), dim3(64 * nw), sc_match_e_lds_bytes(), st, static_cast<const char*>(qpk),
                       static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i, m, n, QG8, DG, nsplit);
  };
  if (single) go(sc_match_e_kernel<false, 8>, 8); else go(sc_match_e_kernel<true, 4>, 4);
}

}  // namespace pr
