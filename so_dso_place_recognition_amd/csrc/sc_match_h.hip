// sc_match_h.hip — all-pairs Scan-Context distance on the gfx950 f16 matrix cores with split-f16 ("hi + lo") operands
// (processSC.m:22-33).  Same mathematics as sc_match.hip (per-ring sector spectra, S_f / P_f, even/odd split of the
// inverse transform, max over the 60 + 60 shifts), different arithmetic:
//
//   every fp32 factor x is carried as  x = hi + lo,  hi = f16(x), lo = f16(x - hi)  (22 significand bits), and every
//   product as the three f16 products  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  accumulated in fp32 (the dropped lo*lo term is
//   2^-22 relative); in stage 1 the three are 60 of the 64 K-slots of TWO MFMAs (kernels.hpp: the operand pairs), in stage 2
//   three MFMAs.  On MI355X the f16 MFMA rate is 16x the fp32 MFMA rate, so the 3x is still a 5x gain in matrix
//   time, and — unlike v_mfma_f32_*_f32, which occupies the SIMD's fp32 VALU datapath — the f16 MFMAs run beside the
//   wave's own VALU work (tools/ubench/f16_feed.hip).  Measured distance error vs fp64: same order as the fp32 kernel
//   (tools/experiments/split_f16_error.py; tests/test_gpu_parity.py).
//
// One wave = 8 queries x 16 DB entries of one channel; spectra are pre-scaled (queries 2^8, DB 2^7, constants 2^10) so
// that hi and lo stay in the normal f16 range; the final correlation is rescaled by 2^-25.
//   stage 1  v_mfma_f32_16x16x32_f16, K = 32 of the 60 ring terms (+4 zero), per frequency f:
//            rows = {Re,Im} x 8 queries, cols = 16 entries
//            T1 = [Qr;Qi].Dr^T = (QrDr | QiDr)       T2 = [Qi;Qr].Di^T = (QiDi | QrDi)     lanes <32 | >=32
//            (T2's row operand = the same LDS image read with row ^ 8)
//            F = T1 + s T2 = (Re S_f | Im S_f)       M = T1 - s T2 = (Re P_f | Im P_f)     s = +1 | -1
//   split    F, M of two consecutive frequencies -> v_cvt_pk_f16_f32 (hi), v_fma_mixlo/hi_f16 (lo = F - hi)
//   swap     v_permlane32_swap(pair j, pair j+4): (Re | Im) x 2 -> (Re, Re') , (Im, Im'): four such registers are, lane
//            for lane, the B operand of a 32x32x16 MFMA over 16 frequencies (k = 8*(lane>>5) + 0..7, pair = lane&31)
//   stage 2  v_mfma_f32_32x32x16_f16 per half (16 frequencies):  E[r][F|M] += Ccos . Re-operand, O[r][F|M] += Csin . Im-operand
//            (A operand = constant [shift 0..31][16 frequencies] tile, hi and lo; 3 MFMAs per chain; 256 accumulators)
//   epilogue max over shifts of E + |O|, max(forward, mirror), d = 0.5 - 0.5 * 2^-25 * max           (processSC.m:30)
// A workgroup (4 waves, one per SIMD) keeps the split spectra of 32 queries of one channel in LDS (159 712 B) and
// sweeps a range of the DB; DB operands stream L2 -> L1 -> VGPR with raw buffer loads (each operand pair takes its 64
// lane pieces from the Re or Im hi + lo tiles of the frequency: kernels.hpp).
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

struct AOps { u32x4 h, l, rh, rl; };          // query row operands: operand pair 1, pair 2 (kernels.hpp), and the same with Re/Im rows exchanged
struct BOps { u32x4 reh, rel, imh, iml; };    // DB column operands: Re pair 1, Re pair 2, Im pair 1, Im pair 2

// one operand tile (4 registers) per call, so that every request can be placed in its own MFMA gap
enum { A_H = 0, A_L = 1, A_RH = 2, A_RL = 3 };
enum { B_REH = 0, B_REL = 1, B_IMH = 2, B_IML = 3 };
// nat / rot: 32-bit LDS byte addresses of this lane's 16 B of operand pair 1 in the block of frequency (P & ~1); the odd frequency of
// the pair is an immediate offset of the ds_read2_b64 (8-bit, in units of 8 B), pair 2 lies dl bytes further (per lane group: kernels.hpp)
typedef const u32x4_a8 __attribute__((address_space(3))) * lds_tile_p;
template <int P, int T>
__device__ __forceinline__ void load_a(AOps& a, unsigned nat, unsigned rot, unsigned dl) {
  unsigned addr = ((T & 2) ? rot : nat) + (P & 1) * SCH_QBLK;
  if (T & 1) addr += dl;
  const u32x4 v = *reinterpret_cast<lds_tile_p>(addr);
  if (T == A_H) a.h = v; else if (T == A_L) a.l = v; else if (T == A_RH) a.rh = v; else a.rl = v;
}
template <int P, int T>
__device__ __forceinline__ void load_b(BOps& b, __amdgpu_buffer_rsrc_t rs, int voff, int voff2) {
  // frequency and Re | Im tile pair in the scalar offset; the lane's place in the pair's hi + lo tiles for operand pair 1 | 2 in two registers
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (T == B_REL || T == B_IML) ? voff2 : voff,
                                                        P * SCH_DFREQ + ((T == B_IMH || T == B_IML) ? 2 : 0) * SCH_DTILE, 0);
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

// Packed results of 16 frequencies (one half) for the 4 stage-1 registers r: [kind][r] is a 4-register MFMA operand
// whose element j holds pair j (lanes 0-31) / pair j+4 (lanes 32-63) after the swap.
struct Half {
  u32x4 reFh[4], reFl[4], imFh[4], imFl[4];   // forward (S): Re / Im operands, hi / lo
  u32x4 reMh[4], reMl[4], imMh[4], imMl[4];   // mirror (P)
};

__device__ __forceinline__ void swap32(u32x4& a, u32x4& b, int e) {   // lanes 32-63 of a[e] <-> lanes 0-31 of b[e]
  const u32x2 v = __builtin_amdgcn_permlane32_swap(a[e], b[e], false, false);
  a[e] = v[0];
  b[e] = v[1];
}

// frequencies (2J, 2J+1) of the half -> element J&3 of the "re" (J < 4) or "im" (J >= 4) registers, still as (Re | Im)
template <int J, int R>
__device__ __forceinline__ void pack_F(Half& hb, const f32x4& Fa, const f32x4& Fb) {
  unsigned h, l;
  split2(Fa[R], Fb[R], h, l);
  if (J < 4) { hb.reFh[R][J & 3] = h; hb.reFl[R][J & 3] = l; } else { hb.imFh[R][J & 3] = h; hb.imFl[R][J & 3] = l; }
}
template <int J, int R>
__device__ __forceinline__ void pack_M(Half& hb, const f32x4& Ma, const f32x4& Mb) {
  unsigned h, l;
  split2(Ma[R], Mb[R], h, l);
  if (J < 4) { hb.reMh[R][J & 3] = h; hb.reMl[R][J & 3] = l; } else { hb.imMh[R][J & 3] = h; hb.imMl[R][J & 3] = l; }
}
template <int R>
__device__ __forceinline__ void swap_r(Half& hb, int e0, int e1) {   // elements e0..e1-1 of the 8 operands of register R
  for (int e = e0; e < e1; e++) {
    swap32(hb.reFh[R], hb.imFh[R], e);
    swap32(hb.reFl[R], hb.imFl[R], e);
    swap32(hb.reMh[R], hb.imMh[R], e);
    swap32(hb.reMl[R], hb.imMl[R], e);
  }
}

__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

struct Consts { u32x4 ch, cl, sh, sl; };   // A operands of one half: cos hi/lo, -sin hi/lo
template <int HALF>
__device__ __forceinline__ void load_consts(Consts& c, __amdgpu_buffer_rsrc_t rc, int lane16) {   // [E|O][half][hi|lo][64] x 16 B
  c.ch = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, ((0 * 2 + HALF) * 2 + 0) * 1024, 0);
  c.cl = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, ((0 * 2 + HALF) * 2 + 1) * 1024, 0);
  c.sh = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, ((1 * 2 + HALF) * 2 + 0) * 1024, 0);
  c.sl = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, ((1 * 2 + HALF) * 2 + 1) * 1024, 0);
}

// One of the 12 stage-2 MFMAs of stage-1 register R: I = 0..3 hi x hi (starts the chain in the first half),
// 4..7 lo(constants) x hi, 8..11 hi x lo; within each four: E forward, O forward, E mirror, O mirror.
template <bool FIRST, int R, int I>
__device__ __forceinline__ void stage2_one(const Half& hb, const Consts& c, f32x16 (&accE)[4][2], f32x16 (&accO)[4][2],
                                           const f32x16& zero) {
  constexpr int V = (I >> 1) & 1, PART = I & 1, T = I >> 2;
  const u32x4& ca = PART ? (T == 1 ? c.sl : c.sh) : (T == 1 ? c.cl : c.ch);
  const u32x4& op = PART ? (V ? (T == 2 ? hb.imMl[R] : hb.imMh[R]) : (T == 2 ? hb.imFl[R] : hb.imFh[R]))
                         : (V ? (T == 2 ? hb.reMl[R] : hb.reMh[R]) : (T == 2 ? hb.reFl[R] : hb.reFh[R]));
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

// ONEQ (m <= 8, the online case of one keyframe per call): all four waves of a workgroup work for the SAME 8-query group
// - wave w takes DB groups g0 + w, g0 + w + 4, ... - instead of four query groups sharing every DB group, so a call
// takes a quarter of the time (three of four waves would otherwise multiply padding).
template <bool ONEQ>
__global__ __launch_bounds__(256, 1) void sc_match_h_kernel(const char* __restrict__ qpk,   // [2][QG32][4][31][1288 B]
                                                            const char* __restrict__ dpk,   // [2][DG][31][4][768 B] + one zero group
                                                            const u32x4* __restrict__ cst,  // [2][2][2][64] x 16 B
                                                            float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                            int m, int n, int QG8, int DG, int nsplit, ScBin bin) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // binary-channel launches (kernels.hpp: ScBin; sc_match_e.hip): gate 2 = the split-f16 pass of channel 1 behind the single-product one -
  // it runs when that one did not (the bound rules it out) or a pair failed its rounding test (the flag word carries this call's number)
  if (bin.gate == 0 && bin.chsel == 0 && bin.viol && blockIdx.x == 0 && threadIdx.x == 0) *bin.viol = 0;   // (channel-0 launch of a binary-channel call: see sc_match_e_kernel)
  if (bin.gate == 2) {
    float bound;
    if (sc_bin_bound(bin, bound) && *bin.viol != bin.gen) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware mapping (workgroups go round-robin to the 8 XCDs, each with its own L2): all workgroups of one XCD work
  // on ONE channel and on the same quarter of the DB ranges, consecutive workgroups of an XCD on consecutive 32-query
  // blocks - so the ~32 resident workgroups of an XCD sweep the same DB range together and share it through that L2.
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ch = bin.chsel < 0 ? (xcd & 1) : bin.chsel;                         // chsel >= 0: one channel on all eight XCDs
  const int range = (bin.chsel < 0 ? (xcd >> 1) + 4 * (idx % nsplit) : xcd + 8 * (idx % nsplit)), qg32 = idx / nsplit;      // nsplit = ranges per XCD slice
  const int nrange = (bin.chsel < 0 ? 4 : 8) * nsplit;
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
  constexpr int GS = ONEQ ? 4 : 1;                                  // DB groups between two units of a wave
  const int wq = ONEQ ? 0 : w;                                      // this wave's query group inside the workgroup's image
  const unsigned nat0 = lds0 + wq * SCH_QIMG + row * 80 + (row >= 8 ? 8 : 0) + sch_a1_byte(kg);
  const unsigned rot0 = lds0 + wq * SCH_QIMG + (row ^ 8) * 80 + (row >= 8 ? 0 : 8) + sch_a1_byte(kg);
  const unsigned dl = sch_a2_byte(kg) - sch_a1_byte(kg);
  const int voff = sch_b1_byte(lane), voff2 = sch_b2_byte(lane);
  const float sg = (lane < 32) ? 1.0f : -1.0f;
  const f32x2 sg2 = {sg, sg};
  float* dist = ch ? dist_i : dist_p;
  const char* dbase = dpk + ((size_t)ch * DG) * SCH_DIMG;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int qrow0 = qg32 * 32 + wq * 8;
  // distances of this wave's 8 query rows: byte offset = ((local row) * n + entry) * 4; local row = R (lanes 0-15) or 4 + R (16-31)
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      dist + (size_t)qrow0 * n, 0, (qrow0 < m ? (m - qrow0 < 8 ? m - qrow0 : 8) : 0) * n * 4, 0x00020000);
  const int pf_slot = (qg32 & 31) * 4 + w;                                  // 0..127
  unsigned pf_sink = 0;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(cst), 0, 8192, 0x00020000);

  // Software pipeline over the slot sequence (4 stage-1 MFMAs per frequency), per operand TILE: the DB tiles of frequency
  // q are requested 8-12 MFMA slots ahead (Re hi, Im hi, Re lo during frequency q-2, Im lo during q-1), the query tiles 4
  // slots ahead (during q-1); buffers rotate with period 4 (DB) and 2 (queries) over 32 positions per group, position 31
  // being a ghost whose requests are issued by hand at the start of the stage-2 phase.  hipcc counts all these loads, so
  // every MFMA waits with the exact vmcnt / lgkmcnt for its own operands only.  The wave issues in order, so the VALU
  // work is placed by hand into the gaps between MFMAs and pinned with sched_barrier: the F/M combination of a
  // frequency runs two MFMAs after its last stage-1 MFMA, the split/pack of pair J-1 under the stage-1 MFMAs of pair J,
  // the permlane swaps of register r+1 and the epilogue of register r-1 under the 12 stage-2 MFMAs of register r.
  AOps At[4];
  BOps Bt[4];
  __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)(g0 + (ONEQ ? w : 0)) * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
  load_b<0, B_REH>(Bt[0], rs, voff, voff2); load_b<0, B_IMH>(Bt[0], rs, voff, voff2); load_b<0, B_REL>(Bt[0], rs, voff, voff2); load_b<0, B_IML>(Bt[0], rs, voff, voff2);
  load_a<0, A_H>(At[0], nat0, rot0, dl); load_a<0, A_RH>(At[0], nat0, rot0, dl); load_a<0, A_L>(At[0], nat0, rot0, dl); load_a<0, A_RL>(At[0], nat0, rot0, dl);
  load_b<1, B_REH>(Bt[1], rs, voff, voff2); load_b<1, B_IMH>(Bt[1], rs, voff, voff2); load_b<1, B_REL>(Bt[1], rs, voff, voff2);
  load_a<1, A_H>(At[1], nat0, rot0, dl); load_a<1, A_RH>(At[1], nat0, rot0, dl);
#ifdef PR_SCH_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev));
#endif
  unsigned nbase = nat0, rbase = rot0;       // carried through the running tile addresses: nothing address-like stays live (= spilled) across a unit
  for (int g = g0 + (ONEQ ? w : 0); g < g1; g += GS) {
    const __amdgpu_buffer_rsrc_t rsn =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)(g + GS) * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
    f32x16 accE[4][2], accO[4][2];
    Half hb;
    Consts c;
    f32x4 Fa, Ma, Fb, Mb, t1a, t2a, t1b, t2b;
    unsigned ncur, rcur, nnxt = nbase, rnxt = rbase;
    TICK(7)
// request tile T of frequency Q of this group (Q >= 31: nothing - the first requests of the next group are issued by
// hand late in the stage-2 phase, when half of the packed registers are free again)
#define LDB(Q, T) { if ((Q) < SC_NF) load_b<((Q) < SC_NF ? (Q) : 0), T>(Bt[(Q) & 3], rs, voff, voff2); }
#define LDA(P, Q, T) { if ((Q) < SC_NF) { if (((Q) >> 1) == ((P) >> 1)) load_a<((Q) < SC_NF ? (Q) : 0), T>(At[(Q) & 3], ncur, rcur, dl); \
                                        else load_a<((Q) < SC_NF ? (Q) : 0), T>(At[(Q) & 3], nnxt, rnxt, dl); } }
// a frequency: operand pair 1, then pair 2, into T1 (Re tiles) and T2 (Im tiles, row-exchanged queries) - four MFMAs, six VALU pieces
#define FREQ(P, t1, t2, W0, W1, W2, W3, W4, W5)                                                   \
  {                                                                                               \
    SB(); MF0(t1, At[(P) & 3].h, Bt[(P) & 3].reh);  SB(); LDB((P) + 1, B_IML); LDA(P, (P) + 1, A_L); W0;   \
    SB(); MF0(t2, At[(P) & 3].rh, Bt[(P) & 3].imh); SB(); LDB((P) + 2, B_REH); W1;                \
    SB(); MFA(t1, At[(P) & 3].l, Bt[(P) & 3].rel);  SB(); LDB((P) + 2, B_IMH); LDA(P, (P) + 1, A_RL); W2; LDA(P, (P) + 2, A_H); W3;  \
    SB(); MFA(t2, At[(P) & 3].rl, Bt[(P) & 3].iml); SB(); LDB((P) + 2, B_REL); W4; LDA(P, (P) + 2, A_RH); W5;              \
    SB();                                                                                         \
  }
#define PKF(J, R) pack_F<J, R>(hb, Fa, Fb)
#define PKM(J, R) pack_M<J, R>(hb, Ma, Mb)
#define PK(J, R) { PKF(J, R); PKM(J, R); }
#define NONE ((void)0)
// LDS bases of this lane's tiles: current pair and next pair (one opaque add per pair and operand kind)
#define ADV() { ncur = nnxt; rcur = rnxt; nnxt = ncur + 2 * SCH_QBLK; rnxt = rcur + 2 * SCH_QBLK; asm("" : "+v"(nnxt)); asm("" : "+v"(rnxt)); }
// Pair J of half H.  The VALU work is spread as evenly as the dependences allow, ~3 instructions per MFMA gap (a
// 16x16x32 MFMA hides two or three; a gap with six costs ~34 cycles instead of ~18): the F/M combination of the previous
// pair's second frequency (its last MFMA is two slots back), the eight split/pack pieces of the previous pair, and in
// the last gap the F/M combination of this pair's first frequency (after the last reader of the old Fa/Ma).
#define FMA_ALL(F, M, t1, t2) { FM2(F, M, t1, t2, 0); FM2(F, M, t1, t2, 2); }
#define PAIR0(H)                                                                                  \
  ADV()                                                                                           \
  FREQ(16 * (H), t1a, t2a, NONE, NONE, NONE, NONE, NONE, NONE)                                    \
  FREQ(16 * (H) + 1, t1b, t2b, NONE, NONE, NONE, NONE, NONE, FMA_ALL(Fa, Ma, t1a, t2a))
#define PAIR(H, J)                                                                                \
  ADV()                                                                                           \
  FREQ(16 * (H) + 2 * (J), t1a, t2a, NONE, FM2(Fb, Mb, t1b, t2b, 0), FM2(Fb, Mb, t1b, t2b, 2), PKF((J) - 1, 0), PKM((J) - 1, 0), PKF((J) - 1, 1)) \
  FREQ(16 * (H) + 2 * (J) + 1, t1b, t2b, PKM((J) - 1, 1), PKF((J) - 1, 2), PKM((J) - 1, 2), PKF((J) - 1, 3), PKM((J) - 1, 3), FMA_ALL(Fa, Ma, t1a, t2a))
// the 12 stage-2 MFMAs of register R with the VALU pieces W0..W11 in their gaps
#define S2(FIRST, R, W0, W1, W2, W3, W4, W5, W6, W7, W8, W9, W10, W11)                             \
  { stage2_one<FIRST, R, 0>(hb, c, accE, accO, zero); SB(); W0; SB();                             \
    stage2_one<FIRST, R, 1>(hb, c, accE, accO, zero); SB(); W1; SB();                             \
    stage2_one<FIRST, R, 2>(hb, c, accE, accO, zero); SB(); W2; SB();                             \
    stage2_one<FIRST, R, 3>(hb, c, accE, accO, zero); SB(); W3; SB();                             \
    stage2_one<FIRST, R, 4>(hb, c, accE, accO, zero); SB(); W4; SB();                             \
    stage2_one<FIRST, R, 5>(hb, c, accE, accO, zero); SB(); W5; SB();                             \
    stage2_one<FIRST, R, 6>(hb, c, accE, accO, zero); SB(); W6; SB();                             \
    stage2_one<FIRST, R, 7>(hb, c, accE, accO, zero); SB(); W7; SB();                             \
    stage2_one<FIRST, R, 8>(hb, c, accE, accO, zero); SB(); W8; SB();                             \
    stage2_one<FIRST, R, 9>(hb, c, accE, accO, zero); SB(); W9; SB();                             \
    stage2_one<FIRST, R, 10>(hb, c, accE, accO, zero); SB(); W10; SB();                           \
    stage2_one<FIRST, R, 11>(hb, c, accE, accO, zero); SB(); W11; SB(); }

    // ---------------------------------------------------------------- first half: frequencies 0..15
    PAIR0(0) PAIR(0, 1) PAIR(0, 2) PAIR(0, 3) PAIR(0, 4) PAIR(0, 5) PAIR(0, 6) PAIR(0, 7)
    TICK(0)
    DRAIN();
    SB();
    FM2(Fb, Mb, t1b, t2b, 0); FM2(Fb, Mb, t1b, t2b, 2);
    PK(7, 0) PK(7, 1) PK(7, 2) PK(7, 3)
    SB();
    load_consts<0>(c, rc, lane * 16);
    swap_r<0>(hb, 0, 4);
    SB();
    TICK(1)
    S2(true, 0, swap_r<1>(hb, 0, 1), NONE, swap_r<1>(hb, 1, 2), NONE, swap_r<1>(hb, 2, 3), NONE, swap_r<1>(hb, 3, 4), NONE, NONE, NONE, NONE, NONE)
    S2(true, 1, swap_r<2>(hb, 0, 1), NONE, swap_r<2>(hb, 1, 2), NONE, swap_r<2>(hb, 2, 3), NONE, swap_r<2>(hb, 3, 4), NONE, NONE, NONE, NONE, NONE)
    S2(true, 2, swap_r<3>(hb, 0, 1), NONE, swap_r<3>(hb, 1, 2), NONE, swap_r<3>(hb, 2, 3), NONE, swap_r<3>(hb, 3, 4), NONE, NONE, NONE, NONE, NONE)
    S2(true, 3, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE)
    TICK(2)
    // ---------------------------------------------------------------- second half: frequencies 16..30
    PAIR0(1) PAIR(1, 1) PAIR(1, 2) PAIR(1, 3) PAIR(1, 4) PAIR(1, 5) PAIR(1, 6)
    ADV()
    FREQ(30, t1a, t2a, NONE, FM2(Fb, Mb, t1b, t2b, 0), FM2(Fb, Mb, t1b, t2b, 2), PK(6, 0), PK(6, 1), PK(6, 2))
    PK(6, 3)
    TICK(3)
    DRAIN();
    SB();
    FM2(Fa, Ma, t1a, t2a, 0); FM2(Fa, Ma, t1a, t2a, 2);
    Fb = f32x4{0.f, 0.f, 0.f, 0.f}; Mb = Fb;
    PK(7, 0) PK(7, 1) PK(7, 2) PK(7, 3)
    SB();
    load_consts<1>(c, rc, lane * 16);
    {  // L2 prefetch of group g + 2 for the whole XCD: this wave's 6 of its 744 cache lines (1/128 of the group), one
       // dword per line into a register nobody reads before the same point of the next group.  The ~128 waves that sweep
       // this range on this XCD cover the group between them, so the demand loads two groups later hit the L2 instead of
       // paying HBM latency in the middle of the in-order load queue.
      asm volatile("" : : "v"(pf_sink));
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(dbase + (size_t)(g + 2 * GS) * SCH_DIMG), 0, (g + 2 * GS < DG) ? SCH_DIMG : 0, 0x00020000);
      int lp = lane;
      asm volatile("" : "+v"(lp));                 // recomputed here, not kept (or spilled) across the unit
      const int pf_off = (lp < 6) ? (pf_slot * 6 + lp) * 128 : (int)0x80000000;   // lines past the group are out of range
      pf_sink = __builtin_amdgcn_raw_buffer_load_b32(rp, pf_off, 0, 0);
    }
    swap_r<0>(hb, 0, 4);
    SB();
    TICK(4)
    // (v_accvgpr_read next to in-flight MFMAs costs ~25 cycles each, so the epilogue is NOT interleaved with stage 2)
#define NB(P, T) load_b<P, T>(Bt[P], rsn, voff, voff2)
#define NA(P, T) load_a<P, T>(At[P], nbase, rbase, dl)
    S2(false, 0, swap_r<1>(hb, 0, 1), NONE, swap_r<1>(hb, 1, 2), NONE, swap_r<1>(hb, 2, 3), NONE, swap_r<1>(hb, 3, 4), NONE, NONE, NONE, NONE, NONE)
    S2(false, 1, swap_r<2>(hb, 0, 1), NONE, swap_r<2>(hb, 1, 2), NONE, swap_r<2>(hb, 2, 3), NONE, swap_r<2>(hb, 3, 4), NONE, NONE, NONE, NONE, NONE)
    S2(false, 2, swap_r<3>(hb, 0, 1), NONE, swap_r<3>(hb, 1, 2), NONE, swap_r<3>(hb, 2, 3), NONE, swap_r<3>(hb, 3, 4), NONE, NONE, NONE, NONE, NONE)
    // first requests of the next group (what its frequencies "-2" and "-1" would have issued)
    S2(false, 3, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE)
    TICK(5)
#define EPILOGUE(R)                                                                               \
  { const int st_base = (le < 32 && g * 16 + (le & 15) < n) ? st_lane : (int)0x80000000;          \
    float mx = -__builtin_inff();                                                                 \
    _Pragma("unroll") for (int e = 0; e < 16; e++) ep_elem<R>(mx, accE, accO, e);                 \
    ep_store<R>(mx, rd, st_base + (R) * 4 * n + g * 64); }
    int le = lane;
    asm volatile("" : "+v"(le));                   // the store addresses are recomputed per unit, not kept (or spilled) across it
    const int st_lane = ((le & 16) ? 4 * n : 0) * 4 + (le & 15) * 4;
    EPILOGUE(0)
    nbase = nnxt - 16 * 2 * SCH_QBLK; rbase = rnxt - 16 * 2 * SCH_QBLK;   // 16 pair advances back: the image's first block
    // first requests of the next group (what its frequencies "-2" and "-1" would have issued): behind the spill reloads of the
    // epilogue's addresses - a reload behind them would wait for these L2 / HBM loads, vmcnt being in order
    SB();
    NB(0, B_REH); NB(0, B_IMH); NB(0, B_REL); NB(0, B_IML); NB(1, B_REH); NB(1, B_IMH); NB(1, B_REL);
    NA(0, A_H); NA(0, A_RH); NA(0, A_L); NA(0, A_RL); NA(1, A_H); NA(1, A_RH);
    SB();
    EPILOGUE(1) EPILOGUE(2) EPILOGUE(3)
    TICK(6)
    rs = rsn;
  }
#ifdef PR_SCH_TIMING
  if (blockIdx.x == 8 * 40 && tid == 0)   // one wave somewhere in the middle of the grid; written over the first distances
    for (int i = 0; i < 8; i++) reinterpret_cast<unsigned long long*>(dist_p)[i] = tacc[i] / (unsigned long long)((g1 - g0) / GS);
#endif
}

}  // namespace

size_t sc_match_h_lds_bytes() { return (size_t)4 * SCH_QIMG + 64; }

void launch_sc_match_h(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p,
                       float* d_i, int nsplit_override, const ScBin* binp) {
  if (m <= 0 || n <= 0) return;
  const int QG8 = sc_qgroups8(m), DG = sc_dgroups(n);
  const int QG32 = QG8 / 4;
  const ScBin bin = binp ? *binp : ScBin{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 1.f, 0, -1};
  // grid = 8 XCD slices (channel x quarter of the ranges) x QG32 query blocks x nsplit ranges per slice
  int nsplit = (128 + QG32 - 1) / QG32;            // >= ~4 workgroups per CU in total, for tail balance
  if (nsplit > DG / 32) nsplit = DG / 32;          // keep >= 8 DB groups (128 entries) per workgroup
  if (nsplit < 1) nsplit = 1;
  if (nsplit_override > 0) nsplit = nsplit_override * 4 <= DG ? nsplit_override : (DG >= 4 ? DG / 4 : 1);
  const int per = bin.chsel < 0 ? 4 : 8;          // ranges per unit of nsplit: a channel takes four of the eight XCDs, or all of them
  if (bin.chsel >= 0) {
    if (nsplit > DG / 64) nsplit = DG / 64 > 0 ? DG / 64 : 1;
    if (nsplit_override > 0) nsplit = nsplit_override * 8 <= DG ? nsplit_override : (DG >= 8 ? DG / 8 : 1);
  }
  if (m <= 8) {   // one query group: the four waves of a workgroup split the DB groups of its range
    nsplit = DG / (8 * per) < 32 ? (DG / (8 * per) > 0 ? DG / (8 * per) : 1) : 32;        // 8 x nsplit workgroups = one per CU, >= 8 units per wave
    if (nsplit_override > 0) nsplit = nsplit_override * per <= DG ? nsplit_override : (DG >= per ? DG / per : 1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_match_h_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)sc_match_h_lds_bytes());
    hipLaunchKernelGGL(sc_match_h_kernel<true>, dim3(8 * nsplit), dim3(256), sc_match_h_lds_bytes(), st, static_cast<const char*>(qpk),
                       static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i, m, n, QG8, DG, nsplit, bin);
    return;
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_match_h_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sc_match_h_lds_bytes());
  hipLaunchKernelGGL(sc_match_h_kernel<false>, dim3(8 * QG32 * nsplit), dim3(256), sc_match_h_lds_bytes(), st,
                     static_cast<const char*>(qpk), static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i,
                     m, n, QG8, DG, nsplit, bin);
}

}  // namespace pr
