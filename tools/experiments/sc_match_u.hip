// sc_match_u.hip — split-f16 Scan-Context matcher, half unit: one wave = 8 queries x 8 DB entries, TWO waves per SIMD.
//
// Same mathematics and arithmetic as sc_match_h.hip (per-ring sector spectra, fp32 carried as f16 hi + lo, three f16 MFMAs per
// product, fp32 accumulate).  What changes is the unit: 128 accumulators + <= 128 VGPRs per wave, so two waves share a SIMD
// and each other's MFMA-only / VALU-only phases, L1 / LDS latencies and s_waitcnt stalls overlap without any hand-placed
// interleaving inside a wave (the full unit of sc_match_h.hip is pinned to one wave per SIMD by its 256 accumulators).
//
//   stage 1  v_mfma_f32_16x16x32_f16, K = 20 rings (+12 zero), per frequency ONE chain of 3 MFMAs:
//            rows = [Qr(8 queries); Qi(8)], cols = [Dr(8 entries) | Di(8)]  ->  X = (QrDr QrDi ; QiDr QiDi), all four real
//            products of the complex dot in one tile (no second row operand)
//   combine  Z = X(lane ^ 40) (ds_bpermute on the LDS crossbar: other row half, other column half), then
//            F = X + s Z = (Re S | Im S), M = X - s Z = (Re P | Im P) in the lanes with column < 8; s = +1 | -1 by lane half
//   merge    a bank-masked DPP row_ror:8 moves the useful half of register r+1 into the unused half of register r:
//            2 F and 2 M registers per frequency = (Re | Im by lane half) x 32 columns (4 queries x 8 entries)
//   split    two consecutive frequencies -> packed f16 hi, packed f16 lo; four such registers (8 frequencies = a quarter)
//            are the B operand of a 32x32x16 MFMA with K = (Re | Im) x 8 frequencies - no cross-lane movement
//   stage 2  v_mfma_f32_32x32x16_f16 per quarter: D1 += [cos | -sin] . operand (shift +k), D2 += [cos | +sin] . operand (shift -k),
//            hi x hi + lo x hi + hi x lo; 24 MFMAs per quarter, 8 chains x 16 = 128 accumulators
//   epilogue max over shifts of D1, D2, forward and mirror -> d = 0.5 - 2^-26 max                                (processSC.m:30)
// A workgroup = 8 waves: wave w works on query group w & 3 of the 32-query LDS image and on entries 8*(w >> 2) .. +7 of every
// 16-entry DB group.  DB image: [ch][group][f][half][hi|lo]{768 B: lane = (ring>>3)<<4 | Im<<3 | entry&7}.
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

#define SB() __builtin_amdgcn_sched_barrier(0)
#ifdef PR_SCH_TIMING
#define TICK(i) { SB(); unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)); tacc[i] += _t - tprev; tprev = _t; SB(); }
#else
#define TICK(i)
#endif
#ifdef EXP_NOLOADB
#define EXP_NOLOADB_V 1
#else
#define EXP_NOLOADB_V 0
#endif
#ifdef EXP_NOPACK
#define EXP_NOPACK_V 1
#else
#define EXP_NOPACK_V 0
#endif
#ifdef EXP_NOSTAGE2
#define EXP_NOSTAGE2_V 1
#else
#define EXP_NOSTAGE2_V 0
#endif

struct AOps { u32x4 h, l; };                         // query rows hi / lo (LDS)
struct BOps { u32x4 h, l; };                         // DB columns hi / lo (global)
struct Quarter { u32x4 h[4], l[4]; };                // packed operands [F01, F23, M01, M23] of 8 frequencies
struct Consts { u32x4 d1h, d1l, d2h, d2l; };

template <int F>
__device__ __forceinline__ void load_a(AOps& o, unsigned nat) {
  o.h = *reinterpret_cast<lds_tile_p>(nat + F * SCH_QBLK);
  o.l = *reinterpret_cast<lds_tile_p>(nat + F * SCH_QBLK + 40);
}
template <int F>
__device__ __forceinline__ void load_b(BOps& o, __amdgpu_buffer_rsrc_t rs, int voff) {
  o.h = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, F * SCH_DFREQ, 0);
  o.l = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + SCH_DTILE, F * SCH_DFREQ, 0);
}

// stage 1 of one frequency (VGPR form: the stage-2 accumulators must not be dragged into the AccVGPR form by the compiler's
// one-form-per-function choice); the trailing s_nop covers MFMA D -> the ds_bpermute / VALU reads that follow
// (cdna_hip_programming.md §5.7)
__device__ __forceinline__ void stage1(const AOps& a, const BOps& b, f32x4& x) {
  asm volatile(
      "v_mfma_f32_16x16x32_f16 %0, %1, %3, 0\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %1, %4, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\t"
      "s_nop 9"
      : "=&v"(x)
      : "v"(a.h), "v"(a.l), "v"(b.h), "v"(b.l));
}

// (hi, lo) split of two fp32 values into packed f16 pairs: lo = f16(x - hi), the residual formed exactly in fp32
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

// Z = X(lane ^ 40): issued right behind stage 1 and consumed one stage-1 chain later, so that the LDS-crossbar latency of
// one frequency hides under the MFMAs of the next
__device__ __forceinline__ void exchange(const f32x4& x, int bperm_addr, f32x4& z) {
#pragma unroll
  for (int r = 0; r < 4; r++) z[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(bperm_addr, __float_as_int(x[r])));
}
// X, Z -> the four merged registers (F01, F23, M01, M23) of one frequency
__device__ __forceinline__ void combine(const f32x4& x, const f32x4& z, f32x2 sg2, f32x4& fm) {
  f32x4 F, M;
#pragma unroll
  for (int r0 = 0; r0 < 4; r0 += 2) {
    const f32x2 a = {x[r0], x[r0 + 1]}, b = {z[r0], z[r0 + 1]};
    const f32x2 f = __builtin_elementwise_fma(b, sg2, a), m = __builtin_elementwise_fma(b, -sg2, a);
    F[r0] = f[0]; F[r0 + 1] = f[1]; M[r0] = m[0]; M[r0 + 1] = m[1];
  }
  // lanes with column >= 8 of register r take the column < 8 lanes of register r + 1 (row_ror:8, banks 2 and 3 only)
#define MERGE(a, b) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(a), __float_as_int(b), 0x128, 0xf, 0xc, false))
  fm[0] = MERGE(F[0], F[1]);
  fm[1] = MERGE(F[2], F[3]);
  fm[2] = MERGE(M[0], M[1]);
  fm[3] = MERGE(M[2], M[3]);
#undef MERGE
}

template <int E>
__device__ __forceinline__ void pack_pair(Quarter& q, const f32x4& a, const f32x4& b) {   // frequencies 2E, 2E+1 of the quarter
#pragma unroll
  for (int o = 0; o < 4; o++) {
    unsigned h, l;
    split2(a[o], b[o], h, l);
    q.h[o][E] = h;
    q.l[o][E] = l;
  }
}

template <bool FIRST>
__device__ __forceinline__ void stage2(const Quarter& q, const Consts& c, f32x16 (&acc)[4][2], const f32x16& zero) {
#define MF(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)
#pragma unroll
  for (int o = 0; o < 4; o++) {
    acc[o][0] = MF(c.d1h, q.h[o], FIRST ? zero : acc[o][0]);
    acc[o][1] = MF(c.d2h, q.h[o], FIRST ? zero : acc[o][1]);
  }
#pragma unroll
  for (int o = 0; o < 4; o++) {
    acc[o][0] = MF(c.d1l, q.h[o], acc[o][0]);
    acc[o][1] = MF(c.d2l, q.h[o], acc[o][1]);
  }
#pragma unroll
  for (int o = 0; o < 4; o++) {
    acc[o][0] = MF(c.d1h, q.l[o], acc[o][0]);
    acc[o][1] = MF(c.d2h, q.l[o], acc[o][1]);
  }
#undef MF
}

template <int Q>
__device__ __forceinline__ void load_consts(Consts& c, __amdgpu_buffer_rsrc_t rc, int lane16) {   // [quarter][D1|D2][hi|lo][64] x 16 B
  c.d1h = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, (Q * 4 + 0) * 1024, 0);
  c.d1l = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, (Q * 4 + 1) * 1024, 0);
  c.d2h = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, (Q * 4 + 2) * 1024, 0);
  c.d2l = __builtin_amdgcn_raw_buffer_load_b128(rc, lane16, (Q * 4 + 3) * 1024, 0);
}

__global__ __launch_bounds__(512, 1) void sc_match_u_kernel(const char* __restrict__ qpk,   // [2][QG32][4][31][1288 B]
                                                            const char* __restrict__ dpk,   // [2][DG][31][2][2][768 B] + 2 zero groups
                                                            const u32x4* __restrict__ cst,  // [4][2][2][64] x 16 B
                                                            float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                            int m, int n, int QG8, int DG, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qgrp = w & 3, dh = w >> 2;
  // XCD-aware mapping, see sc_match_h.hip
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ch = xcd & 1;
  const int range = (xcd >> 1) + 4 * (idx % nsplit), qg32 = idx / nsplit;
  const int nrange = 4 * nsplit;
  const int g0 = (int)((long long)DG * range / nrange), g1 = (int)((long long)DG * (range + 1) / nrange);

  {  // the 4 query groups of this workgroup -> LDS (linear copy; the packed image IS the LDS image) + zeroed tail
    const u32x4* src = reinterpret_cast<const u32x4*>(qpk + ((size_t)ch * QG8 + (size_t)qg32 * 4) * SCH_QIMG);
    u32x4* dst = reinterpret_cast<u32x4*>(lds);
    constexpr int NV = 4 * SCH_QIMG / 16;
    for (int i = tid; i < NV + 4; i += 512) dst[i] = (i < NV) ? src[i] : u32x4{0u, 0u, 0u, 0u};
  }
  __syncthreads();
  if (g0 >= g1) return;

  const int row = lane & 15, kg = lane >> 4;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
  const unsigned nat = lds0 + qgrp * SCH_QIMG + row * 80 + (row >= 8 ? 8 : 0) + kg * 16;
  const int voff = (lane < 48) ? dh * 2 * SCH_DTILE + lane * 16 : (int)0x80000000;   // lanes 48-63: out of range -> zeros (K = 24..31)
  const float sg = (lane < 32) ? 1.0f : -1.0f;
  const f32x2 sg2 = {sg, sg};
  const int bperm_addr = (lane ^ 40) * 4;
  float* dist = ch ? dist_i : dist_p;
  const char* dbase = dpk + ((size_t)ch * DG) * SCH_DIMG;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int qrow0 = qg32 * 32 + qgrp * 8;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(cst), 0, 16384, 0x00020000);
  // distances of this wave's 8 query rows x 8 entries: lane l < 32 of operand pair p (registers 2p, 2p+1):
  // query = 4*(l>>4) + 2p + ((l>>3)&1), entry = 8*dh + (l&7)
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      dist + (size_t)qrow0 * n, 0, (qrow0 < m ? (m - qrow0 < 8 ? m - qrow0 : 8) : 0) * n * 4, 0x00020000);
  const int st_lane = ((4 * (lane >> 4) + ((lane >> 3) & 1)) * n + dh * 8 + (lane & 7)) * 4;
  const int pf_slot = (qg32 & 31) * 8 + w;                                  // 0..255, 3 lines each
  const int pf_off = (lane < 3) ? (pf_slot * 3 + lane) * 128 : (int)0x80000000;
  unsigned pf_sink = 0;
  const int lane16 = lane * 16;

  // operand pipeline: DB tiles of frequency P+2 and query tiles of P+1 are requested while frequency P is computed
  // (positions 0..30, 31 = ghost, 32.. = the next group; buffers by position & 3 / & 1)
  AOps at[2];
  BOps bt[4];
  __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)g0 * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
  load_b<0>(bt[0], rs, voff);
  load_b<1>(bt[1], rs, voff);
  load_a<0>(at[0], nat);
#ifdef PR_SCH_TIMING
  unsigned long long tacc[3] = {0, 0, 0}, tprev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev));
#endif
  for (int g = g0; g < g1; g++) {
    const __amdgpu_buffer_rsrc_t rsn =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dbase + (size_t)(g + 1) * SCH_DIMG), 0, SCH_DIMG, 0x00020000);
    {  // cooperative L2 prefetch of group g + 2 (see sc_match_h.hip): 3 of its 744 lines per wave
      asm volatile("" : : "v"(pf_sink));
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(dbase + (size_t)(g + 2) * SCH_DIMG), 0, (g + 2 < DG) ? SCH_DIMG : 0, 0x00020000);
      pf_sink = __builtin_amdgcn_raw_buffer_load_b32(rp, pf_off, 0, 0);
    }
    f32x16 acc[4][2];
    Quarter qt;
    Consts c;
    f32x4 xa, xb, za, zb, fa, fb;
#define REQ(P)                                                                       \
  {                                                                                  \
    if (!EXP_NOLOADB_V && (P) + 2 < SC_NF) load_b<((P) + 2 < SC_NF ? (P) + 2 : 0)>(bt[((P) + 2) & 3], rs, voff);     \
    if ((P) + 1 < SC_NF) load_a<((P) + 1 < SC_NF ? (P) + 1 : 0)>(at[((P) + 1) & 1], nat);         \
    if (!EXP_NOLOADB_V && (P) == 30) { load_b<0>(bt[0], rsn, voff); load_b<1>(bt[1], rsn, voff); }   /* the next group's first two frequencies */ \
  }
#define PAIR(PA, E)                                                                  \
  {                                                                                  \
    REQ(PA)                                                                          \
    SB();                                                                            \
    stage1(at[(PA) & 1], bt[(PA) & 3], xa);                                          \
    if ((PA) == 30) load_a<0>(at[0], nat);          /* at[0] was frequency 30's own buffer */ \
    exchange(xa, bperm_addr, za);                                                    \
    SB();                                                                            \
    if ((PA) + 1 < SC_NF) {                                                          \
      REQ((PA) + 1)                                                                  \
      SB();                                                                          \
      stage1(at[((PA) + 1) & 1], bt[((PA) + 1) & 3], xb);                            \
      exchange(xb, bperm_addr, zb);                                                  \
      SB();                                                                          \
      combine(xa, za, sg2, fa);                                                      \
      combine(xb, zb, sg2, fb);                                                      \
    } else { combine(xa, za, sg2, fa); fb = f32x4{0.f, 0.f, 0.f, 0.f}; }             \
    if ((E) == 3) { SB(); load_consts<(PA) / 8>(c, rc, lane16); SB(); }   /* hidden under the pack below */ \
    if (!EXP_NOPACK_V || (E) == 3) pack_pair<E>(qt, fa, fb);                                         \
    SB();                                                                            \
  }
#define QUARTER(Q, FIRST)                                                            \
  PAIR(8 * (Q), 0) PAIR(8 * (Q) + 2, 1) PAIR(8 * (Q) + 4, 2) PAIR(8 * (Q) + 6, 3)   \
  TICK(0)                                                                            \
  if (!EXP_NOSTAGE2_V || (Q) == 0) stage2<FIRST>(qt, c, acc, zero);                                  \
  SB();                                                                              \
  TICK(1)
    QUARTER(0, true) QUARTER(1, false) QUARTER(2, false) QUARTER(3, false)
#undef QUARTER
#undef PAIR
#undef REQ
    // epilogue: per operand pair p (registers 2p, 2p+1 of stage 1): max over shifts, D1/D2, forward (operand p) / mirror (2 + p)
#pragma unroll
    for (int p = 0; p < 2; p++) {
      float mx = -__builtin_inff();
#pragma unroll
      for (int e = 0; e < 16; e++) {
        mx = fmaxf(fmaxf(mx, acc[p][0][e]), acc[p][1][e]);
        mx = fmaxf(fmaxf(mx, acc[2 + p][0][e]), acc[2 + p][1][e]);
      }
      const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      const bool ok = lane < 32 && g * 16 + dh * 8 + (lane & 7) < n;
      const unsigned so = ok ? (unsigned)(st_lane + g * 64 + p * 2 * n * 4) : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_fmaf(mx, -0x1p-26f, 0.5f)), rd, (int)so, 0, 0);
    }
    TICK(2)
    rs = rsn;
  }
#ifdef PR_SCH_TIMING
  if (blockIdx.x == 8 * 40 && tid == 0)
    for (int i = 0; i < 3; i++) reinterpret_cast<unsigned long long*>(dist_p)[i] = tacc[i] / (unsigned long long)(g1 - g0);
#endif
}

}  // namespace

void launch_sc_match_u(hipStream_t st, const void* qpk, int m, const void* dpk, int n, const void* cst, float* d_p,
                       float* d_i, int nsplit_override) {
  if (m <= 0 || n <= 0) return;
  const int QG8 = sc_qgroups8(m), DG = sc_dgroups(n);
  const int QG32 = QG8 / 4;
  int nsplit = (128 + QG32 - 1) / QG32;
  if (nsplit > DG / 32) nsplit = DG / 32;
  if (nsplit < 1) nsplit = 1;
  if (nsplit_override > 0) nsplit = nsplit_override * 4 <= DG ? nsplit_override : (DG >= 4 ? DG / 4 : 1);
  const size_t ldsb = (size_t)4 * SCH_QIMG + 64;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_match_u_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipLaunchKernelGGL(sc_match_u_kernel, dim3(8 * QG32 * nsplit), dim3(512), ldsb, st, static_cast<const char*>(qpk),
                     static_cast<const char*>(dpk), static_cast<const u32x4*>(cst), d_p, d_i, m, n, QG8, DG, nsplit);
}

}  // namespace pr
