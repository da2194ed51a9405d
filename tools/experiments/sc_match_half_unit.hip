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
// Mapping to the matrix cores.  One wave = 8 queries x 8 DB entries ("half unit"); the 31 frequencies are processed
// as 16 slots of two: (0,30), (1,2), ..., (27,28), (29,-).
//   stage 1  v_mfma_f32_16x16x4_f32, K = 20 rings, ONE chain per frequency:
//            rows  = [re q0, im q0, re q1, im q1 | q2,q3 | q4,q5 | q6,q7]   (lane group t = lane>>4 owns queries 2t, 2t+1)
//            cols  = [Dre of 8 entries | Dim of 8 entries]
//            so register u of lane (t, c) holds   u=0: (Qre_a.Dre | Qre_a.Dim) = (A | E),  u=1: (C | B) of query a = 2t,
//            u=2, 3: the same for query b = 2t+1   (c < 8 | c >= 8): all four real products without a second chain.
//   combine  v_add/sub_f32 with a DPP row_ror:8 operand:  X1 = r0 + ror8(r1) = (A+B | E+C) = (Re S | Im P),
//            X2 = r0 - ror8(r1) = (A-B | E-C) = (Re P | -Im S); two bank-masked DPP moves then gather the Re S values of
//            queries a and b into one register (64 pairs per register), likewise Im S, Re P, Im P  (8 VALU per frequency)
//   swap     v_permlane32_swap(Y_fa, Y_fb) -> (fa | fb) in the lane halves: each register is now, lane for lane, the B
//            operand (k = lane>>5, pair = lane&31) of a 32x32x2 MFMA over K = {fa, fb}
//   stage 2  v_mfma_f32_32x32x2_f32:  E[r][fwd|mir] += Ccos[slot] . Re-tile,  O[r][fwd|mir] += Csin[slot] . Im-tile
//            (A operand = constant [shift 0..31][fa|fb] tile; 8 MFMAs per slot; 128 accumulator registers)
//   epilogue max over shifts of E + |O| (in-lane over 16 registers + one cross-half exchange) -> d = 0.5 - 0.5*max.
// Stage 1 feeds stage 2 through registers only: no LDS or HBM round trip of the m x n x 124 intermediate.
//
// Why half units: on gfx950 a wave's own VALU / LDS / VMEM instructions do not issue in the shadow of its own fp32 MFMAs,
// but they do hide behind the MFMAs of a second wave on the same SIMD (tools/ubench/mfma_valu.hip).  A full 8 x 16 unit
// needs 256 accumulators = one wave per SIMD (73 % matrix-pipe busy); the half unit needs 128, so a 512-thread
// workgroup puts two waves on every SIMD (<= 256 registers each).  The workgroup keeps the spectra of 32 queries of one
// channel resident in LDS (158 720 B); wave w = (query group w>>1, DB half w&1) sweeps a range of the DB; the DB operand
// stream goes HBM/L2 -> VGPR directly (waves with the same DB half read the same bytes: 3 of 4 hit L1).
// Bound: MFMA fp32 (157.3 TFLOP/s).
#include "kernels.hpp"

namespace pr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct SlotOps {    // operands of one slot (two frequencies fa, fb) for one half unit
  float a[2][5];    // query rows, K-steps s = 0..4                                 (LDS)
  f32x4 b[2];       // DB columns [Re | Im of 8 entries], K-steps 0..3               (global)
  float2 b5;        // K-step 4 of (fa, fb)                                         (global)
  float2 c;         // stage-2 constant tiles (cos, -sin) of the slot               (global, L1-resident)
};
struct Tiles {      // stage-2 B operands of one slot: {forward, mirror} x {Re, Im} x 2 registers (32 pairs each)
  float fre[2], fim[2], mre[2], mim[2];
};

constexpr int SS = 2 * SC_DSTEP;   // floats per slot in the packed DB stream (both halves)

template <int CTRL, int BANK>
__device__ __forceinline__ float dpp(float old, float src) {   // DPP move: lanes of the enabled banks take permuted src
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, BANK, false));
}
__device__ __forceinline__ float ror8(float x) { return dpp<0x128, 0xf>(0.f, x); }   // row_ror:8 in each 16-lane row

template <int SLOT>
__device__ __forceinline__ void load_slot(SlotOps& o, const float* __restrict__ la, const float* __restrict__ db,
                                          const float* __restrict__ cst, int lane) {
#pragma unroll
  for (int h = 0; h < 2; h++) {
    int pos = 2 * SLOT + h;
    if (pos > SC_NF - 1) pos = SC_NF - 1;          // the 32nd position is all-zero on the DB side: any query rows do
    const float* a = la + pos * 320;
#pragma unroll
    for (int s = 0; s < 5; s++) o.a[h][s] = a[s * 64];
  }
  const f32x4* p = reinterpret_cast<const f32x4*>(db);
  o.b[0] = p[lane];
  o.b[1] = p[64 + lane];
  o.b5 = reinterpret_cast<const float2*>(db + 512)[lane];
  o.c = reinterpret_cast<const float2*>(cst)[SLOT * 64 + lane];
}

// stage 1 of both frequencies of a slot: two independent chains, interleaved (the 16x16x4 MFMA has a 40-cycle
// dependent latency against a 32-cycle issue interval)
__device__ __forceinline__ void stage1(const SlotOps& o, f32x4& ta, f32x4& tb) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  ta = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[0][0], o.b[0][0], z, 0, 0, 0);
  tb = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[1][0], o.b[1][0], z, 0, 0, 0);
#pragma unroll
  for (int s = 1; s < 4; s++) {
    ta = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[0][s], o.b[0][s], ta, 0, 0, 0);
    tb = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[1][s], o.b[1][s], tb, 0, 0, 0);
  }
  ta = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[0][4], o.b5.x, ta, 0, 0, 0);
  tb = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[1][4], o.b5.y, tb, 0, 0, 0);
}

// (A|E), (C|B) of queries a, b -> Re S, -Im S, Re P, Im P with 64 pairs per register: lane (t, c) = pair
// (query 2t + (c >= 8), entry c & 7).  The sign of the Im S tile is irrelevant: only |O| is used.
__device__ __forceinline__ void combine(const f32x4& t, float& fre, float& nfim, float& mre, float& mim) {
  const float x1 = t[0] + ror8(t[1]);          // (A+B | E+C) = (Re S_a | Im P_a)
  const float x2 = t[0] - ror8(t[1]);          // (A-B | E-C) = (Re P_a | -Im S_a)
  const float x3 = t[2] + ror8(t[3]);          // query b
  const float x4 = t[2] - ror8(t[3]);
  fre = dpp<0x128, 0xc>(x1, x3);               // c < 8: Re S_a        c >= 8: Re S_b (from c - 8)
  mim = dpp<0x128, 0x3>(x3, x1);               // c < 8: Im P_a (from c + 8)   c >= 8: Im P_b
  mre = dpp<0x128, 0xc>(x2, x4);
  nfim = dpp<0x128, 0x3>(x4, x2);
}

__device__ __forceinline__ void swap_halves(float xa, float xb, float (&out)[2]) {   // lanes 32-63 of xa <-> lanes 0-31 of xb
  const u32x2 v = __builtin_amdgcn_permlane32_swap(__float_as_uint(xa), __float_as_uint(xb), false, false);
  out[0] = __uint_as_float(v[0]);   // (fa | fb) of pairs with lane group t = 0, 1: queries 0..3
  out[1] = __uint_as_float(v[1]);   // queries 4..7
}

__device__ __forceinline__ void make_tiles(const f32x4& ta, const f32x4& tb, Tiles& t) {
  float fa[4], fb[4];
  combine(ta, fa[0], fa[1], fa[2], fa[3]);
  combine(tb, fb[0], fb[1], fb[2], fb[3]);
  swap_halves(fa[0], fb[0], t.fre);
  swap_halves(fa[1], fb[1], t.fim);
  swap_halves(fa[2], fb[2], t.mre);
  swap_halves(fa[3], fb[3], t.mim);
}

// One pipeline step: stage 2 of slot S (tiles tc), stage 1 + combine + swap of slot S+1 (operands X), request of the
// operands of slot S+2 (into Z).  Two operand buffers alternate; the 16 slots of a DB group are fully unrolled.
template <int S>
__device__ __forceinline__ void slot_step(const SlotOps& X, SlotOps& Z, Tiles& tc, float2& cc, f32x16 (&accE)[2][2],
                                          f32x16 (&accO)[2][2], const float* __restrict__ la, const float*& dbn,
                                          const float* __restrict__ cst, int lane) {
  constexpr int N2 = (S + 2) & (SC_NSLOT - 1);
  __builtin_amdgcn_sched_barrier(0);   // keep the steps apart: hoisting later loads only raises register pressure
  load_slot<N2>(Z, la, dbn, cst, lane);
  dbn += SS;
  f32x4 ta, tb;
  stage1(X, ta, tb);
#pragma unroll
  for (int r = 0; r < 2; r++) {
    accE[r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cc.x, tc.fre[r], accE[r][0], 0, 0, 0);
    accO[r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cc.y, tc.fim[r], accO[r][0], 0, 0, 0);
    accE[r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cc.x, tc.mre[r], accE[r][1], 0, 0, 0);
    accO[r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cc.y, tc.mim[r], accO[r][1], 0, 0, 0);
  }
  make_tiles(ta, tb, tc);
  cc = X.c;
  __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(512, 2) void sc_match_kernel(const float* __restrict__ qpk,  // [2][QG8][31][5][64]
                                                          const float* __restrict__ dpk,  // [2][DG][16][2][640]
                                                          const float* __restrict__ cst,  // [16][64][2]
                                                          float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                          int m, int n, int QG8, int DG, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qg = w >> 1, half = w & 1;             // query group of 8 inside the workgroup, DB half of 8 inside a group of 16
  int b = blockIdx.x;
  const int split = b % nsplit;
  b /= nsplit;
  const int ch = b & 1, qg32 = b >> 1;
  const int g0 = (int)((long long)DG * split / nsplit), g1 = (int)((long long)DG * (split + 1) / nsplit);

  {  // the 4 query groups of this workgroup -> LDS (linear copy; the packed image IS the LDS image)
    const f32x4* src = reinterpret_cast<const f32x4*>(qpk + ((size_t)ch * QG8 + (size_t)qg32 * 4) * SC_QIMG);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    for (int i = tid; i < SC_QIMG; i += 512) dst[i] = src[i];
  }
  __syncthreads();
  if (g0 >= g1) return;

  const float* la = lds + qg * SC_QIMG + lane;
  const float* db = dpk + ((size_t)ch * DG + g0) * SC_DIMG + half * SC_DSTEP;   // this wave's half of every slot
  float* dist = ch ? dist_i : dist_p;
  const int qrow0 = qg32 * 32 + qg * 8;

  // Software pipeline over the flat slot sequence (16 per DB group): while stage 2 of slot i runs, stage 1 + combine of
  // slot i+1 are issued and the operands of slot i+2 are requested.  The packed DB buffer has a readable zero tail.
  SlotOps oA, oB;       // at the top of a group: oA = operands(slot 1), oB = free
  Tiles tc;
  float2 cc;
  {
    f32x4 ta, tb;
    load_slot<0>(oB, la, db, cst, lane);
    load_slot<1>(oA, la, db + SS, cst, lane);
    stage1(oB, ta, tb);
    make_tiles(ta, tb, tc);
    cc = oB.c;
  }
  const float* dbn = db + 2 * SS;

  for (int g = g0; g < g1; g++) {
    f32x16 accE[2][2], accO[2][2];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int v = 0; v < 2; v++)
#pragma unroll
        for (int e = 0; e < 16; e++) { accE[r][v][e] = 0.f; accO[r][v][e] = 0.f; }
#define STEP(S, X, Z) slot_step<S>(X, Z, tc, cc, accE, accO, la, dbn, cst, lane)
    STEP(0, oA, oB);  STEP(1, oB, oA);  STEP(2, oA, oB);  STEP(3, oB, oA);
    STEP(4, oA, oB);  STEP(5, oB, oA);  STEP(6, oA, oB);  STEP(7, oB, oA);
    STEP(8, oA, oB);  STEP(9, oB, oA);  STEP(10, oA, oB); STEP(11, oB, oA);
    STEP(12, oA, oB); STEP(13, oB, oA); STEP(14, oA, oB); STEP(15, oB, oA);
#undef STEP
    // end of the DB group: max over the 120 variants = max_k E + |O| over forward and mirror; 8 x 8 distances
#pragma unroll
    for (int r = 0; r < 2; r++) {
      f32x16 v = __builtin_elementwise_max(accE[r][0] + __builtin_elementwise_abs(accO[r][0]),
                                           accE[r][1] + __builtin_elementwise_abs(accO[r][1]));
      float mx = v[0];
#pragma unroll
      for (int e = 1; e < 16; e++) mx = fmaxf(mx, v[e]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (lane < 32) {
        const int qrow = qrow0 + r * 4 + 2 * (lane >> 4) + ((lane >> 3) & 1);
        const int drow = g * 16 + half * 8 + (lane & 7);
        if (qrow < m && drow < n) dist[(size_t)qrow * n + drow] = 0.5f - 0.5f * mx;   // processSC.m:30
      }
    }
  }
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
  hipLaunchKernelGGL(sc_match_kernel, dim3(base * nsplit), dim3(512), sc_match_lds_bytes(), st, qpk, dpk, cst, d_p,
                     d_i, m, n, QG8, DG, nsplit);
}

}  // namespace pr
