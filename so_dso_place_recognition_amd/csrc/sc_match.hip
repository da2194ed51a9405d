// sc_match.hip — all-pairs Scan-Context distance on gfx950 fp32 MFMA (processSC.m:22-33).
//
// Reference arithmetic: d(i,j) = min over the 120 variants (60 column rotations + 60 mirrored rotations,
// permute_sc processSC.m:37-45) of (1 - <variant_k(q_i), d_j>)/2 on L2-normalised 1200-vectors, per channel.
// The 120 inner products are the circular cross-correlation (forward) and circular convolution (mirror) of
// the 60 sectors summed over the 20 rings, so with the per-ring sector spectra Q_r[f], D_r[f] (sc_pack.hip)
//   S_f = sum_r Q_r[f] conj(D_r[f])   (forward)        P_f = sum_r Q_r[f] D_r[f]   (mirror)
//   dot_fwd[k] = sum_f w_f ( Re S_f cos(2 pi f k/60) - Im S_f sin(2 pi f k/60) ),  same with P_f for the mirror
// (w_0 = w_30 = 1, else 2; the 1/60 is folded into the packed spectra).  39 680 FLOP per (query, entry) pair
// instead of the dense 576 000.
//
// Mapping to the matrix cores (one wave = 8 queries x 16 DB entries, all 31 frequencies):
//   stage 1  v_mfma_f32_16x16x4_f32, K = 20 rings:  rows = {Re,Im} x 8 queries, cols = 16 entries
//            T1 = [Qre;Qim] . Dre^T          = (A | C)
//            T2 = [Qim;-Qre] . Dim^T         = (B | -E)      (row operand = DPP row_ror:8 of T1's, sign-flipped)
//            F = T1 + T2 = (Re S_f | Im S_f)    M = T1 - T2 = (Re P_f | Im P_f)        (8 VALU)
//   stage 2  v_mfma_f32_32x32x2_f32, K = {Re,Im}:  the C/D register r of stage 1 holds, lane for lane, the
//            B operand (k = lane>>5, pair = lane&31) of a 32x32x2 MFMA whose A operand is the constant
//            [shift][cos | -sin] tile, so stage 1 feeds stage 2 without any data movement:
//            acc[r][tile][fwd|mir] (32 shifts x 32 pairs) += C_f[tile] . F[r]   (16 MFMAs per frequency)
//   epilogue max over the 60 shifts (15 in-lane v_max + one cross-half) -> d = 0.5 - 0.5*max.
// A workgroup (4 waves, one per SIMD, <=512 VGPR+AGPR each) keeps the spectra of 32 queries of one channel
// resident in LDS (158 720 B) and sweeps a range of the DB; the DB operand stream goes HBM/L2 -> VGPR directly
// (all 4 waves read the same 2.5 KB per frequency, so 3 of 4 hit L1).  Bound: MFMA fp32 (157.3 TFLOP/s).
#include "kernels.hpp"

namespace pr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Ops {        // operands of one frequency step
  float a[5];       // query rows, K-steps s = 0..4                    (LDS)
  f32x4 bre, bim;   // DB Re / Im, K-steps 0..3                        (global)
  float2 b4;        // DB (Re, Im) of K-step 4                         (global)
  float c0, c1;     // stage-2 constant tiles (shifts 0..31, 32..63)    (global, L1-resident)
};

__device__ __forceinline__ float ror8(float x) {  // DPP row_ror:8 inside each 16-lane row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));
}

__device__ __forceinline__ void load_ops(Ops& o, const float* __restrict__ la, const float* __restrict__ db,
                                         const float* __restrict__ cst, int f, int lane) {
  const float* a = la + f * 320;
#pragma unroll
  for (int s = 0; s < 5; s++) o.a[s] = a[s * 64];
  const f32x4* p = reinterpret_cast<const f32x4*>(db);
  o.bre = p[lane];
  o.bim = p[64 + lane];
  o.b4 = reinterpret_cast<const float2*>(db + 512)[lane];
  o.c0 = cst[f * 128 + lane];
  o.c1 = cst[f * 128 + 64 + lane];
}

// Stage 1 in VGPR form.  The 256 accumulators of stage 2 fill the whole AccVGPR half of the register file, and
// hipcc selects one MFMA form per function (AGPR C/D here), so the two small stage-1 accumulators are kept in
// ArchVGPRs by hand: ONE asm statement with the ten MFMAs, its own wait states inside (cdna_hip_programming.md
// §5.7): s_nop 1 covers VALU-written A operands (the DPP products), the trailing s_nop 10 covers the 8-pass
// MFMA D -> VALU read of the add/sub that follows; back-to-back SrcC == vDst chains need none.
__device__ __forceinline__ void stage1(const Ops& o, float sgn, f32x4& F, f32x4& M) {
  f32x4 t1, t2;
  const float r0 = ror8(o.a[0]) * sgn, r1 = ror8(o.a[1]) * sgn, r2 = ror8(o.a[2]) * sgn, r3 = ror8(o.a[3]) * sgn,
              r4 = ror8(o.a[4]) * sgn;
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
        "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4),                              // %7..%11  rotated, sign-flipped rows
        "v"(o.bre[0]), "v"(o.bre[1]), "v"(o.bre[2]), "v"(o.bre[3]), "v"(o.b4.x),  // %12..%16 DB Re
        "v"(o.bim[0]), "v"(o.bim[1]), "v"(o.bim[2]), "v"(o.bim[3]), "v"(o.b4.y)); // %17..%21 DB Im
  F = t1 + t2;
  M = t1 - t2;
}

__global__ __launch_bounds__(256, 1) void sc_match_kernel(const float* __restrict__ qpk,  // [2][QG8][31][5][64]
                                                          const float* __restrict__ dpk,  // [2][DG][31][640]
                                                          const float* __restrict__ cst,  // [31][2][64]
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
  const float* db = dpk + ((size_t)ch * DG + g0) * SC_DIMG;
  float* dist = ch ? dist_i : dist_p;
  const float sgn = ((lane & 15) >= 8) ? -1.0f : 1.0f;

  // software pipeline: operands are loaded two steps ahead, stage 1 runs one step ahead of stage 2.
  // The packed DB buffer carries a readable tail of >= 2 steps, so the prefetch needs no bounds branch.
  Ops o1, o2;
  f32x4 F, M, Fn, Mn;
  float c0, c1;
  {
    Ops o0;
    load_ops(o0, la, db, cst, 0, lane);
    stage1(o0, sgn, F, M);
    c0 = o0.c0; c1 = o0.c1;
  }
  load_ops(o1, la, db + SC_DSTEP, cst, 1, lane);
  const float* dbn = db + 2 * SC_DSTEP;      // operands of step t+2
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  for (int g = g0; g < g1; g++) {
    f32x16 acc[4][2][2];
    // ---- f = 0: accumulators start from the MFMA's zero C operand (no 256-register clear)
    load_ops(o2, la, dbn, cst, 2, lane);
    dbn += SC_DSTEP;
    stage1(o1, sgn, Fn, Mn);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      acc[r][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, F[r], zero, 0, 0, 0);
      acc[r][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, M[r], zero, 0, 0, 0);
      acc[r][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, F[r], zero, 0, 0, 0);
      acc[r][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, M[r], zero, 0, 0, 0);
    }
    F = Fn; M = Mn; c0 = o1.c0; c1 = o1.c1; o1 = o2;
    // ---- f = 1..30
    for (int f = 1; f < SC_NF; f++) {
      int f2 = f + 2;                          // frequency of the step being prefetched
      if (f2 >= SC_NF) f2 -= SC_NF;
      load_ops(o2, la, dbn, cst, f2, lane);
      dbn += SC_DSTEP;
      stage1(o1, sgn, Fn, Mn);                 // stage 1 of step t+1
#pragma unroll
      for (int r = 0; r < 4; r++) {            // stage 2 of step t
        acc[r][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, F[r], acc[r][0][0], 0, 0, 0);
        acc[r][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, M[r], acc[r][0][1], 0, 0, 0);
        acc[r][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, F[r], acc[r][1][0], 0, 0, 0);
        acc[r][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, M[r], acc[r][1][1], 0, 0, 0);
      }
      F = Fn; M = Mn; c0 = o1.c0; c1 = o1.c1; o1 = o2;
    }
    // ---- end of the DB group: max over the 120 variants, write 8 x 16 distances
#pragma unroll
    for (int r = 0; r < 4; r++) {
      f32x16 v = __builtin_elementwise_max(__builtin_elementwise_max(acc[r][0][0], acc[r][0][1]),
                                           __builtin_elementwise_max(acc[r][1][0], acc[r][1][1]));
      float mx = v[0];
#pragma unroll
      for (int e = 1; e < 16; e++) mx = fmaxf(mx, v[e]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (lane < 32) {
        const int qrow = qg32 * 32 + w * 8 + ((lane < 16) ? r : 4 + r);
        const int drow = g * 16 + (lane & 15);
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
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_match_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc_match_lds_bytes());
    attr_set = true;
  }
  hipLaunchKernelGGL(sc_match_kernel, dim3(base * nsplit), dim3(256), sc_match_lds_bytes(), st, qpk, dpk, cst, d_p,
                     d_i, m, n, QG8, DG, nsplit);
}

}  // namespace pr
