// m2dp_match.hip — all-pairs M2DP distance on gfx950 fp32 MFMA (processM2DP.m:12-22).
//
// Reference arithmetic: diff_full = (1 - hist1*hist2')/2 over ALL 4m x 4n variant rows (192-d per channel,
// rows NOT re-normalised), then d(i,j) = min of the 4x4 block (:17-21).  Here: one fp32 GEMM
// [4m x 192] . [192 x 4n] on v_mfma_f32_32x32x2_f32 with the 4x4 block-min fused into the epilogue
// (max of the dot products: 3 in-lane v_max over the 4 query variants + a quad DPP reduction over the 4 DB
// variants), 12 288 FLOP per (query, entry) pair over both channels.
//
// Layout (m2dp_pack): per channel, per tile of 32 rows (8 signatures x 4 variants): [kq=0..23][lane][4] floats,
// component c of (kq, lane) = row (lane&31) of the tile at k = 2*(4*kq+c) + (lane>>5) — i.e. exactly the A (or B)
// operand register of K-step 4*kq+c of a 32x32x2 MFMA, so operands are read with one 16-byte load per 4 K-steps.
// A workgroup (4 waves as 2x2) owns 4 query tiles (32 queries) resident in LDS (96 KB) and sweeps the DB;
// each wave computes 2 query tiles x 2 DB tiles per step, DB operands straight from HBM/L2 into VGPRs.
#include "kernels.hpp"

namespace pr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T>
__global__ __launch_bounds__(256) void m2dp_pack_kernel(const T* __restrict__ sig, int sigs, float* __restrict__ packed,
                                                         int tiles) {
  // one workgroup per signature (4 variant rows x 384)
  const int sg = blockIdx.x, tid = threadIdx.x;
  const int tile = sg >> 3, e = sg & 7;
  for (int o = tid; o < 4 * 384; o += 256) {
    const int var = o / 384, c = o - var * 384, ch = c / 192, k = c - ch * 192;
    const float v = (float)sig[((size_t)sg * 4 + var) * 384 + c];
    const int row = e * 4 + var;
    const int kstep = k >> 1, lane = ((k & 1) << 5) | row;
    packed[((size_t)ch * tiles + tile) * M2_TILE + ((size_t)(kstep >> 2) * 64 + lane) * 4 + (kstep & 3)] = v;
  }
}

__global__ __launch_bounds__(256, 1) void m2dp_match_kernel(const float* __restrict__ qpk, const float* __restrict__ dpk,
                                                            float* __restrict__ dist_p, float* __restrict__ dist_i,
                                                            int m, int n, int QT, int DT, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = w >> 1, wd = w & 1;   // wave's query-tile pair / DB-tile pair
  int b = blockIdx.x;
  const int split = b % nsplit;
  b /= nsplit;
  const int ch = b & 1, qt4 = b >> 1;               // 4 query tiles per workgroup
  const int DT4 = (DT + 3) / 4;                     // DB swept in steps of 4 tiles (2 per wave column)
  const int s0 = (int)((long long)DT4 * split / nsplit), s1 = (int)((long long)DT4 * (split + 1) / nsplit);
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(qpk + ((size_t)ch * QT + (size_t)qt4 * 4) * M2_TILE);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    for (int i = tid; i < M2_TILE; i += 256) dst[i] = src[i];     // 4 tiles * M2_TILE floats = M2_TILE float4
  }
  __syncthreads();
  const f32x4* la0 = reinterpret_cast<const f32x4*>(lds) + (size_t)(wq * 2) * (M2_TILE / 4) + lane;
  const f32x4* la1 = la0 + (M2_TILE / 4);
  float* dist = ch ? dist_i : dist_p;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (s0 >= s1) return;
  // DB operand stream of this wave: tiles (4 s + 2 wd) and (4 s + 2 wd + 1); per tile 24 x 16-byte loads per lane.
  // Software pipeline: DB operands two K-quads ahead (3 rotating register pairs), query operands (LDS) one ahead,
  // loads pinned in front of the 16 MFMAs of the current K-quad with sched_barrier so hipcc cannot sink them next to
  // their first use; the first two K-quads of the NEXT sweep step are requested before this step's epilogue.
  const f32x4* pb = reinterpret_cast<const f32x4*>(dpk + ((size_t)ch * DT + (size_t)s0 * 4 + wd * 2) * M2_TILE) + lane;
  constexpr int TQ = M2_TILE / 4;            // f32x4 per tile
  f32x4 b0[3], b1[3], a0[2], a1[2];
  b0[0] = pb[0];  b1[0] = pb[TQ];
  b0[1] = pb[64]; b1[1] = pb[TQ + 64];
  for (int s = s0; s < s1; s++) {
    const int dt0 = s * 4 + wd * 2;
    const f32x4* pn = pb + 4 * TQ;           // same wave column, next sweep step (the packed buffer has a readable tail)
    f32x16 acc[2][2];
    a0[0] = la0[0]; a1[0] = la1[0];
#pragma unroll
    for (int kq = 0; kq < 24; kq++) {
      const int cb = kq % 3, nb = (kq + 2) % 3, ca = kq & 1, na = (kq + 1) & 1;
      if (kq + 2 < 24) { b0[nb] = pb[(kq + 2) * 64]; b1[nb] = pb[TQ + (kq + 2) * 64]; }
      else             { b0[nb] = pn[(kq - 22) * 64]; b1[nb] = pn[TQ + (kq - 22) * 64]; }
      if (kq + 1 < 24) { a0[na] = la0[(kq + 1) * 64]; a1[na] = la1[(kq + 1) * 64]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const bool first = (kq == 0 && c == 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[ca][c], b0[cb][c], first ? zero : acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[ca][c], b1[cb][c], first ? zero : acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[ca][c], b0[cb][c], first ? zero : acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[ca][c], b1[cb][c], first ? zero : acc[1][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // 24 K-quads advance the period-3 rotation by 0: b0[0], b0[1] already hold K-quads 0, 1 of the next step
    pb = pn;
    // epilogue: C layout col = lane&31 -> (entry = col>>2, variant = col&3); row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    // -> (query = 2*(reg>>2) + (lane>>5), variant = reg&3).  d = min (1-dot)/2 = 0.5 - 0.5*max dot.
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
          float mx = fmaxf(fmaxf(acc[i][j][gq * 4], acc[i][j][gq * 4 + 1]),
                           fmaxf(acc[i][j][gq * 4 + 2], acc[i][j][gq * 4 + 3]));
          mx = fmaxf(mx, __shfl_xor(mx, 1));
          mx = fmaxf(mx, __shfl_xor(mx, 2));
          if ((lane & 3) == 0) {
            const int qrow = (qt4 * 4 + wq * 2 + i) * 8 + 2 * gq + (lane >> 5);
            const int drow = (dt0 + j) * 8 + ((lane & 31) >> 2);
            if (qrow < m && drow < n) dist[(size_t)qrow * n + drow] = 0.5f - 0.5f * mx;   // processM2DP.m:15,19
          }
        }
  }
}

}  // namespace

void launch_m2dp_pack(hipStream_t st, const void* sig, int dtype, int sigs, float* packed, int tiles) {
  if (sigs <= 0) return;
  if (dtype == 0)
    hipLaunchKernelGGL(m2dp_pack_kernel<double>, dim3(sigs), dim3(256), 0, st, (const double*)sig, sigs, packed, tiles);
  else
    hipLaunchKernelGGL(m2dp_pack_kernel<float>, dim3(sigs), dim3(256), 0, st, (const float*)sig, sigs, packed, tiles);
}

void launch_m2dp_match(hipStream_t st, const float* qpk, int m, const float* dpk, int n, float* d_p, float* d_i) {
  if (m <= 0 || n <= 0) return;
  const int QT = m2_qtiles(m), DT = m2_tiles(n);
  const int base = (QT / 4) * 2, DT4 = (DT + 3) / 4;
  int nsplit = (1024 + base - 1) / base;
  if (nsplit > DT4 / 4) nsplit = DT4 / 4;
  if (nsplit < 1) nsplit = 1;
  const size_t lds = (size_t)4 * M2_TILE * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(m2dp_match_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(m2dp_match_kernel, dim3(base * nsplit), dim3(256), lds, st, qpk, dpk, d_p, d_i, m, n, QT, DT,
                     nsplit);
}

}  // namespace pr
