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
  for (int s = s0; s < s1; s++) {
    const int dt0 = s * 4 + wd * 2;                                // this wave's two DB tiles (buffer is padded)
    const f32x4* pb0 = reinterpret_cast<const f32x4*>(dpk + ((size_t)ch * DT + dt0) * M2_TILE) + lane;
    const f32x4* pb1 = pb0 + (M2_TILE / 4);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    f32x4 b0 = pb0[0], b1 = pb1[0];
#pragma unroll 4
    for (int kq = 0; kq < 24; kq++) {
      const f32x4 a0 = la0[kq * 64], a1 = la1[kq * 64];
      const int kn = (kq < 23) ? kq + 1 : 23;
      const f32x4 nb0 = pb0[kn * 64], nb1 = pb1[kn * 64];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b0[c], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b1[c], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b0[c], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b1[c], acc[1][1], 0, 0, 0);
      }
      b0 = nb0;
      b1 = nb1;
    }
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
  const int QT = ((m2_tiles(m) + 3) / 4) * 4, DT = m2_tiles(n);
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
