// delight.hip — DELIGHT descriptor (DELIGHT/DELIGHT.cpp:8-24) and its chi-square matcher (match_signatures/processDELIGHT.m:1-38)
// on gfx950.  SURVEY.md §8 row f3 (a sibling descriptor in the same slot as SC / M2DP; no z-score fusion: run_test.m:26-41).
//
//   delight_gen   : one workgroup per cloud, after cloud_frames (PCA): centre + rotate in fp64, then exactly the reference's
//                   float casts: hist = 8*(|p| > 10) + 4*(z > 0) + 2*(y > 0) + (x > 0), bin = int(intensity); 16 x 256 u32
//                   LDS histogram with LDS atomics -> counts as f64.  HBM-bound (28 B per point).
//   delight_match : d(i,j) = min over the 4 octant permutations of mean over non-empty bins of 2 (a-b)^2 / (a+b).
//                   One wave per (query, entry) pair stream: every lane owns 4 histogram columns x 16 rows of the query in
//                   registers (64 floats) and loads the same columns of the entry with 16 coalesced 16-byte loads; the row
//                   permutations are register renames.  VALU-bound: 4 x 4096 terms per pair, v_rcp_f32 for the division.
#include "kernels.hpp"

namespace pr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void delight_gen_kernel(const double* __restrict__ xyz, const float* __restrict__ inten,
                                                           const int64_t* __restrict__ offs, const double* __restrict__ frames,
                                                           double* __restrict__ out) {
  __shared__ unsigned int hist[16 * 256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int64_t o0 = offs[c];
  const int64_t P = offs[c + 1] - o0;
  for (int b = tid; b < 16 * 256; b += 256) hist[b] = 0u;
  __syncthreads();
  const double* f = frames + (size_t)c * 16;
  const double mx = f[0], my = f[1], mz = f[2];
  const double e00 = f[3], e01 = f[4], e02 = f[5], e10 = f[6], e11 = f[7], e12 = f[8], e20 = f[9], e21 = f[10], e22 = f[11];
  const double* p = xyz + 3 * o0;
  const float* it = inten + o0;
  for (int64_t i = tid; i < P; i += 256) {
    const double x = p[3 * i] - mx, y = p[3 * i + 1] - my, z = p[3 * i + 2] - mz;   // pts_align.h:24-26
    const double ax = (x * e00 + y * e01) + z * e02;                                // :37-39
    const double ay = (x * e10 + y * e11) + z * e12;
    const double az = (x * e20 + y * e21) + z * e22;
    const float fx = (float)ax, fy = (float)ay, fz = (float)az;                     // DELIGHT.cpp:17-19
    const float d = (float)sqrt((ax * ax + ay * ay) + az * az);                     // :20
    const int h = 8 * ((double)d > 10.0) + 4 * (fz > 0) + 2 * (fy > 0) + 1 * (fx > 0);   // :23
    const int bin = (int)it[i];                                                      // :24
    if (bin < 0 || bin >= 256) continue;
    atomicAdd(&hist[h * 256 + bin], 1u);
  }
  __syncthreads();
  double* o = out + (size_t)c * 4096;
  for (int b = tid; b < 16 * 256; b += 256) o[b] = (double)hist[b];
}

template <typename T>
__global__ __launch_bounds__(256) void delight_pack_kernel(const T* __restrict__ sig, size_t n, float* __restrict__ packed) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) packed[i] = (float)sig[i];
}

__device__ __forceinline__ void chi2_perm(const f32x4 (&A)[16], const f32x4 (&B)[16], const int (&mut)[16], float& ts, float& tc) {
  ts = 0.f; tc = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r++) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const float a = A[r][c], b = B[mut[r]][c], s = a + b, d = a - b;
      const float t = 2.f * d * d * __builtin_amdgcn_rcpf(s);
      if (s > 0.f) { ts += t; tc += 1.f; }                                         // processDELIGHT.m:24-28
    }
  }
}

__global__ __launch_bounds__(256) void delight_match_kernel(const float* __restrict__ q, const float* __restrict__ db,
                                                             float* __restrict__ dist, int m, int n, int nsplit) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
  const int j0 = (int)((long long)n * split / nsplit), j1 = (int)((long long)n * (split + 1) / nsplit);
  f32x4 A[16];
  const f32x4* pa = reinterpret_cast<const f32x4*>(q + (size_t)i * 4096) + lane;
#pragma unroll
  for (int r = 0; r < 16; r++) A[r] = pa[r * 64];
  constexpr int M0[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};       // processDELIGHT.m:2-5 (0-based)
  constexpr int M1[16] = {5, 4, 7, 6, 1, 0, 3, 2, 13, 12, 15, 14, 9, 8, 11, 10};
  constexpr int M2[16] = {6, 7, 4, 5, 2, 3, 0, 1, 14, 15, 12, 13, 10, 11, 8, 9};
  constexpr int M3[16] = {3, 2, 1, 0, 7, 6, 5, 4, 11, 10, 9, 8, 15, 14, 13, 12};
  for (int j = j0 + w; j < j1; j += 4) {
    f32x4 B[16];
    const f32x4* pb = reinterpret_cast<const f32x4*>(db + (size_t)j * 4096) + lane;
#pragma unroll
    for (int r = 0; r < 16; r++) B[r] = pb[r * 64];
    float ts[4], tc[4];
    chi2_perm(A, B, M0, ts[0], tc[0]);
    chi2_perm(A, B, M1, ts[1], tc[1]);
    chi2_perm(A, B, M2, ts[2], tc[2]);
    chi2_perm(A, B, M3, ts[3], tc[3]);
#pragma unroll
    for (int k = 0; k < 4; k++)
      for (int d = 32; d > 0; d >>= 1) { ts[k] += __shfl_xor(ts[k], d); tc[k] += __shfl_xor(tc[k], d); }
    float best = __builtin_inff();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float v = ts[k] / tc[k];                                               // :30 (0/0 = NaN never wins, :31)
      if (best > v) best = v;
    }
    if (lane == 0) dist[(size_t)i * n + j] = best;
  }
}

}  // namespace

void launch_delight_gen(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int N,
                        const double* frames, double* out) {
  if (N <= 0) return;
  hipLaunchKernelGGL(delight_gen_kernel, dim3(N), dim3(256), 0, st, xyz, inten, offs, frames, out);
}

void launch_delight_pack(hipStream_t st, const void* sig, int dtype, int sigs, float* packed) {
  if (sigs <= 0) return;
  const size_t n = (size_t)sigs * 4096;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (dtype == 0) hipLaunchKernelGGL(delight_pack_kernel<double>, dim3(blocks), dim3(256), 0, st, (const double*)sig, n, packed);
  else hipLaunchKernelGGL(delight_pack_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)sig, n, packed);
}

void launch_delight_match(hipStream_t st, const float* q, int m, const float* db, int n, float* dist) {
  if (m <= 0 || n <= 0) return;
  int nsplit = (4096 + m - 1) / m;
  if (nsplit > (n + 3) / 4) nsplit = (n + 3) / 4;
  if (nsplit < 1) nsplit = 1;
  hipLaunchKernelGGL(delight_match_kernel, dim3((unsigned)m * nsplit), dim3(256), 0, st, q, db, dist, m, n, nsplit);
}

}  // namespace pr
