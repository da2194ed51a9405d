// plain_match.hip — the two remaining matchers of match_signatures/run_test.m:26-36 (types "gist" and "bow": one distance
// matrix, no z-score fusion) on gfx950.  SURVEY.md §8 row f4.
//   gist : d(i,j) = sum_c (h1[i,c] - h2[j,c])^2 (processGIST.m:1-10), evaluated as written - differences first - in fp64
//          (v_fma_f64 runs at the plain fp32 rate on gfx950; the GEMM form |a|^2 + |b|^2 - 2 a.b cancels for near
//          neighbours, which are the ones that matter).  64 x 64 tile per workgroup, 4 x 4 per thread, K in chunks of 16.
//   bow  : DBoW2 L1 score of two sparse vectors by a merge of their sorted word-id lists (processBoW.m:1-38; rows
//          alternate ids | weights, padded with -1, test_bow.cpp:147-162), d = 1 - score.  One thread per (query, entry)
//          pair, the query's two lists in LDS.  The reference's loop guards never look at the last column - kept.
#include "kernels.hpp"

namespace pr {
namespace {

__global__ __launch_bounds__(256) void gist_distance_kernel(const double* __restrict__ h1, const double* __restrict__ h2,
                                                             int m, int n, int cols, float* __restrict__ dist) {
  __shared__ double A[16][65], B[16][65];
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  double acc[4][4] = {};
  for (int k0 = 0; k0 < cols; k0 += 16) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int e = tid + 256 * t, r = e >> 4, c = e & 15;
      A[c][r] = (i0 + r < m && k0 + c < cols) ? h1[(size_t)(i0 + r) * cols + k0 + c] : 0.0;
      B[c][r] = (j0 + r < n && k0 + c < cols) ? h2[(size_t)(j0 + r) * cols + k0 + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; c++) {
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { a[u] = A[c][ti * 4 + u]; b[u] = B[c][tj * 4 + u]; }
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) { const double d = a[u] - b[v]; acc[u][v] = __builtin_fma(d, d, acc[u][v]); }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; u++)
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int i = i0 + ti * 4 + u, j = j0 + tj * 4 + v;
      if (i < m && j < n) dist[(size_t)i * n + j] = (float)acc[u][v];
    }
}

__global__ __launch_bounds__(256) void bow_distance_kernel(const double* __restrict__ h1, const double* __restrict__ h2,
                                                            int m, int n, int cols, float* __restrict__ dist) {
  extern __shared__ double ql[];              // [2][cols]: ids | weights of query i
  const int i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  for (int c = threadIdx.x; c < 2 * cols; c += 256) ql[c] = h1[(size_t)(2 * i) * cols + c];
  __syncthreads();
  if (j >= n) return;
  const double* i2 = h2 + (size_t)(2 * j) * cols;
  const double* v2 = i2 + cols;
  int a = 1, b = 1;                           // 1-based cursors, as in the reference
  double score = 0.0;
  while (a < cols && ql[a - 1] > -1 && b < cols && i2[b - 1] > -1) {                  // processBoW.m:23
    const double ia = ql[a - 1], ib = i2[b - 1];
    if (ia == ib) {
      const double va = ql[cols + a - 1], vb = v2[b - 1];
      score = score + fabs(va - vb) - fabs(va) - fabs(vb);                            // :25
      a++; b++;
    } else if (ia < ib) a++;
    else b++;
  }
  dist[(size_t)i * n + j] = (float)(1.0 - (-score / 2.0));                            // :37, :14
}

}  // namespace

void launch_gist_distance(hipStream_t st, const double* h1, int m, const double* h2, int n, int cols, float* dist) {
  if (m <= 0 || n <= 0) return;
  hipLaunchKernelGGL(gist_distance_kernel, dim3((n + 63) / 64, (m + 63) / 64), dim3(256), 0, st, h1, h2, m, n, cols, dist);
}

void launch_bow_distance(hipStream_t st, const double* h1, int m, const double* h2, int n, int cols, float* dist) {
  if (m <= 0 || n <= 0) return;
  const size_t lds = (size_t)2 * cols * sizeof(double);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bow_distance_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(bow_distance_kernel, dim3((n + 255) / 256, m), dim3(256), lds, st, h1, h2, m, n, cols, dist);
}

}  // namespace pr
