// delight.hip — DELIGHT descriptor (DELIGHT/DELIGHT.cpp:8-24) and its chi-square matcher (match_signatures/processDELIGHT.m:1-38)
// on gfx950.  SURVEY.md §8 row f3 (a sibling descriptor in the same slot as SC / M2DP; no z-score fusion: run_test.m:26-41).
//
//   delight_gen   : one workgroup per cloud, after cloud_frames (PCA): centre + rotate in fp64, then exactly the reference's
//                   float casts: hist = 8*(|p| > 10) + 4*(z > 0) + 2*(y > 0) + (x > 0), bin = int(intensity); 16 x 256 u32
//                   LDS histogram with LDS atomics -> counts as f64.  HBM-bound (28 B per point).
//   delight_match : d(i,j) = min over the 4 octant permutations of mean over non-empty bins of 2 (a-b)^2 / (a+b).
//                   One wave per query (every lane owns 4 histogram columns x 16 rows = 64 floats in registers), 4 queries
//                   per workgroup, the entry rows staged through LDS; the row permutations are register renames.
//                   VALU-bound (4 x 4096 terms per pair): packed fp32, no compares, one v_rcp_f32 per two terms.
#include "kernels.hpp"

namespace pr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

// Empty-bin masks of the packed histograms: word (r >> 3) of lane l of a signature has bit 4 (r & 7) + c set iff element
// (row r, column 4 l + c) is zero - the lane layout of the matcher.  One thread per (signature, lane).
__global__ __launch_bounds__(256) void delight_mask_kernel(const float* __restrict__ packed, int sigs, unsigned* __restrict__ mask) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)sigs * 64) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(packed) + (t >> 6) * 1024 + (t & 63);
  unsigned w[2] = {0u, 0u};
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const f32x4 v = p[r * 64];
#pragma unroll
    for (int c = 0; c < 4; c++) w[r >> 3] |= (v[c] == 0.f ? 1u : 0u) << (4 * (r & 7) + c);
  }
  mask[2 * t] = w[0];
  mask[2 * t + 1] = w[1];
}

// nibble r of the result = nibble r ^ X of v (X = 0, 5, 6, 3: the four octant permutations are XORs of the row index)
template <int X>
__device__ __forceinline__ unsigned nibble_xor(unsigned v) {
  if (X & 1) v = ((v & 0x0f0f0f0fu) << 4) | ((v >> 4) & 0x0f0f0f0fu);
  if (X & 2) v = ((v & 0x00ff00ffu) << 8) | ((v >> 8) & 0x00ff00ffu);
  if (X & 4) v = (v << 16) | (v >> 16);
  return v;
}

// All four permutations of one query term at a time: rows r of the query against rows r ^ 0, r ^ 5, r ^ 6, r ^ 3 of the
// entry (processDELIGHT.m:2-5: {0..15}, {5,4,7,6,1,0,3,2,..}, {6,7,4,5,2,3,0,1,..}, {3,2,1,0,7,6,5,4,..}, the second eight
// likewise + 8).  A holds a + 2^-30, so that s = a + b and d = a - b are the exact float sums wherever a or b is a count >= 1
// and (2^-30, 2^-30) - a term of 2^-30, invisible next to 1 - where both are empty: no compare, no select, packed fp32
// adds / multiplies.  v_rcp_f32 runs at a quarter of the packed rate, so the four divisions share ONE reciprocal
// (Montgomery's trick): R = 1 / (s0 s1 s2 s3), 1/s0 = (R s2 s3) s1, ... - 9 packed multiplies instead of 3 reciprocals.
// 2^-120 <= s0 s1 s2 s3 <= 2^72 for histogram counts: no underflow, no overflow.
__device__ __forceinline__ void chi2_perm4(const f32x4 (&A)[16], const f32x4 (&B)[16], f32x2 (&acc)[4]) {
#pragma unroll
  for (int r = 0; r < 16; r++) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const f32x2 a = {A[r][2 * h], A[r][2 * h + 1]};
      const f32x2 b0 = {B[r][2 * h], B[r][2 * h + 1]}, b1 = {B[r ^ 5][2 * h], B[r ^ 5][2 * h + 1]};
      const f32x2 b2 = {B[r ^ 6][2 * h], B[r ^ 6][2 * h + 1]}, b3 = {B[r ^ 3][2 * h], B[r ^ 3][2 * h + 1]};
      const f32x2 s0 = a + b0, s1 = a + b1, s2 = a + b2, s3 = a + b3;
      const f32x2 d0 = a - b0, d1 = a - b1, d2 = a - b2, d3 = a - b3;
      const f32x2 p01 = s0 * s1, p23 = s2 * s3, P = p01 * p23;
      const f32x2 R = {__builtin_amdgcn_rcpf(P[0]), __builtin_amdgcn_rcpf(P[1])};
      const f32x2 q23 = R * p23, q01 = R * p01;                                        // 1 / (s0 s1), 1 / (s2 s3)
      acc[0] = __builtin_elementwise_fma(d0 * d0, q23 * s1, acc[0]);                   // processDELIGHT.m:24-28 (x 2 at the end)
      acc[1] = __builtin_elementwise_fma(d1 * d1, q23 * s0, acc[1]);
      acc[2] = __builtin_elementwise_fma(d2 * d2, q01 * s3, acc[2]);
      acc[3] = __builtin_elementwise_fma(d3 * d3, q01 * s2, acc[3]);
    }
  }
}

// Workgroup = 4 queries (one per wave, 64 floats per lane in registers) x a range of entries; the entry rows go through a
// double-buffered 16 KB LDS tile (one global read per workgroup instead of one per wave, requested one row ahead).
__global__ __launch_bounds__(256, 2) void delight_match_kernel(const float* __restrict__ q, const float* __restrict__ db,
                                                                const unsigned* __restrict__ dbmask, float* __restrict__ dist,
                                                                int m, int n, int nsplit) {
  __shared__ f32x4 rowbuf[2][1024];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = (blockIdx.x / nsplit) * 4 + w, split = blockIdx.x % nsplit;
  const int j0 = (int)((long long)n * split / nsplit), j1 = (int)((long long)n * (split + 1) / nsplit);
  if (j0 >= j1) return;
  const bool valid = i < m;
  f32x4 A[16];
  unsigned za0 = 0u, za1 = 0u;
  {
    const f32x4* pa = reinterpret_cast<const f32x4*>(q + (size_t)(valid ? i : 0) * 4096) + lane;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      A[r] = pa[r * 64];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        (r < 8 ? za0 : za1) |= (A[r][c] == 0.f ? 1u : 0u) << (4 * (r & 7) + c);
        A[r][c] += 0x1p-30f;
      }
    }
  }
  f32x4 st[4];
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(db + (size_t)j0 * 4096) + tid;
#pragma unroll
    for (int k = 0; k < 4; k++) rowbuf[0][tid + 256 * k] = src[256 * k];
  }
  __syncthreads();
  for (int j = j0; j < j1; j++) {
    const int cur = (j - j0) & 1;
    if (j + 1 < j1) {
      const f32x4* src = reinterpret_cast<const f32x4*>(db + (size_t)(j + 1) * 4096) + tid;
#pragma unroll
      for (int k = 0; k < 4; k++) st[k] = src[256 * k];
    }
    const unsigned zb0 = dbmask[(size_t)j * 128 + 2 * lane], zb1 = dbmask[(size_t)j * 128 + 2 * lane + 1];
    f32x4 B[16];
#pragma unroll
    for (int r = 0; r < 16; r++) B[r] = rowbuf[cur][r * 64 + lane];
    f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    chi2_perm4(A, B, acc);
    float ts[4];
    // bins empty in both histograms do not count (:25-28): two 13-bit totals per word
    unsigned e01 = (unsigned)(__builtin_popcount(za0 & zb0) + __builtin_popcount(za1 & zb1)) |
                   (unsigned)(__builtin_popcount(za0 & nibble_xor<5>(zb0)) + __builtin_popcount(za1 & nibble_xor<5>(zb1))) << 16;
    unsigned e23 = (unsigned)(__builtin_popcount(za0 & nibble_xor<6>(zb0)) + __builtin_popcount(za1 & nibble_xor<6>(zb1))) |
                   (unsigned)(__builtin_popcount(za0 & nibble_xor<3>(zb0)) + __builtin_popcount(za1 & nibble_xor<3>(zb1))) << 16;
#pragma unroll
    for (int k = 0; k < 4; k++) ts[k] = acc[k][0] + acc[k][1];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
      for (int k = 0; k < 4; k++) ts[k] += __shfl_xor(ts[k], d);
      e01 += (unsigned)__shfl_xor((int)e01, d);
      e23 += (unsigned)__shfl_xor((int)e23, d);
    }
    const float tc[4] = {4096.f - (float)(e01 & 0xffffu), 4096.f - (float)(e01 >> 16), 4096.f - (float)(e23 & 0xffffu),
                         4096.f - (float)(e23 >> 16)};
    float best = __builtin_inff();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float v = 2.f * ts[k] / tc[k];                                         // :30 (0/0 = NaN never wins, :31)
      if (best > v) best = v;
    }
    if (lane == 0 && valid) dist[(size_t)i * n + j] = best;
    if (j + 1 < j1) {
#pragma unroll
      for (int k = 0; k < 4; k++) rowbuf[cur ^ 1][tid + 256 * k] = st[k];
    }
    __syncthreads();
  }
}

}  // namespace

void launch_delight_gen(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int N,
                        const double* frames, double* out) {
  if (N <= 0) return;
  hipLaunchKernelGGL(delight_gen_kernel, dim3(N), dim3(256), 0, st, xyz, inten, offs, frames, out);
}

void launch_delight_pack(hipStream_t st, const void* sig, int dtype, int sigs, float* packed, unsigned* mask) {
  if (sigs <= 0) return;
  const size_t n = (size_t)sigs * 4096;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (dtype == 0) hipLaunchKernelGGL(delight_pack_kernel<double>, dim3(blocks), dim3(256), 0, st, (const double*)sig, n, packed);
  else hipLaunchKernelGGL(delight_pack_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)sig, n, packed);
  hipLaunchKernelGGL(delight_mask_kernel, dim3((unsigned)(((size_t)sigs * 64 + 255) / 256)), dim3(256), 0, st, packed, sigs, mask);
}

void launch_delight_match(hipStream_t st, const float* q, int m, const float* db, const unsigned* dbmask, int n, float* dist) {
  if (m <= 0 || n <= 0) return;
  const int qb = (m + 3) / 4;
  int nsplit = (2048 + qb - 1) / qb;                  // >= 4 rounds of 2 workgroups per CU
  if (nsplit > (n + 7) / 8) nsplit = (n + 7) / 8;     // >= 8 entries per workgroup
  if (nsplit < 1) nsplit = 1;
  hipLaunchKernelGGL(delight_match_kernel, dim3((unsigned)qb * nsplit), dim3(256), 0, st, q, db, dbmask, dist, m, n, nsplit);
}

}  // namespace pr
