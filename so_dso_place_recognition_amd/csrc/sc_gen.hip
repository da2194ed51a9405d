// sc_gen.hip — PCA alignment (utils/pts_align.h:7-46) and Scan-Context signature (SC/SC.cpp:12-76) on gfx950.
//
// Two kernels, one workgroup per cloud each (compiled with -ffp-contract=off so products/sums round as on the CPU):
//   cloud_frames : one streaming pass over the cloud: fp64 raw moments (sum p, sum p p^T) by waves 0..6 with a
//                  fixed reduction tree, while wave 7 walks the intensities IN INPUT ORDER accumulating the
//                  reference's FLOAT sequential average (SC.cpp:60-64; a tree sum differs at ~3e-4 and flips
//                  bins; the loads are wave-uniform -> scalar loads, the chain is one v_add_f32 per
//                  point).  Then mean, scatter matrix cov = sum pp^T - P mean mean^T (un-normalised as :30),
//                  3x3 symmetric Jacobi eigen-solver, eigenvalues ascending, canonical signs (N3)
//                  -> frames[c] = {mean[3], v0[3], v1[3], v2[3], ave, pad} (16 doubles).
//   sc_bin       : second pass: centre, rotate (same association as the oracle), sector = floor((atan2(z,y)+pi)*60/2pi),
//                  ring = floor(sqrt(y^2+z^2)*20/max_rho), idx = sector*20 + ring, dropped iff idx >= 1200 (the
//                  ring-overflow aliasing of SC.cpp:39-44 is kept, H3); LDS-resident 1200-bin grids with LDS
//                  atomics: count (u32 add), min/max of x (order-preserving u64 keys), fp64 intensity sum; epilogue
//                  max-min and mean > ave ? 1 : 0 (:67-75) -> out[c][2400].
// Bound: HBM (28 B per point per pass); the points are never written back.
#include "kernels.hpp"

namespace pr {
namespace {

constexpr int FT = 512;   // threads of cloud_frames (8 waves: 7 reduce, 1 walks the float average)

__device__ void jacobi_eig3(double a[3][3], double v[3][3]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; sweep++) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    const double dia = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off == 0.0 || off <= 1e-36 * dia) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < 3; k++) {
          const double x = a[k][p], y = a[k][q];
          a[k][p] = cs * x - sn * y;
          a[k][q] = sn * x + cs * y;
        }
        for (int k = 0; k < 3; k++) {
          const double x = a[p][k], y = a[q][k];
          a[p][k] = cs * x - sn * y;
          a[q][k] = sn * x + cs * y;
        }
        for (int k = 0; k < 3; k++) {
          const double x = v[k][p], y = v[k][q];
          v[k][p] = cs * x - sn * y;
          v[k][q] = sn * x + cs * y;
        }
      }
  }
}

__global__ __launch_bounds__(FT) void cloud_frames_kernel(const double* __restrict__ xyz, const float* __restrict__ inten,
                                                           const int64_t* __restrict__ offs, double* __restrict__ frames) {
  __shared__ double red[7][9];
  __shared__ float ave_s;
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t o0 = offs[c];
  const int64_t P = offs[c + 1] - o0;
  if (w == 7) {
    // SC.cpp:60-64 / M2DP.cpp:77-81: float accumulator, input order.  Every lane computes the same chain.
    const float* it = inten + o0;
    float ave = 0.f;
    for (int64_t i = 0; i < P; i++) ave += it[i];
    ave = ave / (float)P;
    if (lane == 0) ave_s = ave;
  } else {
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double* p = xyz + 3 * o0;
    for (int64_t i = tid; i < P; i += 7 * 64) {
      const double x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
      s[0] += x; s[1] += y; s[2] += z;
      s[3] += x * x; s[4] += x * y; s[5] += x * z; s[6] += y * y; s[7] += y * z; s[8] += z * z;
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
      double v = s[k];
      for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
      if (lane == 0) red[w][k] = v;
    }
  }
  __syncthreads();
  if (tid == 0) {
    double s[9];
    for (int k = 0; k < 9; k++) {
      double v = 0;
      for (int i = 0; i < 7; i++) v += red[i][k];
      s[k] = v;
    }
    const double n = (double)P;
    const double mx = s[0] / n, my = s[1] / n, mz = s[2] / n;
    double a[3][3], v[3][3];
    a[0][0] = s[3] - n * mx * mx; a[0][1] = s[4] - n * mx * my; a[0][2] = s[5] - n * mx * mz;
    a[1][1] = s[6] - n * my * my; a[1][2] = s[7] - n * my * mz; a[2][2] = s[8] - n * mz * mz;
    a[1][0] = a[0][1]; a[2][0] = a[0][2]; a[2][1] = a[1][2];
    jacobi_eig3(a, v);
    int ord[3] = {0, 1, 2};   // ascending eigenvalues (Eigen::SelfAdjointEigenSolver order, pts_align.h:32-34)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2 - i; j++)
        if (a[ord[j + 1]][ord[j + 1]] < a[ord[j]][ord[j]]) { const int t = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = t; }
    double e[3][3];
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) e[j][k] = v[k][ord[j]];
    for (int j = 0; j < 2; j++) {   // canonical sign: largest-|component| positive (N3)
      int im = 0;
      for (int k = 1; k < 3; k++) if (fabs(e[j][k]) > fabs(e[j][im])) im = k;
      if (e[j][im] < 0) for (int k = 0; k < 3; k++) e[j][k] = -e[j][k];
    }
    const double cx = e[0][1] * e[1][2] - e[0][2] * e[1][1], cy = e[0][2] * e[1][0] - e[0][0] * e[1][2],
                 cz = e[0][0] * e[1][1] - e[0][1] * e[1][0];
    if (cx * e[2][0] + cy * e[2][1] + cz * e[2][2] < 0) for (int k = 0; k < 3; k++) e[2][k] = -e[2][k];   // det = +1
    double* f = frames + (size_t)c * 16;
    f[0] = mx; f[1] = my; f[2] = mz;
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) f[3 + 3 * j + k] = e[j][k];
    f[12] = (double)ave_s;
    f[13] = n; f[14] = 0; f[15] = 0;
  }
}

__device__ __forceinline__ unsigned long long dkey(double x) {   // order-preserving map double -> u64
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

__global__ __launch_bounds__(512) void sc_bin_kernel(const double* __restrict__ xyz, const float* __restrict__ inten,
                                                      const int64_t* __restrict__ offs, const double* __restrict__ frames,
                                                      double max_rho, double* __restrict__ out) {
  __shared__ unsigned int cnt[1200];
  __shared__ unsigned long long lo[1200], hi[1200];
  __shared__ double sum[1200];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int64_t o0 = offs[c];
  const int64_t P = offs[c + 1] - o0;
  for (int b = tid; b < 1200; b += 512) { cnt[b] = 0u; lo[b] = ~0ull; hi[b] = 0ull; sum[b] = 0.0; }
  __syncthreads();
  const double* f = frames + (size_t)c * 16;
  const double mx = f[0], my = f[1], mz = f[2];
  const double e00 = f[3], e01 = f[4], e02 = f[5], e10 = f[6], e11 = f[7], e12 = f[8], e20 = f[9], e21 = f[10], e22 = f[11];
  const double S_res_inv = 60 / (2.0 * M_PI), R_res_inv = 20 / max_rho;   // SC.cpp:5-8
  const double* p = xyz + 3 * o0;
  const float* it = inten + o0;
  for (int64_t i = tid; i < P; i += 512) {
    const double x = p[3 * i] - mx, y = p[3 * i + 1] - my, z = p[3 * i + 2] - mz;   // pts_align.h:24-26
    const double nx = (x * e00 + y * e01) + z * e02;                                // :37-39
    const double yp = (x * e10 + y * e11) + z * e12;
    const double zp = (x * e20 + y * e21) + z * e22;
    const int si = (int)floor((atan2(zp, yp) + M_PI) * S_res_inv);                  // SC.cpp:37
    const int ri = (int)floor(sqrt(yp * yp + zp * zp) * R_res_inv);                 // :38
    const int idx = si * 20 + ri;                                                   // :39
    if (idx >= 1200 || idx < 0) continue;                                           // :42-44
    atomicAdd(&cnt[idx], 1u);
    const unsigned long long k = dkey(nx);
    atomicMin(&lo[idx], k);
    atomicMax(&hi[idx], k);
    atomicAdd(&sum[idx], (double)it[i]);
  }
  __syncthreads();
  const double ave = f[12];   // the float average, widened (comparison double > float promotes the float)
  double* o = out + (size_t)c * 2400;
  for (int b = tid; b < 1200; b += 512) {
    const unsigned int n = cnt[b];
    double st = 0.0, iv = 0.0;
    if (n) {
      st = dunkey(hi[b]) - dunkey(lo[b]);                                           // :74
      iv = (sum[b] / (double)n) > ave ? 1.0 : 0.0;                                  // :69-70
    }
    o[b] = st;
    o[1200 + b] = iv;
  }
}

}  // namespace

void launch_cloud_frames(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int N, double* frames) {
  if (N <= 0) return;
  hipLaunchKernelGGL(cloud_frames_kernel, dim3(N), dim3(FT), 0, st, xyz, inten, offs, frames);
}

void launch_sc_generate(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int N,
                        double max_rho, double* out, double* frames) {
  if (N <= 0) return;
  launch_cloud_frames(st, xyz, inten, offs, N, frames);
  hipLaunchKernelGGL(sc_bin_kernel, dim3(N), dim3(512), 0, st, xyz, inten, offs, frames, max_rho, out);
}

}  // namespace pr
