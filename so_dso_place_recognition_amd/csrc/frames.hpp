// frames.hpp — PCA frame of a cloud (utils/pts_align.h:7-46) from its raw moments, shared by sc_gen.hip (cloud_frames) and prestage.hip
// (the gather pass that emits the frames with the points).  Device code; include in files compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace pr {

constexpr int FRAME_THREADS = 256;   // workgroup of a moments pass: thread t adds points t, t + 256, ... in order, then the fixed tree below

__device__ inline void jacobi_eig3(double a[3][3], double v[3][3]) {
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

// raw moments (sum p, sum p p^T) of a cloud of n points -> frame {mean[3], v0[3], v1[3], v2[3], 0, n, ave, has_ave} (the last two are not
// written here: the reference's sequential FLOAT average of the intensities, SC.cpp:60-64 / M2DP.cpp:77-81, widened, and 1.0 when it is there): scatter matrix
// cov = sum pp^T - n mean mean^T (un-normalised as pts_align.h:30), 3x3 Jacobi, eigenvalues ascending (Eigen::SelfAdjointEigenSolver
// order, :32-34), canonical signs (N3)
__device__ __attribute__((noinline)) inline void finish_frame(const double s[9], double n, double* f) {
  const double mx = s[0] / n, my = s[1] / n, mz = s[2] / n;
  double a[3][3], v[3][3];
  a[0][0] = s[3] - n * mx * mx; a[0][1] = s[4] - n * mx * my; a[0][2] = s[5] - n * mx * mz;
  a[1][1] = s[6] - n * my * my; a[1][2] = s[7] - n * my * mz; a[2][2] = s[8] - n * mz * mz;
  a[1][0] = a[0][1]; a[2][0] = a[0][2]; a[2][1] = a[1][2];
  jacobi_eig3(a, v);
  int ord[3] = {0, 1, 2};
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
  f[0] = mx; f[1] = my; f[2] = mz;
  for (int j = 0; j < 3; j++)
    for (int k = 0; k < 3; k++) f[3 + 3 * j + k] = e[j][k];
  f[12] = 0; f[13] = n;      // slots 14, 15: the cloud's float intensity average and its presence flag - written by whoever computes it
}

// the fixed reduction tree of a moments pass: 64-lane shuffle tree per wave, then the waves in order; thread 0 writes the frame.
// Every moments pass that feeds thread t the same points in the same order (cloud_frames_kernel, gather_frames_kernel) gives the same bits.
__device__ __forceinline__ void reduce_moments_to_frame(const double (&s)[9], double n, double (*red)[9] /* LDS [FRAME_THREADS / 64][9] */, double* frame) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    double v = s[k];
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    if (lane == 0) red[w][k] = v;
  }
  __syncthreads();
  if (tid == 0) {
    double t[9];
    for (int k = 0; k < 9; k++) {
      double v = 0;
      for (int i = 0; i < FRAME_THREADS / 64; i++) v += red[i][k];
      t[k] = v;
    }
    finish_frame(t, n, frame);
  }
}

// The reference's float average of a cloud's intensities (a sequential float sum in input order, SC.cpp:60-64, M2DP.cpp:77-81) by ONE
// workgroup of FRAME_THREADS threads: everybody stages 2048 floats into LDS (two buffers), thread 0 adds them in order, 32 values in
// registers while the next 32 are on their way.  stage: LDS [2][2048 + 32] floats.  Returns the average in every thread... of thread 0
// only (the others return 0); P = 0 gives 0 / 0 = NaN as the reference does.
__device__ inline float block_sequential_average(const float* __restrict__ it, long long P, float (*stage)[2048 + 32]) {
  typedef float f4a __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  float sum = 0.f;
  auto fill = [&](int b, long long c0) {
    for (int s = tid; s < 2048; s += FRAME_THREADS) stage[b][s] = (c0 + s < P) ? it[c0 + s] : 0.f;      // x + 0 = x past the end
  };
  if (P > 0) fill(0, 0);
  __syncthreads();
  for (long long c0 = 0, b = 0; c0 < P; c0 += 2048, b ^= 1) {
    if (c0 + 2048 < P) fill((int)(b ^ 1), c0 + 2048);
    if (tid == 0) {
      const f4a* src = reinterpret_cast<const f4a*>(&stage[b][0]);
      const long long left = P - c0;
      const int nq = (int)((left < 2048 ? left : 2048) + 3) >> 2;         // float4 groups that hold data (the tail of the last one is zeros)
      for (int j = 0; j < nq; j += 8) {
        f4a a[8];
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = src[j + q];                     // (up to 7 groups past nq: zeros or the spare 32 floats)
#pragma unroll
        for (int q = 0; q < 8; q++) { sum += a[q][0]; sum += a[q][1]; sum += a[q][2]; sum += a[q][3]; }
      }
    }
    __syncthreads();
  }
  return tid == 0 ? sum / (float)P : 0.f;
}

}  // namespace pr
