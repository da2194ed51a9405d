// fuse_select.hip — z-score fusion, mask and top-k of run_test.m:38-41, :47-53, :57 on gfx950.
//
// MATLAB normalize(d,2) = (d - mean)/std per row with the N-1 standard deviation, computed over the WHOLE row
// including entries that the mask will later discard (mask is applied after normalisation).  Two kernels:
//   row_moments : per query row and channel (count, mean, M2) in fp64 from ONE pass (shifted sums about the row's first
//                 element), fixed reduction tree (deterministic).  One workgroup per row.  HBM-bound: 8 B per (query, entry) pair.
//   fuse_select : combines the moments of G DB shards in rank order (Chan), then
//                 fused = p_weight*(d_p-mean_p)/std_p + (d_i-mean_i)/std_i  (run_test.m:40), +Inf where
//                 |i-j| < mask_width on GLOBAL indices (:47-53), and selects the k smallest (value, index)
//                 pairs in lexicographic order, i.e. ties go to the lower index like MATLAB min (:57).
#include "div_rn.hpp"
#include "kernels.hpp"

namespace pr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
  red[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  double r = red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void row_moments_kernel(const float* __restrict__ d_p, const float* __restrict__ d_i,
                                                           int n, double* __restrict__ mom) {
  __shared__ double red[256];
  const int tid = threadIdx.x, q = blockIdx.x;
  for (int ch = 0; ch < 2; ch++) {
    // ONE pass over the row: sums of (d - c) and (d - c)^2 in fp64 about the pivot c = first element of the row (distances
    // are fp32 in [0, 1], so with c inside the data range the M2 = S2 - S1^2/n cancellation costs < 1e-13 relative), fixed
    // reduction tree -> deterministic.
    const float* row = (ch ? d_i : d_p) + (size_t)q * n;
    const double c = (double)row[0];
    double s1 = 0.0, s2 = 0.0;
    auto acc = [&](float v) { const double d = (double)v - c; s1 += d; s2 += d * d; };
    // 16-byte loads over the aligned body of the row (4 of them in flight per thread), scalars on the ragged ends
    int head = (int)((4 - ((reinterpret_cast<size_t>(row) >> 2) & 3)) & 3);
    if (head > n) head = n;
    const int nv = (n - head) >> 2;
    if (tid < head) acc(row[tid]);
    const f32x4* rv4 = reinterpret_cast<const f32x4*>(row + head);
    int j = tid;
    for (; j + 768 < nv; j += 1024) {
      const f32x4 a = rv4[j], b = rv4[j + 256], c4 = rv4[j + 512], d4 = rv4[j + 768];
      acc(a[0]); acc(a[1]); acc(a[2]); acc(a[3]);
      acc(b[0]); acc(b[1]); acc(b[2]); acc(b[3]);
      acc(c4[0]); acc(c4[1]); acc(c4[2]); acc(c4[3]);
      acc(d4[0]); acc(d4[1]); acc(d4[2]); acc(d4[3]);
    }
    for (; j < nv; j += 256) { const f32x4 a = rv4[j]; acc(a[0]); acc(a[1]); acc(a[2]); acc(a[3]); }
    for (int t = head + 4 * nv + tid; t < n; t += 256) acc(row[t]);
    const double S1 = block_sum(s1, red, tid);
    const double S2 = block_sum(s2, red, tid);
    if (tid == 0) {
      double* o = mom + ((size_t)q * 2 + ch) * 3;
      o[0] = (double)n;
      o[1] = c + S1 / (double)n;
      o[2] = S2 - S1 * S1 / (double)n;
    }
  }
}

struct Cand {
  double v;
  int j;
};
__device__ __forceinline__ bool cand_less(double av, int aj, double bv, int bj) {   // (a) < (b) lexicographic
  return av < bv || (av == bv && aj < bj);
}

// e_p / e_i / mom2_all: an optional SECOND channel pair over the same (query, entry) grid whose z-scores are added with the
// same weights (BASELINE.json config 5, "fused SC + M2DP scoring": build-defined, no reference counterpart)
__global__ __launch_bounds__(256) void fuse_select_kernel(const float* __restrict__ d_p, const float* __restrict__ d_i,
                                                           const float* __restrict__ e_p, const float* __restrict__ e_i,
                                                           const double* __restrict__ mom2_all,
                                                           int m, int n, const double* __restrict__ mom_all, int G,
                                                           int q_row0, int db_row0, int mask_width, double p_weight,
                                                           int k, int32_t* __restrict__ idx, float* __restrict__ score) {
  __shared__ double st[8];
  __shared__ double rv[256];
  __shared__ int rj[256];
  const int tid = threadIdx.x, q = blockIdx.x;
  const bool plain = (d_i == nullptr);   // single distance matrix, no z-score fusion (run_test.m types other than m2dp/sc)
  const bool two = (e_p != nullptr);
  if (!plain && tid < (two ? 4 : 2)) {  // Chan's parallel combination of the shard moments, fixed (rank) order
    double cn = 0.0, mean = 0.0, m2 = 0.0;
    for (int g = 0; g < G; g++) {
      const double* o = (tid < 2 ? mom_all : mom2_all) + (((size_t)g * m + q) * 2 + (tid & 1)) * 3;
      const double nb = o[0], mb = o[1], m2b = o[2];
      if (nb <= 0.0) continue;
      const double tot = cn + nb, delta = mb - mean;
      mean += delta * (nb / tot);
      m2 += m2b + delta * delta * (cn * nb / tot);
      cn = tot;
    }
    st[tid * 2] = mean;
    st[tid * 2 + 1] = sqrt(m2 / (cn - 1.0));
  }
  __syncthreads();
  const double mp = plain ? 0.0 : st[0], sp = plain ? 1.0 : st[1], mi = plain ? 0.0 : st[2], si = plain ? 1.0 : st[3];
  const float* rp = d_p + (size_t)q * n;
  const float* ri = plain ? rp : d_i + (size_t)q * n;
  const int ig = q_row0 + q;
  // (d - mean) / std is a correctly rounded fp64 division in the reference (MATLAB normalize).  Markstein's sequence gives
  // the same correctly rounded quotient from the row's reciprocal r = RN(1/s): q = x r, e = fma(-q, s, x), fma(e, r, q) -
  // 3 instructions instead of the ~30 of a software fp64 division, which made this kernel VALU-bound.  Rows whose std is
  // zero, denormal or not finite keep the plain division (uniform per workgroup).
  const bool fastdiv = !plain && sp > 1e-290 && sp < 1e290 && si > 1e-290 && si < 1e290;
  const double rsp = 1.0 / sp, rsi = 1.0 / si;
  int head = (int)((4 - ((reinterpret_cast<size_t>(rp) >> 2) & 3)) & 3);
  if (head > n || ((reinterpret_cast<size_t>(rp) ^ reinterpret_cast<size_t>(ri)) & 15)) head = n;   // rows not co-aligned: scalar loads
  const int nv = (n - head) >> 2;
  const f32x4* rp4 = reinterpret_cast<const f32x4*>(rp + head);
  const f32x4* ri4 = reinterpret_cast<const f32x4*>(ri + head);
  double pv = -__builtin_inf();
  int pj = -1;
  for (int t = 0; t < k; t++) {
    double bv = 0.0;
    int bj = -1;
    auto consider = [&](float vp, float vi, int j) {
      const int jg = db_row0 + j;
      double f;
      if (plain) f = (double)vp;
      else if (fastdiv) f = p_weight * div_rn((double)vp - mp, sp, rsp) + div_rn((double)vi - mi, si, rsi);        // run_test.m:40
      else f = p_weight * (((double)vp - mp) / sp) + ((double)vi - mi) / si;
      if (two) f += p_weight * (((double)e_p[(size_t)q * n + j] - st[4]) / st[5]) + ((double)e_i[(size_t)q * n + j] - st[6]) / st[7];
      int dij = ig - jg;
      if (dij < 0) dij = -dij;
      if (dij < mask_width) f = __builtin_inf();                                      // run_test.m:47-53
      if (f != f) return;                                                              // NaN never wins (MATLAB min)
      if (!cand_less(pv, pj, f, jg)) return;                                           // already selected
      if (bj < 0 || cand_less(f, jg, bv, bj)) { bv = f; bj = jg; }
    };
    if (tid < head) consider(rp[tid], ri[tid], tid);
    for (int j = tid; j < nv; j += 256) {
      const f32x4 a = rp4[j], b = ri4[j];
      const int j0 = head + 4 * j;
      consider(a[0], b[0], j0); consider(a[1], b[1], j0 + 1); consider(a[2], b[2], j0 + 2); consider(a[3], b[3], j0 + 3);
    }
    for (int j = head + 4 * nv + tid; j < n; j += 256) consider(rp[j], ri[j], j);
    rv[tid] = bv;
    rj[tid] = bj;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) {
        const int oj = rj[tid + s];
        if (oj >= 0 && (rj[tid] < 0 || cand_less(rv[tid + s], oj, rv[tid], rj[tid]))) {
          rv[tid] = rv[tid + s];
          rj[tid] = oj;
        }
      }
      __syncthreads();
    }
    pv = rv[0];
    pj = rj[0];
    __syncthreads();
    if (tid == 0) {
      idx[(size_t)q * k + t] = pj;
      score[(size_t)q * k + t] = (pj >= 0) ? (float)pv : __builtin_nanf("");
    }
    if (pj < 0) {  // fewer than k candidates: fill the rest
      if (tid == 0)
        for (int u = t + 1; u < k; u++) { idx[(size_t)q * k + u] = -1; score[(size_t)q * k + u] = __builtin_nanf(""); }
      break;
    }
  }
}

}  // namespace

void launch_row_moments(hipStream_t st, const float* d_p, const float* d_i, int m, int n, double* mom) {
  if (m <= 0) return;
  hipLaunchKernelGGL(row_moments_kernel, dim3(m), dim3(256), 0, st, d_p, d_i, n, mom);
}

void launch_fuse_select(hipStream_t st, const float* d_p, const float* d_i, int m, int n, const double* mom_all,
                        int G, int q_row0, int db_row0, int mask_width, double p_weight, int k, int32_t* idx,
                        float* score, const float* e_p, const float* e_i, const double* mom2_all) {
  if (m <= 0) return;
  hipLaunchKernelGGL(fuse_select_kernel, dim3(m), dim3(256), 0, st, d_p, d_i, e_p, e_i, mom2_all, m, n, mom_all, G, q_row0, db_row0,
                     mask_width, p_weight, k, idx, score);
}

}  // namespace pr
