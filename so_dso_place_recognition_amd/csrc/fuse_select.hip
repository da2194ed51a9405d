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
#include <cstdlib>

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
    const float r0 = row[0];
    const double c = (r0 == r0) ? (double)r0 : 0.5;      // NaN first element: any pivot inside the data range will do
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    // NaN entries (distances to / from a zero-norm signature, processSC.m:16,19) are left out of the row statistics,
    // as MATLAB's normalize(.,2) does (mean / std with 'omitnan') [from memory; the reference cannot be run here]
    auto acc = [&](float v) { if (v == v) { const double d = (double)v - c; s1 += d; s2 += d * d; cnt += 1.0; } };
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
    const double N = block_sum(cnt, red, tid);
    if (tid == 0) {
      double* o = mom + ((size_t)q * 2 + ch) * 3;
      o[0] = N;
      o[1] = N > 0.0 ? c + S1 / N : 0.0;
      o[2] = N > 0.0 ? S2 - S1 * S1 / N : 0.0;
    }
  }
}

// Few query rows (a keyframe per call): one workgroup per row leaves the chip idle and pays the row's whole latency in one place.  The row is
// cut into P slices, one workgroup each (grid m x P): raw sums (count, S1, S2) about the row's first element per slice, added in slice order
// by moments_finish_kernel - deterministic for a given (m, n).
__global__ __launch_bounds__(256) void row_moments_slice_kernel(const float* __restrict__ d_p, const float* __restrict__ d_i, int n, int m, int P,
                                                                 double* __restrict__ part /* [P][m][2][3] */) {
  // an online call waits for this kernel: both channels side by side, eight loads of a thread in flight, and ONE reduction tree for the
  // six sums (the per-thread order of the additions and the tree are those of block_sum, so the sums are bit for bit what three
  // block_sum calls per channel gave)
  __shared__ double red[6][256];
  const int tid = threadIdx.x, q = blockIdx.x, sl = blockIdx.y;
  const int c0 = (int)((long long)n * sl / P), c1 = (int)((long long)n * (sl + 1) / P);
  const float* rp = d_p + (size_t)q * n;
  const float* ri = d_i + (size_t)q * n;
  const float p0 = rp[0], i0 = ri[0];
  const double cp = (p0 == p0) ? (double)p0 : 0.5, ci = (i0 == i0) ? (double)i0 : 0.5;
  double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};     // S1, S2, count of d_p; of d_i
  auto acc = [&](float v, double c, int o) { if (v == v) { const double d = (double)v - c; s[o] += d; s[o + 1] += d * d; s[o + 2] += 1.0; } };
  int j = c0 + tid;
  for (; j + 768 < c1; j += 1024) {
    const float a0 = rp[j], a1 = rp[j + 256], a2 = rp[j + 512], a3 = rp[j + 768];
    const float b0 = ri[j], b1 = ri[j + 256], b2 = ri[j + 512], b3 = ri[j + 768];
    acc(a0, cp, 0); acc(a1, cp, 0); acc(a2, cp, 0); acc(a3, cp, 0);
    acc(b0, ci, 3); acc(b1, ci, 3); acc(b2, ci, 3); acc(b3, ci, 3);
  }
  for (; j < c1; j += 256) { acc(rp[j], cp, 0); acc(ri[j], ci, 3); }
#pragma unroll
  for (int o = 0; o < 6; o++) red[o][tid] = s[o];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) {
#pragma unroll
      for (int o = 0; o < 6; o++) red[o][tid] += red[o][tid + st];
    }
    __syncthreads();
  }
  if (tid < 2) {
    double* o = part + (((size_t)sl * m + q) * 2 + tid) * 3;
    o[0] = red[3 * tid + 2][0]; o[1] = red[3 * tid][0]; o[2] = red[3 * tid + 1][0];
  }
}
__global__ __launch_bounds__(64) void moments_finish_kernel(const float* __restrict__ d_p, const float* __restrict__ d_i, int n, int m, int P,
                                                             const double* __restrict__ part, double* __restrict__ mom) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= 2 * m) return;
  const int q = t >> 1, ch = t & 1;
  const float r0 = ((ch ? d_i : d_p) + (size_t)q * n)[0];
  const double c = (r0 == r0) ? (double)r0 : 0.5;
  double N = 0.0, S1 = 0.0, S2 = 0.0;
  const size_t step = (size_t)m * 6;                 // doubles between the partials of consecutive slices
  const double* o = part + ((size_t)q * 2 + ch) * 3;
  int sl = 0;
  for (; sl + 8 <= P; sl += 8) {                     // eight slices' partials requested together, added in slice order
    double a[8][3];
#pragma unroll
    for (int u = 0; u < 8; u++) { a[u][0] = o[(sl + u) * step]; a[u][1] = o[(sl + u) * step + 1]; a[u][2] = o[(sl + u) * step + 2]; }
#pragma unroll
    for (int u = 0; u < 8; u++) { N += a[u][0]; S1 += a[u][1]; S2 += a[u][2]; }
  }
  for (; sl < P; sl++) { N += o[sl * step]; S1 += o[sl * step + 1]; S2 += o[sl * step + 2]; }
  double* w = mom + ((size_t)q * 2 + ch) * 3;
  w[0] = N;
  w[1] = N > 0.0 ? c + S1 / N : 0.0;
  w[2] = N > 0.0 ? S2 - S1 * S1 / N : 0.0;
}

__device__ __forceinline__ bool cand_less(double av, int aj, double bv, int bj) {   // (a) < (b) lexicographic
  return av < bv || (av == bv && aj < bj);
}

// argmin over the workgroup of (v, j) pairs with j >= 0 (j < 0 = no candidate); result in rv[0], rj[0].  Inside a wave by lane exchanges
// (no barrier), the four wave results through LDS: two barriers instead of nine - the selection loops below call this k + r times per row.
__device__ __forceinline__ void block_argmin(double* rv, int* rj, int tid) {
  double v = rv[tid];
  int j = rj[tid];
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    const double ov = __shfl_xor(v, s, 64);
    const int oj = __shfl_xor(j, s, 64);
    if (oj >= 0 && (j < 0 || cand_less(ov, oj, v, j))) { v = ov; j = oj; }
  }
  __syncthreads();                                   // everybody has read its own rv / rj entry
  if ((tid & 63) == 0) { rv[tid >> 6] = v; rj[tid >> 6] = j; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; w++) {
      const int oj = rj[w];
      if (oj >= 0 && (j < 0 || cand_less(rv[w], oj, v, j))) { v = rv[w]; j = oj; }
    }
    rv[0] = v; rj[0] = j;
  }
  __syncthreads();
}

constexpr int FS_CAP = 4096;   // survivors of the threshold pass kept in LDS (a power of two: the fallback of list_topk sorts it whole)
constexpr int NONE_J = 0x7fffffff;   // index of a padding entry (value +Inf): sorts behind every real candidate

__device__ __forceinline__ unsigned long long order_key(double x) {   // order-preserving map double -> u64 (no NaN in the lists)
  const unsigned long long b = (unsigned long long)__double_as_longlong(x + 0.0);   // (-0 + 0 = +0: the two zeros are one value)
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// The k smallest (value, index) pairs of the LDS list lv / lj[0, L), k <= 249, into rv / rj[0, k) in ascending order (rj = NONE_J where the
// list runs out), WITHOUT sorting the list: a most-significant-digit radix selection over the values' order keys (8 bits per pass, LDS
// histogram, one wave picks the bucket that holds rank k) narrows the candidates until at most 256 elements are at or below the bucket -
// three passes for scores that differ - and only those are sorted (bitonic network over 256, one compare-exchange per thread and step).
// A full bitonic sort of 4096 fp64 pairs is 78 steps of 2048 compare-exchanges: 56 us per workgroup against ~4 for this.
// Returns false when more than 256 elements tie around rank k down to the last digit (masses of equal scores): the caller sorts the whole list.
template <typename V>
__device__ bool list_topk(const V* lv, const int* lj, int L, int k, double* rv, int* rj, unsigned* hist, int* ctl, int tid) {
  int C = L;
  if (L > 256) {
    unsigned long long prefix = 0ull;
    int krem = k, below_total = 0, shd = -1;
    for (int d = 0; d < 8 && shd < 0; d++) {
      const int sh = 56 - 8 * d;
      hist[tid] = 0u;
      __syncthreads();
      for (int s = tid; s < L; s += 256) {
        const unsigned long long key = order_key((double)lv[s]);
        if (d == 0 || (key >> (sh + 8)) == prefix) atomicAdd(&hist[(key >> sh) & 255ull], 1u);
      }
      __syncthreads();
      if (tid < 64) {                                  // the bucket that holds rank krem: 4 buckets per lane, inclusive scan over the lanes
        const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
        const unsigned own = (c0 + c1) + (c2 + c3);
        unsigned incl = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned up = __shfl_up(incl, o, 64);
          if (tid >= o) incl += up;
        }
        const unsigned excl = incl - own;
        if (excl < (unsigned)krem && (unsigned)krem <= incl) {
          unsigned b = excl;
          int bin = 4 * tid;
          unsigned cb = c0;
          if (b + c0 < (unsigned)krem) { b += c0; bin++; cb = c1;
            if (b + c1 < (unsigned)krem) { b += c1; bin++; cb = c2;
              if (b + c2 < (unsigned)krem) { b += c2; bin++; cb = c3; } } }
          ctl[0] = bin; ctl[1] = (int)b; ctl[2] = (int)cb;
        }
      }
      __syncthreads();
      prefix = (prefix << 8) | (unsigned long long)ctl[0];
      below_total += ctl[1];
      krem -= ctl[1];
      if (below_total + ctl[2] <= 256) shd = sh;
      __syncthreads();
    }
    if (shd < 0) return false;
    if (tid == 0) ctl[3] = 0;
    __syncthreads();
    for (int s = tid; s < L; s += 256) {
      const double v = (double)lv[s];
      if ((order_key(v) >> shd) <= prefix) {
        const int slot = atomicAdd(&ctl[3], 1);
        rv[slot] = v; rj[slot] = lj[s];
      }
    }
    __syncthreads();
    C = ctl[3];
  } else {
    if (tid < L) { rv[tid] = (double)lv[tid]; rj[tid] = lj[tid]; }
  }
  if (tid >= C) { rv[tid] = __builtin_inf(); rj[tid] = NONE_J; }
  __syncthreads();
  for (int size = 2; size <= 256; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (tid < 128) {
        const int pos = 2 * tid - (tid & (stride - 1)), par = pos + stride;
        const bool up = (pos & size) == 0;
        const double a = rv[pos], b = rv[par];
        const int aj = rj[pos], bj = rj[par];
        if (cand_less(b, bj, a, aj) == up) { rv[pos] = b; rj[pos] = bj; rv[par] = a; rj[par] = aj; }
      }
      __syncthreads();
    }
  return true;
}

// e_p / e_i / mom2_all: an optional SECOND channel pair over the same (query, entry) grid whose z-scores are added with the
// same weights (BASELINE.json config 5, "fused SC + M2DP scoring": build-defined, no reference counterpart).
//
// Selection of the k smallest (score, index) pairs of a row in about ONE pass over it, whatever k:
//   k = 1   every thread keeps the minimum of its elements, block argmin.
//   k > 1   (a) a sample - the first 4096 columns (more for large k, see below), 16 per thread: the r-th smallest of the 256 per-thread sample minima,
//               tau (r = max(k + 7, 16)), is the score of r distinct elements, so at least r >= k elements of the row are
//               <= tau, and about r n / 4096 of them in all;
//           (b) one sweep of the row: every element with score <= tau (ties included) goes to an LDS list;
//           (c) the k best of the list by (score, index): k arg-min rounds for k <= 12, a radix selection + 256-entry sort (list_topk)
//               for more.  A row or slice that fits the list (few-row calls, grid m x P) skips (a) and the atomics of (b).
// Rows that overflow the list (masses of equal scores, e.g. +Inf of the mask) fall back to one sweep per selected element.
// (Measured at 4096 x 100k, k = 9: 0.65 ms against 0.59 ms for k = 1; keeping the 3 best per thread in one sweep instead cost
// 2.5 ms - in a 64-lane wave some lane inserts at nearly every element - and a threshold from a full first sweep 1.13 ms.)
// CAP = capacity of the survivors' list (FS_CAP: 53 KB of LDS, ~3 workgroups per CU)
template <int CAP>
struct SelLds {
  double st[8];
  double rv[256];
  int rj[256];
  double lv[CAP];
  int lj[CAP];
  int lcnt;
  unsigned hist[256];
  int ctl[4];
  double tau_s;
  int last;
};

template <int CAP>
__device__ __forceinline__ void select_row(const float* __restrict__ d_p, const float* __restrict__ d_i,
                                           const float* __restrict__ e_p, const float* __restrict__ e_i,
                                           const double* __restrict__ mom2_all,
                                           int m, int n, const double* __restrict__ mom_all, int G,
                                           int q_row0, int db_row0, int mask_width, double p_weight,
                                           int k, int32_t* __restrict__ idx, float* __restrict__ score, int P,
                                           double* __restrict__ score64, SelLds<CAP>& W) {
  double* st = W.st;
  double* rv = W.rv;
  int* rj = W.rj;
  double* lv = W.lv;
  int* lj = W.lj;
  int& lcnt = W.lcnt;
  unsigned* hist = W.hist;
  int* ctl = W.ctl;
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
  if (tid == 0) lcnt = 0;
  __syncthreads();
  const double mp = plain ? 0.0 : st[0], sp = plain ? 1.0 : st[1], mi = plain ? 0.0 : st[2], si = plain ? 1.0 : st[3];
  // P > 1 (few query rows, grid m x P): this workgroup selects inside columns [c0, c0 + n) of its row only and writes list blockIdx.y of
  // [P][m][k]; slice_merge_kernel merges the P ascending lists (the moments are those of the whole row either way)
  const int n_row = n;
  if (P > 1) {
    const int c0 = (int)((long long)n_row * blockIdx.y / P), c1 = (int)((long long)n_row * (blockIdx.y + 1) / P);
    const size_t ro = (size_t)q * n_row + c0;
    d_p += ro - (size_t)q * (c1 - c0);          // so that the row addressing below (q * n + j with the slice length n) lands on the slice
    if (d_i) d_i += ro - (size_t)q * (c1 - c0);
    if (e_p) { e_p += ro - (size_t)q * (c1 - c0); e_i += ro - (size_t)q * (c1 - c0); }
    db_row0 += c0;
    n = c1 - c0;
    idx += (size_t)blockIdx.y * m * k;
    score += (size_t)blockIdx.y * m * k;
  }
  const float* rp = d_p + (size_t)q * n;
  const float* ri = plain ? rp : d_i + (size_t)q * n;
  const int ig = q_row0 + q;
  // (d - mean) / std is a correctly rounded fp64 division in the reference (MATLAB normalize).  Markstein's sequence gives
  // the same correctly rounded quotient from the row's reciprocal r = RN(1/s): q = x r, e = fma(-q, s, x), fma(e, r, q) -
  // 3 instructions instead of the ~30 of a software fp64 division, which made this kernel VALU-bound.  Rows whose std is
  // zero, denormal or not finite keep the plain division (uniform per workgroup).
  const bool fastdiv = !plain && sp > 1e-290 && sp < 1e290 && si > 1e-290 && si < 1e290;
  const double rsp = 1.0 / sp, rsi = 1.0 / si;
  // (the second channel pair of the fused form the same way: two software fp64 divisions per element were most of its sweep)
  const bool fastdiv2 = two && st[5] > 1e-290 && st[5] < 1e290 && st[7] > 1e-290 && st[7] < 1e290;
  const double m2p = two ? st[4] : 0.0, s2p = two ? st[5] : 1.0, m2i = two ? st[6] : 0.0, s2i = two ? st[7] : 1.0;
  const double rs2p = 1.0 / s2p, rs2i = 1.0 / s2i;
  int head = (int)((4 - ((reinterpret_cast<size_t>(rp) >> 2) & 3)) & 3);
  if (head > n) head = n;
  // rows not co-aligned mod 16 B (d_p and d_i carved out of one allocation with m*n % 4 != 0): no vector body, the scalar
  // tail loop below walks the whole row
  const bool coaligned = !((reinterpret_cast<size_t>(rp) ^ reinterpret_cast<size_t>(ri)) & 15) &&
                         !(e_p && (((reinterpret_cast<size_t>(rp) ^ reinterpret_cast<size_t>(e_p + (size_t)q * n)) |
                                    (reinterpret_cast<size_t>(rp) ^ reinterpret_cast<size_t>(e_i + (size_t)q * n))) & 15));
  if (!coaligned) head = 0;
  const int nv = coaligned ? (n - head) >> 2 : 0;
  const f32x4* rp4 = reinterpret_cast<const f32x4*>(rp + head);
  const f32x4* ri4 = reinterpret_cast<const f32x4*>(ri + head);

  const float* rep = two ? e_p + (size_t)q * n : rp;      // the second channel pair's rows (the fused form), else any valid address
  const float* rei = two ? e_i + (size_t)q * n : rp;
  auto fused4 = [&](float vp, float vi, float wp, float wi, int j) -> double {
    const int jg = db_row0 + j;
    double f;
    if (plain) f = (double)vp;
    else if (fastdiv) f = p_weight * div_rn((double)vp - mp, sp, rsp) + div_rn((double)vi - mi, si, rsi);        // run_test.m:40
    else f = p_weight * (((double)vp - mp) / sp) + ((double)vi - mi) / si;
    if (two) {
      const double xp = (double)wp - m2p, xi = (double)wi - m2i;
      f += fastdiv2 ? p_weight * div_rn(xp, s2p, rs2p) + div_rn(xi, s2i, rs2i) : p_weight * (xp / s2p) + xi / s2i;
    }
    int dij = ig - jg;
    if (dij < 0) dij = -dij;
    if (dij < mask_width) f = __builtin_inf();                                      // run_test.m:47-53
    return f;                                                                        // NaN never wins (MATLAB min)
  };
  auto fused = [&](float vp, float vi, int j) -> double { return fused4(vp, vi, two ? rep[j] : 0.f, two ? rei[j] : 0.f, j); };
  // visit(j, f) over this thread's elements of the row
  const f32x4* ep4 = reinterpret_cast<const f32x4*>(rep + head);
  const f32x4* ei4 = reinterpret_cast<const f32x4*>(rei + head);
  // `maybe(vp, vi, wp, wi)`: a cheap test in front of the exact score - false only where the element cannot be wanted (sweep_if), or
  // always true (sweep)
  auto sweep_if = [&](auto&& maybe, auto&& visit) {
    if (tid < head && maybe(rp[tid], ri[tid], two ? rep[tid] : 0.f, two ? rei[tid] : 0.f)) visit(tid, fused(rp[tid], ri[tid], tid));
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    auto four = [&](const f32x4& a, const f32x4& b, const f32x4& c, const f32x4& d, int j) {
      const int j0 = head + 4 * j;
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (maybe(a[e], b[e], c[e], d[e])) visit(j0 + e, fused4(a[e], b[e], c[e], d[e], j0 + e));
    };
    int j = tid;
    // four 16-byte loads per channel in flight (the kernel runs at ~3 workgroups per CU - its LDS list - so a thread has to cover the
    // latency of its loads by itself); same elements, same order per thread as the plain loop.  The fused form's second channel pair
    // comes the same way (its element-wise 4-byte loads were most of that form's 2.65 ms sweep)
    if (two) {
      for (; j + 768 < nv; j += 1024) {
        const f32x4 a0 = rp4[j], a1 = rp4[j + 256], a2 = rp4[j + 512], a3 = rp4[j + 768];
        const f32x4 b0 = ri4[j], b1 = ri4[j + 256], b2 = ri4[j + 512], b3 = ri4[j + 768];
        const f32x4 c0 = ep4[j], c1 = ep4[j + 256], c2 = ep4[j + 512], c3 = ep4[j + 768];
        const f32x4 d0 = ei4[j], d1 = ei4[j + 256], d2 = ei4[j + 512], d3 = ei4[j + 768];
        four(a0, b0, c0, d0, j); four(a1, b1, c1, d1, j + 256); four(a2, b2, c2, d2, j + 512); four(a3, b3, c3, d3, j + 768);
      }
    } else if (j + 768 < nv) {
      // the round's eight loads are requested while the previous round's 32 scores are evaluated (the kernel spent two thirds of its wave
      // cycles in s_waitcnt with the loads of a round issued only after the arithmetic of the one before)
      f32x4 a0 = rp4[j], a1 = rp4[j + 256], a2 = rp4[j + 512], a3 = rp4[j + 768];
      f32x4 b0 = ri4[j], b1 = ri4[j + 256], b2 = ri4[j + 512], b3 = ri4[j + 768];
      for (;;) {
        const int jn = j + 1024;
        const bool more = jn + 768 < nv;
        const int jl = more ? jn : j;                                  // (the last round re-requests its own lines: no branch around the loads)
        const f32x4 n0 = rp4[jl], n1 = rp4[jl + 256], n2 = rp4[jl + 512], n3 = rp4[jl + 768];
        const f32x4 m0 = ri4[jl], m1 = ri4[jl + 256], m2 = ri4[jl + 512], m3 = ri4[jl + 768];
        four(a0, b0, z4, z4, j); four(a1, b1, z4, z4, j + 256); four(a2, b2, z4, z4, j + 512); four(a3, b3, z4, z4, j + 768);
        j = jn;
        if (!more) break;
        a0 = n0; a1 = n1; a2 = n2; a3 = n3; b0 = m0; b1 = m1; b2 = m2; b3 = m3;
      }
    }
    for (; j < nv; j += 256) {
      const f32x4 a = rp4[j], b = ri4[j];
      if (two) { const f32x4 c = ep4[j], d = ei4[j]; four(a, b, c, d, j); } else four(a, b, z4, z4, j);
    }
    for (int t = head + 4 * nv + tid; t < n; t += 256)
      if (maybe(rp[t], ri[t], two ? rep[t] : 0.f, two ? rei[t] : 0.f)) visit(t, fused(rp[t], ri[t], t));
  };
  auto sweep = [&](auto&& visit) { sweep_if([](float, float, float, float) { return true; }, visit); };
  // (score64: the same fp32-rounded score widened - what the fp64 re-evaluation takes as the candidates' pass scores; saves the caller a
  //  conversion launch, which an online call feels)
  auto emit = [&](int t, double v, int jg) {
    idx[(size_t)q * k + t] = jg;
    const float f32 = (jg >= 0) ? (float)v : __builtin_nanf("");
    score[(size_t)q * k + t] = f32;
    if (score64) score64[(size_t)q * k + t] = (double)f32;
  };

  if (k == 1) {
    double bv = 0.0;
    int bj = -1;
    sweep([&](int j, double f) {
      const int jg = db_row0 + j;
      if (f == f && (bj < 0 || cand_less(f, jg, bv, bj))) { bv = f; bj = jg; }
    });
    rv[tid] = bv; rj[tid] = bj;
    block_argmin(rv, rj, tid);
    if (tid == 0) emit(0, rv[0], rj[0]);
    return;
  }
  // ---- (a) threshold from a sample
  double tau = __builtin_inf();
  if (k <= 249 && n > CAP) {                    // (a row or slice that fits the list needs no threshold)
    double mv = 0.0;
    int mj = -1;
    // sample size: ~r n / ns elements pass the threshold - 4096 columns for the k + 8 <= 16 of the fp32-grade arithmetics (~24 r), more for
    // the k + 56 candidates of PR_SC_ARITH_F16 so that the list stays near a third of its capacity (an overflow costs k sweeps)
    const int r = k + 7 > 16 ? k + 7 : 16;
    long long want = (long long)n * r / 1024;
    const int ns = (int)(want < 4096 ? (n < 4096 ? n : 4096) : (want < n ? want : n));
    auto take = [&](float vp, float vi, int j) {
      const double f = fused(vp, vi, j);
      const int jg = db_row0 + j;
      if (f == f && (mj < 0 || cand_less(f, jg, mv, mj))) { mv = f; mj = jg; }
    };
    int j = tid;
    for (; j + 768 < ns; j += 1024) {
      const float p0 = rp[j], p1 = rp[j + 256], p2 = rp[j + 512], p3 = rp[j + 768];
      const float i0 = ri[j], i1 = ri[j + 256], i2 = ri[j + 512], i3 = ri[j + 768];
      take(p0, i0, j); take(p1, i1, j + 256); take(p2, i2, j + 512); take(p3, i3, j + 768);
    }
    for (; j < ns; j += 256) take(rp[j], ri[j], j);
    // tau = the r-th smallest of the 256 per-thread sample minima: every thread counts the minima in front of its own (broadcast LDS
    // reads, ~1 us) - the thread whose count is r - 1 holds it.  (r arg-min rounds of the whole workgroup, two barriers each, were a third
    // of this kernel's time at k + 56.)  Fewer than r minima: no bound.
    rv[tid] = mv; rj[tid] = mj;
    __syncthreads();
    int before = 0;
    for (int u = 0; u < 256; u++) {
      const int oj = rj[u];
      if (oj >= 0 && (mj < 0 || cand_less(rv[u], oj, mv, mj))) before++;
    }
    if (tid == 0) W.tau_s = __builtin_inf();
    __syncthreads();
    if (mj >= 0 && before == r - 1) W.tau_s = mv;    // (global indices are unique within a row: exactly one thread, if any)
    __syncthreads();
    tau = W.tau_s;
  }
  // ---- (b) everything at or below tau
  const bool whole = n <= CAP && !(tau < __builtin_inf());   // a row or slice that fits: element j IS list entry j (NaN: padding)
  if (whole)
    sweep([&](int j, double f) {
      const bool ok = f == f;
      lv[j] = ok ? f : __builtin_inf();
      lj[j] = ok ? db_row0 + j : NONE_J;
    });
  else {
    // Only ~r n / 4096 elements of the row are at or below tau, and the exact score is ~11 fp64 instructions (two correctly rounded
    // divisions).  In front of it: the same score as ONE affine form fa = a d_p + b d_i + c (two fp64 fmas; a = p / s_p, b = 1 / s_i,
    // c = -(p m_p / s_p + m_i / s_i); the fused form adds its second pair), which differs from the exact score by rounding only
    // (< 1e-13 for the usual row statistics: |a d_p|, |c| ~ 50).  fa > tau + E with E = 1e-10 (1 + |a| + |b| + |c| ...) - five orders above
    // that difference - cannot be wanted: the list is what the unfiltered sweep collects, element for element, and a wave evaluates
    // the exact score only where one of its lanes may pass (~1 element group in 5).  The mask only removes elements, NaN compares false.
    const bool pre = !plain && fastdiv && (!two || fastdiv2) && tau < __builtin_inf();
    const double a1 = p_weight / sp, b1 = 1.0 / si, c1 = -(p_weight * mp / sp + mi / si);
    const double a2 = two ? p_weight / s2p : 0.0, b2 = two ? 1.0 / s2i : 0.0, c2 = two ? -(p_weight * m2p / s2p + m2i / s2i) : 0.0;
    const double tauE = tau + 1e-10 * (1.0 + fabs(a1) + fabs(b1) + fabs(c1) + fabs(a2) + fabs(b2) + fabs(c2) + fabs(tau));
    auto collect = [&](int j, double f) {
      if (f <= tau) {
        const int slot = atomicAdd(&lcnt, 1);
        if (slot < CAP) { lv[slot] = f; lj[slot] = db_row0 + j; }
      }
    };
    if (pre)
      sweep_if([&](float vp, float vi, float wp, float wi) {
        double fa = __builtin_fma(a1, (double)vp, __builtin_fma(b1, (double)vi, c1));
        if (two) fa += __builtin_fma(a2, (double)wp, __builtin_fma(b2, (double)wi, c2));
        return fa <= tauE;
      }, collect);
    else
      sweep(collect);
  }
  __syncthreads();
  const int L = whole ? n : lcnt;
  if (L <= CAP && (k > 12 || whole)) {
    // many results (or a whole slice in the list): the k best by radix selection + a 256-entry sort ...
    if (k <= 249 && list_topk(lv, lj, L, k, rv, rj, hist, ctl, tid)) {
      for (int t = tid; t < k; t += 256) emit(t, rv[t], rj[t] == NONE_J ? -1 : rj[t]);
      return;
    }
    // ... or, when too many entries tie around rank k, by sorting it whole (bitonic network over the next power of two, (+Inf, NONE_J) padding)
    int N = 2;
    while (N < L) N <<= 1;
    for (int s = L + tid; s < N; s += 256) { lv[s] = __builtin_inf(); lj[s] = NONE_J; }
    __syncthreads();
    for (int size = 2; size <= N; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = tid; i < (N >> 1); i += 256) {
          const int pos = 2 * i - (i & (stride - 1)), par = pos + stride;
          const bool up = (pos & size) == 0;
          const double a = lv[pos], b = lv[par];
          const int aj = lj[pos], bj2 = lj[par];
          if (cand_less(b, bj2, a, aj) == up) { lv[pos] = b; lj[pos] = bj2; lv[par] = a; lj[par] = aj; }
        }
        __syncthreads();
      }
    for (int t = tid; t < k; t += 256) emit(t, lv[t < L ? t : 0], (t < L && lj[t] != NONE_J) ? lj[t] : -1);
    return;
  }
  if (L <= CAP) {
    for (int t = 0; t < k; t++) {
      double cv = 0.0;
      int cj = -1, cs = -1;
      for (int s = tid; s < L; s += 256)
        if (lj[s] >= 0 && (cj < 0 || cand_less(lv[s], lj[s], cv, cj))) { cv = lv[s]; cj = lj[s]; cs = s; }
      rv[tid] = cv; rj[tid] = cj;
      block_argmin(rv, rj, tid);
      const double wv = rv[0];
      const int wj = rj[0];
      __syncthreads();
      if (cs >= 0 && cj == wj) lj[cs] = -1;                                          // global indices are unique within a row
      if (tid == 0) emit(t, wv, wj);
      if (wj < 0) {
        if (tid == 0) for (int u = t + 1; u < k; u++) emit(u, 0.0, -1);
        break;
      }
      __syncthreads();
    }
    return;
  }
  // ---- fallback: one pass per selected element
  double pv = -__builtin_inf(), bv = 0.0;
  int pj = -1, bj = -1;
  for (int t = 0; t < k; t++) {
    bv = 0.0; bj = -1;
    sweep([&](int j, double f) {
      const int jg = db_row0 + j;
      if (f != f) return;
      if (!cand_less(pv, pj, f, jg)) return;                                         // already selected
      if (bj < 0 || cand_less(f, jg, bv, bj)) { bv = f; bj = jg; }
    });
    rv[tid] = bv; rj[tid] = bj;
    block_argmin(rv, rj, tid);
    pv = rv[0];
    pj = rj[0];
    __syncthreads();
    if (tid == 0) emit(t, pv, pj);
    if (pj < 0) {  // fewer than k candidates: fill the rest
      if (tid == 0) for (int u = t + 1; u < k; u++) emit(u, 0.0, -1);
      break;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- short rows: one WAVE per row
// select_row's workgroup spends a fixed ~80 us per row on its barriers and LDS rounds (sample threshold, k arg-min rounds), whatever the
// row's length: 0.45 ms per 4096 rows of 12 500 entries (the shard of an 8-GPU run, a KITTI-sized DB) against 0.7 - 0.9 ms for 100 000.
// Here a wave owns a row - four rows per workgroup, no barrier anywhere: the same statistics, the same score expression (bit for bit:
// fusedw below is select_row's fused4 without the second channel pair), the threshold from the r-th smallest of the 64 lanes' sample
// minima (rank counting over v_readlane), survivors compacted into the wave's LDS list by ballot / mbcnt, the k best by k wave arg-min
// rounds; the result is the same k smallest (score, index) pairs in the same order.  For rows of up to 32 768 entries, k <= 16, one channel
// pair, whole rows (launch_fuse_select decides); a list that overflows (masses of ties at the threshold) falls back to one sweep per
// selected element, as select_row does.
constexpr int WS_CAP = 1024;      // survivors per wave: ~r n / 1024 = 16 n / 1024 <= 512 expected at n = 32 768
__device__ __forceinline__ void wave_argmin(double& v, int& j) {     // all lanes end with the smallest (v, j), j >= 0; j < 0 = none
#pragma unroll
  for (int s2 = 32; s2 > 0; s2 >>= 1) {
    const double ov = __shfl_xor(v, s2, 64);
    const int oj = __shfl_xor(j, s2, 64);
    if (oj >= 0 && (j < 0 || cand_less(ov, oj, v, j))) { v = ov; j = oj; }
  }
}
__global__ __launch_bounds__(256) void fuse_select_wave_kernel(const float* __restrict__ d_p, const float* __restrict__ d_i, int m, int n,
                                                                const double* __restrict__ mom_all, int G, int q_row0, int db_row0,
                                                                int mask_width, double p_weight, int k, int32_t* __restrict__ idx,
                                                                float* __restrict__ score, double* __restrict__ score64) {
  __shared__ double lv_[4][WS_CAP];
  __shared__ int lj_[4][WS_CAP];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + w;
  if (q >= m) return;                                  // (whole wave; nothing below synchronises across waves)
  double* lv = lv_[w];
  int* lj = lj_[w];
  const bool plain = (d_i == nullptr);
  double st0 = 0.0, st1 = 1.0;
  if (!plain && lane < 2) {                            // Chan's combination of the shard moments in rank order, as in select_row
    double cn = 0.0, mean = 0.0, m2 = 0.0;
    for (int g = 0; g < G; g++) {
      const double* o = mom_all + (((size_t)g * m + q) * 2 + lane) * 3;
      const double nb = o[0], mb = o[1], m2b = o[2];
      if (nb <= 0.0) continue;
      const double tot = cn + nb, delta = mb - mean;
      mean += delta * (nb / tot);
      m2 += m2b + delta * delta * (cn * nb / tot);
      cn = tot;
    }
    st0 = mean;
    st1 = sqrt(m2 / (cn - 1.0));
  }
  const double mp = plain ? 0.0 : __shfl(st0, 0, 64), sp = plain ? 1.0 : __shfl(st1, 0, 64);
  const double mi = plain ? 0.0 : __shfl(st0, 1, 64), si = plain ? 1.0 : __shfl(st1, 1, 64);
  const float* rp = d_p + (size_t)q * n;
  const float* ri = plain ? rp : d_i + (size_t)q * n;
  const int ig = q_row0 + q;
  const bool fastdiv = !plain && sp > 1e-290 && sp < 1e290 && si > 1e-290 && si < 1e290;
  const double rsp = 1.0 / sp, rsi = 1.0 / si;
  auto fusedw = [&](float vp, float vi, int j) -> double {
    const int jg = db_row0 + j;
    double f;
    if (plain) f = (double)vp;
    else if (fastdiv) f = p_weight * div_rn((double)vp - mp, sp, rsp) + div_rn((double)vi - mi, si, rsi);        // run_test.m:40
    else f = p_weight * (((double)vp - mp) / sp) + ((double)vi - mi) / si;
    int dij = ig - jg;
    if (dij < 0) dij = -dij;
    if (dij < mask_width) f = __builtin_inf();                                      // run_test.m:47-53
    return f;
  };
  auto emit = [&](int t, double v, int jg) {
    idx[(size_t)q * k + t] = jg;
    const float f32 = (jg >= 0) ? (float)v : __builtin_nanf("");
    score[(size_t)q * k + t] = f32;
    if (score64) score64[(size_t)q * k + t] = (double)f32;
  };
  int head = (int)((4 - ((reinterpret_cast<size_t>(rp) >> 2) & 3)) & 3);
  if (head > n) head = n;
  const bool coaligned = !((reinterpret_cast<size_t>(rp) ^ reinterpret_cast<size_t>(ri)) & 15);
  if (!coaligned) head = 0;
  const int nv = coaligned ? (n - head) >> 2 : 0;
  const f32x4* rp4 = reinterpret_cast<const f32x4*>(rp + head);
  const f32x4* ri4 = reinterpret_cast<const f32x4*>(ri + head);
  // visit(valid, j, vp, vi) over the row, called by ALL lanes together (wave-uniform control flow: the visitors use ballots): the aligned
  // body in 16-byte loads (four in flight), scalars on the ragged ends
  auto sweep = [&](auto&& visit) {
    if (head > 0) { const bool v = lane < head; visit(v, lane, v ? rp[lane] : 0.f, v ? ri[lane] : 0.f); }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    auto four = [&](bool v, const f32x4& a, const f32x4& b, int j) {
      const int j0 = head + 4 * j;
#pragma unroll
      for (int e = 0; e < 4; e++) visit(v, j0 + e, a[e], b[e]);
    };
    for (int j0 = 0; j0 < nv; j0 += 256) {
      const int j = j0 + lane;
      const bool v0 = j < nv, v1 = j + 64 < nv, v2 = j + 128 < nv, v3 = j + 192 < nv;
      const f32x4 a0 = v0 ? rp4[j] : z4, a1 = v1 ? rp4[j + 64] : z4, a2 = v2 ? rp4[j + 128] : z4, a3 = v3 ? rp4[j + 192] : z4;
      const f32x4 b0 = v0 ? ri4[j] : z4, b1 = v1 ? ri4[j + 64] : z4, b2 = v2 ? ri4[j + 128] : z4, b3 = v3 ? ri4[j + 192] : z4;
      four(v0, a0, b0, j); four(v1, a1, b1, j + 64); four(v2, a2, b2, j + 128); four(v3, a3, b3, j + 192);
    }
    for (int t0 = head + 4 * nv; t0 < n; t0 += 64) {
      const int t = t0 + lane;
      const bool v = t < n;
      visit(v, t, v ? rp[t] : 0.f, v ? ri[t] : 0.f);
    }
  };
  // ---- (a) threshold: the r-th smallest of the lanes' minima over the first ns columns
  double tau = __builtin_inf();
  const int r = k + 7 > 16 ? k + 7 : 16;                // <= 23 of 64 lanes
  if (n > WS_CAP) {
    double mv = 0.0;
    int mj = -1;
    // sample size: ~r n / ns elements pass the threshold - 1024 columns up to 32k entries, more beyond so that ~512 are expected at most
    long long want = ((long long)n * r / 512 + 63) & ~63ll;
    const int ns = (int)(want < 1024 ? 1024 : (want < n ? want : n));
    for (int j = lane; j < ns; j += 64) {
      const double f = fusedw(rp[j], ri[j], j);
      const int jg = db_row0 + j;
      if (f == f && (mj < 0 || cand_less(f, jg, mv, mj))) { mv = f; mj = jg; }
    }
    int before = 0;
#pragma unroll 8
    for (int u = 0; u < 64; u++) {
      const double ov = __shfl(mv, u, 64);
      const int oj = __shfl(mj, u, 64);
      if (oj >= 0 && (mj < 0 || cand_less(ov, oj, mv, mj))) before++;
    }
    const unsigned long long hit = __ballot(mj >= 0 && before == r - 1);    // global indices are unique within a row: one lane, if any
    if (hit) tau = __shfl(mv, __ffsll((long long)hit) - 1, 64);
  }
  // ---- (b) everything at or below tau into the wave's list (a row that fits: every finite element)
  int L = 0;
  {
    const bool pre = !plain && fastdiv && tau < __builtin_inf();          // the affine test in front of the exact score, as in select_row
    const double a1 = p_weight / sp, b1 = 1.0 / si, c1 = -(p_weight * mp / sp + mi / si);
    const double tauE = tau + 1e-10 * (1.0 + fabs(a1) + fabs(b1) + fabs(c1) + fabs(tau));
    sweep([&](bool valid, int j, float vp, float vi) {
      bool take = false;
      double f = 0.0;
      if (valid && (!pre || __builtin_fma(a1, (double)vp, __builtin_fma(b1, (double)vi, c1)) <= tauE)) {
        f = fusedw(vp, vi, j);
        take = f <= tau;                                                    // (NaN compares false; tau = +Inf takes every finite or +Inf score)
      }
      const unsigned long long mk = __ballot(take);
      if (take) {
        const int slot = L + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
        if (slot < WS_CAP) { lv[slot] = f; lj[slot] = db_row0 + j; }
      }
      L += __popcll(mk);
    });
  }
  if (L <= WS_CAP) {
    // ---- (c) the k best of the list by (score, index): k rounds, each the smallest entry behind the one selected last
    double pv = -__builtin_inf();
    int pj = -1;
    for (int t = 0; t < k; t++) {
      double bv = 0.0;
      int bj = -1;
      for (int s2 = lane; s2 < L; s2 += 64) {
        const double v = lv[s2];
        const int jg = lj[s2];
        if ((pj < 0 || cand_less(pv, pj, v, jg)) && (bj < 0 || cand_less(v, jg, bv, bj))) { bv = v; bj = jg; }
      }
      wave_argmin(bv, bj);
      if (lane == 0) emit(t, bv, bj);
      if (bj < 0) {
        if (lane == 0) for (int u = t + 1; u < k; u++) emit(u, 0.0, -1);
        break;
      }
      pv = bv; pj = bj;
    }
    return;
  }
  // ---- fallback (the list overflowed): one sweep per selected element
  double pv = -__builtin_inf();
  int pj = -1;
  for (int t = 0; t < k; t++) {
    double bv = 0.0;
    int bj = -1;
    sweep([&](bool valid, int j, float vp, float vi) {
      if (!valid) return;
      const double f = fusedw(vp, vi, j);
      const int jg = db_row0 + j;
      if (f != f) return;
      if (pj >= 0 && !cand_less(pv, pj, f, jg)) return;                              // already selected
      if (bj < 0 || cand_less(f, jg, bv, bj)) { bv = f; bj = jg; }
    });
    wave_argmin(bv, bj);
    if (lane == 0) emit(t, bv, bj);
    if (bj < 0) {
      if (lane == 0) for (int u = t + 1; u < k; u++) emit(u, 0.0, -1);
      break;
    }
    pv = bv; pj = bj;
  }
}

// P lists of k (score, index) pairs per query [P][m][k] -> the k best [m][k]; missing entries are -1.  The P k entries (at most 8192) go
// through a radix selection / a bitonic network in LDS (lv: N floats, lj: N ints, N = the next power of two).
__device__ void merge_slices(const int32_t* sidx, const float* sscore, int P, int m, int k, int q, float* lv, int* lj, double* rv, int* rj,
                             unsigned* hist, int* ctl, int32_t* __restrict__ idx, float* __restrict__ score, double* __restrict__ score64,
                             int tid) {
  const int T = P * k;
  int N = 2;
  while (N < T) N <<= 1;
  for (int s = tid; s < N; s += 256) {
    int j = -1;
    float v = 0.f;
    if (s < T) {
      const size_t o = ((size_t)(s / k) * m + q) * k + (s % k);
      j = sidx[o]; v = sscore[o];
    }
    const bool ok = j >= 0 && v == v;
    lv[s] = ok ? v : __builtin_inff();
    lj[s] = ok ? j : NONE_J;
  }
  __syncthreads();
  auto put = [&](int t, bool ok, float v, int j) {
    idx[(size_t)q * k + t] = ok ? j : -1;
    const float f32 = ok ? v : __builtin_nanf("");
    score[(size_t)q * k + t] = f32;
    if (score64) score64[(size_t)q * k + t] = (double)f32;
  };
  if (k <= 249 && list_topk(lv, lj, T, k, rv, rj, hist, ctl, tid)) {      // radix selection + a 256-entry sort (see list_topk)
    for (int t = tid; t < k; t += 256) put(t, rj[t] != NONE_J, (float)rv[t], rj[t]);
    return;
  }
  for (int size = 2; size <= N; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (N >> 1); i += 256) {
        const int pos = 2 * i - (i & (stride - 1)), par = pos + stride;
        const bool up = (pos & size) == 0;
        const float a = lv[pos], b = lv[par];
        const int aj = lj[pos], bj = lj[par];
        if (cand_less((double)b, bj, (double)a, aj) == up) { lv[pos] = b; lj[pos] = bj; lv[par] = a; lj[par] = aj; }
      }
      __syncthreads();
    }
  for (int t = tid; t < k; t += 256) put(t, lj[t] != NONE_J, lv[t], lj[t]);
}

__global__ __launch_bounds__(256) void slice_merge_kernel(const int32_t* __restrict__ sidx, const float* __restrict__ sscore, int P, int m, int k,
                                                           int32_t* __restrict__ idx, float* __restrict__ score, double* __restrict__ score64) {
  extern __shared__ __attribute__((aligned(8))) char smem[];
  int N = 2;
  while (N < P * k) N <<= 1;
  __shared__ double rv[256];
  __shared__ int rj[256];
  __shared__ unsigned hist[256];
  __shared__ int ctl[4];
  merge_slices(sidx, sscore, P, m, k, blockIdx.x, reinterpret_cast<float*>(smem), reinterpret_cast<int*>(smem) + N, rv, rj, hist, ctl, idx, score,
                      score64, threadIdx.x);
}

// The selection of a row or - few query rows, grid m x P - of one slice of it (slice_merge_kernel then merges the P lists).
// (Round 4 measured the merge - and the moments' finish, and the wave selection behind the re-evaluation - as last-workgroup hand-offs
//  inside the launch in front of them, self-resetting tickets: an online call of one query gained nothing, 8 - 32 queries lost 10 - 15 % -
//  the fence, the ticket and the agent-scope re-reads cost what a launch costs, and the finishing workgroup starts later than a fresh
//  kernel would.  tools/experiments/README.md.)
template <int CAP>
#ifdef FS_NOLB
#define FS_BOUNDS __launch_bounds__(256)
#else
#define FS_BOUNDS __launch_bounds__(256, 3)
#endif
__global__ FS_BOUNDS void fuse_select_kernel(const float* __restrict__ d_p, const float* __restrict__ d_i,
                                                           const float* __restrict__ e_p, const float* __restrict__ e_i,
                                                           const double* __restrict__ mom2_all,
                                                           int m, int n, const double* __restrict__ mom_all, int G,
                                                           int q_row0, int db_row0, int mask_width, double p_weight,
                                                           int k, int32_t* __restrict__ idx, float* __restrict__ score, int P,
                                                           double* __restrict__ score64) {
  __shared__ SelLds<CAP> L;
  select_row<CAP>(d_p, d_i, e_p, e_i, mom2_all, m, n, mom_all, G, q_row0, db_row0, mask_width, p_weight, k, idx, score, P, score64, L);
}

}  // namespace

// slices per row for m query rows of n columns: 1 = one workgroup per row (enough rows to fill the chip, or short rows)
int select_slices(int m, int n) {
  if (m > 64 || n < 16384) return 1;
  // slices that fit the selection's LDS list skip its sampling phase; up to 8 rows (an online call) get slices of half the list: 49 slices
  // of a 100k row instead of 25 (m = 1: 0.353 -> 0.346 ms, m = 8: 0.374 -> 0.356 ms); from 16 rows up that loses (m = 32, f16 arithmetic:
  // 0.52 -> 0.59 ms in alternating runs on one box).  PR_SLICE_COLS overrides the width for A/B runs
  static const int forced = getenv("PR_SLICE_COLS") ? atoi(getenv("PR_SLICE_COLS")) : 0;
  const int cols = forced > 0 ? forced : (m <= 8 ? FS_CAP / 2 : FS_CAP);
  int P = (n + cols - 1) / cols;      // slices of half the selection's LDS list (they skip its sampling phase; 2048 columns: 49 slices of a 100k row, measured against 25)
  if (P > 64) P = 64;
  while (P > 1 && m * P > 1024) P >>= 1;            // capacity of the scratch (partial moments: 1024 (row, slice) pairs), ~4 workgroups per CU
  return P;
}
// slice lists (index + score) | partial moments of 1024 (row, slice) pairs
size_t select_scratch_bytes() { return (size_t)64 * 16 * 128 * 8 + (size_t)64 * 16 * 6 * 8; }

void launch_row_moments(hipStream_t st, const float* d_p, const float* d_i, int m, int n, double* mom, void* scratch) {
  if (m <= 0) return;
  const int P = scratch ? select_slices(m, n) : 1;
  if (P > 1) {
    double* part = reinterpret_cast<double*>(static_cast<char*>(scratch) + (size_t)64 * 16 * 128 * 8);
    hipLaunchKernelGGL(row_moments_slice_kernel, dim3(m, P), dim3(256), 0, st, d_p, d_i, n, m, P, part);
    hipLaunchKernelGGL(moments_finish_kernel, dim3((2 * m + 63) / 64), dim3(64), 0, st, d_p, d_i, n, m, P, part, mom);
    return;
  }
  hipLaunchKernelGGL(row_moments_kernel, dim3(m), dim3(256), 0, st, d_p, d_i, n, mom);
}

void launch_fuse_select(hipStream_t st, const float* d_p, const float* d_i, int m, int n, const double* mom_all,
                        int G, int q_row0, int db_row0, int mask_width, double p_weight, int k, int32_t* idx,
                        float* score, const float* e_p, const float* e_i, const double* mom2_all, void* scratch, double* score64) {
  if (m <= 0) return;
  int P = (scratch && k <= 128) ? select_slices(m, n) : 1;
  if ((size_t)P * m * k > (size_t)64 * 16 * 128) P = 1;      // capacity of the slice lists
  if (P > 1) {
    int32_t* sidx = static_cast<int32_t*>(scratch);
    float* ssc = reinterpret_cast<float*>(sidx + (size_t)64 * 16 * 128);
    int N = 2;
    while (N < P * k) N <<= 1;
    hipLaunchKernelGGL(fuse_select_kernel<FS_CAP>, dim3(m, P), dim3(256), 0, st, d_p, d_i, e_p, e_i, mom2_all, m, n, mom_all, G, q_row0, db_row0,
                       mask_width, p_weight, k, sidx, ssc, P, nullptr);
    // (up to 8192 entries = 64 KB of dynamic LDS next to ~4 KB of static arrays: above the 64 KB a launch gets without asking)
    if ((size_t)N * 8 > 40 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(slice_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)N * 8));
    hipLaunchKernelGGL(slice_merge_kernel, dim3(m), dim3(256), (size_t)N * 8, st, sidx, ssc, P, m, k, idx, score, score64);
    return;
  }
  static const bool wave_rows = !(getenv("PR_SELECT_WAVE") && atoi(getenv("PR_SELECT_WAVE")) == 0);     // A/B: PR_SELECT_WAVE=0
  static const int wave_max_n = getenv("PR_SELECT_WAVE_N") ? atoi(getenv("PR_SELECT_WAVE_N")) : 32768;
  if (wave_rows && !e_p && n <= wave_max_n && k <= 16 && m >= 64) {       // short rows: one wave per row (fuse_select_wave_kernel)
    hipLaunchKernelGGL(fuse_select_wave_kernel, dim3((m + 3) / 4), dim3(256), 0, st, d_p, d_i, m, n, mom_all, G, q_row0, db_row0, mask_width,
                       p_weight, k, idx, score, score64);
    return;
  }
  // (a 1024-entry list - 17 KB of LDS, 9 workgroups per CU instead of 3 - for whole rows with few results measured the same 0.87 ms per
  //  4096 x 100k: the sweep is bound by its ~30 fp64 instructions per element, not by resident waves)
  hipLaunchKernelGGL(fuse_select_kernel<FS_CAP>, dim3(m), dim3(256), 0, st, d_p, d_i, e_p, e_i, mom2_all, m, n, mom_all, G, q_row0, db_row0,
                     mask_width, p_weight, k, idx, score, 1, score64);
}

}  // namespace pr
