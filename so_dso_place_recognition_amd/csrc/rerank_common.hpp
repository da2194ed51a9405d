// rerank_common.hpp - device helpers shared by rerank.hip (fp64 re-evaluation of the selection's survivors) and exact_row.hip (the
// flagged queries' exact rows): the reference's pair distances in fp64 (processSC.m:15-33, processM2DP.m:12-22), Chan combination of
// the shards' moments, the (score, index) order of run_test.m:57, and the flagged-query list of a call.
#pragma once
#include "kernels.hpp"

namespace pr {
namespace {

__device__ __forceinline__ double ld(const void* p, int dtype, size_t i) {
  return dtype == 0 ? static_cast<const double*>(p)[i] : (double)static_cast<const float*>(p)[i];
}

// Sum over the 256 threads: inside a wave by lane exchanges (an xor butterfly: both partners add the same two numbers, so all 64 lanes hold
// the same bits - no barrier), the four wave sums through LDS, added in wave order by every thread.  Two barriers instead of ten: the pair
// functions below are chains of such reductions, and an online call waits for one pair per workgroup.
__device__ __forceinline__ double block_sum256(double v, double* red, int tid) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  const double r = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return r;
}

// MATLAB min: NaN only if every element is NaN
__device__ __forceinline__ double nanmin(double a, double b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }
__device__ __forceinline__ double block_min256(double v, double* red, int tid) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v = nanmin(v, __shfl_xor(v, s, 64));
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  const double r = nanmin(nanmin(red[0], red[1]), nanmin(red[2], red[3]));
  __syncthreads();
  return r;
}

// processSC.m:15-33 for one channel of one pair, fp64.  buf: 60 x 21 (query, padded sector stride) + 1200 (entry) doubles.
// Thread = (ring r, block of 5 consecutive shifts): it walks the 60 sectors of the entry once and keeps the 5 forward and
// 5 mirrored variants of its shifts in registers - a sliding window over the query's ring column, so every step costs three
// LDS reads (entry value, one new query value per direction) for ten multiply-adds; the first version read two operands
// per multiply-add and was LDS-bound at 3.2 ms per 4096 x 9 pairs.
__device__ double sc_pair_exact(const void* qsig, int qdt, size_t qoff, const void* dsig, int ddt, size_t doff,
                                double* buf /*60*21 + 1200*/, double* red, int tid) {
  double* qs = buf;
  double* ds = buf + 60 * 21;
  double pq = 0.0, pd = 0.0;
  for (int i = tid; i < 1200; i += 256) {
    const double x = ld(qsig, qdt, qoff + i), y = ld(dsig, ddt, doff + i);
    qs[(i / 20) * 21 + (i % 20)] = x;
    ds[i] = y;
    pq += x * x;
    pd += y * y;
  }
  const double nq = sqrt(block_sum256(pq, red, tid));
  const double nd = sqrt(block_sum256(pd, red, tid));
  for (int i = tid; i < 1200; i += 256) {                       // processSC.m:16,19 (0/0 = NaN stays NaN)
    const int a = (i / 20) * 21 + (i % 20);
    qs[a] = qs[a] / nq;
    ds[i] = ds[i] / nd;
  }
  __syncthreads();
  const int r = tid % 20, blk = tid / 20;                        // blk 0..11 (tid < 240): shifts 5 blk .. 5 blk + 4
  double af[5] = {0, 0, 0, 0, 0}, am[5] = {0, 0, 0, 0, 0};
  if (tid < 240) {
    double wf[5], wm[5];
    const double* qc = qs + r;
#pragma unroll
    for (int j = 0; j < 5; j++) { wf[j] = qc[(5 * blk + j) * 21]; wm[j] = wf[j]; }
    int nf = (5 * blk + 5) % 60;                                 // sector entering the forward window next
    int nm = (5 * blk + 59) % 60;                                // ... and the mirrored one
    for (int c0 = 0; c0 < 60; c0 += 5) {
#pragma unroll
      for (int u = 0; u < 5; u++) {                              // sector c = c0 + u of the entry (permute_sc, processSC.m:37-45)
        const double dv = ds[(c0 + u) * 20 + r];
#pragma unroll
        for (int j = 0; j < 5; j++) {
          af[j] += wf[(j + u) % 5] * dv;                         // q[(k0 + c) % 60], k0 = 5 blk + j
          am[j] += wm[(j + 5 - u) % 5] * dv;                     // q[(k0 - c) % 60]
        }
        wf[u] = qc[nf * 21];
        wm[(4 - u + 5) % 5] = qc[nm * 21];
        nf = nf == 59 ? 0 : nf + 1;
        nm = nm == 0 ? 59 : nm - 1;
      }
    }
  }
  __syncthreads();                                               // all reads of qs / ds done: the buffer becomes [variant][ring]
  if (tid < 240) {
#pragma unroll
    for (int j = 0; j < 5; j++) {
      buf[(2 * (5 * blk + j)) * 20 + r] = af[j];
      buf[(2 * (5 * blk + j) + 1) * 20 + r] = am[j];
    }
  }
  __syncthreads();
  double diff = __builtin_nan("");
  if (tid < 120) {
    double dot = 0.0;
#pragma unroll
    for (int q = 0; q < 20; q++) dot += buf[tid * 20 + q];
    diff = (1.0 - dot) / 2.0;                                    // processSC.m:30
  }
  const double best = block_min256(diff, red, tid);              // processSC.m:31 (its barriers also free buf for the next channel)
  return best;
}

// processM2DP.m:12-22 for one channel of one pair: rows [4][384], channel columns [192 ch, 192 ch + 192)
__device__ double m2dp_pair_exact(const void* qsig, int qdt, size_t qoff, const void* dsig, int ddt, size_t doff, int ch,
                                  double* red, int tid) {
  // thread = (a, b, part of 16 x 12 columns)
  const int ab = tid >> 4, part = tid & 15, a = ab >> 2, b = ab & 3;
  double s = 0.0;
  for (int c = part * 12; c < part * 12 + 12; c++)
    s += ld(qsig, qdt, qoff + (size_t)a * 384 + ch * 192 + c) * ld(dsig, ddt, doff + (size_t)b * 384 + ch * 192 + c);
  red[tid] = s;
  __syncthreads();
  double diff = __builtin_nan("");
  if (part == 0) {
    double dot = 0.0;
    for (int p = 0; p < 16; p++) dot += red[tid + p];
    diff = (1.0 - dot) / 2.0;                                   // processM2DP.m:15
  }
  __syncthreads();
  return block_min256(diff, red, tid);                          // processM2DP.m:19
}

__device__ void chan_combine(const double* mom_all, int G, int m, int q, int ch, double& mean, double& sd, double* count = nullptr) {
  double cn = 0.0, mu = 0.0, m2 = 0.0;
  for (int g = 0; g < G; g++) {                                 // rank order, as fuse_select_kernel
    const double* o = mom_all + (((size_t)g * m + q) * 2 + ch) * 3;
    const double nb = o[0], mb = o[1], m2b = o[2];
    if (nb <= 0.0) continue;
    const double tot = cn + nb, delta = mb - mu;
    mu += delta * (nb / tot);
    m2 += m2b + delta * delta * (cn * nb / tot);
    cn = tot;
  }
  mean = mu;
  sd = sqrt(m2 / (cn - 1.0));
  if (count) *count = cn;
}

__device__ __forceinline__ bool cand_before(double av, int aj, double bv, int bj) {   // NaN / -1 entries sort last
  const bool abad = (aj < 0) || (av != av), bbad = (bj < 0) || (bv != bv);
  if (abad != bbad) return bbad;
  if (abad) return false;
  return av < bv || (av == bv && aj < bj);
}

}  // namespace
}  // namespace pr
