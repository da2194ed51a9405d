// rerank.hip — the fp64 end of run_test.m:38-57 on gfx950: what makes the returned top-k and its scores those of the
// reference's double arithmetic although the all-pairs matchers work in fp32.
//
//   nan_fixup     distances to / from a zero-norm signature are NaN in MATLAB (processSC.m:16,19: 0/0); the matchers see
//                 such rows as all-zero operands, this kernel writes the NaNs.
//   rerank        for every (query, survivor of the fp32 top-(k+8) selection): the distance of that pair again, now
//                 straight from the raw signatures in fp64 and in the reference's own formulation - SC: rows / L2 norm,
//                 the 120 shifted / mirrored variants, (1 - dot)/2, minimum (processSC.m:15-33); M2DP: (1 - dot)/2 over
//                 the 4 x 4 sign variants (processM2DP.m:12-22) - and the fused z-score of run_test.m:40 with the row
//                 moments of all shards.  One workgroup per pair; ~0.6 M fp64 multiply-adds per SC pair.
//   rerank_sort   orders the survivors of a query by (fp64 score, index) and keeps k (run_test.m:57, ties -> lower index); one thread
//                 per query, or one wave per query when there are at most 64 of them (an online call).
//   merge_topk    k-way merge of the per-shard top-k lists of G database shards by (score, global index).
#include "kernels.hpp"

namespace pr {
namespace {

// zero-norm signatures (processSC.m:16,19: MATLAB divides by zero, the row / column of distances is NaN): thread = DB column, walking the
// query rows of its grid row; one launch for both kinds of flag
__global__ __launch_bounds__(256) void nan_fixup_kernel(float* __restrict__ d_p, float* __restrict__ d_i, int m, int n,
                                                         const int* __restrict__ qbad, const int* __restrict__ dbad) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int bc = dbad ? dbad[j] : 0;
  const float nanv = __builtin_nanf("");
  for (int q = blockIdx.y; q < m; q += gridDim.y) {
    const int b = bc | (qbad ? qbad[q] : 0);
    if (!b) continue;
    if (b & 1) d_p[(size_t)q * n + j] = nanv;
    if ((b & 2) && d_i) d_i[(size_t)q * n + j] = nanv;
  }
}

__device__ __forceinline__ double ld(const void* p, int dtype, size_t i) {
  return dtype == 0 ? static_cast<const double*>(p)[i] : (double)static_cast<const float*>(p)[i];
}

__device__ __forceinline__ double block_sum256(double v, double* red, int tid) {
  red[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// MATLAB min: NaN only if every element is NaN
__device__ __forceinline__ double block_min256(double v, double* red, int tid) {
  red[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      const double a = red[tid], b = red[tid + s];
      red[tid] = (a != a) ? b : ((b != b) ? a : (b < a ? b : a));
    }
    __syncthreads();
  }
  const double r = red[0];
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

struct RerankArgs {
  const void* q_sc; const void* db_sc; int sc_dt;               // raw SC signatures [m][2400] / [n_local][2400] or null
  const void* q_m2; const void* db_m2; int m2_dt;               // raw M2DP signatures [4 m][384] / [4 n_local][384] or null
  const double* mom_sc; const double* mom_m2;                   // [G][m][2][3] moments of all shards per descriptor type
  int m, n_local, G, q_row0, db_row0, mask_width, kin;
  double p_weight;
  const double* cand_sc; int k;                                 // fp32-pass scores of the candidates [m][kin], ascending, or null; the k wanted
  double eps_d, eps_mult;                                       // distance error bound of the all-pairs pass per channel, and the safety factor on it
  double* cand_part;                                            // [m][4][kin] or null: the four weighted channel z-scores (SC structure, SC intensity, M2DP count, M2DP intensity; 0 for an absent type) of every evaluated candidate, NaN in [0] otherwise (order_check_kernel)
};

// |all-pairs-pass score - exact score| <= this for a pair whose exact score is s, given the row statistics: distance error eps_d per
// channel over its sigma (w = sum of weight / sigma over the channels), amplified by 1 + s^2 / (n - 1) through the statistics, + the
// rounding of the score itself
__device__ __forceinline__ double score_err_bound(double eps_d, double w, double s, double cn) {
  return eps_d * w * (1.0 + s * s / fmax(cn - 1.0, 1.0)) + 1e-6 * fabs(s) + 1e-5;
}
__device__ double row_weight(const double* mom_sc, const double* mom_m2, int G, int m, int q, double p_weight, double* cn_out);

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

__global__ __launch_bounds__(256) void rerank_kernel(RerankArgs A, const int32_t* __restrict__ idx_in,
                                                      double* __restrict__ cand_score) {
  __shared__ double buf[60 * 21 + 1200];
  __shared__ double red[256];
  const int tid = threadIdx.x, q = blockIdx.x / A.kin, t = blockIdx.x % A.kin;
  const int jg = idx_in[(size_t)q * A.kin + t];
  double* out = cand_score + (size_t)q * A.kin + t;
  if (A.cand_part && tid < 4) A.cand_part[((size_t)q * 4 + tid) * A.kin + t] = __builtin_nan("");   // stays NaN unless the pair is evaluated (tid 0 overwrites its own store below: program order)
  if (jg < 0) { if (tid == 0) *out = __builtin_nan(""); return; }
  int dij = (A.q_row0 + q) - jg;
  if (dij < 0) dij = -dij;
  if (dij < A.mask_width) { if (tid == 0) *out = __builtin_inf(); return; }   // run_test.m:47-53
  const int jl = jg - A.db_row0;
  if (jl < 0 || jl >= A.n_local) { if (tid == 0) *out = __builtin_nan(""); return; }   // another shard's row: its owner evaluates it
  if (A.cand_sc && t >= A.k) {
    // Candidates beyond the k-th of the fp32 pass whose fp32 score is above the k-th by more than 64 x the error bound of an fp32
    // score cannot enter the exact top-k: they keep their fp32 score (it only has to sort behind the evaluated ones).  Bound of
    // |fp32 score - exact score| given the row statistics: distance error 1e-6 per channel over its sigma,
    // amplified by 1 + s^2 / (n - 1) through the statistics, + the rounding of the score itself.
    const double sk = A.cand_sc[(size_t)q * A.kin + A.k - 1], st = A.cand_sc[(size_t)q * A.kin + t];
    double cn = 2.0;
    const double w = row_weight(A.q_sc ? A.mom_sc : nullptr, A.q_m2 ? A.mom_m2 : nullptr, A.G, A.m, q, A.p_weight, &cn);
    const double delta = A.eps_mult * score_err_bound(A.eps_d, w, sk, cn);
    if (st > sk + delta) { if (tid == 0) *out = st; return; }   // (NaN / Inf statistics or scores: the comparison is false, the pair is evaluated)
  }
  double f = 0.0, z4[4] = {0.0, 0.0, 0.0, 0.0};
  if (A.q_sc) {
    for (int ch = 0; ch < 2; ch++) {
      const double d = sc_pair_exact(A.q_sc, A.sc_dt, (size_t)q * 2400 + ch * 1200, A.db_sc, A.sc_dt, (size_t)jl * 2400 + ch * 1200,
                                     buf, red, tid);
      double mean, sd;
      chan_combine(A.mom_sc, A.G, A.m, q, ch, mean, sd);
      z4[ch] = (ch == 0 ? A.p_weight : 1.0) * ((d - mean) / sd);    // run_test.m:40
      f += z4[ch];
    }
  }
  if (A.q_m2) {
    for (int ch = 0; ch < 2; ch++) {
      const double d = m2dp_pair_exact(A.q_m2, A.m2_dt, (size_t)q * 4 * 384, A.db_m2, A.m2_dt, (size_t)jl * 4 * 384, ch, red, tid);
      double mean, sd;
      chan_combine(A.mom_m2, A.G, A.m, q, ch, mean, sd);
      z4[2 + ch] = (ch == 0 ? A.p_weight : 1.0) * ((d - mean) / sd);
      f += z4[2 + ch];
    }
  }
  if (tid == 0) *out = f;
  if (A.cand_part && tid < 4) A.cand_part[((size_t)q * 4 + tid) * A.kin + t] = z4[tid];    // (every thread holds the reduced values)
}

__device__ double row_weight(const double* mom_sc, const double* mom_m2, int G, int m, int q, double p_weight, double* cn_out) {
  double w = 0.0, cn = 2.0;
  for (int ch = 0; ch < 2; ch++) {
    double mean, sd;
    if (mom_sc) { chan_combine(mom_sc, G, m, q, ch, mean, sd, &cn); w += (ch == 0 ? p_weight : 1.0) / sd; }
    if (mom_m2) { chan_combine(mom_m2, G, m, q, ch, mean, sd, &cn); w += (ch == 0 ? p_weight : 1.0) / sd; }
  }
  *cn_out = cn;
  return w;
}

// PR_SC_ARITH_F16: is the exact top-k of the candidates provably the exact top-k of ALL entries?  cand_sc = the all-pairs-pass scores of
// the k_in candidates (ascending: every entry that is NOT a candidate has a pass score >= the last one, T), score = the exact scores of
// the k selected.  Every non-candidate's exact score is >= T - err(T); if the exact k-th best is below that, nothing outside the list can
// enter the top-k (pruned candidates: rerank_kernel).  Otherwise - or when statistics / scores are not finite - the query is flagged.
__global__ __launch_bounds__(64) void margin_check_kernel(const double* __restrict__ mom_sc, const double* __restrict__ mom_m2, int G, int m,
                                                           double p_weight, int kin, const double* __restrict__ cand_sc, int k,
                                                           const double* __restrict__ score, double eps_d, int32_t* __restrict__ flags,
                                                           int32_t* __restrict__ count, const int32_t* __restrict__ order_flags) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= m) return;
  const double* cs = cand_sc + (size_t)q * kin;
  int flag = order_flags ? (order_flags[q] != 0) : 0;          // order_check_kernel: the re-evaluated order hangs on the pass's sigmas
  const double T = cs[kin - 1];
  if (T == T) {                                       // a full candidate list (NaN = fewer than k_in entries exist: nothing is outside it)
    double cn = 2.0;
    const double w = row_weight(mom_sc, mom_m2, G, m, q, p_weight, &cn);
    const double sk = score[(size_t)q * k + k - 1];
    const double lim = T - score_err_bound(eps_d, w, T, cn);
    if (T < __builtin_inf() && !(sk < lim)) flag = 1;   // (T = +Inf: everything left is masked)
  }
  flags[q] = flag;
  if (flag) atomicAdd(count, 1);
}

__device__ __forceinline__ bool cand_before(double av, int aj, double bv, int bj) {   // NaN / -1 entries sort last
  const bool abad = (aj < 0) || (av != av), bbad = (bj < 0) || (bv != bv);
  if (abad != bbad) return bbad;
  if (abad) return false;
  return av < bv || (av == bv && aj < bj);
}

// one thread per query: selection of the k best of `cnt` candidates laid out with the given strides
__device__ void select_k(const int32_t* idx, const double* sc, int cnt, int k, int32_t* oidx, double* osc, float* osc32) {
  unsigned long long taken_lo = 0, taken_hi = 0;                // cnt <= 128
  for (int t = 0; t < k; t++) {
    int best = -1;
    for (int c = 0; c < cnt; c++) {
      const bool tk = c < 64 ? (taken_lo >> c) & 1 : (taken_hi >> (c - 64)) & 1;
      if (tk) continue;
      if (best < 0 || cand_before(sc[c], idx[c], sc[best], idx[best])) best = c;
    }
    const bool ok = best >= 0 && idx[best] >= 0 && sc[best] == sc[best];
    if (best >= 0) { if (best < 64) taken_lo |= 1ull << best; else taken_hi |= 1ull << (best - 64); }
    oidx[t] = ok ? idx[best] : -1;
    const double v = ok ? sc[best] : __builtin_nan("");
    if (osc) osc[t] = v;
    if (osc32) osc32[t] = (float)v;
  }
}

__global__ __launch_bounds__(64) void rerank_sort_kernel(const int32_t* __restrict__ idx_in, const double* __restrict__ cand_score,
                                                          int m, int kin, int k, int32_t* __restrict__ idx, double* __restrict__ score,
                                                          float* __restrict__ score32) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= m) return;
  select_k(idx_in + (size_t)q * kin, cand_score + (size_t)q * kin, kin, k, idx + (size_t)q * k, score ? score + (size_t)q * k : nullptr,
           score32 ? score32 + (size_t)q * k : nullptr);
}

// The same with one WAVE per query (few query rows - an online call: one thread walking its 57 candidates alone is 16 us of load latency):
// lane l holds candidates l and l + 64, k rounds of a wave arg-min by cand_before with the winner retired.
// per-channel relative sigma error the order check allows (order_check_kernel below has the derivation)
__device__ __forceinline__ void order_eps(const double* mom_sc, const double* mom_m2, int Gmom, int m, int q, double eps_floor, double noise,
                                          double (&eps)[4]) {
  for (int c = 0; c < 4; c++) {
    eps[c] = 0.0;
    const double* mom = c < 2 ? mom_sc : mom_m2;
    if (!mom) continue;
    double mean, sd, cn = 2.0;
    chan_combine(mom, Gmom, m, q, c & 1, mean, sd, &cn);
    const double e = 4.0 * noise / (sd * sqrt(fmax(cn - 1.0, 1.0)));
    eps[c] = (e == e) ? fmax(eps_floor, e) : 1.0;             // (sigma = 0 or NaN: nothing about the order is certain)
  }
}

__global__ __launch_bounds__(64) void rerank_sort_wave_kernel(const int32_t* __restrict__ idx_in, const double* __restrict__ cand_score,
                                                               int m, int kin, int k, int32_t* __restrict__ idx, double* __restrict__ score,
                                                               float* __restrict__ score32, const double* __restrict__ cand_part,
                                                               const double* __restrict__ mom_sc, const double* __restrict__ mom_m2, int Gmom,
                                                               double eps_floor, double noise, int32_t* __restrict__ order_flags) {
  // cand_part != null: the order check of order_check_kernel folded into the selection rounds (k + 1 of them: the last one finds the best
  // candidate left out) - every winner is compared with the previous one; an online call saves a launch
  const int q = blockIdx.x, lane = threadIdx.x;
  double v[2], z[2][4];
  int j[2];
  double eps[4] = {0.0, 0.0, 0.0, 0.0};
  if (cand_part) order_eps(mom_sc, mom_m2, Gmom, m, q, eps_floor, noise, eps);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int c = lane + 64 * h;
    j[h] = c < kin ? idx_in[(size_t)q * kin + c] : -1;
    v[h] = c < kin ? cand_score[(size_t)q * kin + c] : __builtin_nan("");
    for (int cc = 0; cc < 4; cc++) z[h][cc] = (cand_part && c < kin) ? cand_part[((size_t)q * 4 + cc) * kin + c] : __builtin_nan("");
  }
  double pv = 0.0, pz[4] = {0.0, 0.0, 0.0, 0.0};
  bool have_prev = false;
  int flag = 0;
  const int rounds = cand_part ? k + 1 : k;
  for (int t = 0; t < rounds; t++) {
    const bool first = cand_before(v[0], j[0], v[1], j[1]) || !cand_before(v[1], j[1], v[0], j[0]);
    double bv = first ? v[0] : v[1];
    int bj = first ? j[0] : j[1];
    int bc = lane + (first ? 0 : 64);                       // the candidate's position: retires exactly one entry even among equal bad ones
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const double ov = __shfl_xor(bv, s, 64);
      const int oj = __shfl_xor(bj, s, 64), oc = __shfl_xor(bc, s, 64);
      if (cand_before(ov, oj, bv, bj) || (!cand_before(bv, bj, ov, oj) && oc < bc)) { bv = ov; bj = oj; bc = oc; }
    }
    const bool ok = bj >= 0 && bv == bv;
    if (lane == 0 && t < k) {
      idx[(size_t)q * k + t] = ok ? bj : -1;
      const double o = ok ? bv : __builtin_nan("");
      if (score) score[(size_t)q * k + t] = o;
      if (score32) score32[(size_t)q * k + t] = (float)o;
    }
    if (cand_part && ok) {                                   // (no further entries: nothing left to compare, -1 / NaN fill the rest)
      double wz[4];
      for (int cc = 0; cc < 4; cc++) wz[cc] = __shfl(bc < 64 ? z[0][cc] : z[1][cc], bc & 63, 64);
      const bool evd = wz[0] == wz[0];                      // evaluated (masked: +Inf, pruned: pass score - NaN parts)
      if (have_prev && evd) {
        double lim = 0.0, span = 0.0;
        for (int cc = 0; cc < 4; cc++) { const double dz = fabs(wz[cc] - pz[cc]); lim += eps[cc] * dz; span += dz; }
        if (span > 0.0 && !(bv - pv > lim)) flag = 1;
      }
      have_prev = evd;
      pv = bv;
      for (int cc = 0; cc < 4; cc++) pz[cc] = wz[cc];
    }
    if (bc == lane) { j[0] = -1; v[0] = __builtin_nan(""); }
    if (bc == lane + 64) { j[1] = -1; v[1] = __builtin_nan(""); }
  }
  if (order_flags && lane == 0) order_flags[q] = flag;
}

// cand_idx [m][kin] + the partial evaluations of G shards [G][m][kin] (NaN where the candidate is not the shard's) -> the k best.
// Every candidate has exactly one owner; a masked pair is +Inf at its owner.
__global__ __launch_bounds__(64) void rerank_finish_kernel(const int32_t* __restrict__ cand_idx, const double* __restrict__ part_all,
                                                            int G, int m, int kin, int k, int32_t* __restrict__ idx,
                                                            double* __restrict__ score) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= m) return;
  int32_t ci[128];
  double cs[128];
  for (int t = 0; t < kin; t++) {
    ci[t] = cand_idx[(size_t)q * kin + t];
    double v = __builtin_nan("");
    for (int g = 0; g < G; g++) {
      const double x = part_all[((size_t)g * m + q) * kin + t];
      if (x == x) { v = x; break; }
    }
    cs[t] = v;
  }
  select_k(ci, cs, kin, k, idx + (size_t)q * k, score + (size_t)q * k, nullptr);
}

// PR_SC_ARITH_F16: is the ORDER of the re-evaluated candidates certain?  Their exact scores are s = sum_c z_c over the (two or four)
// channels with the pair's distances exact (fp64) but the row statistics those of the single-product pass: z_c scales with 1 / sigma_c,
// and the pass's sigma_c is off by a relative eps_c.  Two candidates a, b with s_a <= s_b keep that order under the true sigmas if
//      s_b - s_a > sum_c eps_c |z_c(b) - z_c(a)|
// (always true when all channels agree on the order; the means shift both alike).  Walking the selected k and the best candidate left
// out, every ADJACENT pair must pass - then the whole chain is in its true order and the k-th / (k+1)-th boundary is the true one.
// eps_c: the pass's distance noise e (rms nu ~ 3e-5, |e| < 1.3e-4 observed) enters sigma^2 as (2 / n) sum (d_j - mu) e_j, i.e. a relative
// nu / (sigma_c sqrt(n)) on sigma - 1e-4 at n = 10^4, 1e-2 on a row of 24 - so eps_c = max(PR_F16_SIGMA_REL, 4 PR_F16_NOISE / (sigma_c
// sqrt(n - 1))) with PR_F16_NOISE = 1e-4 (include/place_recognition.h).  Pairs equal in every channel (duplicated signatures) are ordered
// by index and certain; candidates that were not evaluated (pruned: their pass score is beyond the k-th by more than the pass's error)
// are certain by that bound.  Anything else flags the query for the split-f16 pass.  One wave per query; score_all [G][m][kin] /
// part_all [G][m][4][kin] hold values at the candidate's owner, NaN elsewhere (G = 1: the scratch arrays of pr_rerank_dev).
__global__ __launch_bounds__(64) void order_check_kernel(const double* __restrict__ mom_sc, const double* __restrict__ mom_m2, int Gmom,
                                                          const int32_t* __restrict__ cand_idx, const double* __restrict__ score_all,
                                                          const double* __restrict__ part_all, int G, int m, int kin, int k,
                                                          const int32_t* __restrict__ idx_sel, double eps_floor, double noise,
                                                          int32_t* __restrict__ flags) {
  // one WAVE per query (a thread walking its 57 - 128 candidates alone is ~0.1 ms of dependent loads - an online call has one query):
  // lane l holds candidates l and l + 64; the members of S = selected k + best one left out are ranked by (score, index) through an LDS
  // copy, scattered to their rank, and lane p checks the adjacent pair (p, p + 1)
  __shared__ double s_v[128], t_v[129], t_z[4][129];
  __shared__ int s_j[128], s_in[128], t_ev[129];
  const int q = blockIdx.x, lane = threadIdx.x;
  double eps[4];
  order_eps(mom_sc, mom_m2, Gmom, m, q, eps_floor, noise, eps);
  double v[2], z[2][4];
  int j[2], ev[2], sel[2];
  for (int h = 0; h < 2; h++) {
    const int c = lane + 64 * h;
    j[h] = c < kin ? cand_idx[(size_t)q * kin + c] : -1;
    v[h] = __builtin_nan("");
    int own = -1;
    if (j[h] >= 0)
      for (int g = 0; g < G; g++) {
        const double x = score_all[((size_t)g * m + q) * kin + c];
        if (x == x) { v[h] = x; own = g; break; }
      }
    for (int cc = 0; cc < 4; cc++) z[h][cc] = own >= 0 ? part_all[(((size_t)own * m + q) * 4 + cc) * kin + c] : __builtin_nan("");
    ev[h] = own >= 0 && z[h][0] == z[h][0];                    // evaluated (masked pairs are +Inf, pruned ones keep their pass score: NaN parts)
    sel[h] = 0;
  }
  for (int u = 0; u < k; u++) {                                // (wave-uniform loads)
    const int want = idx_sel[(size_t)q * k + u];
    if (want < 0) break;                                       // fewer than k entries exist
    if (j[0] == want) sel[0] = 1;
    if (j[1] == want) sel[1] = 1;
  }
  {                                                            // the best candidate that was left out joins S
    const bool c0 = j[0] >= 0 && v[0] == v[0] && !sel[0], c1 = j[1] >= 0 && v[1] == v[1] && !sel[1];
    const bool first = c0 && (!c1 || cand_before(v[0], j[0], v[1], j[1]));
    double bv = first ? v[0] : v[1];
    int bj = (c0 || c1) ? (first ? j[0] : j[1]) : -1;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const double ov = __shfl_xor(bv, s, 64);
      const int oj = __shfl_xor(bj, s, 64);
      if (oj >= 0 && (bj < 0 || cand_before(ov, oj, bv, bj))) { bv = ov; bj = oj; }
    }
    if (bj >= 0) { if (j[0] == bj) sel[0] = 1; if (j[1] == bj) sel[1] = 1; }
  }
  for (int h = 0; h < 2; h++) { s_v[lane + 64 * h] = v[h]; s_j[lane + 64 * h] = j[h]; s_in[lane + 64 * h] = sel[h]; }
  __syncthreads();
  int rank[2] = {0, 0}, nS = 0;
  for (int d = 0; d < kin; d++) {                              // (broadcast reads)
    if (!s_in[d]) continue;
    nS++;
    const double dv = s_v[d];
    const int dj = s_j[d];
    if (cand_before(dv, dj, v[0], j[0])) rank[0]++;
    if (cand_before(dv, dj, v[1], j[1])) rank[1]++;
  }
  for (int h = 0; h < 2; h++)
    if (sel[h]) {
      t_v[rank[h]] = v[h]; t_ev[rank[h]] = ev[h];
      for (int cc = 0; cc < 4; cc++) t_z[cc][rank[h]] = z[h][cc];
    }
  __syncthreads();
  int flag = 0;
  for (int p = lane; p + 1 < nS; p += 64) {
    if (!t_ev[p] || !t_ev[p + 1]) continue;
    double lim = 0.0, span = 0.0;
    for (int cc = 0; cc < 4; cc++) {
      const double dz = fabs(t_z[cc][p + 1] - t_z[cc][p]);
      lim += eps[cc] * dz; span += dz;
    }
    if (span > 0.0 && !(t_v[p + 1] - t_v[p] > lim)) flag = 1;
  }
  flag = __any(flag);
  if (lane == 0) flags[q] = flag ? 1 : 0;
}

// ---- resolution of a query whose order check failed (single-shard calls): the row statistics of run_test.m:40 EXACTLY.  exact_row_kernel
// evaluates the distance of the query to every entry of the shard in fp64, in the reference's own formulation (the device functions of
// rerank_kernel), into d64 [channels: SC p, SC i, M2DP p, M2DP i][n]; moments64_kernel turns each present channel into (count, mean, M2)
// - two passes over the doubles, NaN left out (zero-norm signatures) - in the layout of pr_row_moments_dev.  ~23 ns per pair: 2.3 ms per
// flagged query and 100 000 entries.
__global__ __launch_bounds__(256) void exact_row_kernel(const void* __restrict__ q_sc, const void* __restrict__ db_sc, int sc_dt,
                                                         const void* __restrict__ q_m2, const void* __restrict__ db_m2, int m2_dt, int n,
                                                         double* __restrict__ d64) {
  __shared__ double buf[60 * 21 + 1200];
  __shared__ double red[256];
  const int tid = threadIdx.x;
  for (int j = blockIdx.x; j < n; j += gridDim.x) {
    if (q_sc)
      for (int ch = 0; ch < 2; ch++) {
        const double d = sc_pair_exact(q_sc, sc_dt, (size_t)ch * 1200, db_sc, sc_dt, (size_t)j * 2400 + ch * 1200, buf, red, tid);
        if (tid == 0) d64[(size_t)ch * n + j] = d;
      }
    if (q_m2)
      for (int ch = 0; ch < 2; ch++) {
        const double d = m2dp_pair_exact(q_m2, m2_dt, 0, db_m2, m2_dt, (size_t)j * 4 * 384, ch, red, tid);
        if (tid == 0) d64[(size_t)(2 + ch) * n + j] = d;
      }
  }
}

__global__ __launch_bounds__(256) void moments64_kernel(const double* __restrict__ d64, int n, double* __restrict__ mom_sc,
                                                         double* __restrict__ mom_m2) {
  __shared__ double red[256];
  const int tid = threadIdx.x, c = blockIdx.x;                  // channel 0..3
  double* out = (c < 2 ? mom_sc : mom_m2);
  if (!out) return;
  out += (c & 1) * 3;
  const double* x = d64 + (size_t)c * n;
  double s = 0.0, cnt = 0.0;
  for (int j = tid; j < n; j += 256) { const double v = x[j]; if (v == v) { s += v; cnt += 1.0; } }
  const double N = block_sum256(cnt, red, tid);
  const double mean = block_sum256(s, red, tid) / N;
  double m2 = 0.0;
  for (int j = tid; j < n; j += 256) { const double v = x[j]; if (v == v) m2 += (v - mean) * (v - mean); }
  const double M2 = block_sum256(m2, red, tid);
  if (tid == 0) { out[0] = N; out[1] = N > 0.0 ? mean : 0.0; out[2] = N > 0.0 ? M2 : 0.0; }
}

__global__ __launch_bounds__(256) void widen_kernel(const float* __restrict__ a, long long n, double* __restrict__ b) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = (double)a[i];
}

// idx_all [G][m][k], score_all [G][m][k] -> idx [m][k], score [m][k].  Every shard's list is ascending by (score, index) with its missing
// entries (-1 / NaN) last - as pr_fuse_select_dev and pr_rerank_dev write them - so this is a G-way merge with one cursor per list: no
// limit on G * k (the selection over a gathered array capped it at 128, which 8 shards x the k + 56 candidates of PR_SC_ARITH_F16 exceed).
__global__ __launch_bounds__(64) void merge_topk_kernel(const int32_t* __restrict__ idx_all, const double* __restrict__ score_all,
                                                         int G, int m, int k, int32_t* __restrict__ idx, double* __restrict__ score) {
  __shared__ unsigned char curs[64][65];                      // [cursor of list g][thread]: G <= 64, k <= 128 (a private array indexed at run
  const int q = blockIdx.x * 64 + threadIdx.x;                // time would live in scratch memory)
  if (q >= m) return;
  unsigned char (&cur)[64][65] = curs;
  const int th = threadIdx.x;
  for (int g = 0; g < G; g++) cur[g][th] = 0;
  for (int t = 0; t < k; t++) {
    int bg = -1, bj = -1;
    double bv = 0.0;
    for (int g = 0; g < G; g++) {
      if (cur[g][th] >= k) continue;
      const size_t o = ((size_t)g * m + q) * k + cur[g][th];
      const int j = idx_all[o];
      const double v = score_all[o];
      if (bg < 0 || cand_before(v, j, bv, bj)) { bg = g; bj = j; bv = v; }
    }
    const bool ok = bg >= 0 && bj >= 0 && bv == bv;
    if (bg >= 0) cur[bg][th]++;
    idx[(size_t)q * k + t] = ok ? bj : -1;
    score[(size_t)q * k + t] = ok ? bv : __builtin_nan("");
  }
}

}  // namespace

void launch_nan_fixup(hipStream_t st, float* d_p, float* d_i, int m, int n, const int* qbad, const int* dbad) {
  if (m <= 0 || n <= 0) return;
  if (qbad || dbad)
    hipLaunchKernelGGL(nan_fixup_kernel, dim3((n + 255) / 256, m < 64 ? m : 64), dim3(256), 0, st, d_p, d_i, m, n, qbad, dbad);
}

void launch_rerank(hipStream_t st, const void* q_sc, const void* db_sc, int sc_dt, const void* q_m2, const void* db_m2, int m2_dt,
                   const double* mom_sc, const double* mom_m2, int m, int n_local, int G, int q_row0, int db_row0, int mask_width,
                   double p_weight, int kin, const int32_t* idx_in, double* cand_score, int k, int32_t* idx, double* score,
                   float* score32, const double* cand_sc32, double eps_d, double* cand_part, double order_floor, double order_noise,
                   int32_t* order_flags) {
  if (m <= 0) return;
  RerankArgs A{q_sc, db_sc, sc_dt, q_m2, db_m2, m2_dt, mom_sc, mom_m2, m, n_local, G, q_row0, db_row0, mask_width, kin, p_weight, cand_sc32, k,
               eps_d > 0 ? eps_d : 1e-6, eps_d > 0 ? 2.0 : 64.0, cand_part};
  hipLaunchKernelGGL(rerank_kernel, dim3((unsigned)m * kin), dim3(256), 0, st, A, idx_in, cand_score);
  // order check (order_flags != null, needs cand_part): inside the wave selection for few queries, its own launch otherwise
  const bool oc = order_flags && cand_part;
  if (m <= 64 && kin <= 128) {
    hipLaunchKernelGGL(rerank_sort_wave_kernel, dim3(m), dim3(64), 0, st, idx_in, cand_score, m, kin, k, idx, score, score32,
                       oc ? cand_part : nullptr, mom_sc, mom_m2, G, order_floor, order_noise, oc ? order_flags : nullptr);
    return;
  }
  hipLaunchKernelGGL(rerank_sort_kernel, dim3((m + 63) / 64), dim3(64), 0, st, idx_in, cand_score, m, kin, k, idx, score, score32);
  if (oc)
    hipLaunchKernelGGL(order_check_kernel, dim3(m), dim3(64), 0, st, mom_sc, mom_m2, G, idx_in, cand_score, cand_part, 1, m, kin, k, idx,
                       order_floor, order_noise, order_flags);
}

void launch_rerank_partial(hipStream_t st, const void* q_sc, const void* db_sc, int sc_dt, const void* q_m2, const void* db_m2, int m2_dt,
                           const double* mom_sc, const double* mom_m2, int m, int n_local, int G, int q_row0, int db_row0, int mask_width,
                           double p_weight, int kin, const int32_t* idx_in, double* cand_score, const double* cand_sc32, int k, double eps_d,
                           double* cand_part) {
  if (m <= 0) return;
  RerankArgs A{q_sc, db_sc, sc_dt, q_m2, db_m2, m2_dt, mom_sc, mom_m2, m, n_local, G, q_row0, db_row0, mask_width, kin, p_weight, cand_sc32, k,
               eps_d > 0 ? eps_d : 1e-6, eps_d > 0 ? 2.0 : 64.0, cand_part};
  hipLaunchKernelGGL(rerank_kernel, dim3((unsigned)m * kin), dim3(256), 0, st, A, idx_in, cand_score);
}

void launch_margin_check(hipStream_t st, const double* mom_sc, const double* mom_m2, int G, int m, double p_weight, int kin,
                         const double* cand_sc, int k, const double* score, double eps_d, int32_t* flags, int32_t* count,
                         const int32_t* order_flags) {
  if (m <= 0) return;
  hipLaunchKernelGGL(margin_check_kernel, dim3((m + 63) / 64), dim3(64), 0, st, mom_sc, mom_m2, G, m, p_weight, kin, cand_sc, k, score, eps_d,
                     flags, count, order_flags);
}

void launch_order_check(hipStream_t st, const double* mom_sc, const double* mom_m2, int Gmom, const int32_t* cand_idx, const double* score_all,
                        const double* part_all, int G, int m, int kin, int k, const int32_t* idx_sel, double eps_floor, double noise,
                        int32_t* flags) {
  if (m <= 0) return;
  hipLaunchKernelGGL(order_check_kernel, dim3(m), dim3(64), 0, st, mom_sc, mom_m2, Gmom, cand_idx, score_all, part_all, G, m, kin, k,
                     idx_sel, eps_floor, noise, flags);
}

void launch_exact_row_moments(hipStream_t st, const void* q_sc, const void* db_sc, int sc_dt, const void* q_m2, const void* db_m2, int m2_dt,
                              int n, double* d64, double* mom_sc, double* mom_m2) {
  if (n <= 0) return;
  hipLaunchKernelGGL(exact_row_kernel, dim3(n < 8192 ? n : 8192), dim3(256), 0, st, q_sc, db_sc, sc_dt, q_m2, db_m2, m2_dt, n, d64);
  hipLaunchKernelGGL(moments64_kernel, dim3(4), dim3(256), 0, st, d64, n, q_sc ? mom_sc : nullptr, q_m2 ? mom_m2 : nullptr);
}

void launch_rerank_finish(hipStream_t st, const int32_t* cand_idx, const double* part_all, int G, int m, int kin, int k, int32_t* idx,
                          double* score) {
  if (m <= 0) return;
  hipLaunchKernelGGL(rerank_finish_kernel, dim3((m + 63) / 64), dim3(64), 0, st, cand_idx, part_all, G, m, kin, k, idx, score);
}

void launch_widen(hipStream_t st, const float* a, long long n, double* b) {
  if (n <= 0) return;
  hipLaunchKernelGGL(widen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, n, b);
}

void launch_merge_topk(hipStream_t st, const int32_t* idx_all, const double* score_all, int G, int m, int k, int32_t* idx, double* score) {
  if (m <= 0) return;
  hipLaunchKernelGGL(merge_topk_kernel, dim3((m + 63) / 64), dim3(64), 0, st, idx_all, score_all, G, m, k, idx, score);
}

}  // namespace pr
