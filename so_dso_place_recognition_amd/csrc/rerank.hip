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
#include "rerank_common.hpp"

namespace pr {
namespace {

// zero-norm signatures (processSC.m:16,19: MATLAB divides by zero, the row / column of distances is NaN).  One launch: workgroups [0, m)
// look at one QUERY each and leave at once unless it is flagged (then the workgroup writes the NaN row), the workgroups behind them look at
// 256 DB columns each and walk the rows only for a flagged column - m + n / 256 workgroups that read one flag each (the first version gave
// every column thread the whole list of query flags to read: 0.16 ms per 4096 x 100k step for nothing)
__global__ __launch_bounds__(256) void nan_fixup_kernel(float* __restrict__ d_p, float* __restrict__ d_i, int m, int n,
                                                         const int* __restrict__ qbad, const int* __restrict__ dbad) {
  const float nanv = __builtin_nanf("");
  if ((int)blockIdx.x < m) {
    const int q = blockIdx.x;
    const int b = qbad ? (qbad[2 * q] | qbad[2 * q + 1]) : 0;          // [row][channel] entries, see sc_pack.hip
    if (!b) return;
    for (int j = threadIdx.x; j < n; j += 256) {
      if (b & 1) d_p[(size_t)q * n + j] = nanv;
      if ((b & 2) && d_i) d_i[(size_t)q * n + j] = nanv;
    }
    return;
  }
  const int j = (blockIdx.x - m) * 256 + threadIdx.x;
  if (j >= n || !dbad) return;
  const int b = dbad[2 * j] | dbad[2 * j + 1];
  if (!b) return;
  for (int q = 0; q < m; q++) {
    if (b & 1) d_p[(size_t)q * n + j] = nanv;
    if ((b & 2) && d_i) d_i[(size_t)q * n + j] = nanv;
  }
}


// the wave-per-query selection + order check behind the re-evaluation (sort_wave_body below)
struct SortArgs {
  const int32_t* idx_in; const double* p5; int m, kin, k;
  int32_t* idx; double* score; float* score32;                  // [m][k] outputs (score / score32 may be null)
  int check;                                                    // fold the order check into the selection rounds
  const double* mom_sc; const double* mom_m2; int Gmom;
  double p_weight, eps_floor, noise;
  int32_t* order_flags;
  const double* cand_sc; double eps_d;                          // containment check: the candidates' all-pairs-pass scores [m][kin] (ascending) or null, the pass's distance error bound
};
struct RerankArgs {
  const void* q_sc; const void* db_sc; int sc_dt;               // raw SC signatures [m][2400] / [n_local][2400] or null
  const void* q_m2; const void* db_m2; int m2_dt;               // raw M2DP signatures [4 m][384] / [4 n_local][384] or null
  const double* mom_sc; const double* mom_m2;                   // [G][m][2][3] moments of all shards per descriptor type
  int m, n_local, G, q_row0, db_row0, mask_width, kin;
  double p_weight;
  const double* cand_sc; int k;                                 // fp32-pass scores of the candidates [m][kin], ascending, or null; the k wanted
  double eps_d, eps_mult;                                       // distance error bound of the all-pairs pass per channel, and the safety factor on it
  double* p5;                                                   // [m][5][kin]: per query the candidates' scores [kin], then their four exact channel distances [4][kin] (SC structure, SC intensity, M2DP count, M2DP intensity; 0 for an absent type; NaN in the first one = not evaluated: masked, pruned, another shard's)
};
// the "p5" layout: what a shard knows about the candidates of a query after its re-evaluation.  Shard g, query q, candidate t:
//   score = p5[(((size_t)g * m + q) * 5 + 0) * kin + t],   distance of channel c = p5[(((size_t)g * m + q) * 5 + 1 + c) * kin + t]
// One array travels through the sharded protocol's last all-gather, and it is all the order check and the fp64-statistics resolution need.
__device__ __forceinline__ size_t p5_at(int g, int m, int q, int c5, int kin, int t) { return (((size_t)g * m + q) * 5 + c5) * kin + t; }

// |all-pairs-pass score - exact score| <= this for a pair whose exact score is s, given the row statistics: distance error eps_d per
// channel over its sigma (w = sum of weight / sigma over the channels), amplified by 1 + s^2 / (n - 1) through the statistics, + the
// rounding of the score itself
__device__ __forceinline__ double score_err_bound(double eps_d, double w, double s, double cn) {
  return eps_d * w * (1.0 + s * s / fmax(cn - 1.0, 1.0)) + 1e-6 * fabs(s) + 1e-5;
}
__device__ double row_weight(const double* mom_sc, const double* mom_m2, int G, int m, int q, double p_weight, double* cn_out);


// Is candidate t of query q evaluated by this shard?  false: *skipv is the score it keeps (NaN: no such candidate / another shard's row,
// +Inf: masked, its pass score: pruned).
__device__ __forceinline__ bool pair_is_evaluated(const RerankArgs& A, const int32_t* idx_in, int q, int t, int* jl_out, double* skipv) {
  const int jg = idx_in[(size_t)q * A.kin + t];
  if (jg < 0) { *skipv = __builtin_nan(""); return false; }
  int dij = (A.q_row0 + q) - jg;
  if (dij < 0) dij = -dij;
  if (dij < A.mask_width) { *skipv = __builtin_inf(); return false; }   // run_test.m:47-53
  const int jl = jg - A.db_row0;
  *jl_out = jl;
  if (jl < 0 || jl >= A.n_local) { *skipv = __builtin_nan(""); return false; }   // another shard's row: its owner evaluates it
  if (A.cand_sc && t >= A.k) {
    // Candidates beyond the k-th of the fp32 pass whose fp32 score is above the k-th by more than 64 x the error bound of an fp32
    // score cannot enter the exact top-k: they keep their fp32 score (it only has to sort behind the evaluated ones).  Bound of
    // |fp32 score - exact score| given the row statistics: distance error 1e-6 per channel over its sigma,
    // amplified by 1 + s^2 / (n - 1) through the statistics, + the rounding of the score itself.
    const double sk = A.cand_sc[(size_t)q * A.kin + A.k - 1], st = A.cand_sc[(size_t)q * A.kin + t];
    double cn = 2.0;
    const double w = row_weight(A.q_sc ? A.mom_sc : nullptr, A.q_m2 ? A.mom_m2 : nullptr, A.G, A.m, q, A.p_weight, &cn);
    const double delta = A.eps_mult * score_err_bound(A.eps_d, w, sk, cn);
    if (st > sk + delta) { *skipv = st; return false; }         // (NaN / Inf statistics or scores: the comparison is false, the pair is evaluated)
  }
  return true;
}

// Batches: one WAVE per query decides for all of its candidates (lane = candidate), writes the scores of those that are not evaluated and
// appends the others to a work list - most candidates of a query with a clear winner are pruned, and a workgroup of 256 threads that only
// finds that out costs what ~10 us of dependent loads cost (8 of 9 workgroups of the metric workload, 56 of 57 in PR_SC_ARITH_F16).
// (16 queries per workgroup and ONE atomicAdd per workgroup for its share of the work list: one atomic per evaluated pair on the same word
// cost this kernel 52 - 61 us per 4096 - 5000 queries - same-address atomics from eight XCDs serialise at ~25 ns each)
constexpr int PLAN_QB = 16;
__global__ __launch_bounds__(256) void rerank_plan_kernel(RerankArgs A, const int32_t* __restrict__ idx_in, int32_t* __restrict__ work,
                                                           unsigned* __restrict__ count) {
  __shared__ unsigned wsum[4];
  __shared__ unsigned base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int q0 = blockIdx.x * PLAN_QB, items = PLAN_QB * A.kin;
  int mine[8];                                         // kin <= 128: at most 16 x 128 / 256 items per thread
  int nmine = 0;
  for (int it = tid; it < items; it += 256) {
    const int q = q0 + it / A.kin, t = it % A.kin;
    if (q >= A.m) continue;
    int jl = 0;
    double skipv = 0.0;
    if (pair_is_evaluated(A, idx_in, q, t, &jl, &skipv)) mine[nmine++] = q * A.kin + t;
    else { A.p5[p5_at(0, A.m, q, 0, A.kin, t)] = skipv; A.p5[p5_at(0, A.m, q, 1, A.kin, t)] = __builtin_nan(""); }
  }
  unsigned pre = (unsigned)nmine;                      // inclusive scan over the wave, then over the four waves
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(pre, o, 64); if (lane >= o) pre += v; }
  if (lane == 63) wsum[wv] = pre;
  __syncthreads();
  if (tid == 0) base = atomicAdd(count, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  __syncthreads();
  unsigned off = base + pre - (unsigned)nmine;
  for (int w2 = 0; w2 < wv; w2++) off += wsum[w2];
  for (int i = 0; i < nmine; i++) work[off + i] = mine[i];
}

// The exact distance(s) of a pair into its p5 slots, and its fused score = the sum of the channel terms in channel order (run_test.m:40).
//   MODE 0  one workgroup per (query, candidate), decision included (few queries per call would not fill the chip otherwise - unused now)
//   MODE 1  SPLIT, calls of few queries: one workgroup per (query, candidate, CHANNEL) - an online call waits for ONE channel's chain of
//           reductions instead of two or four in a row; the last of the pair's workgroups to finish (a self-resetting ticket per pair)
//           forms the score from the stored distances: bit for bit the same sum
//   MODE 2  batches: a fixed grid walks the work list of rerank_plan_kernel
template <int MODE>
__global__ __launch_bounds__(256) void rerank_kernel(RerankArgs A, const int32_t* __restrict__ idx_in, unsigned* __restrict__ tick,
                                                      const int32_t* __restrict__ work, const unsigned* __restrict__ count) {
  __shared__ double buf[60 * 21 + 1200];
  __shared__ double red[256];
  __shared__ int s_last;
  constexpr bool SPLIT = MODE == 1;
  const int tid = threadIdx.x;
  const int nch = SPLIT ? (A.q_sc ? 2 : 0) + (A.q_m2 ? 2 : 0) : 1;
  const unsigned total = MODE == 2 ? *count : 1u;
  for (unsigned w = MODE == 2 ? blockIdx.x : 0u; w < total; w += gridDim.x) {
    const int pair = MODE == 2 ? work[w] : (int)(blockIdx.x / nch), cl = MODE == 2 ? 0 : (int)(blockIdx.x % nch);
    const int q = pair / A.kin, t = pair % A.kin;
    double* out = A.p5 + p5_at(0, A.m, q, 0, A.kin, t);
    double* dout = A.p5 + p5_at(0, A.m, q, 1, A.kin, t);       // + c * kin: channel c
    int jl;
    if constexpr (MODE == 2) jl = idx_in[(size_t)q * A.kin + t] - A.db_row0;
    else {
      // not evaluated: the first distance slot is NaN, the score says why (the pair's first workgroup writes, the others just leave)
      double skipv = 0.0;
      if (!pair_is_evaluated(A, idx_in, q, t, &jl, &skipv)) {
        if (cl == 0 && tid == 0) { *out = skipv; dout[0] = __builtin_nan(""); }
        return;
      }
    }
    auto one = [&](int c) -> double {                            // channel 0, 1: SC structure / intensity; 2, 3: M2DP count / intensity
      if (c < 2) return sc_pair_exact(A.q_sc, A.sc_dt, (size_t)q * 2400 + c * 1200, A.db_sc, A.sc_dt, (size_t)jl * 2400 + c * 1200, buf, red, tid);
      return m2dp_pair_exact(A.q_m2, A.m2_dt, (size_t)q * 4 * 384, A.db_m2, A.m2_dt, (size_t)jl * 4 * 384, c - 2, red, tid);
    };
    auto term = [&](int c, double d) -> double {
      double mean, sd;
      chan_combine(c < 2 ? A.mom_sc : A.mom_m2, A.G, A.m, q, c & 1, mean, sd);
      return ((c & 1) ? 1.0 : A.p_weight) * ((d - mean) / sd);
    };
    if constexpr (!SPLIT) {
      double f = 0.0, d4[4] = {0.0, 0.0, 0.0, 0.0};
      for (int c = 0; c < 4; c++) {
        if (c < 2 ? !A.q_sc : !A.q_m2) continue;
        d4[c] = one(c);
        f += term(c, d4[c]);
      }
      if (tid == 0) *out = f;
      if (tid < 4) dout[(size_t)tid * A.kin] = d4[tid];          // (every thread holds the reduced values)
      __syncthreads();                                           // (MODE 2: buf / red serve the next pair)
    } else {
      const int c = A.q_sc ? cl : 2 + cl;
      const double d = one(c);
      if (tid == 0) {
        dout[(size_t)c * A.kin] = d;
        if (cl == 0) for (int a = 0; a < 4; a++) if (a < 2 ? !A.q_sc : !A.q_m2) dout[(size_t)a * A.kin] = 0.0;   // absent descriptor type
        __threadfence();
        const unsigned old = atomicAdd(&tick[pair], 1u);
        s_last = (old == (unsigned)nch - 1u);
        if (s_last) tick[pair] = 0u;                            // ready for the next launch
      }
      __syncthreads();
      if (!s_last) return;
      __threadfence();
      if (tid == 0) {
        double f = 0.0;                                         // (the other channels' distances: agent-scope loads, out of L2)
        for (int a = 0; a < 4; a++) {
          if (a < 2 ? !A.q_sc : !A.q_m2) continue;
          f += term(a, __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long*>(dout + (size_t)a * A.kin), __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT)));
        }
        *out = f;
      }
    }
  }
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

__global__ __launch_bounds__(64) void rerank_sort_kernel(const int32_t* __restrict__ idx_in, const double* __restrict__ p5,
                                                          int m, int kin, int k, int32_t* __restrict__ idx, double* __restrict__ score,
                                                          float* __restrict__ score32) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= m) return;
  select_k(idx_in + (size_t)q * kin, p5 + p5_at(0, m, q, 0, kin, 0), kin, k, idx + (size_t)q * k, score ? score + (size_t)q * k : nullptr,
           score32 ? score32 + (size_t)q * k : nullptr);
}

// Row statistics of the (up to) four channels of a query - SC structure, SC intensity, M2DP count, M2DP intensity - from the shards' moments,
// combined in rank order: weight (p or 1; 0 = channel absent), mean, sigma, and the relative sigma error the order check allows
// (order_check_kernel below has the derivation)
struct RowStats { double w[4], mean[4], sd[4], eps[4], cn; };
__device__ __forceinline__ void row_stats(const double* mom_sc, const double* mom_m2, int Gmom, int m, int q, double p_weight, double eps_floor,
                                          double noise, RowStats& S) {
  S.cn = 2.0;
  for (int c = 0; c < 4; c++) {
    S.w[c] = 0.0; S.mean[c] = 0.0; S.sd[c] = 1.0; S.eps[c] = 0.0;
    const double* mom = c < 2 ? mom_sc : mom_m2;
    if (!mom) continue;
    double cn = 2.0;
    chan_combine(mom, Gmom, m, q, c & 1, S.mean[c], S.sd[c], &cn);
    S.cn = cn;
    S.w[c] = (c & 1) ? 1.0 : p_weight;
    // |sigma(d + e) - sigma(d)| <= max |e| for ANY error pattern (no independence assumed: the entries of a cluster of near-copies share
    // their error), + the systematic compression of the row by the minimum over noisy variants
    const double e = eps_floor + noise / S.sd[c];
    S.eps[c] = (e == e && e < 1.0) ? e : 1.0;                   // (sigma = 0 or NaN: nothing about the order is certain)
  }
}
// Containment (run_test.m:57 takes the minimum over the WHOLE row; the re-evaluation sees the k_in best of the all-pairs pass): every entry
// outside the candidate list has a pass score >= T = the last candidate's, hence an exact-distance score >= T - err(T) under the pass's row
// statistics (score_err_bound), and under the TRUE sigmas it moves against a listed entry by at most sum_c eps_c w_c R_c / sigma_c (eps_c: the
// relative sigma error the order check allows; R_c: the range of a channel's distances - 1 for SC's (1 - cos)/2, 2 for M2DP's [-0.5, 1.5]).
// If the exact k-th best score sk is not below T by more than both, an entry the list does not hold could belong to the top-k: 1 = not
// certain -> the query is answered from its exact row (exact_row.hip).  T NaN: fewer than k_in entries exist; T = +Inf: the rest is masked.
__device__ __forceinline__ int not_contained(const RowStats& S, double eps_d, double T, double sk) {
  if (!(T == T) || !(T < __builtin_inf())) return 0;
  double w = 0.0, slack = 0.0;
  for (int c = 0; c < 4; c++) {
    if (S.w[c] == 0.0) continue;
    w += S.w[c] / S.sd[c];
    slack += S.eps[c] * S.w[c] * (c < 2 ? 1.0 : 2.0) / S.sd[c];
  }
  return !(sk < T - score_err_bound(eps_d, w, T, S.cn) - slack) ? 1 : 0;
}
// PR_SC_ARITH_F16: is the exact top-k of the candidates provably the exact top-k of ALL entries?  cand_sc = the all-pairs-pass scores of
// the k_in candidates (ascending: every entry that is NOT a candidate has a pass score >= the last one, T), score = the exact scores of
// the k selected.  Every non-candidate's exact score is >= T - err(T); if the exact k-th best is below that, nothing outside the list can
// enter the top-k (pruned candidates: rerank_kernel).  Otherwise - or when statistics / scores are not finite - the query is flagged.
__global__ __launch_bounds__(64) void margin_check_kernel(const double* __restrict__ mom_sc, const double* __restrict__ mom_m2, int G, int m,
                                                           double p_weight, int kin, const double* __restrict__ cand_sc, int k,
                                                           const double* __restrict__ score, double eps_d, int32_t* __restrict__ flags,
                                                           int32_t* __restrict__ count, const int32_t* __restrict__ order_flags,
                                                           double eps_floor, double noise) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= m) return;
  const double* cs = cand_sc + (size_t)q * kin;
  int flag = order_flags ? (order_flags[q] != 0) : 0;          // order_check_kernel: the re-evaluated order hangs on the pass's sigmas
  const double T = cs[kin - 1];
  if (T == T) {                                       // a full candidate list (NaN = fewer than k_in entries exist: nothing is outside it)
    // Round 6 tried not_contained()'s rigorous sigma slack here (sum_c eps_c w_c R_c / sigma_c with the f16 pass's eps_c = 2e-4 + 2e-4 / sigma_c
    // and the FULL range R_c of a channel's distances): ~0.7 z-units at the usual sigma ~ 0.03 - more than the k-th to (k + 56)-th score
    // gap of a row's dense part, i.e. every query of a k > 1 call fell back to split-f16 (tests/test_gpu_f16.py: 56 of 56) and the arithmetic
    // was pointless.  Kept instead: eps_d = PR_F16_DISTANCE_BOUND = 2e-3, the WORST-CASE distance error of the pass (16 x the 1.25e-4
    // observed), which puts ~0.2 z-units of margin here - five times what the pass's sigma error (relative 7e-3) moves an entry of
    // |z_c| <= 5 against a listed one; entries further out than that in one channel AND inside the margin in the sum are the case this
    // arithmetic does not bound (the split-f16 default does: not_contained).  eps_floor / noise stay in the signature for that experiment.
    (void)eps_floor; (void)noise;
    double cn = 2.0;
    const double w = row_weight(mom_sc, mom_m2, G, m, q, p_weight, &cn);
    const double sk = score[(size_t)q * k + k - 1];
    const double lim = T - score_err_bound(eps_d, w, T, cn);
    if (T < __builtin_inf() && !(sk < lim)) flag = 1;   // (T = +Inf: everything left is masked)
  }
  flags[q] = flag;
  if (flag) atomicAdd(count, 1);
}


// the weighted channel z-score exactly as rerank_kernel forms it (run_test.m:40)
__device__ __forceinline__ double chan_z(const RowStats& S, int c, double d) { return S.w[c] == 0.0 ? 0.0 : S.w[c] * ((d - S.mean[c]) / S.sd[c]); }

// The same with one WAVE per query (few query rows - an online call: one thread walking its 57 candidates alone is 16 us of load latency):
// lane l holds candidates l and l + 64, k rounds of a wave arg-min by cand_before with the winner retired.
__device__ void sort_wave_body(const SortArgs& a, int q, int lane) {
  // check: the order check of order_check_kernel folded into the selection rounds (k + 1 of them: the last one finds the best
  // candidate left out) - every winner is compared with the previous one; an online call saves a launch
  const int m = a.m, kin = a.kin, k = a.k, check = a.check;
  const double* p5 = a.p5;
  double v[2], z[2][4];
  int j[2];
  RowStats S;
  if (check) row_stats(a.mom_sc, a.mom_m2, a.Gmom, m, q, a.p_weight, a.eps_floor, a.noise, S);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int c = lane + 64 * h;
    j[h] = c < kin ? a.idx_in[(size_t)q * kin + c] : -1;
    v[h] = c < kin ? p5[p5_at(0, m, q, 0, kin, c)] : __builtin_nan("");
    const double d0 = (check && c < kin) ? p5[p5_at(0, m, q, 1, kin, c)] : __builtin_nan("");
    for (int cc = 0; cc < 4; cc++) {
      const double d = (d0 == d0) ? p5[p5_at(0, m, q, 1 + cc, kin, c)] : __builtin_nan("");
      z[h][cc] = (d0 == d0) ? chan_z(S, cc, d) : __builtin_nan("");
    }
  }
  double pv = 0.0, pz[4] = {0.0, 0.0, 0.0, 0.0}, sk = __builtin_nan(""), kth_v = 0.0, kth_z[4] = {0.0, 0.0, 0.0, 0.0};
  bool have_prev = false, kth_ev = false;
  int flag = 0;
  const int rounds = check ? k + 1 : k;
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
    if (t == k - 1 && ok) sk = bv;
    if (lane == 0 && t < k) {
      a.idx[(size_t)q * k + t] = ok ? bj : -1;
      const double o = ok ? bv : __builtin_nan("");
      if (a.score) a.score[(size_t)q * k + t] = o;
      if (a.score32) a.score32[(size_t)q * k + t] = (float)o;
    }
    if (check && ok) {                                       // (no further entries: nothing left to compare, -1 / NaN fill the rest)
      double wz[4];
      for (int cc = 0; cc < 4; cc++) wz[cc] = __shfl(bc < 64 ? z[0][cc] : z[1][cc], bc & 63, 64);
      const bool evd = wz[0] == wz[0];                      // evaluated (masked: +Inf, pruned: pass score - NaN distances)
      if (t == k - 1) { kth_ev = evd; kth_v = bv; for (int cc = 0; cc < 4; cc++) kth_z[cc] = wz[cc]; }
      if (have_prev && evd) {
        double lim = 0.0, span = 0.0;
        for (int cc = 0; cc < 4; cc++) { const double dz = fabs(wz[cc] - pz[cc]); lim += S.eps[cc] * dz; span += dz; }
        if (span > 0.0 && !(bv - pv > lim)) flag = 1;
      }
      have_prev = evd;
      pv = bv;
      for (int cc = 0; cc < 4; cc++) pz[cc] = wz[cc];
    }
    if (bc == lane) { j[0] = -1; v[0] = __builtin_nan(""); }
    if (bc == lane + 64) { j[1] = -1; v[1] = __builtin_nan(""); }
  }
  if (check && kth_ev) {                                     // no candidate still in the lanes may overtake the k-th selected one (order_check_kernel)
    int f2 = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (j[h] < 0 || !(z[h][0] == z[h][0])) continue;       // retired, absent, or not evaluated
      double lim = 0.0, span = 0.0;
      for (int cc = 0; cc < 4; cc++) { const double dz = fabs(z[h][cc] - kth_z[cc]); lim += S.eps[cc] * dz; span += dz; }
      if (span > 0.0 && !(v[h] - kth_v > lim)) f2 = 1;
    }
    if (__any(f2)) flag |= 1;
  }
  if (check && a.cand_sc && not_contained(S, a.eps_d, a.cand_sc[(size_t)q * kin + kin - 1], sk)) flag |= 2;
  if (a.order_flags && lane == 0) a.order_flags[q] = flag;
}
__global__ __launch_bounds__(64) void rerank_sort_wave_kernel(SortArgs a) { sort_wave_body(a, blockIdx.x, threadIdx.x); }
// cand_idx [m][kin] + the evaluations of G shards, p5_all [G][m][5][kin] (NaN scores where the candidate is not the shard's) -> the k best.
// Every candidate has exactly one owner; a masked pair is +Inf at its owner.
__global__ __launch_bounds__(64) void rerank_finish_kernel(const int32_t* __restrict__ cand_idx, const double* __restrict__ p5_all,
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
      const double x = p5_all[p5_at(g, m, q, 0, kin, t)];
      if (x == x) { v = x; break; }
    }
    cs[t] = v;
  }
  select_k(ci, cs, kin, k, idx + (size_t)q * k, score + (size_t)q * k, nullptr);
}

// Is the ORDER of the re-evaluated candidates certain?  Their exact scores are s = sum_c z_c over the (two or four) channels with the pair's
// distances exact (fp64) but the row statistics those of the all-pairs pass: z_c scales with 1 / sigma_c, and the pass's sigma_c is off by a
// relative eps_c.  Two candidates a, b with s_a <= s_b keep that order under the true sigmas if
//      s_b - s_a > sum_c eps_c |z_c(b) - z_c(a)|
// (always true when all channels agree on the order; the means shift both alike).  Walking the selected k and the best candidate left
// out, every ADJACENT pair must pass - then the whole chain is in its true order and the k-th / (k+1)-th boundary is the true one.
// eps_c = SIGMA_REL + DIST_ERR / sigma_c (the PR_F16_* or PR_F32_* constants of include/place_recognition.h): the pass's distance errors e_j
// move sigma by at most max |e_j| whatever their pattern - also when a cluster of near-copies shares ONE error, where the round-4 form
// 4 rms / (sigma sqrt(n - 1)) (independent errors) was 3 - 40 x too small and let a swapped pair through (tools/fuzz_all.py seed 13 case 21).
// Pairs equal in every channel (duplicated signatures) are ordered by index and certain; candidates that were not evaluated (pruned: their
// pass score is beyond the k-th by more than the pass's error) are certain by that bound.  A flagged query is answered from its exact fp64
// row (exact_row.hip) or, in PR_SC_ARITH_F16, by the split-f16 pass.  One wave per query; p5_all [G][m][5][kin] holds values at the candidate's owner, NaN elsewhere.
__global__ __launch_bounds__(64) void order_check_kernel(const double* __restrict__ mom_sc, const double* __restrict__ mom_m2, int Gmom,
                                                          const int32_t* __restrict__ cand_idx, const double* __restrict__ p5_all, int G, int m,
                                                          int kin, int k, const int32_t* __restrict__ idx_sel, double p_weight, double eps_floor,
                                                          double noise, int32_t* __restrict__ flags, const double* __restrict__ cand_sc,
                                                          const double* __restrict__ score_sel, double eps_d) {
  // one WAVE per query (a thread walking its 57 - 128 candidates alone is ~0.1 ms of dependent loads - an online call has one query):
  // lane l holds candidates l and l + 64; the members of S = selected k + best one left out are ranked by (score, index) through an LDS
  // copy, scattered to their rank, and lane p checks the adjacent pair (p, p + 1)
  __shared__ double s_v[128], t_v[129], t_z[4][129];
  __shared__ int s_j[128], s_in[128], t_ev[129];
  const int q = blockIdx.x, lane = threadIdx.x;
  RowStats S;
  row_stats(mom_sc, mom_m2, Gmom, m, q, p_weight, eps_floor, noise, S);
  double v[2], z[2][4];
  int j[2], ev[2], sel[2];
  for (int h = 0; h < 2; h++) {
    const int c = lane + 64 * h;
    j[h] = c < kin ? cand_idx[(size_t)q * kin + c] : -1;
    v[h] = __builtin_nan("");
    int own = -1;
    if (j[h] >= 0)
      for (int g = 0; g < G; g++) {
        const double x = p5_all[p5_at(g, m, q, 0, kin, c)];
        if (x == x) { v[h] = x; own = g; break; }
      }
    const double d0 = own >= 0 ? p5_all[p5_at(own, m, q, 1, kin, c)] : __builtin_nan("");
    ev[h] = d0 == d0;                                          // evaluated (masked pairs are +Inf, pruned ones keep their pass score: NaN distances)
    for (int cc = 0; cc < 4; cc++) z[h][cc] = ev[h] ? chan_z(S, cc, p5_all[p5_at(own, m, q, 1 + cc, kin, c)]) : __builtin_nan("");
    sel[h] = 0;
  }
  bool has_out = false;
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
    has_out = bj >= 0;
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
      lim += S.eps[cc] * dz; span += dz;
    }
    if (span > 0.0 && !(t_v[p + 1] - t_v[p] > lim)) flag = 1;
  }
  // ... and no candidate further down may overtake the k-th selected one: the chain above orders the selected k and the best one left out,
  // but a candidate of another channel composition anywhere in the list can leap over all of them when the sigmas move (tools/fuzz_all.py
  // seed 13 case 21: rank 43 of the f16-statistics order was rank 2 of the true one)
  {
    const int kk = has_out ? nS - 2 : nS - 1;
    if (kk >= 0 && t_ev[kk])
      for (int h = 0; h < 2; h++) {
        if (sel[h] || !ev[h] || j[h] < 0) continue;
        double lim = 0.0, span = 0.0;
        for (int cc = 0; cc < 4; cc++) { const double dz = fabs(z[h][cc] - t_z[cc][kk]); lim += S.eps[cc] * dz; span += dz; }
        if (span > 0.0 && !(v[h] - t_v[kk] > lim)) flag = 1;
      }
  }
  flag = __any(flag) ? 1 : 0;
  // (bit 1) the containment check of not_contained(): cand_sc = the candidates' pass scores [m][kin], score_sel = the exact scores of the k selected
  if (cand_sc && score_sel && not_contained(S, eps_d, cand_sc[(size_t)q * kin + kin - 1], score_sel[(size_t)q * k + k - 1])) flag |= 2;
  if (lane == 0) flags[q] = flag;
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
    hipLaunchKernelGGL(nan_fixup_kernel, dim3(m + (n + 255) / 256), dim3(256), 0, st, d_p, d_i, m, n, qbad, dbad);
}

// few queries: one workgroup per (pair, channel); batches: a plan (one wave per query) + a fixed grid over the pairs that ARE evaluated
static bool rerank_split(int m) { return m <= 64; }
static void launch_rerank_kernel(hipStream_t st, const RerankArgs& A, const int32_t* idx_in, unsigned* tick, size_t tick_cap) {
  const unsigned nch = (A.q_sc ? 2u : 0u) + (A.q_m2 ? 2u : 0u);
  if (rerank_split(A.m)) {
    hipLaunchKernelGGL(rerank_kernel<1>, dim3((unsigned)A.m * A.kin * nch), dim3(256), 0, st, A, idx_in, tick, (const int32_t*)nullptr, (const unsigned*)nullptr);
    return;
  }
  // tick: [cap] tickets | [cap] work list | [1] its length
  const size_t pairs = (size_t)A.m * A.kin;
  int32_t* work = reinterpret_cast<int32_t*>(tick + tick_cap);
  unsigned* count = tick + 2 * tick_cap;
  launch_zero_ints(st, reinterpret_cast<int*>(count), 1);     // (a kernel: tiny memset nodes of a captured graph have been seen not to replay, see pr_sigset_pack)
  hipLaunchKernelGGL(rerank_plan_kernel, dim3((A.m + PLAN_QB - 1) / PLAN_QB), dim3(256), 0, st, A, idx_in, work, count);
  const unsigned grid = pairs < 7168 ? (unsigned)pairs : 7168u;      // 256 CUs x 7 workgroups (their LDS) x 4 rounds
  hipLaunchKernelGGL(rerank_kernel<2>, dim3(grid), dim3(256), 0, st, A, idx_in, tick, work, count);
}

void launch_rerank(hipStream_t st, const void* q_sc, const void* db_sc, int sc_dt, const void* q_m2, const void* db_m2, int m2_dt,
                   const double* mom_sc, const double* mom_m2, int m, int n_local, int G, int q_row0, int db_row0, int mask_width,
                   double p_weight, int kin, const int32_t* idx_in, double* p5, unsigned* tick, size_t tick_cap, int k, int32_t* idx, double* score,
                   float* score32, const double* cand_sc32, double eps_d, double order_floor, double order_noise, int32_t* order_flags) {
  if (m <= 0) return;
  // (containment: fp32-grade passes only - eps_d = 0; the single-product arithmetic has pr_f16_margin_dev for it)
  const double* contain_sc = eps_d > 0 ? nullptr : cand_sc32;
  SortArgs S{idx_in, p5, m, kin, k, idx, score, score32, order_flags ? 1 : 0, q_sc ? mom_sc : nullptr, q_m2 ? mom_m2 : nullptr, G, p_weight,
             order_floor, order_noise, order_flags, contain_sc, 1e-6};
  RerankArgs A{q_sc, db_sc, sc_dt, q_m2, db_m2, m2_dt, mom_sc, mom_m2, m, n_local, G, q_row0, db_row0, mask_width, kin, p_weight, cand_sc32, k,
               eps_d > 0 ? eps_d : 1e-6, eps_d > 0 ? 2.0 : 64.0, p5};
  launch_rerank_kernel(st, A, idx_in, tick, tick_cap);
  if (m <= 64) { hipLaunchKernelGGL(rerank_sort_wave_kernel, dim3(m), dim3(64), 0, st, S); return; }
  hipLaunchKernelGGL(rerank_sort_kernel, dim3((m + 63) / 64), dim3(64), 0, st, idx_in, p5, m, kin, k, idx, score, score32);
  if (order_flags)
    hipLaunchKernelGGL(order_check_kernel, dim3(m), dim3(64), 0, st, q_sc ? mom_sc : nullptr, q_m2 ? mom_m2 : nullptr, G, idx_in, p5, 1, m, kin, k,
                       idx, p_weight, order_floor, order_noise, order_flags, score ? contain_sc : nullptr, (const double*)score, 1e-6);
}

void launch_rerank_partial(hipStream_t st, const void* q_sc, const void* db_sc, int sc_dt, const void* q_m2, const void* db_m2, int m2_dt,
                           const double* mom_sc, const double* mom_m2, int m, int n_local, int G, int q_row0, int db_row0, int mask_width,
                           double p_weight, int kin, const int32_t* idx_in, double* p5, unsigned* tick, size_t tick_cap, const double* cand_sc32, int k,
                           double eps_d) {
  if (m <= 0) return;
  RerankArgs A{q_sc, db_sc, sc_dt, q_m2, db_m2, m2_dt, mom_sc, mom_m2, m, n_local, G, q_row0, db_row0, mask_width, kin, p_weight, cand_sc32, k,
               eps_d > 0 ? eps_d : 1e-6, eps_d > 0 ? 2.0 : 64.0, p5};
  launch_rerank_kernel(st, A, idx_in, tick, tick_cap);
}

void launch_margin_check(hipStream_t st, const double* mom_sc, const double* mom_m2, int G, int m, double p_weight, int kin,
                         const double* cand_sc, int k, const double* score, double eps_d, int32_t* flags, int32_t* count,
                         const int32_t* order_flags, double eps_floor, double noise) {
  if (m <= 0) return;
  hipLaunchKernelGGL(margin_check_kernel, dim3((m + 63) / 64), dim3(64), 0, st, mom_sc, mom_m2, G, m, p_weight, kin, cand_sc, k, score, eps_d,
                     flags, count, order_flags, eps_floor, noise);
}

void launch_order_check(hipStream_t st, const double* mom_sc, const double* mom_m2, int Gmom, const int32_t* cand_idx, const double* p5_all,
                        int G, int m, int kin, int k, const int32_t* idx_sel, double p_weight, double eps_floor, double noise, int32_t* flags,
                        const double* cand_sc, const double* score_sel) {
  if (m <= 0) return;
  hipLaunchKernelGGL(order_check_kernel, dim3(m), dim3(64), 0, st, mom_sc, mom_m2, Gmom, cand_idx, p5_all, G, m, kin, k, idx_sel, p_weight,
                     eps_floor, noise, flags, cand_sc, score_sel, 1e-6);
}


void launch_rerank_finish(hipStream_t st, const int32_t* cand_idx, const double* p5_all, int G, int m, int kin, int k, int32_t* idx,
                          double* score) {
  if (m <= 0) return;
  hipLaunchKernelGGL(rerank_finish_kernel, dim3((m + 63) / 64), dim3(64), 0, st, cand_idx, p5_all, G, m, kin, k, idx, score);
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
