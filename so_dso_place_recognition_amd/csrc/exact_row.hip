// exact_row.hip — the queries a top-k call cannot answer from its all-pairs pass, answered in fp64 THROUGHOUT (run_test.m:38-57 are
// doubles): their distances to EVERY entry of the shard, the row statistics of normalize(.,2), the fused scores, the mask and the k
// smallest - the reference's own computation for that query, stream-ordered (no host round trip) and hipGraph-capturable.
//
// Two reasons flag a query (rerank.hip: order_check_kernel / sort_wave_body, one word per query):
//   bit 0  the ORDER of its re-evaluated candidates hangs on the all-pairs pass's row sigmas (two neighbours whose channels disagree);
//   bit 1  CONTAINMENT: the k_in best of the pass do not provably hold the exact top-k (more than k_in entries whose fused scores agree
//          to the pass's resolution - near-copies of one place: a vehicle that stood still, then came back).
// The flagged queries form an ascending list (built inside the kernels for calls of up to RESOLVE_SMALL_M queries - every workgroup
// scans the flags -, by flag_compact_kernel otherwise); a pass serves RESOLVE_SLOTS list slots from `offset`:
//   xrow_kernel     the distances of the slot's query to every entry of THIS shard, channel by channel, into rows [slot][4][n_local]
//                   (SC structure, SC intensity, M2DP count, M2DP intensity), and their shifted sums (count, sum (d - K), sum (d - K)^2)
//                   about K = the pass's mean of the whole row (the same number on every shard) per workgroup in a FIXED entry order;
//                   the last workgroup to finish (a self-resetting ticket) adds the partials in workgroup order -> exact [m][4][3] =
//                   this shard's (count, mean, M2): deterministic.  With nothing flagged every workgroup leaves at once.
//   xrow_select     (after the all-gather of `exact` in a sharded call) one workgroup per slot: the row statistics of ALL shards (Chan
//                   combination in rank order), score_j = sum_c w_c (d_c(j) - mean_c) / sigma_c in rerank_kernel's operation order,
//                   +Inf under the mask (global indices), NaN never selected, and the k smallest by (score, index) in k sweeps ->
//                   sel [slot][2][k] (scores | global indices as doubles), or straight into idx / score of a single-shard call.
//   xrow_merge      (sharded, after the all-gather of `sel`) one wave per slot: G-way merge of the shards' lists by (score, index) ->
//                   idx / score of the query.  Identical inputs on every rank: identical results.
// NaN distances (zero-norm signatures, processSC.m:16,19) stay out of the statistics and are never selected (MATLAB's normalize / min).
#include "kernels.hpp"
#include "rerank_common.hpp"

namespace pr {
namespace {

__global__ __launch_bounds__(256) void flag_compact_kernel(const int32_t* __restrict__ flags, int m, int32_t* __restrict__ list /* [m] */,
                                                            int32_t* __restrict__ cnt /* [1] */) {
  __shared__ int wsum[4], base;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int q0 = 0; q0 < m; q0 += 256) {
    const int q = q0 + tid;
    const int f = (q < m && flags[q] != 0) ? 1 : 0;
    const unsigned long long b = __ballot(f);
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(b);
    __syncthreads();
    int off = base;
    for (int u = 0; u < w; u++) off += wsum[u];
    if (f) list[off + before] = q;
    __syncthreads();
    if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) *cnt = base;
}

// the flagged queries of slots [offset, offset + RESOLVE_SLOTS) into s_list (LDS), their total number as return value: from the flags
// themselves (flags != null: every thread of the workgroup calls this; nt = its size, a multiple of 64) or from the compacted list
__device__ int flagged_slots(const int32_t* flags, int m, const int32_t* list, const int32_t* cnt, int offset, int* s_list, int* s_tmp /* [6] */,
                             int tid, int nt) {
  if (!flags) {
    const int total = *cnt;
    for (int s = tid; s < RESOLVE_SLOTS && offset + s < total; s += nt) s_list[s] = list[offset + s];
    __syncthreads();
    return total;
  }
  const int lane = tid & 63, w = tid >> 6, nw = nt >> 6;
  if (tid == 0) s_tmp[4] = 0;
  __syncthreads();
  for (int q0 = 0; q0 < m; q0 += nt) {
    const int q = q0 + tid;
    const int f = (q < m && flags[q] != 0) ? 1 : 0;
    const unsigned long long b = __ballot(f);
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) s_tmp[w] = __popcll(b);
    __syncthreads();
    int off = s_tmp[4];
    for (int u = 0; u < w; u++) off += s_tmp[u];
    const int slot = off + before - offset;
    if (f && slot >= 0 && slot < RESOLVE_SLOTS) s_list[slot] = q;
    __syncthreads();
    if (tid == 0) { int t = s_tmp[4]; for (int u = 0; u < nw; u++) t += s_tmp[u]; s_tmp[4] = t; }
    __syncthreads();
  }
  return s_tmp[4];
}

__device__ __forceinline__ double pivot_of(const double* mom_all, int G, int m, int q, int ch) {
  double mean, sd;
  chan_combine(mom_all, G, m, q, ch, mean, sd);
  return (mean == mean) ? mean : 0.5;
}
// Chan combination in rank order of exact_all [G][m][4][3] (chan_combine's arithmetic)
__device__ void exact_combine(const double* exact_all, int G, int m, int q, int c, double& mean, double& sd, double* loc /* [3] or null: the totals */) {
  double cn = 0.0, mu = 0.0, m2 = 0.0;
  for (int g = 0; g < G; g++) {
    const double* o = exact_all + (((size_t)g * m + q) * 4 + c) * 3;
    const double nb = o[0], mb = o[1], m2b = o[2];
    if (nb <= 0.0) continue;
    const double tot = cn + nb, delta = mb - mu;
    mu += delta * (nb / tot);
    m2 += m2b + delta * delta * (cn * nb / tot);
    cn = tot;
  }
  mean = mu;
  sd = sqrt(m2 / (cn - 1.0));
  if (loc) { loc[0] = cn; loc[1] = mu; loc[2] = m2; }
}

struct XrowArgs {
  const void* q_sc; const void* db_sc; int sc_dt;
  const void* q_m2; const void* db_m2; int m2_dt;
  const double* mom_sc; const double* mom_m2;                   // [G][m][2][3] the all-pairs pass's moments (the pivot K = their combined mean)
  int G, m, n_local;
  const int32_t* flags;                                         // [m] flags, or null: the compacted list below
  const int32_t* list; const int32_t* cnt; int offset, NB;
  double* partial;                                              // [RESOLVE_SLOTS][NB][4][3]
  double* exact;                                                // [m][4][3]
  double* rows;                                                 // [RESOLVE_SLOTS][4][n_local]
  unsigned* tick;                                               // [1] zero between launches
  int* dflags;                                                  // deferred warning bits of the context ([2] resolved, [3] more flagged than one pass)
};

// partial [slot][NB][4][3] of this pass -> exact [q][4][3]; run by the last workgroup of a rows kernel (all of its threads)
__device__ void xrow_finish(const XrowArgs& A, const int* s_list, int ns, int total, double* red, int tid) {
  if (tid == 0) {
    *A.tick = 0u;                                               // ready for the next launch
    if (A.dflags) { A.dflags[2] = 1; if (total - A.offset > RESOLVE_SLOTS) A.dflags[3] = 1; }
  }
  auto part_at = [&](size_t i) {                                 // written by other workgroups: agent-scope loads (L2), no copy of this CU's L1
    return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long*>(A.partial + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  };
  for (int s = 0; s < ns; s++) {
    const int q = s_list[s];
    if (tid < 12) {
      const int c = tid / 3, e = tid % 3;
      double sum = 0.0;
      for (int bb = 0; bb < A.NB; bb++) sum += part_at((((size_t)s * A.NB + bb) * 4 + c) * 3 + e);   // workgroup order: deterministic
      red[tid] = sum;
    }
    __syncthreads();
    if (tid < 4) {
      const int c = tid;
      const bool present = c < 2 ? A.q_sc != nullptr : A.q_m2 != nullptr;
      const double N = red[3 * c], S1 = red[3 * c + 1], S2 = red[3 * c + 2];
      const double K = present ? pivot_of(c < 2 ? A.mom_sc : A.mom_m2, A.G, A.m, q, c & 1) : 0.0;
      double* w = A.exact + ((size_t)q * 4 + c) * 3;
      w[0] = N;
      w[1] = N > 0.0 ? K + S1 / N : 0.0;
      w[2] = N > 0.0 ? S2 - S1 * S1 / N : 0.0;
    }
    __syncthreads();
  }
}

// The rows in the reference's own formulation on the vector ALUs (the device functions of rerank_kernel: ~23 ns per pair and channel pair):
// workgroup b takes entries b, b + NB, ... of every slot.
__global__ __launch_bounds__(256) void xrow_valu_kernel(XrowArgs A) {
  __shared__ double buf[60 * 21 + 1200];
  __shared__ double red[256];
  __shared__ int s_list[RESOLVE_SLOTS], s_tmp[6];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int total = flagged_slots(A.flags, A.m, A.list, A.cnt, A.offset, s_list, s_tmp, tid, 256);
  const int ns = total - A.offset < RESOLVE_SLOTS ? total - A.offset : RESOLVE_SLOTS;
  if (ns <= 0) return;                                          // nothing flagged: the usual case
  const size_t esc = A.sc_dt == 0 ? 8 : 4, em2 = A.m2_dt == 0 ? 8 : 4;
  for (int s = 0; s < ns; s++) {
    const int q = s_list[s];
    const void* qs = A.q_sc ? static_cast<const char*>(A.q_sc) + (size_t)q * 2400 * esc : nullptr;
    const void* qm = A.q_m2 ? static_cast<const char*>(A.q_m2) + (size_t)q * 4 * 384 * em2 : nullptr;
    double* row = A.rows + (size_t)s * 4 * A.n_local;
    double K[4] = {0.0, 0.0, 0.0, 0.0}, acc[4][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
    if (qs) { K[0] = pivot_of(A.mom_sc, A.G, A.m, q, 0); K[1] = pivot_of(A.mom_sc, A.G, A.m, q, 1); }
    if (qm) { K[2] = pivot_of(A.mom_m2, A.G, A.m, q, 0); K[3] = pivot_of(A.mom_m2, A.G, A.m, q, 1); }
    for (int j = b; j < A.n_local; j += A.NB) {
      for (int c = 0; c < 4; c++) {
        if (c < 2 ? !qs : !qm) continue;
        const double d = c < 2 ? sc_pair_exact(qs, A.sc_dt, (size_t)c * 1200, A.db_sc, A.sc_dt, (size_t)j * 2400 + c * 1200, buf, red, tid)
                               : m2dp_pair_exact(qm, A.m2_dt, 0, A.db_m2, A.m2_dt, (size_t)j * 4 * 384, c - 2, red, tid);
        if (tid == 0) row[(size_t)c * A.n_local + j] = d;       // (every thread holds the block-wide value)
        if (d == d) { const double x = d - K[c]; acc[c][0] += 1.0; acc[c][1] += x; acc[c][2] += x * x; }
      }
    }
    if (tid < 12) {
      const int c = tid / 3, e = tid % 3;
      A.partial[(((size_t)s * A.NB + b) * 4 + c) * 3 + e] = acc[c][e];
    }
  }
  // the last workgroup to arrive finishes: partials -> exact moments
  __threadfence();
  __syncthreads();
  if (tid == 0) s_tmp[5] = (atomicAdd(A.tick, 1u) == (unsigned)gridDim.x - 1u) ? 1 : 0;
  __syncthreads();
  if (!s_tmp[5]) return;
  __threadfence();
  xrow_finish(A, s_list, ns, total, red, tid);
}

struct SelArgs {
  const int32_t* flags; const int32_t* list; const int32_t* cnt; int offset;
  const double* exact_all; int G, m, n_local;                   // [G][m][4][3] the shards' exact moments (G = 1: this context's own)
  int q_row0, db_row0, mask_width;
  double p_weight; int has_sc, has_m2, k;
  const double* rows;                                           // [RESOLVE_SLOTS][4][n_local]
  double* sel;                                                  // [RESOLVE_SLOTS][2][k] (sharded) or null
  int32_t* idx; double* score;                                  // [m][k], written when sel is null (single shard)
  double* mom_sc; double* mom_m2;                               // single shard, or null: the caller's moments rows [m][2][3] become the exact ones
};

__device__ __forceinline__ bool cand_less(double av, int aj, double bv, int bj) { return av < bv || (av == bv && aj < bj); }

__global__ __launch_bounds__(256) void xrow_select_kernel(SelArgs A) {
  __shared__ int s_list[RESOLVE_SLOTS], s_tmp[6];
  __shared__ double rv[256];
  __shared__ int rj[256];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int total = flagged_slots(A.flags, A.m, A.list, A.cnt, A.offset, s_list, s_tmp, tid, 256);
  if (A.offset + s >= total) return;
  const int q = s_list[s], n = A.n_local, k = A.k;
  double w[4] = {A.has_sc ? A.p_weight : 0.0, A.has_sc ? 1.0 : 0.0, A.has_m2 ? A.p_weight : 0.0, A.has_m2 ? 1.0 : 0.0}, mean[4], sd[4];
  for (int c = 0; c < 4; c++) {
    mean[c] = 0.0; sd[c] = 1.0;
    if (w[c] == 0.0) continue;
    double loc[3];
    exact_combine(A.exact_all, A.G, A.m, q, c, mean[c], sd[c], loc);
    double* mo = c < 2 ? A.mom_sc : A.mom_m2;
    if (mo && tid < 3) mo[((size_t)q * 2 + (c & 1)) * 3 + tid] = loc[tid];
  }
  const double* row = A.rows + (size_t)s * 4 * n;
  const int ig = A.q_row0 + q;
  auto fused = [&](int j) -> double {                          // run_test.m:40 in rerank_kernel's operation order, then the mask (:47-53)
    double f = 0.0;
#pragma unroll
    for (int c = 0; c < 4; c++)
      if (w[c] != 0.0) f += w[c] * ((row[(size_t)c * n + j] - mean[c]) / sd[c]);
    int dij = ig - (A.db_row0 + j);
    if (dij < 0) dij = -dij;
    if (dij < A.mask_width) f = __builtin_inf();
    return f;
  };
  auto emit = [&](int t, double v, int jg) {
    if (A.sel) { A.sel[((size_t)s * 2 + 0) * k + t] = jg >= 0 ? v : __builtin_nan(""); A.sel[((size_t)s * 2 + 1) * k + t] = (double)jg; }
    else { A.idx[(size_t)q * k + t] = jg; A.score[(size_t)q * k + t] = jg >= 0 ? v : __builtin_nan(""); }
  };
  // the k smallest (score, index) pairs, one sweep of the row per element (run_test.m:57: ties -> lower index, NaN never)
  double pv = -__builtin_inf();
  int pj = -1;
  for (int t = 0; t < k; t++) {
    double bv = 0.0;
    int bj = -1;
    for (int j = tid; j < n; j += 256) {
      const double f = fused(j);
      const int jg = A.db_row0 + j;
      if (f != f || !cand_less(pv, pj, f, jg)) continue;       // NaN, or selected already
      if (bj < 0 || cand_less(f, jg, bv, bj)) { bv = f; bj = jg; }
    }
    rv[tid] = bv; rj[tid] = bj;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (tid < st) {
        const int oj = rj[tid + st];
        if (oj >= 0 && (rj[tid] < 0 || cand_less(rv[tid + st], oj, rv[tid], rj[tid]))) { rv[tid] = rv[tid + st]; rj[tid] = oj; }
      }
      __syncthreads();
    }
    pv = rv[0];
    pj = rj[0];
    __syncthreads();
    if (tid == 0) emit(t, pv, pj);
    if (pj < 0) {                                               // fewer than k candidates: -1 / NaN fill the rest
      if (tid == 0) for (int u = t + 1; u < k; u++) emit(u, 0.0, -1);
      break;
    }
  }
}

// sel_all [G][RESOLVE_SLOTS][2][k] -> idx / score [m][k] of the slot's query: every shard's list is ascending by (score, index) with its
// missing entries (-1 / NaN) last; lane g walks list g
__global__ __launch_bounds__(64) void xrow_merge_kernel(const int32_t* __restrict__ flags, const int32_t* __restrict__ list, const int32_t* __restrict__ cnt,
                                                         int offset, const double* __restrict__ sel_all, int G, int m, int k,
                                                         int32_t* __restrict__ idx, double* __restrict__ score) {
  __shared__ int s_list[RESOLVE_SLOTS], s_tmp[6];
  const int s = blockIdx.x, lane = threadIdx.x;
  const int total = flagged_slots(flags, m, list, cnt, offset, s_list, s_tmp, lane, 64);
  if (offset + s >= total) return;
  const int q = s_list[s];
  int cur = 0;
  for (int t = 0; t < k; t++) {
    double v = __builtin_nan("");
    int j = -1;
    if (lane < G && cur < k) {
      const double* L = sel_all + (((size_t)lane * RESOLVE_SLOTS + s) * 2) * k;
      v = L[cur];
      j = (int)L[k + cur];
    }
    double bv = v;
    int bj = j, bl = lane;
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
      const double ov = __shfl_xor(bv, sft, 64);
      const int oj = __shfl_xor(bj, sft, 64), ol = __shfl_xor(bl, sft, 64);
      if (cand_before(ov, oj, bv, bj) || (!cand_before(bv, bj, ov, oj) && ol < bl)) { bv = ov; bj = oj; bl = ol; }
    }
    const bool ok = bj >= 0 && bv == bv;
    if (lane == 0) {
      idx[(size_t)q * k + t] = ok ? bj : -1;
      score[(size_t)q * k + t] = ok ? bv : __builtin_nan("");
    }
    if (ok && bl == lane) cur++;
  }
}

}  // namespace

int exact_partial_blocks(int n_local) { return n_local < RESOLVE_NB ? (n_local > 0 ? n_local : 1) : RESOLVE_NB; }

// flags [m] -> the flagged-query list the resolution kernels read: small calls scan the flags themselves (returns flags), larger ones get
// the compacted list (one more launch; returns null)
static const int32_t* resolve_list(hipStream_t st, const int32_t* flags, int m, int32_t* list, int32_t* cnt) {
  if (m <= RESOLVE_SMALL_M) return flags;
  hipLaunchKernelGGL(flag_compact_kernel, dim3(1), dim3(256), 0, st, flags, m, list, cnt);
  return nullptr;
}

void launch_flag_compact(hipStream_t st, const int32_t* flags, int m, int32_t* list, int32_t* cnt) {
  hipLaunchKernelGGL(flag_compact_kernel, dim3(1), dim3(256), 0, st, flags, m, list, cnt);
}

void launch_xrow(hipStream_t st, const void* q_sc, const void* db_sc, int sc_dt, const void* q_m2, const void* db_m2, int m2_dt,
                 const double* mom_sc, const double* mom_m2, int G, int m, int n_local, const int32_t* flags, int32_t* list, int32_t* cnt,
                 int offset, bool compacted, double* partial, double* exact, double* rows, unsigned* tick, int* dflags) {
  if (m <= 0 || n_local <= 0) return;
  const int32_t* fl = compacted ? nullptr : resolve_list(st, flags, m, list, cnt);
  int NB = exact_partial_blocks(n_local);
  if (m <= 64 && NB > 256) NB = 256;      // an online call pays for the launch every time and for the resolution once in 10^5 calls: a small grid
  XrowArgs A{q_sc, db_sc, sc_dt, q_m2, db_m2, m2_dt, mom_sc, mom_m2, G, m, n_local, fl, list, cnt, offset, NB, partial, exact, rows, tick, dflags};
  hipLaunchKernelGGL(xrow_valu_kernel, dim3(NB), dim3(256), 0, st, A);
}

void launch_xrow_select(hipStream_t st, const int32_t* flags, const int32_t* list, const int32_t* cnt, int offset, const double* exact_all, int G,
                        int m, int n_local, int q_row0, int db_row0, int mask_width, double p_weight, int has_sc, int has_m2, int k,
                        const double* rows, double* sel, int32_t* idx, double* score, double* out_mom_sc, double* out_mom_m2) {
  if (m <= 0 || n_local <= 0) return;
  SelArgs A{m <= RESOLVE_SMALL_M ? flags : nullptr, list, cnt, offset, exact_all, G, m, n_local, q_row0, db_row0, mask_width, p_weight, has_sc, has_m2, k,
            rows, sel, idx, score, out_mom_sc, out_mom_m2};
  hipLaunchKernelGGL(xrow_select_kernel, dim3(m < RESOLVE_SLOTS ? m : RESOLVE_SLOTS), dim3(256), 0, st, A);
}

void launch_xrow_merge(hipStream_t st, const int32_t* flags, const int32_t* list, const int32_t* cnt, int offset, const double* sel_all, int G, int m,
                       int k, int32_t* idx, double* score) {
  if (m <= 0) return;
  hipLaunchKernelGGL(xrow_merge_kernel, dim3(m < RESOLVE_SLOTS ? m : RESOLVE_SLOTS), dim3(64), 0, st, m <= RESOLVE_SMALL_M ? flags : nullptr, list, cnt,
                     offset, sel_all, G, m, k, idx, score);
}

}  // namespace pr
