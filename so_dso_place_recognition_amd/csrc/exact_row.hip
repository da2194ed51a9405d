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
  int* dflags;                                                  // deferred warning bits of the context ([2] resolved, [3] flagged queries left unresolved)
  int last_pass;                                                // the caller runs no pass behind this one: what it leaves is reported in dflags[3]
};

// partial [slot][NB][4][3] of this pass -> exact [q][4][3]; run by the last workgroup of a rows kernel (all of its threads)
__device__ void xrow_finish(const XrowArgs& A, const int* s_list, int ns, int total, double* red, int tid) {
  if (tid == 0) {
    *A.tick = 0u;                                               // ready for the next launch
    // ([3] is only ever SET here - an earlier call's bit stays until pr_take_warnings reads it - and only by a pass its caller declares its last)
    if (A.dflags) { A.dflags[2] = 1; if (A.last_pass && total - A.offset > RESOLVE_SLOTS) A.dflags[3] = 1; }
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

// ---- the same rows through the sector spectra (the factorisation of sc_match*.hip, here in fp64 on the vector ALUs: gfx950's
// v_mfma_f64_16x16x4 issues every ~156 cycles = 32 TFLOP/s, v_fma_f64 does 78 - tools/ubench/mfma_f64_rate.hip).  With X_r(f) the DFT over the
// 60 sectors of ring r (f = 0 .. 30) and Q conj(D) = (ac + bd) + i (bc - ad) =: P + i R, Q D = (ac - bd) + i (ad + bc) =: P' + i R' summed
// over the 20 rings, the 120 variant products of processSC.m:24-30 are
//     forward  k: sum_f w_f/60 (P cos(th_f k) - R sin(th_f k)),    mirrored k: the same with (P', R'),    th_f = 2 pi f / 60, w_0 = w_30 = 1, else 2
// and the values at 60 - k come with the sine sums negated: E = sum w P cos, O = sum w R sin for k = 0 .. 30 give E - O and E + O.
// 6.3 k multiply-adds per pair and channel instead of 144 k, + 38 k per DB entry and channel for its spectra (once per pass: a workgroup
// takes a tile of XE entries, transforms it into LDS and walks the pass's slots over it).  Distances agree with the direct form to ~1e-15.
constexpr int XE = 8;                         // entries per tile
constexpr int XQ = 20 * 62;                   // doubles of one (slot, channel) query spectrum: [ring][f]{a, b}, scaled by w_f / (60 |q|)

// query spectra of the pass's slots: qspec [slot][ch][ring][f]{re, im} x w_f / (60 |q_ch|)   (grid: slots x 2)
__global__ __launch_bounds__(256) void xrow_qspec_kernel(XrowArgs A, const double* __restrict__ tw, double* __restrict__ qspec) {
  __shared__ double x[20][61];
  __shared__ double red[256];
  __shared__ int s_list[RESOLVE_SLOTS], s_tmp[6];
  const int s = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
  const int total = flagged_slots(A.flags, A.m, A.list, A.cnt, A.offset, s_list, s_tmp, tid, 256);
  if (A.offset + s >= total) return;
  const int q = s_list[s];
  double ssq = 0.0;
  for (int i = tid; i < 1200; i += 256) {                       // bin = sector * 20 + ring (SC.cpp:39)
    const double v = ld(A.q_sc, A.sc_dt, (size_t)q * 2400 + ch * 1200 + i);
    x[i % 20][i / 20] = v;
    ssq += v * v;
  }
  const double nq = sqrt(block_sum256(ssq, red, tid));         // (its barriers also publish x)
  for (int it = tid; it < 20 * 31; it += 256) {
    const int r = it / 31, f = it % 31;
    double re = 0.0, im = 0.0;
    int t = 0;
    for (int sct = 0; sct < 60; sct++) {
      re += x[r][sct] * tw[t];
      im -= x[r][sct] * tw[60 + t];
      t += f; if (t >= 60) t -= 60;
    }
    const double sc = ((f == 0 || f == 30) ? 1.0 : 2.0) / (60.0 * nq);      // |q| = 0: inf -> NaN spectra -> NaN distances (processSC.m:16)
    double* o = qspec + (((size_t)s * 2 + ch) * 20 + r) * 62 + 2 * f;
    o[0] = re * sc; o[1] = im * sc;
  }
}

// One ring row of a DB entry -> its 31 sector frequencies.  Sectors s and 60 - s share their cosine and have opposite sines (even / odd
// parts xe, xo); s and 30 - s then share both up to (-1)^f: even and odd frequencies see different folded inputs, 14 cosine and 14 sine terms each
// (868 multiply-adds per row instead of 3600).  The row sits in this lane's registers, the twiddles are wave-uniform (scalar loads at constant
// offsets: the loops are fully unrolled).  twfs: [f = 0 .. 30][s = 1 .. 14]{cos, sin}(2 pi f s / 60).
__constant__ double g_twfs[31 * 14 * 2];                       // set per device by xrow_set_twiddles (pr_create): constant address space = scalar loads
__device__ __forceinline__ void xrow_dft_row(const double (&x)[60], double* __restrict__ out /* [31][2] */, double& ssq) {
  int tz = 0;
  asm volatile("" : "+s"(tz));                                  // (an opaque zero in the index keeps the 868 twiddle loads inside the call: hoisted out of the tile loop they spill ~3000 SGPRs)
  const double* twfs = g_twfs + tz;
  ssq = 0.0;
#pragma unroll
  for (int sct = 0; sct < 60; sct++) ssq += x[sct] * x[sct];
  double ce[15], co[15], se[15], so[15];                       // folded inputs of the even / odd frequencies, s = 1 .. 14 (index 0 unused)
#pragma unroll
  for (int sct = 1; sct < 15; sct++) {
    const double xe1 = x[sct] + x[60 - sct], xe2 = x[30 - sct] + x[30 + sct];
    const double xo1 = x[sct] - x[60 - sct], xo2 = x[30 - sct] - x[30 + sct];
    ce[sct] = xe1 + xe2; co[sct] = xe1 - xe2;                    // cos(th f (30 - s)) = (-1)^f cos(th f s)
    se[sct] = xo1 - xo2; so[sct] = xo1 + xo2;                    // sin(th f (30 - s)) = -(-1)^f sin(th f s)
  }
  const double xe15 = x[15] + x[45], xo15 = x[15] - x[45];
  const double r0e = x[0] + x[30], r0o = x[0] - x[30];
#pragma unroll
  for (int f = 0; f < 31; f++) {
    // s = 15: cos(pi f / 2), sin(pi f / 2) in {0, +-1}
    double re = (f & 1) ? r0o : r0e + ((f & 2) ? -xe15 : xe15);
    double im = (f & 1) ? ((f & 2) ? xo15 : -xo15) : 0.0;        // Im = - sum xo sin: f = 1 (mod 4): -xo15, f = 3 (mod 4): +xo15
#pragma unroll
    for (int sct = 1; sct < 15; sct++) {
      const double c = twfs[(f * 14 + sct - 1) * 2], sn = twfs[(f * 14 + sct - 1) * 2 + 1];
      re += ((f & 1) ? co[sct] : ce[sct]) * c;
      im -= ((f & 1) ? so[sct] : se[sct]) * sn;
    }
    out[2 * f] = re; out[2 * f + 1] = im;
  }
}

// rows + moment partials of the SC channels; `finish`: this launch is the pass's last rows kernel (its last workgroup turns the partials
// of all four channels into the exact moments).  Per (tile, channel): 160 lanes transform the tile's rows into LDS; then the pass's slots in
// groups of XG = 4, two barriers per group:
//   stage 1  thread = (frequency, 2 entries, 2 slots): the four ring sums ac, bd, bc, ad of its four (slot, entry) pairs - every spectrum
//            value read from LDS feeds two pairs - combined into (P, R) = Q conj(D) and (P', R') = Q D and stored per COLUMN
//            (slot, entry, variant) = 4 x 8 x 2 = 64 columns of 31 x (X, Y)
//   stage 2  lane = column, wave w = shifts k = 4 w .. 4 w + 3: the lane takes its column's 62 numbers into registers ONCE and forms, per
//            shift, the even- and odd-frequency halves of sum X cos and sum Y sin - together they give the correlation at k, 60 - k, 30 - k
//            and 30 + k (cos(th f (30 - k)) = (-1)^f cos(th f k), sin likewise with the other sign) - with the twiddles wave-uniform, i.e.
//            scalar loads from constant memory: no LDS traffic for them, and a column's numbers leave LDS four times instead of 32.
// (The first version - thread = entry x shift, every lane reading its entry's 124 sums - was bound by those LDS reads: 3.3 k cycles of LDS
// return bandwidth per slot and tile against 0.8 k of fp64 FMAs.)  The next group's query spectra are fetched while stage 1 runs.
constexpr int XG = 4;                         // slots per group
constexpr int XT = XG * XE * 2 * 62;          // doubles of the column store [XG][XE][variant][31]{X, Y}
constexpr size_t XROW_SC_LDS = ((size_t)XE * XQ + XG * XQ + XT + 4 * 64 + RESOLVE_SLOTS * XE + XE + RESOLVE_SLOTS * 2 * 3 + RESOLVE_SLOTS * 2) * 8;   // 160 KB less 2.8 KB
__constant__ double g_tw2[16 * 31 * 2];       // [k = 0 .. 15][f]{cos, sin}(2 pi f k / 60), set with g_twfs
template <typename T>
__global__ __launch_bounds__(256) void xrow_sc_kernel(XrowArgs A, const double* __restrict__ qspec, int finish) {
  extern __shared__ __attribute__((aligned(16))) double xl[];
  double* spec = xl;                                            // [XE][20][31][2]
  double* qs = spec + (size_t)XE * XQ;                          // [XG][20][31][2]
  double* tcol = qs + XG * XQ;                                  // [XG][XE][2][31][2]
  double* mpart = tcol + XT;                                    // [4 waves][64 columns]
  double* dtile = mpart + 4 * 64;                               // [RESOLVE_SLOTS][XE] this (tile, channel)'s distances
  double* part = tcol;                                          // [160] the rows' square sums (before the group loop: the column store is idle)
  double* rnd = dtile + RESOLVE_SLOTS * XE;                     // [XE] 1 / |d|
  double* acc = rnd + XE;                                       // [RESOLVE_SLOTS][2][3]
  double* piv = acc + RESOLVE_SLOTS * 2 * 3;                    // [RESOLVE_SLOTS][2]
  __shared__ int s_list[RESOLVE_SLOTS], s_tmp[6];
  const int tid = threadIdx.x, b = blockIdx.x, n = A.n_local;
  const int total = flagged_slots(A.flags, A.m, A.list, A.cnt, A.offset, s_list, s_tmp, tid, 256);
  const int ns = total - A.offset < RESOLVE_SLOTS ? total - A.offset : RESOLVE_SLOTS;
  if (ns <= 0) return;                                          // nothing flagged: the usual case
  for (int i = tid; i < ns * 6; i += 256) acc[i] = 0.0;
  for (int i = tid; i < ns * 2; i += 256) piv[i] = pivot_of(A.mom_sc, A.G, A.m, s_list[i >> 1], i & 1);
  // stage 1: frequency f1 (lane 31 of a group of 32 repeats frequency 30 and stores nothing), entries 2 ep, 2 ep + 1, slots 2 sp, 2 sp + 1 of the group
  const int f1 = (tid & 31) == 31 ? 30 : (tid & 31), ep = (tid >> 5) & 3, sp = tid >> 7;
  const bool live = (tid & 31) != 31;
  // stage 2: column = lane (slot of the group, entry, variant), shifts 4 wv .. 4 wv + 3
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ngroups = (ns + XG - 1) / XG;
  const int ntile = (n + XE - 1) / XE;
  // (Holding the NEXT (tile, channel)'s ring row in registers while the current one is worked on - 60 loads under a whole unit of
  //  arithmetic - was built and measured: with fp64 signatures the 120 extra registers spill 161 VGPRs to scratch and the pass takes 3.2
  //  instead of 1.3 ms for one slot.  The row is loaded where it is used.)
  for (int tile = b; tile < ntile; tile += A.NB) {
    const int j0 = tile * XE, ne = n - j0 < XE ? n - j0 : XE;
    for (int ch = 0; ch < 2; ch++) {
      double qn[XG * XQ / 256 + 1];                              // the next group's query spectra on their way to LDS (4960 doubles / 256 threads)
      auto q_fetch = [&](int g) {
#pragma unroll
        for (int u = 0; u < XG * XQ / 256 + 1; u++) {
          const int i = tid + 256 * u;
          int sl = g * XG + i / XQ;                              // (a ragged last group repeats the pass's last slot: its results are not stored)
          if (sl >= ns) sl = ns - 1;
          qn[u] = i < XG * XQ ? qspec[((size_t)sl * 2 + ch) * XQ + i % XQ] : 0.0;
        }
      };
      auto q_store = [&]() {
#pragma unroll
        for (int u = 0; u < XG * XQ / 256 + 1; u++) { const int i = tid + 256 * u; if (i < XG * XQ) qs[i] = qn[u]; }
      };
      q_fetch(0);
      __syncthreads();                                          // the previous (tile, channel)'s readers of spec / rnd / dtile / qs are done
      q_store();
      if (tid < 160) {
        const int e = tid / 20;
        double ssq = 0.0;
        double* out = spec + (size_t)tid * 62;
        if (e < ne) {
          const T* px = static_cast<const T*>(A.db_sc) + (size_t)(j0 + e) * 2400 + ch * 1200 + tid % 20;
          double x[60];
#pragma unroll
          for (int sct = 0; sct < 60; sct++) x[sct] = (double)px[sct * 20];
          xrow_dft_row(x, out, ssq);
        } else {
          for (int i = 0; i < 62; i++) out[i] = 0.0;
        }
        part[tid] = ssq;
      }
      __syncthreads();
      if (tid < XE) {
        double sum = 0.0;
        for (int r = 0; r < 20; r++) sum += part[tid * 20 + r];
        rnd[tid] = 1.0 / sqrt(sum);                              // |d| = 0: inf, and inf x 0 = NaN below (processSC.m:19)
      }
      // (rnd is read behind the barriers of the group loop)
      for (int g = 0; g < ngroups; g++) {
        if (g + 1 < ngroups) q_fetch(g + 1);
        {                                                        // stage 1
          double ac[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, bd[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, bc[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, ad[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
          const double* q0 = qs + (size_t)(2 * sp) * XQ + 2 * f1;
          const double* d0 = spec + (size_t)(2 * ep) * XQ + 2 * f1;
#pragma unroll 4
          for (int r = 0; r < 20; r++) {
            double qa[2], qb[2], dc[2], dd[2];
#pragma unroll
            for (int u = 0; u < 2; u++) { qa[u] = q0[u * XQ + r * 62]; qb[u] = q0[u * XQ + r * 62 + 1]; dc[u] = d0[u * XQ + r * 62]; dd[u] = d0[u * XQ + r * 62 + 1]; }
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
              for (int v = 0; v < 2; v++) { ac[u][v] += qa[u] * dc[v]; bd[u][v] += qb[u] * dd[v]; bc[u][v] += qb[u] * dc[v]; ad[u][v] += qa[u] * dd[v]; }
          }
          if (live) {
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
              for (int v = 0; v < 2; v++) {
                double* o = tcol + ((size_t)((2 * sp + u) * XE + 2 * ep + v) * 2) * 62 + 2 * f1;
                o[0] = ac[u][v] + bd[u][v]; o[1] = bc[u][v] - ad[u][v];            // forward:  (P, R)   = Q conj(D)
                o[62] = ac[u][v] - bd[u][v]; o[63] = ad[u][v] + bc[u][v];          // mirrored: (P', R') = Q D
              }
          }
        }
        __syncthreads();
        if (g + 1 < ngroups) q_store();                          // (every reader of this group's spectra is past the barrier)
        {                                                        // stage 2
          double X[31], Y[31];
          const double* cp = tcol + (size_t)lane * 62;
#pragma unroll
          for (int f = 0; f < 31; f++) { X[f] = cp[2 * f]; Y[f] = cp[2 * f + 1]; }
          int tz = wv * 4 * 62;
          asm volatile("" : "+s"(tz));                           // (keeps the twiddle loads inside the group loop)
          const double* t2 = g_tw2 + tz;
          double mm = -__builtin_inf();
          bool isnan_ = false;
#pragma unroll
          for (int kk = 0; kk < 4; kk++) {
            double pe = 0.0, po = 0.0, re = 0.0, ro = 0.0;
#pragma unroll
            for (int f = 0; f < 31; f++) {
              const double c = t2[(kk * 31 + f) * 2], sn = t2[(kk * 31 + f) * 2 + 1];
              if (f & 1) { po += X[f] * c; ro += Y[f] * sn; } else { pe += X[f] * c; re += Y[f] * sn; }
            }
            const double Ek = pe + po, E30 = pe - po, Ok = re + ro, O30 = ro - re;
            const double c4 = fmax(fmax(Ek - Ok, Ek + Ok), fmax(E30 - O30, E30 + O30));
            isnan_ = isnan_ || (c4 != c4);
            mm = fmax(mm, c4);
          }
          mpart[wv * 64 + lane] = isnan_ ? __builtin_nan("") : mm;
        }
        __syncthreads();
        if (tid < XG * XE) {                                     // (slot of the group, entry): the maximum over both variants and the four waves' shifts
          const int sg = tid / XE, e = tid % XE, sl = g * XG + sg;
          double mm = -__builtin_inf();
          bool bad_ = false;
#pragma unroll
          for (int v = 0; v < 2; v++)
#pragma unroll
            for (int w = 0; w < 4; w++) { const double x = mpart[w * 64 + tid * 2 + v]; bad_ = bad_ || (x != x); mm = fmax(mm, x); }
          if (sl < ns) dtile[sl * XE + e] = bad_ ? __builtin_nan("") : (1.0 - mm * rnd[e]) / 2.0;   // processSC.m:30-31 (a zero-norm row on either side: NaN)
        }
      }
      __syncthreads();
      for (int i = tid; i < ns * XE; i += 256) {
        const int s = i / XE, e = i % XE;
        if (e < ne) A.rows[((size_t)s * 4 + ch) * n + j0 + e] = dtile[i];
      }
      if (tid < ns) {                                            // slot tid's moments: the tile's entries in order
        const double K = piv[tid * 2 + ch];
        double* a3 = acc + (tid * 2 + ch) * 3;
        double c0 = a3[0], c1 = a3[1], c2 = a3[2];
        for (int e = 0; e < ne; e++) {
          const double d = dtile[tid * XE + e];
          if (d == d) { const double xd = d - K; c0 += 1.0; c1 += xd; c2 += xd * xd; }
        }
        a3[0] = c0; a3[1] = c1; a3[2] = c2;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < ns * 6; i += 256) {
    const int s = i / 6, c = (i % 6) / 3, e = i % 3;
    A.partial[(((size_t)s * A.NB + b) * 4 + c) * 3 + e] = acc[i];
  }
  if (!A.q_m2) for (int i = tid; i < ns * 6; i += 256) A.partial[(((size_t)(i / 6) * A.NB + b) * 4 + 2 + (i % 6) / 3) * 3 + i % 3] = 0.0;
  if (!finish) return;
  __threadfence();
  __syncthreads();
  if (tid == 0) s_tmp[5] = (atomicAdd(A.tick, 1u) == (unsigned)gridDim.x - 1u) ? 1 : 0;
  __syncthreads();
  if (!s_tmp[5]) return;
  __threadfence();
  xrow_finish(A, s_list, ns, total, tcol, tid);
}

// processM2DP.m:12-22 for the pass's slots over tiles of XE entries: the 4 x 4 row products of a (slot, entry) pair per channel,
// thread = (entry, 2 query rows, 2 entry rows, an eighth of the 192 columns)
constexpr int XMR = 196;                      // row stride of the LDS copies in doubles: 2 XMR = 8 and 4 XMR = 16 (mod 32 doubles = the 64 banks), so the four
                                              // (entry, row pair) combinations of a wave and its eight column lanes read 32 different bank pairs (a stride of
                                              // 192 with contiguous column parts put them on four: 8-way conflicts, 7.4 k cycles per slot and tile)
constexpr size_t XROW_M2_LDS = ((size_t)XE * 4 * XMR + 2 /* XGM */ * 4 * XMR + RESOLVE_SLOTS * XE + RESOLVE_SLOTS * 2 * 3 + RESOLVE_SLOTS * 2) * 8;
constexpr int XGM = 2;                        // slots per group here: 71 KB of LDS, two workgroups (two waves per SIMD) per CU - the kernel is latency-bound
template <typename T>
__global__ __launch_bounds__(256, 2) void xrow_m2dp_kernel(XrowArgs A, int sc_too) {
  extern __shared__ __attribute__((aligned(16))) double xl[];
  double* dr = xl;                                              // [XE][4][XMR] the tile's rows of this channel
  double* qr = dr + (size_t)XE * 4 * XMR;                       // [XGM][4][XMR] the query rows of a group of slots
  double* dtile = qr + XGM * 4 * XMR;                             // [RESOLVE_SLOTS][XE] this (tile, channel)'s distances
  double* acc = dtile + RESOLVE_SLOTS * XE;                     // [RESOLVE_SLOTS][2][3]
  double* piv = acc + RESOLVE_SLOTS * 2 * 3;                    // [RESOLVE_SLOTS][2]
  __shared__ double red[256];
  __shared__ int s_list[RESOLVE_SLOTS], s_tmp[6];
  const int tid = threadIdx.x, b = blockIdx.x, n = A.n_local;
  const int total = flagged_slots(A.flags, A.m, A.list, A.cnt, A.offset, s_list, s_tmp, tid, 256);
  const int ns = total - A.offset < RESOLVE_SLOTS ? total - A.offset : RESOLVE_SLOTS;
  if (ns <= 0) return;
  for (int i = tid; i < ns * 6; i += 256) acc[i] = 0.0;
  for (int i = tid; i < ns * 2; i += 256) piv[i] = pivot_of(A.mom_m2, A.G, A.m, s_list[i >> 1], i & 1);
  const int cp = tid & 7, bb = (tid >> 3) & 1, ab = (tid >> 4) & 1, e = tid >> 5;
  const int ntile = (n + XE - 1) / XE;
  for (int tile = b; tile < ntile; tile += A.NB) {
    const int j0 = tile * XE, ne = n - j0 < XE ? n - j0 : XE;
    for (int ch = 0; ch < 2; ch++) {
      const T* qsig = static_cast<const T*>(A.q_m2);
      const T* dsig = static_cast<const T*>(A.db_m2);
      T qn[3 * XGM];                                             // a group's query rows on their way to LDS: XGM slots x 768 values / 256 threads
      auto q_fetch = [&](int g) {                                // (one slot per barrier left every iteration waiting for its own ~2 k cycles of load latency)
#pragma unroll
        for (int u = 0; u < 3 * XGM; u++) {
          const int i = tid + 256 * u;
          int sl = g * XGM + i / 768;
          if (sl >= ns) sl = ns - 1;                             // (a ragged last group repeats the pass's last slot: its results are not stored)
          qn[u] = qsig[((size_t)s_list[sl] * 4 + (i % 768) / 192) * 384 + ch * 192 + i % 192];
        }
      };
      auto q_store = [&]() {
#pragma unroll
        for (int u = 0; u < 3 * XGM; u++) { const int i = tid + 256 * u; qr[((i / 768) * 4 + (i % 768) / 192) * XMR + i % 192] = (double)qn[u]; }
      };
      q_fetch(0);
      __syncthreads();                                          // the previous (tile, channel)'s readers of dr / qr / dtile are done
      q_store();
      {                                                         // the tile: 24 values per thread, all requested before the first is stored
        T tv[24];
#pragma unroll
        for (int u = 0; u < 24; u++) {
          const int i = tid + 256 * u, ee = i / (4 * 192), rr = (i / 192) & 3, c = i % 192;
          tv[u] = ee < ne ? dsig[((size_t)(j0 + ee) * 4 + rr) * 384 + ch * 192 + c] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < 24; u++) {
          const int i = tid + 256 * u, ee = i / (4 * 192), rr = (i / 192) & 3, c = i % 192;
          dr[(ee * 4 + rr) * XMR + c] = (double)tv[u];
        }
      }
      __syncthreads();
      const int ngroups = (ns + XGM - 1) / XGM;
      for (int g = 0; g < ngroups; g++) {
        if (g + 1 < ngroups) q_fetch(g + 1);
#pragma unroll
        for (int sg = 0; sg < XGM; sg++) {
          double d00 = 0.0, d01 = 0.0, d10 = 0.0, d11 = 0.0;
          const double* q0 = qr + (size_t)(sg * 4 + 2 * ab) * XMR + cp;     // columns cp, cp + 8, ...: the eight column lanes read 64 contiguous bytes
          const double* d0 = dr + ((size_t)e * 4 + 2 * bb) * XMR + cp;
#pragma unroll
          for (int c = 0; c < 192; c += 8) {
            const double qa = q0[c], qb = q0[XMR + c], da = d0[c], db = d0[XMR + c];
            d00 += qa * da; d01 += qa * db; d10 += qb * da; d11 += qb * db;
          }
#pragma unroll
          for (int sft = 1; sft < 8; sft <<= 1) {                // the eight column parts: an xor butterfly (all lanes end with the same sums)
            d00 += __shfl_xor(d00, sft, 64); d01 += __shfl_xor(d01, sft, 64); d10 += __shfl_xor(d10, sft, 64); d11 += __shfl_xor(d11, sft, 64);
          }
          double mn = nanmin(nanmin((1.0 - d00) / 2.0, (1.0 - d01) / 2.0), nanmin((1.0 - d10) / 2.0, (1.0 - d11) / 2.0));   // processM2DP.m:15,19
          mn = nanmin(mn, __shfl_xor(mn, 8, 64));
          mn = nanmin(mn, __shfl_xor(mn, 16, 64));
          const int sl = g * XGM + sg;
          if ((tid & 31) == 0 && sl < ns) dtile[sl * XE + e] = mn;
        }
        __syncthreads();                                        // every reader of this group's rows is done
        if (g + 1 < ngroups) { q_store(); __syncthreads(); }
      }
      for (int i = tid; i < ns * XE; i += 256) {
        const int s = i / XE, ee = i % XE;
        if (ee < ne) A.rows[((size_t)s * 4 + 2 + ch) * n + j0 + ee] = dtile[i];
      }
      if (tid < ns) {                                            // slot tid's moments: the tile's entries in order
        const double K = piv[tid * 2 + ch];
        double* a3 = acc + (tid * 2 + ch) * 3;
        double c0 = a3[0], c1 = a3[1], c2 = a3[2];
        for (int ee = 0; ee < ne; ee++) {
          const double d = dtile[tid * XE + ee];
          if (d == d) { const double xd = d - K; c0 += 1.0; c1 += xd; c2 += xd * xd; }
        }
        a3[0] = c0; a3[1] = c1; a3[2] = c2;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < ns * 6; i += 256) {
    const int s = i / 6, c = (i % 6) / 3, e3 = i % 3;
    A.partial[(((size_t)s * A.NB + b) * 4 + 2 + c) * 3 + e3] = acc[i];
    if (!sc_too) A.partial[(((size_t)s * A.NB + b) * 4 + c) * 3 + e3] = 0.0;
  }
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
  double* part;                                                 // [P][RESOLVE_SLOTS][2][k] the slices' lists (xrow_merge_kernel's input layout: slice = "shard")
  int P;                                                        // slices per row: workgroup (slot, p) sweeps entries [n p / P, n (p + 1) / P)
  double* mom_sc; double* mom_m2;                               // single shard, or null: the caller's moments rows [m][2][3] become the exact ones
};

__device__ __forceinline__ bool cand_less(double av, int aj, double bv, int bj) { return av < bv || (av == bv && aj < bj); }

__global__ __launch_bounds__(256) void xrow_select_kernel(SelArgs A) {
  __shared__ int s_list[RESOLVE_SLOTS], s_tmp[6];
  __shared__ double rv[256];
  __shared__ int rj[256];
  const int s = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
  const int total = flagged_slots(A.flags, A.m, A.list, A.cnt, A.offset, s_list, s_tmp, tid, 256);
  if (A.offset + s >= total) return;
  const int q = s_list[s], n = A.n_local, k = A.k;
  const int j_lo = (int)((long long)n * sl / A.P), j_hi = (int)((long long)n * (sl + 1) / A.P);
  double w[4] = {A.has_sc ? A.p_weight : 0.0, A.has_sc ? 1.0 : 0.0, A.has_m2 ? A.p_weight : 0.0, A.has_m2 ? 1.0 : 0.0}, mean[4], sd[4];
  for (int c = 0; c < 4; c++) {
    mean[c] = 0.0; sd[c] = 1.0;
    if (w[c] == 0.0) continue;
    double loc[3];
    exact_combine(A.exact_all, A.G, A.m, q, c, mean[c], sd[c], loc);
    double* mo = c < 2 ? A.mom_sc : A.mom_m2;
    if (mo && sl == 0 && tid < 3) mo[((size_t)q * 2 + (c & 1)) * 3 + tid] = loc[tid];
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
  double* mine = A.part + (((size_t)sl * RESOLVE_SLOTS + s) * 2) * k;
  auto emit = [&](int t, double v, int jg) { mine[t] = jg >= 0 ? v : __builtin_nan(""); mine[k + t] = (double)jg; };
  // the k smallest (score, index) pairs, one sweep of the row per element (run_test.m:57: ties -> lower index, NaN never)
  double pv = -__builtin_inf();
  int pj = -1;
  for (int t = 0; t < k; t++) {
    double bv = 0.0;
    int bj = -1;
    for (int j4 = j_lo + tid; j4 < j_hi; j4 += 1024) {         // four entries per round: their loads are in flight together
      double f4[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int j = j4 + 256 * u; f4[u] = j < j_hi ? fused(j) : __builtin_nan(""); }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const double f = f4[u];
        const int jg = A.db_row0 + j4 + 256 * u;
        if (f != f || !cand_less(pv, pj, f, jg)) continue;     // NaN (or past the end), or selected already
        if (bj < 0 || cand_less(f, jg, bv, bj)) { bv = f; bj = jg; }
      }
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

// sel_all [G][RESOLVE_SLOTS][2][k] -> idx / score [m][k] of the slot's query (or the merged list sel_out [RESOLVE_SLOTS][2][k]): every
// list is ascending by (score, index) with its missing entries (-1 / NaN) last; lane g walks list g.  Two uses: the G shards' lists of a
// sharded call, and the P row slices of xrow_select_kernel
__global__ __launch_bounds__(64) void xrow_merge_kernel(const int32_t* __restrict__ flags, const int32_t* __restrict__ list, const int32_t* __restrict__ cnt,
                                                         int offset, const double* __restrict__ sel_all, int G, int m, int k,
                                                         int32_t* __restrict__ idx, double* __restrict__ score, double* __restrict__ sel_out) {
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
      if (sel_out) {                                            // (a shard's own list, for the all-gather: [RESOLVE_SLOTS][2][k])
        sel_out[((size_t)s * 2 + 0) * k + t] = ok ? bv : __builtin_nan("");
        sel_out[((size_t)s * 2 + 1) * k + t] = ok ? (double)bj : -1.0;
      } else {
        idx[(size_t)q * k + t] = ok ? bj : -1;
        score[(size_t)q * k + t] = ok ? bv : __builtin_nan("");
      }
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

size_t xrow_qspec_doubles() { return (size_t)RESOLVE_SLOTS * 2 * XQ; }

// the DB-side twiddles [f = 0 .. 30][s = 1 .. 14]{cos, sin}(2 pi f s / 60) into the current device's constant memory (pr_create)
hipError_t xrow_set_twiddles(const double* cos60, const double* sin60) {
  double t[31 * 14 * 2];
  for (int f = 0; f < 31; f++)
    for (int sct = 1; sct < 15; sct++) { const int i = (f * sct) % 60; t[(f * 14 + sct - 1) * 2] = cos60[i]; t[(f * 14 + sct - 1) * 2 + 1] = sin60[i]; }
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_twfs), t, sizeof t);
  if (e != hipSuccess) return e;
  double t2[16 * 31 * 2];
  for (int k = 0; k < 16; k++)
    for (int f = 0; f < 31; f++) { const int i = (f * k) % 60; t2[(k * 31 + f) * 2] = cos60[i]; t2[(k * 31 + f) * 2 + 1] = sin60[i]; }
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tw2), t2, sizeof t2);
}

void launch_xrow(hipStream_t st, const void* q_sc, const void* db_sc, int sc_dt, const void* q_m2, const void* db_m2, int m2_dt,
                 const double* mom_sc, const double* mom_m2, int G, int m, int n_local, const int32_t* flags, int32_t* list, int32_t* cnt,
                 int offset, bool compacted, double* partial, double* exact, double* rows, unsigned* tick, int* dflags, int last_pass,
                 double* qspec, const double* tw, int direct) {
  if (m <= 0 || n_local <= 0) return;
  if (!direct) {                           // the spectral kernels ask for more dynamic LDS than a launch gets by default: if the device refuses, the direct form
    static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(xrow_sc_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XROW_SC_LDS) == hipSuccess &&
                               hipFuncSetAttribute(reinterpret_cast<const void*>(xrow_sc_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XROW_SC_LDS) == hipSuccess;
    if (!lds_ok) direct = 1;
  }
  const int32_t* fl = compacted ? nullptr : resolve_list(st, flags, m, list, cnt);
  if (direct) {                            // PR_XROW=direct: the reference's own formulation (the cross-check of the spectral form)
    int NB = exact_partial_blocks(n_local);
    if (m <= 64 && NB > 256) NB = 256;      // an online call pays for the launch every time and for the resolution once in 10^5 calls: a small grid
    XrowArgs A{q_sc, db_sc, sc_dt, q_m2, db_m2, m2_dt, mom_sc, mom_m2, G, m, n_local, fl, list, cnt, offset, NB, partial, exact, rows, tick, dflags, last_pass};
    hipLaunchKernelGGL(xrow_valu_kernel, dim3(NB), dim3(256), 0, st, A);
    return;
  }
  const int ntile = (n_local + XE - 1) / XE;
  const int NB = ntile < 256 ? ntile : 256;                     // one workgroup per CU (its LDS), each walking tiles b, b + NB, ...
  XrowArgs A{q_sc, db_sc, sc_dt, q_m2, db_m2, m2_dt, mom_sc, mom_m2, G, m, n_local, fl, list, cnt, offset, NB, partial, exact, rows, tick, dflags, last_pass};
  const int slots = m < RESOLVE_SLOTS ? m : RESOLVE_SLOTS;
  if (q_sc) {
    hipLaunchKernelGGL(xrow_qspec_kernel, dim3(slots, 2), dim3(256), 0, st, A, tw, qspec);
    if (sc_dt == 0) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xrow_sc_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XROW_SC_LDS);
      hipLaunchKernelGGL(xrow_sc_kernel<double>, dim3(NB), dim3(256), XROW_SC_LDS, st, A, (const double*)qspec, q_m2 ? 0 : 1);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xrow_sc_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XROW_SC_LDS);
      hipLaunchKernelGGL(xrow_sc_kernel<float>, dim3(NB), dim3(256), XROW_SC_LDS, st, A, (const double*)qspec, q_m2 ? 0 : 1);
    }
  }
  if (q_m2) {                               // (alone it runs two workgroups per CU; behind the SC kernel it shares that kernel's partial-sum geometry)
    if (!q_sc) { A.NB = ntile < 512 ? ntile : 512; }
    if (m2_dt == 0) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xrow_m2dp_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XROW_M2_LDS);
      hipLaunchKernelGGL(xrow_m2dp_kernel<double>, dim3(A.NB), dim3(256), XROW_M2_LDS, st, A, q_sc ? 1 : 0);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xrow_m2dp_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)XROW_M2_LDS);
      hipLaunchKernelGGL(xrow_m2dp_kernel<float>, dim3(A.NB), dim3(256), XROW_M2_LDS, st, A, q_sc ? 1 : 0);
    }
  }
}

int xrow_select_slices(int m) { const int slots = m < RESOLVE_SLOTS ? m : RESOLVE_SLOTS; int P = 512 / slots; return P < 1 ? 1 : (P > 64 ? 64 : P); }
size_t xrow_select_part_doubles(int m, int k) { return (size_t)xrow_select_slices(m) * RESOLVE_SLOTS * 2 * (size_t)k; }

// the row slices' lists, then their merge: into sel (sharded: the shard's list for the all-gather) or, sel = null, straight into idx / score
void launch_xrow_select(hipStream_t st, const int32_t* flags, const int32_t* list, const int32_t* cnt, int offset, const double* exact_all, int G,
                        int m, int n_local, int q_row0, int db_row0, int mask_width, double p_weight, int has_sc, int has_m2, int k,
                        const double* rows, double* part, double* sel, int32_t* idx, double* score, double* out_mom_sc, double* out_mom_m2) {
  if (m <= 0 || n_local <= 0) return;
  const int slots = m < RESOLVE_SLOTS ? m : RESOLVE_SLOTS;
  int P = xrow_select_slices(m);
  while (P > 1 && n_local / P < 1024) P >>= 1;                  // (short rows: a slice of under a thousand entries is not worth a workgroup)
  const int32_t* fl = m <= RESOLVE_SMALL_M ? flags : nullptr;
  SelArgs A{fl, list, cnt, offset, exact_all, G, m, n_local, q_row0, db_row0, mask_width, p_weight, has_sc, has_m2, k, rows, part, P, out_mom_sc, out_mom_m2};
  hipLaunchKernelGGL(xrow_select_kernel, dim3(slots, P), dim3(256), 0, st, A);
  hipLaunchKernelGGL(xrow_merge_kernel, dim3(slots), dim3(64), 0, st, fl, list, cnt, offset, (const double*)part, P, m, k, idx, score, sel);
}

void launch_xrow_merge(hipStream_t st, const int32_t* flags, const int32_t* list, const int32_t* cnt, int offset, const double* sel_all, int G, int m,
                       int k, int32_t* idx, double* score) {
  if (m <= 0) return;
  hipLaunchKernelGGL(xrow_merge_kernel, dim3(m < RESOLVE_SLOTS ? m : RESOLVE_SLOTS), dim3(64), 0, st, m <= RESOLVE_SMALL_M ? flags : nullptr, list, cnt,
                     offset, sel_all, G, m, k, idx, score, (double*)nullptr);
}

}  // namespace pr
