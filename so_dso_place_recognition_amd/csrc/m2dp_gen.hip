// m2dp_gen.hip — M2DP signature (M2DP/M2DP.cpp:38-109 + the 4-variant loop of test_m2dp.cpp:44-68) on gfx950.
//
// After cloud_frames (sc_gen.hip; PCA frame + float average, once per cloud as test_m2dp.cpp:45):
//   m2dp_bin  : grid (cloud, variant, group of 16 or 4 planes).  Each thread centres + rotates its points, applies the
//               variant signs pt' = (dx x, dy y, dx dy z) (test_m2dp.cpp:52-56), and classifies the group's projections
//               xp = xProj.pt', yp = yProj.pt' (the reference: a0*b0 + (a1*b1 + a2*b2) with NO zero seed so that the
//               degenerate plane 32, xProj = yProj = +0, keeps its signed-zero behaviour):
//               si = floor((atan2(yp,xp)+pi)*16/2pi), ri = floor(sqrt(xp^2+yp^2)*8/max_rho), idx = ri*16+si,
//               dropped iff idx >= 128 (M2DP.cpp:66-68) - in fp32 where that provably gives the reference's integers
//               (fast_bins.hpp), in the reference's fp64 expressions otherwise.  LDS-resident per-plane 128-bin grids: ONE
//               64-bit LDS atomic per projection (count << 47 | fixed-point intensity) or, where that cannot be vouched
//               for, u32 count + f64 sum (see the kernel); then mean > ave ? 1 : 0 (M2DP.cpp:84-91) and both slabs go to
//               the scratch matrices [cloud][variant][{count,intensity}][64][128] (f64).
//   m2dp_svd  : grid (cloud, variant, channel): leading singular pair of the 64x128 matrix (JacobiSVD U.col(0),
//               V.col(0), M2DP.cpp:94-103) in fp64: G = A A^T (64x64) in LDS, 8 normalised squarings (G^256),
//               u = dominant column, then polished with u <- A(A^T u)/||.|| on the original matrix until the
//               update is below 4e-15; v = A^T u / sigma.  Sign: sum(u) >= 0 (N6).  Zero matrix -> (e0, e0).
// Bound: VALU (~50 fp32 instructions per projection, 256 P projections per cloud) + LDS atomics, not HBM and not MFMA.
#include <type_traits>

#include "fast_bins.hpp"
#include "kernels.hpp"

namespace pr {
namespace {

constexpr int MAT = 64 * 128;

// PPB planes per workgroup: 16 for throughput (every point is read by 16 workgroups of its cloud), 4 when there are few
// clouds - a small batch has nothing else to hide a workgroup's ~50 instructions per (point, plane) behind.
//
// What the kernel's time is made of (MI355X, 128 clouds x 50 000 points, PPB = 4; tools/gen_only.py under rocprofv3):
//   * the projections are classified in fp32 (fast_bins.hpp: proj_bins16_fast - the dots included, with a rigorous error budget in the
//     accept test), NPT = 4 points side by side without control flow, the plane constants read once per NPT projections; the ~0.03 % it does
//     not vouch for evaluate the reference's fp64 expressions behind ONE rare branch: 55 VALU instructions per projection (92 before);
//   * the LDS atomics: a 64-bit LDS atomic costs what ~25 VALU instructions cost (5.6 ms with u32 count + f64 sum, 3.6 ms with neither), so
//     the FAST mode adds ONE u64 per projection: count << 47 | intensity in fixed point.  The scale comes from the cloud's float average
//     (2^s with 2 ave P 2^s < 2^46), every point's intensity is rounded to that grid once, and the epilogue decides mean > ave
//     (M2DP.cpp:84-91) from the fixed-point sum only where it is certain: | sum - n ave | above the accumulated rounding (n/2 grid steps)
//     plus the fp64 slack of the reference's own sum and division.  Anything else - an uncertain bin (probability ~1e-9 per bin), a
//     negative / non-finite / huge intensity, a sum that does not fit 47 bits, a cloud of 2^17 points or more - reruns the workgroup in the
//     EXACT mode: u32 count + f64 sum per projection as before.  The counts are exact in both modes.
template <int PPB>
__global__ __launch_bounds__(256) void m2dp_bin_kernel(const double* __restrict__ xyz, const float* __restrict__ inten,
                                                        const int64_t* __restrict__ offs, const double* __restrict__ frames,
                                                        const float* __restrict__ ave_in,
                                                        const double* __restrict__ planes, double max_rho, int c0,
                                                        double* __restrict__ mats) {
  constexpr int NB = PPB * 128, SPARE = 32;     // + spare bins: dropped projections and padding lanes add there instead of branching
  __shared__ unsigned long long acc[NB + SPARE];   // FAST: count << 47 | fixed-point sum;  EXACT: the f64 sums (same storage)
  __shared__ unsigned int cnt[NB + SPARE];         // EXACT: counts
  __shared__ double pl[PPB][6];
  __shared__ __attribute__((aligned(16))) float plf[PPB][8];   // the same rounded to float (fast classifier), padded to two 16-byte reads
  __shared__ unsigned long long wtot[4];
  __shared__ int need_exact;
  double* isum = reinterpret_cast<double*>(acc);
  constexpr int NPG = 64 / PPB;              // plane groups per (cloud, variant)
  constexpr int NPT = 4;                     // points per thread and round
  const int tid = threadIdx.x;
  const int pg = blockIdx.x % NPG, var = (blockIdx.x / NPG) & 3, cl = blockIdx.x / (4 * NPG);
  const int c = c0 + cl;
  const int64_t o0 = offs[c];
  const int64_t P = offs[c + 1] - o0;
  for (int b = tid; b < NB + SPARE; b += 256) acc[b] = 0ull;
  if (tid < PPB * 6) {
    const int k = tid / 6, a = tid % 6;
    const double v = (a < 3) ? planes[(pg * PPB + k) * 3 + a] : planes[64 * 3 + (pg * PPB + k) * 3 + (a - 3)];
    pl[k][a] = v;
    plf[k][a] = (float)v;
  }
  if (tid == 0) need_exact = 0;
  __syncthreads();
  const double* f = frames + (size_t)c * 16;
  const double mx = f[0], my = f[1], mz = f[2];
  const double e00 = f[3], e01 = f[4], e02 = f[5], e10 = f[6], e11 = f[7], e12 = f[8], e20 = f[9], e21 = f[10], e22 = f[11];
  // variants in the order (-,-), (-,+), (+,-), (+,+)  (test_m2dp.cpp:47-48)
  const double dx = (var & 2) ? 1.0 : -1.0, dy = (var & 1) ? 1.0 : -1.0, dz = dx * dy;
  const double S_res_inv = 16 / (2.0 * M_PI), R_res_inv = 8 / max_rho;   // M2DP.cpp:32-33
  const float R_f = (float)R_res_inv;
  const double* p = xyz + 3 * o0;
  const float* it = inten + o0;
  // the float average, widened (double > float promotes the float, M2DP.cpp:88); ave_in = NULL: the caller's frame carries it (frames.hpp)
  const double ave = ave_in ? (double)ave_in[c] : f[14];
  // fixed-point grid of the FAST mode: 2^-s with (2 ave P) 2^s < 2^46
  bool fast = P < (1 << 17) && ave >= 0.0 && ave < 1e300;
  double scale = 1.0, inv_scale = 1.0;
  if (fast) {
    int e;
    (void)frexp(2.0 * ave * (double)P + 1e-300, &e);
    int sh = 46 - e;
    sh = sh > 1000 ? 1000 : sh;
    scale = ldexp(1.0, sh);
    inv_scale = ldexp(1.0, -sh);
  }
  // one pass over the cloud; FAST selects the accumulation
  auto bin_points = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    unsigned long long ltot = 0ull;
    bool bad = false;
    // (lane = consecutive point: coalesced loads.  The clouds of the reference's pipeline arrive in unordered_map iteration order, i.e.
    // shuffled, so the 64 lanes of an LDS atomic spread over the bins; a spatially sorted cloud serialises them on a few addresses.)
    // the aligned point in fp64 (pts_align.h:24-39, test_m2dp.cpp:52-56); not kept in registers across the plane loop - the rare exact
    // evaluation of a projection computes it again from memory (the same expressions, the same bits)
    auto aligned = [&](int64_t j, double (&q)[3]) {
      const double x = p[3 * j] - mx, y = p[3 * j + 1] - my, z = p[3 * j + 2] - mz;
      q[0] = dx * ((x * e00 + y * e01) + z * e02);
      q[1] = dy * ((x * e10 + y * e11) + z * e12);
      q[2] = dz * ((x * e20 + y * e21) + z * e22);
    };
    for (int64_t i0 = tid; i0 < P; i0 += 256 * NPT) {
      double iv[NPT];
      unsigned long long fx[NPT];
      PointF pt[NPT];
      bool live[NPT];
#pragma unroll
      for (int u = 0; u < NPT; u++) {
        const int64_t i = i0 + 256 * u;
        live[u] = i < P;
        const int64_t j = live[u] ? i : i0;
        double q[3];
        aligned(j, q);
        iv[u] = (double)it[j];
        pt[u] = make_pointf(q[0], q[1], q[2], R_f);
        fx[u] = 0ull;
        if constexpr (FAST) {
          const double g = rint(iv[u] * scale);
          const bool okv = g >= 0.0 && g < 0x1p46;                 // (NaN fails)
          bad |= live[u] && !okv;
          const unsigned long long gv = (live[u] && okv) ? (unsigned long long)g : 0ull;
          ltot += gv;
          fx[u] = live[u] ? (1ull << 47) + gv : 0ull;              // what a projection of this point adds (a padding lane: nothing)
        }
      }
#pragma unroll 2
      for (int k = 0; k < PPB; k++) {
        float pf[8];
        *reinterpret_cast<float4*>(pf) = *reinterpret_cast<const float4*>(&plf[k][0]);
        *reinterpret_cast<float4*>(pf + 4) = *reinterpret_cast<const float4*>(&plf[k][4]);
        // the NPT projections side by side: floor((atan2(yp, xp) + pi) * S_res_inv), floor(sqrt(xp^2 + yp^2) * R_res_inv) (M2DP.cpp:56-61)
        // from the fp32 classifier; one rare branch for whatever it does not vouch for
        int si[NPT], ri[NPT];
        bool ok[NPT];
#pragma unroll
        for (int u = 0; u < NPT; u++) ok[u] = proj_bins16_fast(pt[u], pf, R_f, si[u], ri[u]);
        bool all_ok = true;
#pragma unroll
        for (int u = 0; u < NPT; u++) all_ok &= ok[u];
        if (!all_ok) {
#pragma unroll
          for (int u = 0; u < NPT; u++)
            if (!ok[u]) {
              double q[3];
              aligned((i0 + 256 * u < P) ? i0 + 256 * u : i0, q);
              proj_bins16_exact(q[0], q[1], q[2], pl[k], S_res_inv, R_res_inv, si[u], ri[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < NPT; u++) {
          const int idx = ri[u] * 16 + si[u];
          if constexpr (FAST) {
            const int slot = ((unsigned)idx < 128u) ? k * 128 + idx : NB + (tid & (SPARE - 1));               // dropped: M2DP.cpp:66-68
            atomicAdd(&acc[slot], fx[u]);
          } else {
            const int slot = ((unsigned)idx < 128u && live[u]) ? k * 128 + idx : NB + (tid & (SPARE - 1));
            atomicAdd(&cnt[slot], 1u);
            atomicAdd(&isum[slot], iv[u]);
          }
        }
      }
    }
    if constexpr (FAST) {
      // the workgroup's total of the fixed-point values bounds every bin's sum: it has to fit the 47-bit field
      for (int d = 32; d > 0; d >>= 1) ltot += __shfl_down(ltot, d);
      if ((tid & 63) == 0) wtot[tid >> 6] = ltot;
      if (bad) need_exact = 1;
    }
  };
  double* mc = mats + (((size_t)cl * 4 + var) * 2) * MAT + (size_t)pg * PPB * 128;
  double* mi = mc + MAT;
  if (fast) {
    bin_points(std::true_type{});
    __syncthreads();
    if (tid == 0 && ((wtot[0] + wtot[1]) + (wtot[2] + wtot[3])) >= (1ull << 47)) need_exact = 1;
    // certain decisions only: | sum - n ave | against n/2 grid steps of rounding + the fp64 slack of the reference's sum and division
    bool unsure = false;
    for (int b = tid; b < NB; b += 256) {
      const unsigned long long v = acc[b];
      const double n = (double)(v >> 47), sd = (double)(v & ((1ull << 47) - 1ull)) * inv_scale;
      const double diff = sd - n * ave, margin = n * inv_scale + (n + 4.0) * 0x1p-51 * (sd + n * ave);
      unsure |= (n > 0.0) && !(fabs(diff) > margin);
    }
    if (unsure) need_exact = 1;
    __syncthreads();
    fast = need_exact == 0;
    if (fast) {
      for (int b = tid; b < NB; b += 256) {
        const unsigned long long v = acc[b];
        const double n = (double)(v >> 47), sd = (double)(v & ((1ull << 47) - 1ull)) * inv_scale;
        mc[b] = n;
        mi[b] = (n > 0.0 && sd > n * ave) ? 1.0 : 0.0;
      }
      return;
    }
    __syncthreads();
    for (int b = tid; b < NB + SPARE; b += 256) acc[b] = 0ull;
  }
  for (int b = tid; b < NB + SPARE; b += 256) cnt[b] = 0u;
  __syncthreads();
  bin_points(std::false_type{});
  __syncthreads();
  for (int b = tid; b < NB; b += 256) {
    const unsigned int n = cnt[b];
    mc[b] = (double)n;
    mi[b] = n ? ((isum[b] / (double)n) > ave ? 1.0 : 0.0) : 0.0;
  }
}

// Sum / maximum over the 256 threads: inside a wave by lane exchanges (an xor butterfly: both partners combine the same two numbers, so
// all 64 lanes hold the same bits, no barrier), the four wave results through LDS, combined in wave order by every thread: two barriers
// instead of ten.  (Round 4: the leading-pair iteration below is a chain of such reductions; with the tree in LDS it spent most of its
// time in barriers.)
__device__ __forceinline__ double block_sum256(double v, double* red, int tid) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  const double r = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return r;
}

// The leading singular pair of one 64 x 128 bin matrix (M2DP.cpp:94-103).  One LDS buffer serves A (64 x 129, padded rows), then the Gram
// matrix G (64 x 65) and its eight squarings, then A again (re-read from global memory: 64 KB out of L2) for the polish on the original
// matrix - 70 KB of LDS instead of 103, so that TWO workgroups share a CU (the Gram product keeps its 4 x 4 block per thread in registers
// until every read of A is done).  The polish step u <- A (A^T u) runs on all 256 threads (A^T u: two half-sums per column, A v: four
// quarter-sums per row, added in a fixed order) with three barriers per iteration.
__global__ __launch_bounds__(256) void m2dp_svd_kernel(const double* __restrict__ mats, int c0, double* __restrict__ out,
                                                        int* __restrict__ flags, int* __restrict__ svd_rows) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* A = sm;               // 64 x 129 (padded rows) | G: 64 x 65
  double* G = sm;
  double* u = A + 64 * 129;     // 64
  double* v = u + 64;           // 128
  double* red = v + 128;        // 256 (partial sums of the polish; [0..3] the wave results of block_sum256)
  double* bc = red + 256;       // 4: broadcast of (norm, max difference)
  const int tid = threadIdx.x;
  const int ch = blockIdx.x & 1, var = (blockIdx.x >> 1) & 3, cl = blockIdx.x >> 3;
  const double* src = mats + (((size_t)cl * 4 + var) * 2 + ch) * MAT;
  double fro = 0.0;
  for (int e = tid; e < MAT; e += 256) {
    const double a = src[e];
    A[(e >> 7) * 129 + (e & 127)] = a;
    fro += a * a;
  }
  fro = block_sum256(fro, red, tid);   // also orders the A stores
  double* o = out + ((size_t)(c0 + cl) * 4 + var) * 384 + ch * 192;
  if (!(fro > 0.0)) {                  // JacobiSVD of a zero matrix: identity factors
    for (int e = tid; e < 192; e += 256) o[e] = (e == 0 || e == 64) ? 1.0 : 0.0;
    return;
  }
  // G = A A^T / fro.  Every thread owns a 4 x 4 block of the 64 x 64 result: 8 LDS reads per 16 multiply-adds instead
  // of 32 (the kernel is LDS-bound); every entry is summed over k in ascending order.
  // rows i0 .. i0 + 3 (four values per wave: broadcasts) x rows jl, jl + 16, jl + 32, jl + 48: the 16 lanes that differ in jl read 16 CONSECUTIVE
  // rows (pitch 129 | 65 doubles = bank step 2: conflict-free); round 6 - they owned 4 jl .. 4 jl + 3 before (bank step 8: two lanes per bank pair,
  // SQ_LDS_BANK_CONFLICT 0.42 per CU cycle).  Every entry is still the same sum over k in the same order: the result is bit for bit the old one.
  const int i0 = (tid >> 4) * 4, jl = tid & 15;
  {
    double acc[4][4] = {};
    for (int k = 0; k < 128; k++) {
      double ai[4], aj[4];
#pragma unroll
      for (int a = 0; a < 4; a++) { ai[a] = A[(i0 + a) * 129 + k]; aj[a] = A[(jl + 16 * a) * 129 + k]; }
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] += ai[a] * aj[b];
    }
    __syncthreads();                   // every read of A is done: G takes its place
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) G[(i0 + a) * 65 + jl + 16 * b] = acc[a][b] / fro;
  }
  __syncthreads();
  for (int it = 0; it < 8; it++) {     // G <- G*G / ||G*G||_F   (G symmetric)
    double acc[4][4] = {};
    for (int k = 0; k < 64; k++) {
      double gi[4], gj[4];
#pragma unroll
      for (int a = 0; a < 4; a++) { gi[a] = G[(i0 + a) * 65 + k]; gj[a] = G[(jl + 16 * a) * 65 + k]; }
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] += gi[a] * gj[b];
    }
    double part = 0.0;                 // Frobenius norm (it only scales G; u is renormalised below)
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) part += acc[a][b] * acc[a][b];
    const double nf = sqrt(block_sum256(part, red, tid));   // (its barriers also order the reads of G before the writes)
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) G[(i0 + a) * 65 + jl + 16 * b] = acc[a][b] / nf;
    __syncthreads();
  }
  // start vector: G * ones (non-negative matrices keep it in the Perron cone), normalised
  if (tid < 64) {
    double s = 0.0;
    for (int k = 0; k < 64; k++) s += G[tid * 65 + k];
    u[tid] = s;
  }
  __syncthreads();
  {
    const double nn = sqrt(block_sum256(tid < 64 ? u[tid] * u[tid] : 0.0, red, tid));
    if (tid < 64) u[tid] = u[tid] / nn;
  }
  // A again, over G (the sums above have passed block_sum256's barriers: nobody reads G any more)
  for (int e = tid; e < MAT; e += 256) A[(e >> 7) * 129 + (e & 127)] = src[e];
  __syncthreads();
  double sigma = 0.0;
  bool converged = false;
  const int col = tid & 127, hv = tid >> 7;          // A^T u: column col, rows 32 hv .. 32 hv + 31
  const int row = tid & 63, qv = tid >> 6;           // A v:   row row, columns 32 qv .. 32 qv + 31
  for (int it = 0; it < 400; it++) {   // polish on the original matrix: u <- A (A^T u)
    {                                   // (four independent partial sums: the chain of 32 dependent LDS read -> multiply-adds was the iteration's time)
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      const double* ac = A + (32 * hv) * 129 + col;
      const double* uc = u + 32 * hv;
#pragma unroll
      for (int i = 0; i < 32; i += 4) {
        s0 += ac[i * 129] * uc[i]; s1 += ac[(i + 1) * 129] * uc[i + 1]; s2 += ac[(i + 2) * 129] * uc[i + 2]; s3 += ac[(i + 3) * 129] * uc[i + 3];
      }
      red[tid] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (tid < 128) v[tid] = red[tid] + red[tid + 128];
    __syncthreads();
    {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      const double* ar = A + row * 129 + 32 * qv;
      const double* vc = v + 32 * qv;
#pragma unroll
      for (int k = 0; k < 32; k += 4) { s0 += ar[k] * vc[k]; s1 += ar[k + 1] * vc[k + 1]; s2 += ar[k + 2] * vc[k + 2]; s3 += ar[k + 3] * vc[k + 3]; }
      red[tid] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    double un = 0.0;
    if (tid < 64) {                    // one wave: norm and largest change by lane exchanges
      un = (red[tid] + red[tid + 64]) + (red[tid + 128] + red[tid + 192]);
      double n2 = un * un;
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) n2 += __shfl_xor(n2, s, 64);
      const double nn = sqrt(n2);      // = sigma^2 at convergence
      un = un / nn;
      double df = fabs(un - u[tid]);
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) df = fmax(df, __shfl_xor(df, s, 64));
      u[tid] = un;
      if (tid == 0) { bc[0] = nn; bc[1] = df; }
    }
    __syncthreads();
    sigma = sqrt(bc[0]);
    const double dmax = bc[1];
    if (dmax < 4e-15 && it >= 1) { converged = true; break; }
  }
  // sigma_2 / sigma_1 > ~0.99: the leading pair is ill-defined in the reference too (JacobiSVD returns whichever of the two
  // near-equal directions its sweeps end on); the vector reached so far is written and the call reports it
  // (pr_take_warnings bit PR_WARN_M2DP_SVD; the row joins the list pr_m2dp_svd_rows returns: [0] = count, then up to M2DP_SVD_ROWS_CAP rows)
  if (!converged && tid == 0) {
    atomicOr(flags + 1, 1);
    const int slot = atomicAdd(svd_rows, 1);
    if (slot < M2DP_SVD_ROWS_CAP) svd_rows[1 + slot] = (c0 + cl) * 4 + var;
  }
  // v = A^T u / sigma, sign so that sum(u) >= 0
  __syncthreads();
  {
    double s = 0.0;
    for (int i = 32 * hv; i < 32 * hv + 32; i++) s += A[i * 129 + col] * u[i];
    red[tid] = s;
  }
  __syncthreads();
  if (tid < 128) v[tid] = (red[tid] + red[tid + 128]) / sigma;
  __syncthreads();
  const double su = block_sum256(tid < 64 ? u[tid] : 0.0, red, tid);
  const double sg = (su < 0.0) ? -1.0 : 1.0;
  if (tid < 64) o[tid] = sg * u[tid];
  if (tid < 128) o[64 + tid] = sg * v[tid];
}

constexpr size_t SVD_LDS = (size_t)(64 * 129 + 64 + 128 + 256 + 4) * sizeof(double);   // 69 856 B: two workgroups per CU
constexpr int GEN_BATCH = 256;   // clouds per scratch batch (256 * 4 * 2 * 64 KiB = 128 MiB)

}  // namespace

size_t m2dp_generate_scratch_bytes(int N) {     // one buffer of matrices per batch in flight: two when there is more than one batch
  const int nb = N < GEN_BATCH ? N : GEN_BATCH;
  return (size_t)(N > GEN_BATCH ? 2 : 1) * nb * 4 * 2 * MAT * sizeof(double);
}

// Batches of GEN_BATCH clouds: the binning of batch b + 1 (VALU-bound, stream st) runs beside the singular pairs of batch b (LDS-bound, one
// 100 KB workgroup per CU, stream st2) - two matrix buffers, ev_bin[p] / ev_svd[p] order the two streams per buffer.  On return st has
// joined st2.  (st2 / events null, or a single batch: everything on st.)
void launch_m2dp_bin_svd(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int N,
                         double max_rho, const double* frames, const float* ave, const double* planes, double* mats,
                         double* out, int* flags, int* svd_rows, hipStream_t st2, hipEvent_t* ev_bin, hipEvent_t* ev_svd) {
  if (N <= 0) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(m2dp_svd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)SVD_LDS);
  const bool two = N > GEN_BATCH && st2 && ev_bin && ev_svd;
  const size_t buf = (size_t)GEN_BATCH * 4 * 2 * MAT;
  int b = 0;
  for (int c0 = 0; c0 < N; c0 += GEN_BATCH, b++) {
    const int nc = (N - c0) < GEN_BATCH ? (N - c0) : GEN_BATCH;
    const int p = two ? (b & 1) : 0;
    double* m = mats + (size_t)p * buf;
    if (two && b >= 2) (void)hipStreamWaitEvent(st, ev_svd[p], 0);          // the matrices of batch b - 2 have been consumed
    if (nc * 16 >= 768)        // one full round of 3 workgroups per CU: the 16-plane form reads and aligns a point once per 16 projections
      hipLaunchKernelGGL(m2dp_bin_kernel<16>, dim3(nc * 16), dim3(256), 0, st, xyz, inten, offs, frames, ave, planes, max_rho,
                         c0, m);
    else
      hipLaunchKernelGGL(m2dp_bin_kernel<4>, dim3(nc * 64), dim3(256), 0, st, xyz, inten, offs, frames, ave, planes, max_rho,
                         c0, m);
    hipStream_t ss = st;
    if (two) {
      (void)hipEventRecord(ev_bin[p], st);
      (void)hipStreamWaitEvent(st2, ev_bin[p], 0);
      ss = st2;
    }
    hipLaunchKernelGGL(m2dp_svd_kernel, dim3(nc * 8), dim3(256), SVD_LDS, ss, m, c0, out, flags, svd_rows);
    if (two) (void)hipEventRecord(ev_svd[p], st2);
  }
  if (two) {
    (void)hipStreamWaitEvent(st, ev_svd[0], 0);
    (void)hipStreamWaitEvent(st, ev_svd[1], 0);
  }
}

}  // namespace pr
