// sc_gen.hip — PCA alignment (utils/pts_align.h:7-46) and Scan-Context signature (SC/SC.cpp:12-76) on gfx950.
//
// Three kernels, one workgroup per cloud (ave_chain: one lane per cloud), compiled with -ffp-contract=off so products/sums round as on the CPU:
//   ave_chain    : the reference's FLOAT sequential average of the intensities IN INPUT ORDER (SC.cpp:60-64,
//                  M2DP.cpp:77-81; a tree sum differs at ~3e-4 and flips bins).  The chain of P
//                  dependent v_add_f32 cannot be parallelised inside a cloud, so every LANE walks a different
//                  cloud (8 independent chains per wave, per-lane 16-byte loads, two batches of 8 loads in flight per lane).
//                  Runs on a high-priority side stream beside the moments AND the binning pass (11 cycles per dependent add: 0.27 ms per
//                  50 000 points - longer than either pass).  -> ave[c] (float) and / or frames[c][14..15] (frames.hpp).
//   cloud_frames : one streaming pass over the cloud: fp64 raw moments (sum p, sum p p^T), fixed reduction tree;
//                  mean, scatter matrix cov = sum pp^T - P mean mean^T (un-normalised as :30), 3x3 symmetric Jacobi
//                  eigen-solver, eigenvalues ascending, canonical signs (N3)
//                  -> frames[c] = {mean[3], v0[3], v1[3], v2[3], 0, P, [float average, flag]} (16 doubles, frames.hpp).  Callers that
//                  bring the frames (the GPU pre-stage emits them, averages included) skip this pass and the chain.
//   sc_bin       : second pass: centre, rotate (same association as the oracle), sector = floor((atan2(z,y)+pi)*60/2pi),
//                  ring = floor(sqrt(y^2+z^2)*20/max_rho), idx = sector*20 + ring, dropped iff idx >= 1200 (the
//                  ring-overflow aliasing of SC.cpp:39-44 is kept, H3); LDS-resident 1200-bin grids with LDS
//                  atomics: count (u32 add), min/max of x (order-preserving u64 keys), fp64 intensity sum; epilogue
//                  max-min and mean > ave ? 1 : 0 (:67-75) -> out[c][2400].  When the average is still on its way the pass writes the
//                  bin MEANS (sc_bin_kernel<true>) and sc_finish_kernel applies the averages afterwards.
// Bound: HBM (28 B per point per pass); the points are never written back.
//
// The DEFAULT is exactly these two streaming passes over all clouds at once (moments, then binning: 24 + 28 B per point at ~5 TB/s each,
// four points of a thread in flight); with the caller's frames it is the binning pass alone (0.28 ms per 1024 x 50k points = 0.64 of 8 TB/s).
// Two one-pass designs are kept in this file for A/B runs only - both parity-green, both slower (tools/experiments/
// README.md): PR_SC_GEN=batched (batches of ~96 MB so that the binning pass re-reads the Infinity Cache; a cloud split over W workgroups with
// last-arriver merges: cloud_frames_split / sc_bin_split / sc_finish; 6.25 ms against 2.52 ms at 5000 x 50k points) and PR_SC_GEN=cluster
// (eight workgroups of one XCD hold a cloud in registers between the moments and the binning: sc_gen_cluster_kernel; 6.6 ms).
#include "fast_bins.hpp"
#include "frames.hpp"
#include "kernels.hpp"

namespace pr {
namespace {

constexpr int FT = FRAME_THREADS;   // threads of cloud_frames
constexpr int RW = FT / 64;

// The float average of the reference is a sequential sum in input order (SC.cpp:60-64, M2DP.cpp:77-81): one dependent
// v_add_f32 per point, ~5 cycles each - 0.1 ms for 50k points whatever the number of clouds.  One wave per CPW = 8
// clouds: all 64 lanes stream the next CH = 512 floats of each of the 8 clouds (coalesced 16-byte loads, requested one
// round = 512 adds ahead) into LDS, then lane k < 8 adds cloud k's chunk in order out of LDS.  (Lane-private loads - one
// cache line per lane and instruction, 64 floats in flight - left the chain waiting on memory: 32 cycles per point.)
constexpr int CPW = 8;
constexpr int ACH = 512;                     // floats per cloud and round
__global__ __launch_bounds__(64) void ave_chain_kernel(const float* __restrict__ inten, const int64_t* __restrict__ offs,
                                                        int N, float* __restrict__ ave_out, double* __restrict__ frames_out) {
  // rows of ACH + 4 floats: the eight adding lanes read eight different bank groups; one more row: the read-ahead of the last block stays inside
  __shared__ __attribute__((aligned(16))) float buf[CPW + 1][ACH + 4];
  typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
  typedef float f4a __attribute__((ext_vector_type(4)));
  // the chain is the critical path of a generate call and shares its CU with the waves of the moments / binning pass: issue priority
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x;
  const int cb = blockIdx.x * CPW;
  const float* base[CPW];
  int64_t Pk[CPW];
  int64_t Pmax = 0;
#pragma unroll
  for (int k = 0; k < CPW; k++) {
    const bool live = cb + k < N;
    const int64_t o0 = live ? offs[cb + k] : 0;
    Pk[k] = live ? offs[cb + k + 1] - o0 : 0;
    base[k] = inten + o0;
    Pmax = Pk[k] > Pmax ? Pk[k] : Pmax;
  }
  f4 nx[CPW][2];
  auto request = [&](int64_t r0) {            // floats [r0 + 8 lane, + 8) of every cloud; zeros past the end (x + 0 = x)
#pragma unroll
    for (int k = 0; k < CPW; k++) {
      const int64_t i = r0 + 8 * lane;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int64_t j = i + 4 * h;
        if (j + 4 <= Pk[k]) nx[k][h] = *reinterpret_cast<const f4*>(base[k] + j);
        else {
          f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; e++) if (j + e < Pk[k]) v[e] = base[k][j + e];
          nx[k][h] = v;
        }
      }
    }
  };
  float ave = 0.f;
  request(0);
  for (int64_t r0 = 0; r0 < Pmax; r0 += ACH) {
#pragma unroll
    for (int k = 0; k < CPW; k++) {
      *reinterpret_cast<f4a*>(&buf[k][8 * lane]) = nx[k][0];
      *reinterpret_cast<f4a*>(&buf[k][8 * lane + 4]) = nx[k][1];
    }
    if (r0 + ACH < Pmax) request(r0 + ACH);
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the LDS writes of this wave are done (one wave per workgroup)
    if (lane < CPW) {
      // 32 floats in registers while the next 32 are on their way from LDS: the chain waits for the adds only (~4.5 cycles each).  The
      // scheduling barriers keep hipcc from hoisting a block's reads above the adds that still use its registers (it then copies 32 registers
      // per block); the last block reads 32 floats ahead for nothing.
      const f4a* src = reinterpret_cast<const f4a*>(&buf[lane][0]);
      f4a a[8], b[8];
#pragma unroll
      for (int q = 0; q < 8; q++) a[q] = src[q];
#pragma unroll 1
      for (int j = 0; j < ACH / 4; j += 16) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; q++) b[q] = src[j + 8 + q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; q++) { ave += a[q][0]; ave += a[q][1]; ave += a[q][2]; ave += a[q][3]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = src[j + 16 + q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; q++) { ave += b[q][0]; ave += b[q][1]; ave += b[q][2]; ave += b[q][3]; }
      }
    }
  }
  if (lane < CPW && cb + lane < N) {
    int64_t P = 0;
#pragma unroll
    for (int k = 0; k < CPW; k++) P = (lane == k) ? Pk[k] : P;
    const float a = ave / (float)P;          // SC.cpp:64
    if (ave_out) ave_out[cb + lane] = a;
    if (frames_out) { frames_out[(size_t)(cb + lane) * 16 + 14] = (double)a; frames_out[(size_t)(cb + lane) * 16 + 15] = 1.0; }   // frames.hpp
  }
}

__global__ __launch_bounds__(FT) void cloud_frames_kernel(const double* __restrict__ xyz,
                                                           const int64_t* __restrict__ offs, double* __restrict__ frames) {
  __shared__ double red[RW][9];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int64_t o0 = offs[c];
  const int64_t P = offs[c + 1] - o0;
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const double* p = xyz + 3 * o0;
  auto add = [&](double x, double y, double z) {
    s[0] += x; s[1] += y; s[2] += z;
    s[3] += x * x; s[4] += x * y; s[5] += x * z; s[6] += y * y; s[7] += y * z; s[8] += z * z;
  };
  // four points of the thread in flight (one point per trip left the pass at 4.8 TB/s: 4 waves per SIMD x 24 B); the adds keep the
  // thread's point order t, t + 256, ... that frames.hpp's contract names
  int64_t i = tid;
  for (; i + 3 * FT < P; i += 4 * FT) {
    double v[4][3];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int64_t j = i + u * FT; v[u][0] = p[3 * j]; v[u][1] = p[3 * j + 1]; v[u][2] = p[3 * j + 2]; }
#pragma unroll
    for (int u = 0; u < 4; u++) add(v[u][0], v[u][1], v[u][2]);
  }
  for (; i < P; i += FT) add(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
  reduce_moments_to_frame(s, (double)P, red, frames + (size_t)c * 16);
}

__device__ __forceinline__ unsigned long long dkey(double x) {   // order-preserving map double -> u64
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// MEAN: out[1200 + b] = the bin's mean intensity (-inf for an empty bin) instead of the 0/1 of SC.cpp:69-70 - sc_finish_kernel applies the
// float average afterwards, so that the binning does not wait for the average chain (which runs beside it on a side stream)
template <bool MEAN>
__global__ __launch_bounds__(512) void sc_bin_kernel(const double* __restrict__ xyz, const float* __restrict__ inten,
                                                      const int64_t* __restrict__ offs, const double* __restrict__ frames,
                                                      const float* __restrict__ ave_in, double max_rho,
                                                      double* __restrict__ out) {
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
  const float S_f = (float)S_res_inv, R_f = (float)R_res_inv;
  const double* p = xyz + 3 * o0;
  const float* it = inten + o0;
  auto bin = [&](double px, double py, double pz, float iv) {
    const double x = px - mx, y = py - my, z = pz - mz;                             // pts_align.h:24-26
    const double nx = (x * e00 + y * e01) + z * e02;                                // :37-39
    const double yp = (x * e10 + y * e11) + z * e12;
    const double zp = (x * e20 + y * e21) + z * e22;
    const int si = polar_sector(zp, yp, S_res_inv, S_f);    // floor((atan2(zp, yp) + pi) * S_res_inv), SC.cpp:37
    const int ri = polar_ring(yp, zp, R_res_inv, R_f);      // floor(sqrt(yp^2 + zp^2) * R_res_inv),   SC.cpp:38
    const int idx = si * 20 + ri;                                                   // :39
    if (idx >= 1200 || idx < 0) return;                                             // :42-44
    atomicAdd(&cnt[idx], 1u);
    const unsigned long long k = dkey(nx);
    atomicMin(&lo[idx], k);
    atomicMax(&hi[idx], k);
    atomicAdd(&sum[idx], (double)iv);
  };
  int64_t i = tid;
  for (; i + 3 * 512 < P; i += 4 * 512) {          // four points of the thread in flight
    double v[4][3];
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int64_t j = i + u * 512; v[u][0] = p[3 * j]; v[u][1] = p[3 * j + 1]; v[u][2] = p[3 * j + 2]; w[u] = it[j]; }
#pragma unroll
    for (int u = 0; u < 4; u++) bin(v[u][0], v[u][1], v[u][2], w[u]);
  }
  for (; i < P; i += 512) bin(p[3 * i], p[3 * i + 1], p[3 * i + 2], it[i]);
  __syncthreads();
  // the float average, widened (double > float promotes the float, SC.cpp:70); ave_in = NULL: the caller's frame carries it (frames.hpp)
  const double ave = MEAN ? 0.0 : (ave_in ? (double)ave_in[c] : f[14]);
  double* o = out + (size_t)c * 2400;
  for (int b = tid; b < 1200; b += 512) {
    const unsigned int n = cnt[b];
    double st = 0.0, iv = MEAN ? -__builtin_inf() : 0.0;
    if (n) {
      st = dunkey(hi[b]) - dunkey(lo[b]);                                           // :74
      iv = MEAN ? sum[b] / (double)n : ((sum[b] / (double)n) > ave ? 1.0 : 0.0);    // :69-70
    }
    o[b] = st;
    o[1200 + b] = iv;
  }
}


// ---------------------------------------------------------------------------------------------------- split (batched) path
constexpr int PART_BYTES = 1200 * 4 + 3 * 1200 * 8;   // partial bin grids of one (cloud, slice): cnt u32 | lo u64 | hi u64 | sum f64

// last-arriver protocol (MI355X_MICROARCH.md "Valid forms"): plain stores -> __syncthreads -> lane-0 agent-scope release ->
// ticket; the winner: agent-scope acquire -> __syncthreads -> plain loads.  The ticket counter resets itself for the next batch.
__device__ __forceinline__ bool last_arriver(unsigned* ticket, int W, int tid, int* flag) {
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == (unsigned)(W - 1));
    if (last) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}

__global__ __launch_bounds__(FT) void cloud_frames_split_kernel(const double* __restrict__ xyz, const int64_t* __restrict__ offs,
                                                                 int c0, int W, double* __restrict__ partial,
                                                                 unsigned* __restrict__ ticket, double* __restrict__ frames) {
  __shared__ double red[RW][9];
  __shared__ int flag;
  const int cl = blockIdx.x / W, sl = blockIdx.x - cl * W, c = c0 + cl;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t o0 = offs[c];
  const int64_t P = offs[c + 1] - o0;
  const int64_t i0 = P * sl / W, i1 = P * (sl + 1) / W;
  {
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double* p = xyz + 3 * o0;
    for (int64_t i = i0 + tid; i < i1; i += FT) {
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
  if (tid < 9) {
    double v = 0;
    for (int i = 0; i < RW; i++) v += red[i][tid];
    partial[((size_t)cl * W + sl) * 9 + tid] = v;
  }
  if (!last_arriver(ticket + cl, W, tid, &flag)) return;
  if (tid == 0) {
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int u = 0; u < W; u++)                                   // slice order: the sum does not depend on the arrival order
      for (int k = 0; k < 9; k++) s[k] += partial[((size_t)cl * W + u) * 9 + k];
    finish_frame(s, (double)P, frames + (size_t)c * 16);
  }
}

// binning of slice sl of cloud c; the last slice to finish merges all W partial grids and writes, per bin,
// out[b] = max - min of the aligned x (SC.cpp:74) and out[1200 + b] = MEAN intensity (-inf for an empty bin); sc_finish turns
// the mean into the 0/1 of SC.cpp:69-70 once the float average is known.
__global__ __launch_bounds__(512) void sc_bin_split_kernel(const double* __restrict__ xyz, const float* __restrict__ inten,
                                                            const int64_t* __restrict__ offs, int c0, int W,
                                                            const double* __restrict__ frames, double max_rho,
                                                            char* __restrict__ partial, unsigned* __restrict__ ticket,
                                                            double* __restrict__ out) {
  __shared__ unsigned int cnt[1200];
  __shared__ unsigned long long lo[1200], hi[1200];
  __shared__ double sum[1200];
  __shared__ int flag;
  const int cl = blockIdx.x / W, sl = blockIdx.x - cl * W, c = c0 + cl, tid = threadIdx.x;
  const int64_t o0 = offs[c];
  const int64_t P = offs[c + 1] - o0;
  const int64_t i0 = P * sl / W, i1 = P * (sl + 1) / W;
  for (int b = tid; b < 1200; b += 512) { cnt[b] = 0u; lo[b] = ~0ull; hi[b] = 0ull; sum[b] = 0.0; }
  __syncthreads();
  const double* f = frames + (size_t)c * 16;
  const double mx = f[0], my = f[1], mz = f[2];
  const double e00 = f[3], e01 = f[4], e02 = f[5], e10 = f[6], e11 = f[7], e12 = f[8], e20 = f[9], e21 = f[10], e22 = f[11];
  const double S_res_inv = 60 / (2.0 * M_PI), R_res_inv = 20 / max_rho;   // SC.cpp:5-8
  const float S_f = (float)S_res_inv, R_f = (float)R_res_inv;
  const double* p = xyz + 3 * o0;
  const float* it = inten + o0;
  for (int64_t i = i0 + tid; i < i1; i += 512) {
    const double x = p[3 * i] - mx, y = p[3 * i + 1] - my, z = p[3 * i + 2] - mz;   // pts_align.h:24-26
    const double nx = (x * e00 + y * e01) + z * e02;                                // :37-39
    const double yp = (x * e10 + y * e11) + z * e12;
    const double zp = (x * e20 + y * e21) + z * e22;
    const int si = polar_sector(zp, yp, S_res_inv, S_f);    // SC.cpp:37
    const int ri = polar_ring(yp, zp, R_res_inv, R_f);      // SC.cpp:38
    const int idx = si * 20 + ri;                                                   // :39
    if (idx >= 1200 || idx < 0) continue;                                           // :42-44
    atomicAdd(&cnt[idx], 1u);
    const unsigned long long k = dkey(nx);
    atomicMin(&lo[idx], k);
    atomicMax(&hi[idx], k);
    atomicAdd(&sum[idx], (double)it[i]);
  }
  __syncthreads();
  if (W > 1) {
    char* mine = partial + ((size_t)cl * W + sl) * PART_BYTES;
    unsigned* pc = reinterpret_cast<unsigned*>(mine);
    unsigned long long* pl = reinterpret_cast<unsigned long long*>(mine + 4800);
    unsigned long long* ph = pl + 1200;
    double* ps = reinterpret_cast<double*>(ph + 1200);
    for (int b = tid; b < 1200; b += 512) { pc[b] = cnt[b]; pl[b] = lo[b]; ph[b] = hi[b]; ps[b] = sum[b]; }
    if (!last_arriver(ticket + cl, W, tid, &flag)) return;
    for (int u = 0; u < W; u++) {
      if (u == sl) continue;
      const char* o = partial + ((size_t)cl * W + u) * PART_BYTES;
      const unsigned* qc = reinterpret_cast<const unsigned*>(o);
      const unsigned long long* ql = reinterpret_cast<const unsigned long long*>(o + 4800);
      const unsigned long long* qh = ql + 1200;
      const double* qs = reinterpret_cast<const double*>(qh + 1200);
      for (int b = tid; b < 1200; b += 512) {
        cnt[b] += qc[b];
        lo[b] = ql[b] < lo[b] ? ql[b] : lo[b];
        hi[b] = qh[b] > hi[b] ? qh[b] : hi[b];
        sum[b] += qs[b];                                         // exact: floats of <= 24 bits, a few thousand per bin
      }
    }
  }
  double* o = out + (size_t)c * 2400;
  for (int b = tid; b < 1200; b += 512) {
    const unsigned int n = cnt[b];
    o[b] = n ? dunkey(hi[b]) - dunkey(lo[b]) : 0.0;                                 // :74
    o[1200 + b] = n ? sum[b] / (double)n : -__builtin_inf();                        // the mean; sc_finish compares it with the average
  }
}

// ---------------------------------------------------------------------------------------------------- cluster path: ONE HBM pass
// A cloud is read from HBM ONCE: CW workgroups (a cluster, all on one XCD, resident for the whole launch) each pull one slice
// of it - up to 16 points per thread - into REGISTERS, add up the slice's moments, hand them to the cluster (last arriver
// computes the PCA frame and publishes it with a sequence number the others spin on), and then bin their register-resident
// points; the last workgroup to finish merges the CW partial bin grids and writes the signature.  Algorithmic bytes = HBM bytes
// (28 per point + the output); the hand-offs stay in the XCD's L2.  A cluster walks clouds cl, cl + ncl, ...; scratch is
// double-buffered by the parity of the walk (nobody can get two clouds ahead: the next frame needs everybody's moments).
// Spins are bounded: a lost hand-off sets *err instead of hanging the GPU.
constexpr int CK = 13;                        // points per thread held in registers
constexpr int CL_PART = PART_BYTES;           // partial grid of one slice
struct ClusterScratch {                       // per cluster, x 2 parities: see sc_cluster_scratch_bytes
  unsigned ticket_m, ticket_b, seq[2], xcc[4];
};
// Hand-offs INSIDE one XCD: its L2 is the coherence point (the vector L1 is write-through), so a producer only waits for its stores
// to be acknowledged (vmcnt 0) before the ticket / sequence atomic (executed in L2), and a consumer only invalidates its CU's L1
// (buffer_inv sc0) before reading - no L2 write-back / invalidate as an agent-scope fence would do across XCDs.  The kernel
// checks with HW_REG_XCC_ID that a cluster's workgroups really share an XCD and reports err = 2 otherwise (host: two-pass path).
__device__ __forceinline__ void l1_invalidate() { asm volatile("buffer_inv sc0" ::: "memory"); }
__device__ __forceinline__ bool last_arriver_xcd(unsigned* ticket, int W, int tid, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == (unsigned)(W - 1));
    if (last) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      l1_invalidate();
    }
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}
__device__ __forceinline__ bool spin_until(unsigned* seq, unsigned want, int* err) {
  for (int i = 0; i < (1 << 22); i++) {
    if (__hip_atomic_load(seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  *err = 1;
  return false;
}
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void sc_gen_cluster_kernel(const double* __restrict__ xyz, const float* __restrict__ inten, const int64_t* __restrict__ offs, int N,
                           double max_rho, int CW, int ncl, ClusterScratch* __restrict__ cs, double* __restrict__ pmom,
                           double* __restrict__ pframe, char* __restrict__ pgrid, double* __restrict__ frames,
                           double* __restrict__ out, int* __restrict__ err) {
  __shared__ unsigned int cnt[1200];
  __shared__ unsigned long long lo[1200], hi[1200];
  __shared__ double sum[1200];
  __shared__ double red[8][9];
  __shared__ double frs[16];
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int cl = xcd + 8 * (j / CW), sl = j % CW;       // the cluster's workgroups share an XCD (workgroups go round-robin to XCDs)
  if (cl >= ncl) return;
  const double S_res_inv = 60 / (2.0 * M_PI), R_res_inv = 20 / max_rho;   // SC.cpp:5-8
  const float S_f = (float)S_res_inv, R_f = (float)R_res_inv;
  ClusterScratch* my = cs + cl;
  if (tid == 0) {   // all workgroups of a cluster must sit on one XCD (that is what the cheap hand-offs assume)
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x = (x & 0xf) + 1;
    const unsigned prev = __hip_atomic_exchange(&my->xcc[0], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev != 0 && prev != x) *err = 2;
  }
  unsigned it = 0;
  for (int c = cl; c < N; c += ncl, it++) {
    const int par = it & 1;
    const int64_t o0 = offs[c];
    const int64_t P = offs[c + 1] - o0;
    const int64_t i0 = P * sl / CW, i1 = P * (sl + 1) / CW;
    const double* p = xyz + 3 * o0;
    const float* itn = inten + o0;
    double px[CK], py[CK], pz[CK];
    float pv[CK];
#pragma unroll
    for (int u = 0; u < CK; u++) {              // the whole slice is requested at once: 16 x 28 bytes per thread in flight
      const int64_t i = i0 + tid + 512 * u;
      const bool ok = i < i1;
      px[u] = ok ? p[3 * i] : 0.0; py[u] = ok ? p[3 * i + 1] : 0.0; pz[u] = ok ? p[3 * i + 2] : 0.0;
      pv[u] = ok ? itn[i] : 0.f;
    }
    for (int b = tid; b < 1200; b += 512) { cnt[b] = 0u; lo[b] = ~0ull; hi[b] = 0ull; sum[b] = 0.0; }
    {   // moments of the slice (pts_align.h:9-21); points past the slice are zeros
      double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < CK; u++) {
        const double x = px[u], y = py[u], z = pz[u];
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
    double* mom = pmom + ((size_t)(cl * 2 + par) * CW) * 9;
    double* fr = pframe + (size_t)(cl * 2 + par) * 16;
    if (tid < 9) {
      double v = 0;
      for (int q = 0; q < 8; q++) v += red[q][tid];
      mom[sl * 9 + tid] = v;
    }
    if (last_arriver_xcd(&my->ticket_m, CW, tid, &flag)) {
      if (tid == 0) {
        double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int u = 0; u < CW; u++)                                  // slice order: the sum does not depend on the arrival order
          for (int k = 0; k < 9; k++) s[k] += mom[u * 9 + k];
        double f[16];
        finish_frame(s, (double)P, f);
        for (int k = 0; k < 16; k++) { frs[k] = f[k]; fr[k] = f[k]; frames[(size_t)c * 16 + k] = f[k]; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&my->seq[par], it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (tid == 0) {
      spin_until(&my->seq[par], it + 1, err);
      l1_invalidate();
      for (int k = 0; k < 16; k++) frs[k] = __hip_atomic_load(fr + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const double mx = frs[0], my_ = frs[1], mz = frs[2];
    const double e00 = frs[3], e01 = frs[4], e02 = frs[5], e10 = frs[6], e11 = frs[7], e12 = frs[8], e20 = frs[9], e21 = frs[10], e22 = frs[11];
#pragma unroll
    for (int u = 0; u < CK; u++) {
      if (i0 + tid + 512 * u >= i1) continue;
      const double x = px[u] - mx, y = py[u] - my_, z = pz[u] - mz;                   // pts_align.h:24-26
      const double nx = (x * e00 + y * e01) + z * e02;                                // :37-39
      const double yp = (x * e10 + y * e11) + z * e12;
      const double zp = (x * e20 + y * e21) + z * e22;
      const int si = polar_sector(zp, yp, S_res_inv, S_f);    // SC.cpp:37
      const int ri = polar_ring(yp, zp, R_res_inv, R_f);      // SC.cpp:38
      const int idx = si * 20 + ri;                                                   // :39
      if (idx >= 1200 || idx < 0) continue;                                           // :42-44
      atomicAdd(&cnt[idx], 1u);
      const unsigned long long k = dkey(nx);
      atomicMin(&lo[idx], k);
      atomicMax(&hi[idx], k);
      atomicAdd(&sum[idx], (double)pv[u]);
      __builtin_amdgcn_sched_barrier(0);        // one point at a time: interleaving the 16 bodies costs more registers than it hides
    }
    __syncthreads();
    if (CW > 1) {
      char* base = pgrid + ((size_t)(cl * 2 + par) * CW) * CL_PART;
      char* mine = base + (size_t)sl * CL_PART;
      unsigned* pc = reinterpret_cast<unsigned*>(mine);
      unsigned long long* pl = reinterpret_cast<unsigned long long*>(mine + 4800);
      unsigned long long* ph = pl + 1200;
      double* ps = reinterpret_cast<double*>(ph + 1200);
      for (int b = tid; b < 1200; b += 512) { pc[b] = cnt[b]; pl[b] = lo[b]; ph[b] = hi[b]; ps[b] = sum[b]; }
      if (!last_arriver_xcd(&my->ticket_b, CW, tid, &flag)) continue;
      for (int u = 0; u < CW; u++) {
        if (u == sl) continue;
        const char* o = base + (size_t)u * CL_PART;
        const unsigned* qc = reinterpret_cast<const unsigned*>(o);
        const unsigned long long* ql = reinterpret_cast<const unsigned long long*>(o + 4800);
        const unsigned long long* qh = ql + 1200;
        const double* qs = reinterpret_cast<const double*>(qh + 1200);
        for (int b = tid; b < 1200; b += 512) {
          cnt[b] += qc[b];
          lo[b] = ql[b] < lo[b] ? ql[b] : lo[b];
          hi[b] = qh[b] > hi[b] ? qh[b] : hi[b];
          sum[b] += qs[b];                                         // exact: floats of <= 24 bits, a few thousand per bin
        }
      }
    }
    double* o = out + (size_t)c * 2400;
    for (int b = tid; b < 1200; b += 512) {
      const unsigned int n = cnt[b];
      o[b] = n ? dunkey(hi[b]) - dunkey(lo[b]) : 0.0;                                 // :74
      o[1200 + b] = n ? sum[b] / (double)n : -__builtin_inf();                        // the mean; sc_finish compares it with the average
    }
    __syncthreads();
  }
}

// out[c][1200 + b]: mean intensity of the bin -> (mean > float average) ? 1 : 0   (SC.cpp:69-70)
__global__ __launch_bounds__(256) void sc_finish_kernel(const float* __restrict__ ave_in, int N, double* __restrict__ out) {
  const int c = blockIdx.x;
  const double ave = (double)ave_in[c];   // the float average, widened (double > float promotes the float, SC.cpp:70)
  double* o = out + (size_t)c * 2400 + 1200;
  for (int b = threadIdx.x; b < 1200; b += 256) o[b] = o[b] > ave ? 1.0 : 0.0;
}

}  // namespace

void launch_ave_chain(hipStream_t st, const float* inten, const int64_t* offs, int N, float* ave, double* frames) {
  if (N <= 0) return;
  hipLaunchKernelGGL(ave_chain_kernel, dim3((N + CPW - 1) / CPW), dim3(64), 0, st, inten, offs, N, ave, frames);
}


size_t sc_generate_scratch_bytes() { return (size_t)2 * SC_SCRATCH_PER_STREAM; }

// Clouds c0..c1-1 as ONE batch on stream st: moments then bins, W workgroups per cloud.  scratch: this stream's
// SC_SCRATCH_PER_STREAM bytes = [tickets frames | tickets bins | partial moments | partial grids], tickets zero once.
void launch_sc_batch(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int c0, int c1, int W, double max_rho,
                     double* frames, char* scratch, double* out) {
  const int nb = c1 - c0;
  if (nb <= 0) return;
  unsigned* tk_f = reinterpret_cast<unsigned*>(scratch);
  unsigned* tk_b = tk_f + SC_MAX_SPLIT_CLOUDS;
  double* pm = reinterpret_cast<double*>(scratch + 2 * SC_MAX_SPLIT_CLOUDS * 4);
  char* pg = reinterpret_cast<char*>(pm + (size_t)SC_MAX_SPLIT_CLOUDS * SC_MAX_W * 9);
  if (W <= 1) {
    hipLaunchKernelGGL(cloud_frames_kernel, dim3(nb), dim3(FT), 0, st, xyz, offs + c0, frames + (size_t)c0 * 16);
    hipLaunchKernelGGL(sc_bin_split_kernel, dim3(nb), dim3(512), 0, st, xyz, inten, offs, c0, 1, frames, max_rho, pg, tk_b, out);
    return;
  }
  hipLaunchKernelGGL(cloud_frames_split_kernel, dim3(nb * W), dim3(FT), 0, st, xyz, offs, c0, W, pm, tk_f, frames);
  hipLaunchKernelGGL(sc_bin_split_kernel, dim3(nb * W), dim3(512), 0, st, xyz, inten, offs, c0, W, frames, max_rho, pg, tk_b, out);
}

// scratch of the cluster path for ncl clusters of CW workgroups: [ClusterScratch x ncl | moments | frames | partial grids | err]
size_t sc_cluster_scratch_bytes(int ncl, int CW) {
  return (size_t)ncl * sizeof(ClusterScratch) + (size_t)ncl * 2 * CW * 9 * 8 + (size_t)ncl * 2 * 16 * 8 + (size_t)ncl * 2 * CW * CL_PART + 64;
}
int sc_cluster_points_per_workgroup() { return 512 * CK; }

// clusters of CW workgroups, one workgroup per CU: grid = 8 XCDs x (ncu / 8 / CW) clusters x CW.  The first
// ncl * sizeof(ClusterScratch) bytes of scratch (tickets, sequence numbers) and the err word must be zero at entry
// (launch_zero_ints); returns the device address of the err word.
int* launch_sc_cluster(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int N, double max_rho, int CW, int ncu,
                       char* scratch, double* frames, double* out) {
  const int per_xcd = (ncu / 8) / CW;                  // clusters per XCD
  const int ncl = 8 * per_xcd;
  ClusterScratch* cs = reinterpret_cast<ClusterScratch*>(scratch);
  double* pmom = reinterpret_cast<double*>(scratch + (size_t)ncl * sizeof(ClusterScratch));
  double* pframe = pmom + (size_t)ncl * 2 * CW * 9;
  char* pgrid = reinterpret_cast<char*>(pframe + (size_t)ncl * 2 * 16);
  int* err = reinterpret_cast<int*>(pgrid + (size_t)ncl * 2 * CW * CL_PART);
  launch_zero_ints(st, reinterpret_cast<int*>(cs), (int)(ncl * sizeof(ClusterScratch) / 4));
  launch_zero_ints(st, err, 1);
  hipLaunchKernelGGL(sc_gen_cluster_kernel, dim3(8 * per_xcd * CW), dim3(512), 0, st, xyz, inten, offs, N, max_rho, CW, ncl, cs, pmom,
                     pframe, pgrid, frames, out, err);
  return err;
}

void launch_sc_finish(hipStream_t st, const float* ave, int N, double* out) {
  if (N <= 0) return;
  hipLaunchKernelGGL(sc_finish_kernel, dim3(N), dim3(256), 0, st, ave, N, out);
}

void launch_cloud_frames(hipStream_t st, const double* xyz, const int64_t* offs, int N, double* frames) {
  if (N <= 0) return;
  hipLaunchKernelGGL(cloud_frames_kernel, dim3(N), dim3(FT), 0, st, xyz, offs, frames);
}

void launch_sc_bin(hipStream_t st, const double* xyz, const float* inten, const int64_t* offs, int N, double max_rho,
                   const double* frames, const float* ave, double* out, int ave_in_frames) {
  if (N <= 0) return;
  if (ave || ave_in_frames) hipLaunchKernelGGL(sc_bin_kernel<false>, dim3(N), dim3(512), 0, st, xyz, inten, offs, frames, ave, max_rho, out);
  else hipLaunchKernelGGL(sc_bin_kernel<true>, dim3(N), dim3(512), 0, st, xyz, inten, offs, frames, ave, max_rho, out);   // + launch_sc_finish
}

}  // namespace pr
