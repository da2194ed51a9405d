// sc_pack.hip — Scan-Context signature packing for the matcher (gfx950).
//
// Follows processSC.m:15-20: every 1200-vector (one channel of a signature) is divided by its L2 norm.
// The matcher (sc_match.hip) evaluates the 120 column-shift / mirror variants of processSC.m:24-32 through a
// length-60 real DFT over the sector axis, so the normalised rows are stored here as their
// per-ring sector spectra  X_r[f] = (1/sqrt 60) * sum_s x[s*20+r] * exp(-2 pi i f s / 60),  f = 0..30,
// computed in fp64 and rounded once to fp32, laid out as the MFMA operand image of the set's role:
//   query image  [ch][group of 8][pos][s=0..4][lane]    lane = (k<<4) | (part<<3) | e   ring = 4s+k
//   (pos = processing order of the frequencies: 0,30,1,2,...,29; the DB image has a 32nd, all-zero position)
//   DB image     [ch][group of 16][pos]{ q4a[lane][4] (Re, s=0..3) | q4b[lane][4] (Im, s=0..3) | d2[lane][2] (Re,Im of s=4) }
//                                                       lane = (k<<4) | j               ring = 4s+k
// One workgroup per (row, channel).  HBM-trivial: 2400 values in, 2480 floats out per row.
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"

namespace pr {
namespace {

constexpr int SC_PACK_FEW = 8;    // signatures up to which a set is packed one workgroup per (signature, channel): see sc_pack_h_few_kernel

template <typename T>
__global__ __launch_bounds__(256) void sc_pack_kernel(const T* __restrict__ sig, int rows, int role,
                                                       float* __restrict__ packed, int groups,
                                                       const double* __restrict__ tw, int* __restrict__ flags,
                                                       int* __restrict__ bad) {
  __shared__ double x[1200];
  __shared__ double red[256];
  __shared__ double tws[120];
  const int tid = threadIdx.x;
  const int row = blockIdx.x >> 1, ch = blockIdx.x & 1;
  const T* src = sig + (size_t)row * 2400 + ch * 1200;
  double part = 0.0;
  for (int i = tid; i < 1200; i += 256) {
    double v = (double)src[i];
    x[i] = v;
    part += v * v;
  }
  if (tid < 120) tws[tid] = tw[tid];
  red[tid] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double nr = sqrt(red[0]);
  // zero (or NaN) norm: MATLAB produces a NaN row (processSC.m:16,19).  The row is packed as zeros and
  // marked in bad[row] (bit = channel); launch_nan_fixup writes the NaN distances behind the matcher.
  const bool isbad = !(nr > 0.0) || !(nr < __builtin_inf());
  if (tid == 0) { bad[2 * row + ch] = isbad ? (1 << ch) : 0; if (isbad) atomicOr(flags, 1); }
  for (int i = tid; i < 1200; i += 256) x[i] = isbad ? 0.0 : x[i] / nr;   // processSC.m:16,19
  __syncthreads();
  const double scale = 0.12909944487358055;  // 1/sqrt(60)
  for (int o = tid; o < SC_NF * 40; o += 256) {
    const int f = o / 40, rem = o - f * 40, ring = rem >> 1, im = rem & 1;
    double acc = 0.0;
    int t = 0;  // (f*s) mod 60
    for (int s = 0; s < 60; s++) {
      const double w = im ? -tws[60 + t] : tws[t];
      acc += x[s * 20 + ring] * w;
      t += f;
      if (t >= 60) t -= 60;
    }
    const float val = (float)(acc * scale);
    const int s4 = ring >> 2, k = ring & 3;
    size_t dst;
    if (role == 0) {  // query image
      const int g = row >> 3, e = row & 7;
      const int lane = (k << 4) | (im << 3) | e;
      dst = ((size_t)ch * groups + g) * SC_QIMG + (size_t)sc_fpos(f) * 320 + s4 * 64 + lane;
    } else {          // DB image
      const int g = row >> 4, j = row & 15;
      const int lane = (k << 4) | j;
      const size_t base = ((size_t)ch * groups + g) * SC_DIMG + (size_t)sc_fpos(f) * SC_DSTEP;
      dst = (s4 < 4) ? base + im * 256 + lane * 4 + s4 : base + 512 + lane * 2 + im;
    }
    packed[dst] = val;
  }
}


// Split-f16 images for sc_match_h.hip: every spectrum value x (queries scaled by 2^8, DB by 2^7) is stored as
// hi = f16(x), lo = f16(x - hi).
//   query image  [ch][group of 8][f = 0..30]{ 16 rows x 80 B: Qhi ring 0..19 | Qlo ring 16..19, 0..15 }, row = (Im<<3) | e,
//                rows 8..15 shifted by 8 B (LDS bank spread), 1288 B per frequency
//   DB image     [ch][group of 16][f = 0..30][Re hi | Re lo | Im hi | Im lo]{ 768 B: lane = (ring>>3)<<4 | j, 16 B per
//                lane = rings 8g..8g+7; rings 20..23 stay zero in the hi tiles and hold D hi 16..19 again in the lo tiles }
//   (kernels.hpp: how the matchers' two operand pairs per frequency read these)
template <typename T>
__global__ __launch_bounds__(320) void sc_pack_h_kernel(const T* __restrict__ sig, int rows, int role,
                                                         unsigned short* __restrict__ packed, int groups,
                                                         const double* __restrict__ tw, int* __restrict__ flags,
                                                         int* __restrict__ bad) {
  __shared__ double x[1200];
  __shared__ double red[320];
  __shared__ double tws[120];
  const int tid = threadIdx.x;                            // 320 threads = one per (ring, frequency <= 15)
  const int row = blockIdx.x >> 1, ch = blockIdx.x & 1;
  const T* src = sig + (size_t)row * 2400 + ch * 1200;
  double part = 0.0;
  for (int i = tid; i < 1200; i += 320) {
    double v = (double)src[i];
    x[i] = v;
    part += v * v;
  }
  if (tid < 120) tws[tid] = tw[tid];
  red[tid] = part;
  __syncthreads();
  if (tid < 64) red[tid] += red[tid + 256];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double nr = sqrt(red[0]);
  const bool isbad = !(nr > 0.0) || !(nr < __builtin_inf());   // see sc_pack_kernel
  if (tid == 0) { bad[2 * row + ch] = isbad ? (1 << ch) : 0; if (isbad) atomicOr(flags, 1); }
  for (int i = tid; i < 1200; i += 320) x[i] = isbad ? 0.0 : x[i] / nr;   // processSC.m:16,19
  __syncthreads();
  const double scale = 0.12909944487358055 * (role == 0 ? 256.0 : 128.0);  // 1/sqrt(60) x 2^8 | 2^7
  // One thread per (ring, frequency f <= 15): the bins f and 30 - f share every product, because
  // cos(2 pi (30-f) s/60) = (-1)^s cos(2 pi f s/60) and sin(2 pi (30-f) s/60) = -(-1)^s sin(2 pi f s/60): with the sums over
  // even and odd sectors kept apart, X_f = (Ce + Co) - i (Se + So) and X_{30-f} = (Ce - Co) + i (Se - So).
  {
    const int o = tid;
    const int f = o / 20, ring = o - f * 20;
    double ce = 0.0, co = 0.0, se = 0.0, so = 0.0;
    int t = 0;  // (f*s) mod 60
#pragma unroll 3
    for (int s2 = 0; s2 < 60; s2 += 2) {
      const double x0 = x[s2 * 20 + ring];
      ce += x0 * tws[t];
      se += x0 * tws[60 + t];
      t += f; if (t >= 60) t -= 60;
      const double x1 = x[(s2 + 1) * 20 + ring];
      co += x1 * tws[t];
      so += x1 * tws[60 + t];
      t += f; if (t >= 60) t -= 60;
    }
    auto put = [&](double val, int ff, int im) {
      const _Float16 hi = (_Float16)val;
      const _Float16 lo = (_Float16)(val - (double)hi);
      size_t bh, bl;   // byte offsets of hi and lo
      if (role == 0) {
        const int g = row >> 3, rr = (im << 3) | (row & 7);
        const size_t base = ((size_t)ch * groups + g) * SCH_QIMG + (size_t)ff * SCH_QBLK + rr * 80 + (rr >= 8 ? 8 : 0);
        bh = base + ring * 2;
        bl = base + sch_qlo_byte(ring);
      } else {
        const int g = row >> 4, j = row & 15;
        const size_t base = ((size_t)ch * groups + g) * SCH_DIMG + (size_t)ff * SCH_DFREQ + (size_t)im * 2 * SCH_DTILE +
                            (((ring >> 3) << 4) | j) * 16 + (ring & 7) * 2;
        bh = base;
        bl = base + SCH_DTILE;
        if (ring >= 16) packed[(bl + 8) >> 1] = __builtin_bit_cast(unsigned short, hi);    // D hi 16..19 again, behind D lo 16..19
      }
      packed[bh >> 1] = __builtin_bit_cast(unsigned short, hi);
      packed[bl >> 1] = __builtin_bit_cast(unsigned short, lo);
    };
    put((ce + co) * scale, f, 0);
    put(-(se + so) * scale, f, 1);
    if (f != 15) {                                      // 30 - 15 = 15: the same bin
      put((ce - co) * scale, 30 - f, 0);
      put((se - so) * scale, 30 - f, 1);
    }
  }
}

// The shipped split-f16 pack.  What bounded the kernel above: 1.5 LDS reads per multiply-add and 2-byte global stores.
// (The fp64 matrix cores are no way out: v_mfma_f64_16x16x4_f64 issues every ~156 cycles on gfx950 = 32 TFLOP/s, against
// 78 TFLOP/s of plain v_fma_f64 - tools/ubench/mfma_f64_rate.hip; a version of this kernel built on it ran at 1.85 ms.)
//   lane      = one ring column of one signature: 16 signatures x 20 rings = 320 threads; the lane walks its 60 sectors
//               straight from global memory (8 B per sector, 160-byte runs per signature) - no staging, no LDS reads
//   workgroup = 16 signatures x one channel x 4 frequencies f = 4 fb .. 4 fb + 3 and their partners 30 - f (even / odd
//               sector sums kept apart as above): 16 accumulators per lane
//   twiddles  = wave-uniform (every lane of the workgroup works on the same f), so they come through the scalar cache
//               into SGPRs and cost no vector or LDS bandwidth: 16 v_fma_f64 per 8-byte load, nothing else in the loop
//   norm      = every lane sums the squares of its column, the 20 partial sums of a signature are added in ring order
//   output    = hi / lo halves scattered into an LDS copy of the workgroup's 8 frequency slices of the group image
//               (a slice is contiguous: 3072 B per DB group, 1288 B per query group), then copied out linearly.
// Binary-channel statistics (kernels.hpp: ScBin; LO images only, channel 1, binfo != null): the frequency-block-0 workgroup of a row block
// decides whether each row is binary (all non-zero entries equal and positive) and leaves binfo and, in the block's slot behind bstat, the
// non-binary flag and the largest ones; every workgroup adds up the w-weighted squares of the rounding residuals val - hi of its own
// frequencies (in units of the normalised spectrum) per row and leaves the largest of its 16 rows in the same slot (entry 2 + fb);
// sc_bstat_finalize_kernel folds the slots into bstat[0..5] (OR / max: the result does not depend on any order).
template <typename T, int ROLE, bool LO>
__global__ __launch_bounds__(320) void sc_pack_h_col_kernel(const T* __restrict__ sig, int rows,
                                                             unsigned short* __restrict__ packed, int groups,
                                                             const double* __restrict__ tw, int* __restrict__ flags,
                                                             int* __restrict__ bad, float* __restrict__ binfo, int* __restrict__ bstat) {
  // LO = false: the single-product images (hi halves only, kernels.hpp SCF_*)
  constexpr int SL = ROLE == 0 ? (LO ? SCH_QBLK : SCF_QBLK) : (LO ? SCH_DFREQ : SCF_DFREQ);   // bytes of one (group, frequency) slice
  constexpr int NG = ROLE == 0 ? 2 : 1;                         // groups per 16 signatures
  constexpr int IMGB = ROLE == 0 ? (LO ? SCH_QIMG : SCF_QIMG) : (LO ? SCH_DIMG : SCF_DIMG);
  constexpr int QROW = LO ? 80 : 40;                            // bytes per query row: Q hi | Q lo
  __shared__ __attribute__((aligned(16))) char img[NG * 8 * SL];
  __shared__ double part[16 * 20];
  __shared__ int bnz[LO ? 320 : 1];                      // the columns' non-zero counts
  __shared__ float epart[LO ? 320 : 1];                  // ... and residual sums (not the same array: the statistics lanes still read bnz while the others store)
  const int tid = threadIdx.x, lrow = tid / 20, ring = tid - 20 * lrow;
  // workgroups go round-robin to the 8 XCDs (each with its own L2): the four frequency blocks of the same 16 signatures -
  // which read the same input - are consecutive workgroups of ONE XCD, so HBM sees the input once
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int fb = idx & 3, ch = (idx >> 2) & 1, blk = (idx >> 3) * 8 + xcd;
  if (blk * 16 >= rows) return;
  const int row = blk * 16 + lrow;
  const bool valid = row < rows;
  for (int i = tid; i < NG * 8 * SL / 8; i += 320) reinterpret_cast<unsigned long long*>(img)[i] = 0ull;   // padding stays zero
  const T* src = sig + (size_t)(valid ? row : 0) * 2400 + ch * 1200 + ring;   // lanes past the end redo signature 0 and drop it
  double ce[4] = {0, 0, 0, 0}, co[4] = {0, 0, 0, 0}, se[4] = {0, 0, 0, 0}, so[4] = {0, 0, 0, 0};
  double nsq = 0.0;
  const bool do_eps = LO && binfo != nullptr && ch == 1, do_bin = do_eps && fb == 0;     // workgroup-uniform
  int nz = 0;
  double vmn = __builtin_inf(), vmx = -__builtin_inf();
  float e2 = 0.f;
  const double* twc = tw + 120 + (size_t)fb * 60 * 8;           // [sector][j]{cos, sin}, wave-uniform
  // rolled loop over 4 sectors at a time, the column values requested two rounds (8 sectors) ahead of their use
#ifndef SCP_AHEAD
#define SCP_AHEAD 2     // rounds the column values are requested ahead of their use (the ring holds SCP_AHEAD + 1 rounds; 15 rounds: the ring size divides 15)
#endif
  constexpr int XR = SCP_AHEAD + 1;
  static_assert(15 % XR == 0, "the ring size must divide the 15 rounds");
  T xq[XR][4];
#pragma unroll
  for (int r = 0; r < SCP_AHEAD; r++)
#pragma unroll
    for (int k = 0; k < 4; k++) xq[r][k] = src[(r * 4 + k) * 20];
#pragma unroll XR
  for (int it = 0; it < 15; it++) {
    if (it + SCP_AHEAD < 15) {
#pragma unroll
      for (int k = 0; k < 4; k++) xq[(it + SCP_AHEAD) % XR][k] = src[((it + SCP_AHEAD) * 4 + k) * 20];
    }
    // (the 32 twiddles of a round are addressed through an opaque zero that the empty asm redefines together with the accumulators: the scalar
    //  loads of round it cannot start before round it - 1 has finished.  hipcc otherwise requests three rounds' worth ahead - 192 SGPRs, of
    //  which it spills ~100 through v_writelane / v_readlane: ~70 VALU instructions per round beside the 32 multiply-adds, 1.36 ms per 100k
    //  signatures.  The other workgroups of the CU cover the scalar cache's latency.)
    int tz = 0;
    asm volatile("" : "+s"(tz), "+v"(ce[0]), "+v"(ce[1]), "+v"(ce[2]), "+v"(ce[3]), "+v"(co[0]), "+v"(co[1]), "+v"(co[2]), "+v"(co[3]),
                      "+v"(se[0]), "+v"(se[1]), "+v"(se[2]), "+v"(se[3]), "+v"(so[0]), "+v"(so[1]), "+v"(so[2]), "+v"(so[3]));
    const double* tp = twc + it * 32 + tz;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const double x0 = (double)xq[it % XR][2 * h], x1 = (double)xq[it % XR][2 * h + 1];
      nsq += x0 * x0;
      nsq += x1 * x1;
      if (do_bin) {
        if (x0 != 0.0) { nz++; vmn = fmin(vmn, x0); vmx = fmax(vmx, x0); }
        if (x1 != 0.0) { nz++; vmn = fmin(vmn, x1); vmx = fmax(vmx, x1); }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        ce[j] += x0 * tp[h * 16 + 2 * j];
        se[j] += x0 * tp[h * 16 + 2 * j + 1];
        co[j] += x1 * tp[h * 16 + 8 + 2 * j];
        so[j] += x1 * tp[h * 16 + 8 + 2 * j + 1];
      }
    }
  }
  part[tid] = nsq;
  if (LO && do_bin) bnz[tid] = (nz == 0 || (vmn == vmx && vmn > 0.0)) ? nz : -1;      // this column: its non-zero entries are one positive value | not
  __syncthreads();
  double n2 = 0.0;
#pragma unroll
  for (int r = 0; r < 20; r++) n2 += part[lrow * 20 + r];
  const double nr = sqrt(n2);
  const bool isbad = !(nr > 0.0) || !(nr < __builtin_inf());   // MATLAB: NaN row -> zeros here + bad[row], see sc_pack_kernel
  if (valid && ring == 0 && fb == 0) { bad[2 * row + ch] = isbad ? (1 << ch) : 0; if (isbad) atomicOr(flags, 1); }
  // the statistics lanes: thread t < 16 speaks for row t of the block (one wave: their results are combined by shuffles) and the workgroup
  // leaves them in its own slot behind bstat - plain stores; sc_bstat_finalize_kernel folds the slots into bstat[0..5].  (Atomics on the
  // set's six numbers cost 0.7 - 0.9 ms per 100 000 rows: same-address atomics from eight XCDs serialise at ~25 ns each.)
  const int srow = blk * 16 + tid;
  bool sbad = true;
  if (LO && do_eps && tid < 16) {
    double n2t = 0.0;
#pragma unroll
    for (int r = 0; r < 20; r++) n2t += part[tid * 20 + r];
    const double nrt = sqrt(n2t);
    sbad = srow >= rows || !(nrt > 0.0) || !(nrt < __builtin_inf());
  }
  if (LO && do_bin) {                                   // (workgroup-uniform) the columns' values through `part` again: two more barriers in one workgroup of eight
    __syncthreads();
    part[tid] = vmx;
    __syncthreads();
    if (tid < 16) {
      int ones = 0;
      bool binary = true;
      double a = -__builtin_inf();
      for (int r = 0; r < 20; r++) {
        const int c = bnz[tid * 20 + r];
        const double v = part[tid * 20 + r];
        if (c < 0) binary = false;
        if (c > 0) { ones += c; if (a == -__builtin_inf()) a = v; else if (v != a) binary = false; }
      }
      binary = binary && ones > 0;
      float s1 = 1.f, rs = 1.f;
      if (!sbad && binary) { s1 = (float)sqrt((double)ones); rs = (float)(1.0 / sqrt((double)ones)); }
      if (srow < rows) { binfo[2 * srow] = s1; binfo[2 * srow + 1] = rs; }
      int mo = (!sbad && binary) ? ones : 0, nb = (!sbad && !binary) ? 1 : 0;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { mo = max(mo, __shfl_xor(mo, o)); nb |= __shfl_xor(nb, o); }
      if (tid == 0) { bstat[SC_BSTAT_INTS + 6 * blk] = nb; bstat[SC_BSTAT_INTS + 6 * blk + 1] = mo; }
    }
  }
  const double sc = isbad ? 0.0 : 0.12909944487358055 * (ROLE == 0 ? 256.0 : 128.0) / nr;   // 1/sqrt(60) x 2^8 | 2^7, over the norm (processSC.m:16,19)
  auto put = [&](double val, int slice, int im) {
    if (isbad) val = 0.0;
    // through fp32: two hardware converts instead of ~80 instructions of software f64 -> f16; the double rounding can
    // move hi by one f16 ulp in rare halfway cases, and lo = f16(val - hi) takes up the difference either way
    // (the empty asm keeps the compiler from folding the two casts back into its software f64 -> f16 sequence)
    float vf = (float)val;
    asm volatile("" : "+v"(vf));
    const _Float16 hi = (_Float16)vf;
    float rf = (float)(val - (double)hi);
    asm volatile("" : "+v"(rf));
    const _Float16 lo = (_Float16)rf;
    int bh, bl;   // byte offsets of hi and lo in the LDS copy
    if (ROLE == 0) {
      const int rr = (im << 3) | (lrow & 7);
      bh = ((lrow >> 3) * 8 + slice) * SL + rr * QROW + ((LO && rr >= 8) ? 8 : 0) + ring * 2;
      bl = bh - ring * 2 + sch_qlo_byte(ring);
    } else {
      bh = slice * SL + im * (LO ? 2 : 1) * SCH_DTILE + (((ring >> 3) << 4) | lrow) * 16 + (ring & 7) * 2;
      bl = bh + SCH_DTILE;
      if (LO && ring >= 16) *reinterpret_cast<_Float16*>(img + bl + 8) = hi;      // D hi 16..19 again, behind D lo 16..19 (kernels.hpp)
    }
    *reinterpret_cast<_Float16*>(img + bh) = hi;
    if (LO) *reinterpret_cast<_Float16*>(img + bl) = lo;
    if (LO) {                                             // residual of the hi half (rf: what lo is rounded from), weight w_f = 1 (f = 0, 30) | 2, in fp32
      const bool edge = (fb == 0 && (slice & 3) == 0);    // slice 0 of block 0: f = 0, slice 4: f = 30
      e2 = __builtin_fmaf(edge ? rf : 2.f * rf, rf, e2);
    }
  };
  if (valid) {
#pragma unroll
    for (int j = 0; j < 4; j++) {                         // slices 0..3: f = 4 fb + j, slices 4..7: 30 - f
      put((ce[j] + co[j]) * sc, j, 0);
      put(-(se[j] + so[j]) * sc, j, 1);
      if (4 * fb + j != 15) {                             // 30 - 15 = 15: the same bin
        put((ce[j] - co[j]) * sc, 4 + j, 0);
        put((se[j] - so[j]) * sc, 4 + j, 1);
      }
    }
  }
  if (LO && do_eps) epart[tid] = e2;
  __syncthreads();
  if (LO && do_eps && tid < 16) {
    float e = 0.f;
    for (int r = 0; r < 20; r++) e += epart[tid * 20 + r];
    if (sbad) e = 0.f;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) e = fmaxf(e, __shfl_xor(e, o));
    // (scaled values: 2^-16 | 2^-14 brings the squares back to the normalised spectrum; 1 + 2^-10 covers the fp32 sums and rf's own rounding;
    // non-negative floats order like their bit patterns)
    const int eb = __float_as_int(e * (ROLE == 0 ? 0x1p-16f : 0x1p-14f) * (1.f + 0x1p-10f));
    if (tid == 0) bstat[SC_BSTAT_INTS + 6 * blk + 2 + fb] = eb;
  }
  // slice (group gg, k) -> frequency (k < 4 ? 4 fb + k : 30 - 4 fb - (k - 4)) of group NG blk + gg
  constexpr int W = SL / 8;                               // 8-byte words per slice
  for (int i = tid; i < NG * 8 * W; i += 320) {
    const int sl = i / W, wd = i - sl * W, gg = sl >> 3, k = sl & 7;
    const int f = k < 4 ? 4 * fb + k : 30 - 4 * fb - (k - 4);
    if (k >= 4 && f == 15) continue;
    const int g = NG * blk + gg;
    if (g >= groups) continue;
    reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(packed) + ((size_t)ch * groups + g) * IMGB + (size_t)f * SL)[wd] =
        reinterpret_cast<const unsigned long long*>(img)[i];
  }
}

// A handful of signatures (the queries of an online call): the kernel above is built for throughput - its twiddles come through the scalar
// cache, 15 dependent cold misses per workgroup when nothing has warmed it: 16 - 21 us for ONE query.  Here one workgroup per (signature,
// channel), 320 threads = (frequency f <= 15, ring): column and twiddles staged in LDS once, then EXACTLY the arithmetic of the kernel above
// - raw sums over the even / odd sectors in sector order, the column's square sum in the same order, the 20 column sums added in ring order,
// one scale factor - so the image is bit for bit the one that kernel writes.  The workgroup writes only its own row of the group image (the
// caller has zeroed the image).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <typename T, int ROLE, bool LO>
__global__ __launch_bounds__(320) void sc_pack_h_few_kernel(const T* __restrict__ sig, int rows, unsigned short* __restrict__ packed, int groups,
                                                             const double* __restrict__ tw, int* __restrict__ flags, int* __restrict__ bad,
                                                             float* __restrict__ binfo, int* __restrict__ bstat,
                                                             int row0 /* place of sig's first row in the image, bad and binfo (an append; 0 otherwise) */) {
  constexpr int SL = ROLE == 0 ? (LO ? SCH_QBLK : SCF_QBLK) : (LO ? SCH_DFREQ : SCF_DFREQ);
  constexpr int IMGB = ROLE == 0 ? (LO ? SCH_QIMG : SCF_QIMG) : (LO ? SCH_DIMG : SCF_DIMG);
  constexpr int QROW = LO ? 80 : 40;
  __shared__ double x[1200];
  __shared__ double tws[120];
  __shared__ double part[20];
  __shared__ double wred[5][3];
  const int tid = threadIdx.x, f = tid / 20, ring = tid - 20 * f;
  const int row = row0 + (blockIdx.x >> 1), ch = blockIdx.x & 1;
  const T* src = sig + (size_t)(blockIdx.x >> 1) * 2400 + ch * 1200;
  const bool do_bin = LO && binfo != nullptr && ch == 1;    // binary-channel statistics, as in sc_pack_h_col_kernel (the row's whole residual
  double bz = 0.0, bmn_ = __builtin_inf(), bmx_ = -__builtin_inf();   // sum goes to bstat[2])
  for (int i = tid; i < 1200; i += 320) {
    const double v = (double)src[i];
    x[i] = v;
    if (do_bin && v != 0.0) { bz += 1.0; bmn_ = fmin(bmn_, v); bmx_ = fmax(bmx_, v); }
  }
  if (do_bin) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { bz += __shfl_xor(bz, o); bmn_ = fmin(bmn_, __shfl_xor(bmn_, o)); bmx_ = fmax(bmx_, __shfl_xor(bmx_, o)); }
    if ((tid & 63) == 0) { wred[tid >> 6][0] = bz; wred[tid >> 6][1] = bmn_; wred[tid >> 6][2] = bmx_; }
  }
  if (tid < 120) tws[tid] = tw[tid];
  __syncthreads();
  double ce = 0.0, co = 0.0, se = 0.0, so = 0.0, nsq = 0.0;
  int t = 0;  // (f*s) mod 60
#pragma unroll 1
  for (int s2 = 0; s2 < 60; s2 += 2) {      // (rolled: a cold kernel's instructions come from HBM too, ~1 us per few hundred bytes)
    const double x0 = x[s2 * 20 + ring], x1 = x[(s2 + 1) * 20 + ring];
    nsq += x0 * x0;
    nsq += x1 * x1;
    ce += x0 * tws[t];
    se += x0 * tws[60 + t];
    t += f; if (t >= 60) t -= 60;
    co += x1 * tws[t];
    so += x1 * tws[60 + t];
    t += f; if (t >= 60) t -= 60;
  }
  if (f == 0) part[ring] = nsq;
  __syncthreads();
  double n2 = 0.0;
#pragma unroll
  for (int r = 0; r < 20; r++) n2 += part[r];
  const double nr = sqrt(n2);
  const bool isbad = !(nr > 0.0) || !(nr < __builtin_inf());
  if (tid == 0) { bad[2 * row + ch] = isbad ? (1 << ch) : 0; if (isbad) atomicOr(flags, 1); }
  if (do_bin && tid == 0) {
    double ones = 0.0, mn = __builtin_inf(), mx = -__builtin_inf();
    for (int w5 = 0; w5 < 5; w5++) { ones += wred[w5][0]; mn = fmin(mn, wred[w5][1]); mx = fmax(mx, wred[w5][2]); }
    const bool binary = ones > 0.0 && mn == mx && mn > 0.0;
    float s = 1.f, rs = 1.f;
    if (!isbad && binary) { s = (float)sqrt(ones); rs = (float)(1.0 / sqrt(ones)); atomicMax(bstat + 1, (int)ones); }
    if (!isbad && !binary) atomicOr(bstat, 1);
    binfo[2 * row] = s; binfo[2 * row + 1] = rs;
  }
  const double sc = isbad ? 0.0 : 0.12909944487358055 * (ROLE == 0 ? 256.0 : 128.0) / nr;
  char* out = reinterpret_cast<char*>(packed);
  double e2 = 0.0;
  auto put = [&](double val, int ff, int im) {
    if (isbad) val = 0.0;
    float vf = (float)val;
    asm volatile("" : "+v"(vf));
    const _Float16 hi = (_Float16)vf;
    float rf = (float)(val - (double)hi);
    asm volatile("" : "+v"(rf));
    const _Float16 lo = (_Float16)rf;
    if (do_bin) {
      const double r = (val - (double)hi) * (ROLE == 0 ? 0x1p-8 : 0x1p-7);
      e2 += ((ff == 0 || ff == 30) ? 1.0 : 2.0) * r * r;
    }
    size_t bh;
    if (ROLE == 0) {
      const int g = row >> 3, rr = (im << 3) | (row & 7);
      bh = ((size_t)ch * groups + g) * IMGB + (size_t)ff * SL + rr * QROW + ((LO && rr >= 8) ? 8 : 0) + ring * 2;
      *reinterpret_cast<_Float16*>(out + bh) = hi;
      if (LO) *reinterpret_cast<_Float16*>(out + bh - ring * 2 + sch_qlo_byte(ring)) = lo;
    } else {
      const int g = row >> 4, j = row & 15;
      bh = ((size_t)ch * groups + g) * IMGB + (size_t)ff * SL + (size_t)im * (LO ? 2 : 1) * SCH_DTILE + (((ring >> 3) << 4) | j) * 16 + (ring & 7) * 2;
      *reinterpret_cast<_Float16*>(out + bh) = hi;
      if (LO) *reinterpret_cast<_Float16*>(out + bh + SCH_DTILE) = lo;
      if (LO && ring >= 16) *reinterpret_cast<_Float16*>(out + bh + SCH_DTILE + 8) = hi;
    }
  };
  put((ce + co) * sc, f, 0);
  put(-(se + so) * sc, f, 1);
  if (f != 15) {                                          // 30 - 15 = 15: the same bin
    put((ce - co) * sc, 30 - f, 0);
    put((se - so) * sc, 30 - f, 1);
  }
  if (do_bin) {                                           // (workgroup-uniform)
    e2 = wave_sum(e2);
    __syncthreads();                                      // wred was read by thread 0 above
    if ((tid & 63) == 0) wred[tid >> 6][0] = e2;
    __syncthreads();
    if (tid == 0 && !isbad) {
      double e = 0.0;
      for (int w5 = 0; w5 < 5; w5++) e += wred[w5][0];
      atomicMax(bstat + 2, __float_as_int((float)(e * (1.0 + 0x1p-20))));
    }
  }
}

// bstat[0..5] of a set packed by sc_pack_h_col_kernel: OR / max over its blocks' slots (one workgroup)
__global__ __launch_bounds__(256) void sc_bstat_finalize_kernel(int* __restrict__ bstat, int nblk) {
  __shared__ int red[6][256];
  int v[6] = {0, 0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < nblk; b += 256) {
    const int* w = bstat + SC_BSTAT_INTS + 6 * b;
    v[0] |= w[0];
#pragma unroll
    for (int i = 1; i < 6; i++) v[i] = max(v[i], w[i]);      // (ones, and non-negative floats as their bit patterns)
  }
#pragma unroll
  for (int i = 0; i < 6; i++) red[i][threadIdx.x] = v[i];
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) {
      red[0][threadIdx.x] |= red[0][threadIdx.x + s2];
#pragma unroll
      for (int i = 1; i < 6; i++) red[i][threadIdx.x] = max(red[i][threadIdx.x], red[i][threadIdx.x + s2]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 6) bstat[threadIdx.x] = red[threadIdx.x][0];
}

template <typename T, int ROLE, bool LO = true>
void launch_pack_col(hipStream_t st, const T* sig, int rows, unsigned short* packed, int groups, const double* tw, int* flags, int* bad,
                     float* binfo = nullptr, int* bstat = nullptr) {
  if (rows <= SC_PACK_FEW) {
    hipLaunchKernelGGL((sc_pack_h_few_kernel<T, ROLE, LO>), dim3((unsigned)rows * 2), dim3(320), 0, st, sig, rows, packed, groups, tw, flags, bad,
                       binfo, bstat, 0);
    return;
  }
  hipLaunchKernelGGL((sc_pack_h_col_kernel<T, ROLE, LO>), dim3((unsigned)(((rows + 15) / 16 + 7) / 8) * 64), dim3(320), 0, st, sig, rows, packed,
                     groups, tw, flags, bad, binfo, bstat);
  if (LO && bstat) hipLaunchKernelGGL(sc_bstat_finalize_kernel, dim3(1), dim3(256), 0, st, bstat, (rows + 15) / 16);
}

__global__ __launch_bounds__(256) void zero_ints_kernel(int* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0;
}

__global__ __launch_bounds__(256) void fill_ints_kernel(int* __restrict__ p, int n, int v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

void launch_zero_ints(hipStream_t st, int* p, int n) {
  if (n > 0) hipLaunchKernelGGL(zero_ints_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p, n);
}
void launch_fill_ints(hipStream_t st, int* p, int n, int v) {
  if (n > 0) hipLaunchKernelGGL(fill_ints_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p, n, v);
}

void launch_sc_pack_h(hipStream_t st, const void* sig, int dtype, int rows, int role, void* packed, int groups,
                      const double* twiddle, int* flags, int* bad, int single, float* binfo, int* bstat) {
  if (rows <= 0) return;
  if (single) {
    if (dtype == 0 && role == 0) launch_pack_col<double, 0, false>(st, (const double*)sig, rows, (unsigned short*)packed, groups, twiddle, flags, bad);
    else if (dtype == 0) launch_pack_col<double, 1, false>(st, (const double*)sig, rows, (unsigned short*)packed, groups, twiddle, flags, bad);
    else if (role == 0) launch_pack_col<float, 0, false>(st, (const float*)sig, rows, (unsigned short*)packed, groups, twiddle, flags, bad);
    else launch_pack_col<float, 1, false>(st, (const float*)sig, rows, (unsigned short*)packed, groups, twiddle, flags, bad);
    return;
  }
  static const bool valu = getenv("PR_SC_PACK") && !strcmp(getenv("PR_SC_PACK"), "valu");   // the per-thread DFT, kept for A/B runs
  if (valu && bstat) hipLaunchKernelGGL(fill_ints_kernel, dim3(1), dim3(256), 0, st, bstat, 1, 1);   // (no statistics there: "not binary")
  if (valu && dtype == 0)
    hipLaunchKernelGGL(sc_pack_h_kernel<double>, dim3(rows * 2), dim3(320), 0, st, (const double*)sig, rows, role,
                       (unsigned short*)packed, groups, twiddle, flags, bad);
  else if (valu)
    hipLaunchKernelGGL(sc_pack_h_kernel<float>, dim3(rows * 2), dim3(320), 0, st, (const float*)sig, rows, role,
                       (unsigned short*)packed, groups, twiddle, flags, bad);
  else if (dtype == 0 && role == 0) launch_pack_col<double, 0>(st, (const double*)sig, rows, (unsigned short*)packed, groups, twiddle, flags, bad, binfo, bstat);
  else if (dtype == 0) launch_pack_col<double, 1>(st, (const double*)sig, rows, (unsigned short*)packed, groups, twiddle, flags, bad, binfo, bstat);
  else if (role == 0) launch_pack_col<float, 0>(st, (const float*)sig, rows, (unsigned short*)packed, groups, twiddle, flags, bad, binfo, bstat);
  else launch_pack_col<float, 1>(st, (const float*)sig, rows, (unsigned short*)packed, groups, twiddle, flags, bad, binfo, bstat);
}

// rows [row0, row0 + rows) of a DB image whose other rows stay as they are (pr_sigset_append): one workgroup per (row, channel) of the
// few-rows kernel, whatever the count - bit for bit what a pack of the whole set writes for these rows.  The binary-channel statistics are
// FOLDED into bstat (or / max; the row's whole residual sum into slot [2], whose sum with [3..5] keeps bounding every row).
void launch_sc_pack_h_rows(hipStream_t st, const void* sig, int dtype, int rows, int row0, void* packed, int groups, const double* twiddle,
                           int* flags, int* bad, int single, float* binfo, int* bstat) {
  if (rows <= 0) return;
  unsigned short* pk = (unsigned short*)packed;
  const dim3 grid((unsigned)rows * 2), blk(320);
  if (single) {
    if (dtype == 0) hipLaunchKernelGGL((sc_pack_h_few_kernel<double, 1, false>), grid, blk, 0, st, (const double*)sig, rows, pk, groups, twiddle, flags, bad, nullptr, nullptr, row0);
    else hipLaunchKernelGGL((sc_pack_h_few_kernel<float, 1, false>), grid, blk, 0, st, (const float*)sig, rows, pk, groups, twiddle, flags, bad, nullptr, nullptr, row0);
  } else {
    if (dtype == 0) hipLaunchKernelGGL((sc_pack_h_few_kernel<double, 1, true>), grid, blk, 0, st, (const double*)sig, rows, pk, groups, twiddle, flags, bad, binfo, bstat, row0);
    else hipLaunchKernelGGL((sc_pack_h_few_kernel<float, 1, true>), grid, blk, 0, st, (const float*)sig, rows, pk, groups, twiddle, flags, bad, binfo, bstat, row0);
  }
}

void launch_sc_pack(hipStream_t st, const void* sig, int dtype, int rows, int role, float* packed, int groups,
                    const double* twiddle, int* flags, int* bad) {
  if (rows <= 0) return;
  if (dtype == 0)
    hipLaunchKernelGGL(sc_pack_kernel<double>, dim3(rows * 2), dim3(256), 0, st, (const double*)sig, rows, role,
                       packed, groups, twiddle, flags, bad);
  else
    hipLaunchKernelGGL(sc_pack_kernel<float>, dim3(rows * 2), dim3(256), 0, st, (const float*)sig, rows, role,
                       packed, groups, twiddle, flags, bad);
}

}  // namespace pr
