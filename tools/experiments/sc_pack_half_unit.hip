// sc_pack.hip — Scan-Context signature packing for the matcher (gfx950).
//
// Follows processSC.m:15-20: every 1200-vector (one channel of a signature) is divided by its L2 norm.
// The matcher (sc_match.hip) evaluates the 120 column-shift / mirror variants of processSC.m:24-32 through a
// length-60 real DFT over the sector axis, so the normalised rows are stored here as their
// per-ring sector spectra  X_r[f] = (1/sqrt 60) * sum_s x[s*20+r] * exp(-2 pi i f s / 60),  f = 0..30,
// computed in fp64 and rounded once to fp32, laid out as the MFMA operand image of the set's role:
//   query image  [ch][group of 8][pos][s=0..4][lane]    lane = (k<<4) | row,  row = 4*(q>>1) + 2*(q&1) + part,  ring = 4s+k
//   (pos = processing order of the frequencies: 0,30,1,2,...,29; two positions = one slot of the matcher)
//   DB image     [ch][group of 16][slot][half of 8 entries]{ fa: [lane][4] (s=0..3) | fb: [lane][4] | [lane][2] = s=4 of (fa, fb) }
//                                                       lane = (k<<4) | col,  col = (d&7) + 8*part: [Re of 8 entries | Im of 8 entries]
// One workgroup per (row, channel).  HBM-trivial: 2400 values in, 2480 floats out per row.
#include "kernels.hpp"

namespace pr {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void sc_pack_kernel(const T* __restrict__ sig, int rows, int role,
                                                       float* __restrict__ packed, int groups,
                                                       const double* __restrict__ tw, int* __restrict__ flags) {
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
  if (!(nr > 0.0) && tid == 0) atomicOr(flags, 1);   // MATLAB would produce a NaN row
  for (int i = tid; i < 1200; i += 256) x[i] = x[i] / nr;   // processSC.m:16,19
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
    const int pos = sc_fpos(f);
    size_t dst;
    if (role == 0) {  // query image: A-operand row i = 4*(q>>1) + 2*(q&1) + part  (lane group t = q>>1 owns queries 2t, 2t+1)
      const int g = row >> 3, q = row & 7;
      const int lane = (k << 4) | (4 * (q >> 1) + 2 * (q & 1) + im);
      dst = ((size_t)ch * groups + g) * SC_QIMG + (size_t)pos * 320 + s4 * 64 + lane;
    } else {          // DB image: per slot two halves of 8 entries; B-operand column c = (d & 7) + 8*part
      const int g = row >> 4, d = row & 15;
      const int lane = (k << 4) | ((d & 7) + 8 * im);
      const int h = pos & 1;
      const size_t base = ((size_t)ch * groups + g) * SC_DIMG + (size_t)(pos >> 1) * (2 * SC_DSTEP) + (size_t)(d >> 3) * SC_DSTEP;
      dst = (s4 < 4) ? base + h * 256 + lane * 4 + s4 : base + 512 + lane * 2 + h;
    }
    packed[dst] = val;
  }
}

}  // namespace

void launch_sc_pack(hipStream_t st, const void* sig, int dtype, int rows, int role, float* packed, int groups,
                    const double* twiddle, int* flags) {
  if (rows <= 0) return;
  if (dtype == 0)
    hipLaunchKernelGGL(sc_pack_kernel<double>, dim3(rows * 2), dim3(256), 0, st, (const double*)sig, rows, role,
                       packed, groups, twiddle, flags);
  else
    hipLaunchKernelGGL(sc_pack_kernel<float>, dim3(rows * 2), dim3(256), 0, st, (const float*)sig, rows, role,
                       packed, groups, twiddle, flags);
}

}  // namespace pr
