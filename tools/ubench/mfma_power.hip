// microbenchmark: what does the chip SUSTAIN on back-to-back v_mfma_f32_32x32x16_f16 (no loads, no LDS, no VALU in the loop) as a function of
// the operand DATA?  The 2.5 PFLOP/s f16 peak is 1024 SIMDs x 1024 FLOP/clk x 2.4 GHz; under a dense MFMA stream the chip clocks against its
// power limit, and how far down depends on how much the operands toggle.  m2dp_match_h_kernel reads 0.54 of the 2.4 GHz peak = 76 % pipe-busy at
// ~1.7 GHz: this program says what 100 % pipe-busy reads with the same kind of operands - the ceiling that kernel is to be held against.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o tools/ubench/mfma_power ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// NA x NB accumulator tiles per wave (4 x 2 = the register tile of m2dp_match_h_kernel), PROD products per tile and K-step:
// PROD = 3: a_hi b_hi + a_hi b_lo + a_lo b_hi (split-f16), PROD = 1: a_hi b_hi only
// ORDER 0: the matcher's order (A operand fixed over the NB tiles of a product, B changes with every MFMA); ORDER 1: a B operand fixed over all
// NA tiles and the products that use it (B changes every 4 - 8 MFMAs, A with every MFMA) - does the operand-reuse pattern change the clock?
template <int PROD, int ORDER = 0>
__global__ __launch_bounds__(256, 2) void k(const f16x8* __restrict__ src, float* out, int iters) {
  constexpr int NA = 4, NB = 2;
  const int lane = threadIdx.x & 63;
  // operands: [kind a_hi, a_lo, b_hi, b_lo][tile][lane]; the same registers every iteration: consecutive MFMAs still see different operands
  f16x8 ah[NA], al[NA], bh[NB], bl[NB];
  for (int t = 0; t < NA; t++) { ah[t] = src[(0 * 4 + t) * 64 + lane]; al[t] = src[(1 * 4 + t) * 64 + lane]; }
  for (int t = 0; t < NB; t++) { bh[t] = src[(2 * 4 + t) * 64 + lane]; bl[t] = src[(3 * 4 + t) * 64 + lane]; }
  f32x16 acc[NA][NB];
  for (int i = 0; i < NA; i++) for (int j = 0; j < NB; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  for (int it = 0; it < iters; it++) {
    if constexpr (ORDER == 1) {
#pragma unroll
      for (int j = 0; j < NB; j++) {
#pragma unroll
        for (int i = 0; i < NA; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        if (PROD == 3) {
#pragma unroll
          for (int i = 0; i < NA; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < NA; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        }
      }
      continue;
    }
#pragma unroll
    for (int i = 0; i < NA; i++) {
#pragma unroll
      for (int j = 0; j < NB; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      if (PROD == 3) {
#pragma unroll
        for (int j = 0; j < NB; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < NA; i++) for (int j = 0; j < NB; j++) for (int e = 0; e < 16; e++) s += acc[i][j][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double gauss() {
  double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
  return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v);
}

int main(int argc, char** argv) {
  const double target_ms = argc > 1 ? atof(argv[1]) : 150.0;
  f16x8* d; float* o;
  hipMalloc(&d, 16 * 64 * 16); hipMalloc(&o, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"zeros", "ones (constant, exact products)", "M2DP-like: hi = f16(256 x), lo = f16(256 x - hi), x ~ N(0, 0.07)", "random bits (finite f16, all exponents)",
                         "M2DP-like, lo rounded to 6 significant bits", "M2DP-like, lo rounded to 4 significant bits", "M2DP-like, lo rounded to 2 significant bits",
                         "M2DP-like, lo = 0 (two of three products with a zero operand)", "M2DP-like, hi AND lo rounded to 6 significant bits (not usable: the data dependence)"};
  for (int data = 0; data < 9; data++) {
    std::vector<_Float16> h(16 * 64 * 8);
    srand(7);
    for (size_t i = 0; i < h.size(); i++) {
      const int kind = (int)(i / (4 * 64 * 8));          // 0 a_hi, 1 a_lo, 2 b_hi, 3 b_lo
      if (data == 0) h[i] = (_Float16)0.f;
      else if (data == 1) h[i] = (_Float16)1.f;
      else if (data == 2 || data >= 4) {
        const double x = 256.0 * 0.07 * gauss();
        _Float16 hi = (_Float16)x;
        _Float16 lo = (_Float16)(x - (double)hi);
        auto keep = [](_Float16 v, int bits) {           // round to `bits` significant bits (implicit one included)
          unsigned short b = __builtin_bit_cast(unsigned short, v);
          const int drop = 11 - bits;
          b = (unsigned short)((b + (1u << (drop - 1))) & ~((1u << drop) - 1u));
          return __builtin_bit_cast(_Float16, b);
        };
        if (data == 4) lo = keep(lo, 6);
        if (data == 5) lo = keep(lo, 4);
        if (data == 6) lo = keep(lo, 2);
        if (data == 7) lo = (_Float16)0.f;
        if (data == 8) { hi = keep(hi, 6); lo = keep(lo, 6); }
        h[i] = (kind & 1) ? lo : hi;
      } else {
        unsigned short b = (unsigned short)(rand() & 0xffff);
        if ((b & 0x7c00) == 0x7c00) b &= 0xbfff;          // no Inf / NaN
        h[i] = __builtin_bit_cast(_Float16, b);
      }
    }
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int order = 0; order <= (data == 2 ? 1 : 0); order++)
    for (int prod = 3; prod >= ((data >= 4 || order) ? 3 : 1); prod -= 2)
      for (int wg = 256; wg <= ((data >= 4 || order) ? 256 : 512); wg *= 2) {           // 256 workgroups = one wave per SIMD; 512 = two
        auto go = [&](int iters) {
          if (prod == 3 && order) hipLaunchKernelGGL((k<3, 1>), dim3(wg), dim3(256), 0, 0, d, o, iters);
          else if (prod == 3) hipLaunchKernelGGL(k<3>, dim3(wg), dim3(256), 0, 0, d, o, iters);
          else hipLaunchKernelGGL(k<1>, dim3(wg), dim3(256), 0, 0, d, o, iters);
        };
        const int mf = (prod == 3 ? 24 : 8);
        int iters = 20000;
        go(iters); hipDeviceSynchronize();
        hipEventRecord(e0); go(iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        iters = (int)(iters * target_ms / ms);
        double best = 0, last = 0;
        for (int rep = 0; rep < 3; rep++) {
          hipEventRecord(e0); go(iters); hipEventRecord(e1); hipEventSynchronize(e1);
          hipEventElapsedTime(&ms, e0, e1);
          last = (double)wg * 4 * iters * mf * 32768.0 / (ms * 1e-3) / 1e12;
          if (last > best) best = last;
        }
        printf("%-90s order=%d products=%d waves/SIMD=%d : %7.1f TFLOP/s (last of three %.0f ms launches; best %7.1f) = %.3f of 2500 = %.3f busy-GHz\n", names[data], order, prod,
               wg / 256, last, target_ms, best, last / 2500.0, last / 2500.0 * 2.4);
      }
  }
  return 0;
}
