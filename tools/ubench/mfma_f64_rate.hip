// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: NACC independent accumulator chains per wave, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = f64x4{0, 0, 0, 0};
  const double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(double* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000, blocks = 256;
  k<NACC><<<blocks, 256>>>(d, 10);
  (void)hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * NACC;   // MFMAs per wave
  printf("NACC=%d: %.3f ms, %.1f ns per MFMA per wave (= %.0f cycles at 2.4 GHz), %.1f TFLOP/s\n", NACC, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4,
         n * blocks * 4 * 2048.0 / (ms * 1e-3) / 1e12);
}
int main() {
  double* d; (void)hipMalloc(&d, 256 * 256 * 8);
  run<1>(d); run<2>(d); run<5>(d); run<8>(d);
  return 0;
}
