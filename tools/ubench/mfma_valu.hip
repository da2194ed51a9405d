// microbenchmark: do VALU instructions of the SAME wave hide behind v_mfma_f32_32x32x2_f32 (one wave per SIMD)?
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o tools/ubench/mfma_valu ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int K, int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ float sh[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = i * 1e-3f;
  __syncthreads();
  const float* lp = sh + (threadIdx.x & 63);
  f32x16 acc[8];
  for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float v[16];
  for (int i = 0; i < 16; i++) v[i] = a + i;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < K; j++) {
        float& x = v[(i * K + j) & 15];
        if (MODE == 0) x = x * b + a;                                   // plain VALU
        else if (MODE == 1) x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false)) + a;   // DPP VALU
        else x += lp[((i * K + j) & 15) * 64];                          // LDS read + VALU
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
  for (int i = 0; i < 16; i++) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int K, int MODE = 0> void run(float* d, unsigned long long* c, int threads = 256) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<K, MODE>), dim3(256), dim3(threads), 0, 0, d, c, iters);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("mode=%d threads=%d K=%2d fillers per MFMA: %.1f cycles per MFMA (per wave)\n", MODE, threads, K, (double)h / (iters * 8.0));
}
int main() {
  float* d; unsigned long long* c;
  hipMalloc(&d, 256 * 512 * 4); hipMalloc(&c, 8);
  run<0>(d, c); run<0>(d, c); run<2>(d, c); run<4>(d, c); run<8>(d, c); run<12>(d, c); run<16>(d, c); run<24>(d, c);
  run<0>(d, c, 512); run<4>(d, c, 512); run<8>(d, c, 512); run<16>(d, c, 512); run<24>(d, c, 512);
  run<8, 1>(d, c, 256); run<8, 1>(d, c, 512); run<16, 1>(d, c, 512);
  run<4, 2>(d, c, 256); run<4, 2>(d, c, 512); run<8, 2>(d, c, 512);
  return 0;
}
