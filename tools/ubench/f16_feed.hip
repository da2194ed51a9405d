// microbenchmarks for the split-f16 SC matcher design (gfx950):
//  (a) do VALU instructions of the SAME wave hide behind f16 MFMAs (16x16x32 / 32x32x16)?  one wave per SIMD
//  (b) L1-hit bandwidth per CU: 4 waves of a workgroup stream the same small region with global_load_dwordx4
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/f16_feed.hip -o tools/ubench/f16_feed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int K, int SHAPE, int FILL, int T>
__global__ __launch_bounds__(T, 1) void mf(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc32[4];
  f32x4 acc16[8];
  for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) acc32[i][e] = 0.f;
  for (int i = 0; i < 8; i++) for (int e = 0; e < 4; e++) acc16[i][e] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (_Float16)(threadIdx.x * 1e-3f + e); b[e] = (_Float16)(1.0f + e * 0.1f); }
  float x0 = threadIdx.x * 1e-3f, x1 = 1.0f + threadIdx.x * 1e-4f;
  float v[16];
  for (int i = 0; i < 16; i++) v[i] = x0 + i;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {   // asm volatile keeps the exact order: MFMA, then K independent VALU
      if (SHAPE == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc16[i]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc32[i & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < K; j++) {
        float& x = v[(i * K + j) & 15];
        if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(x1), "v"(x0));
        else if (FILL == 1) { unsigned r; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(x1)); x = __uint_as_float(r); }
        else if (FILL == 2) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(x1));
        else asm volatile("v_accvgpr_write_b32 a255, %0" : : "v"(x) : "a255");
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) s += acc32[i][e];
  for (int i = 0; i < 8; i++) for (int e = 0; e < 4; e++) s += acc16[i][e];
  for (int i = 0; i < 16; i++) s += v[i];
  out[blockIdx.x * T + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int K, int SHAPE, int FILL = 0, int T = 256> void run_mf(float* d, unsigned long long* c) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mf<K, SHAPE, FILL, T>), dim3(256), dim3(T), 0, 0, d, c, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((mf<K, SHAPE, FILL, T>), dim3(256), dim3(T), 0, 0, d, c, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("[%.3f ms, %.0f TFLOP/s, %.2f GHz] ", ms, 256.0 * (T / 64) * iters * 8.0 * 32768.0 / (ms * 1e-3) / 1e12, (double)h / (ms * 1e-3) / 1e9);
  printf("%s  %d waves/SIMD %2d fillers (kind %d) per MFMA: %.1f cycles per MFMA (per wave)\n", SHAPE ? "32x32x16_f16" : "16x16x32_f16", T / 256, K, FILL, (double)h / (iters * 8.0));
}

// (b) every wave of the workgroup reads the same REGION bytes (one dwordx4 per lane = 1 KiB per instruction), U loads in flight
template <int U, bool SAME>
__global__ __launch_bounds__(256, 1) void l1(const f32x4* __restrict__ src, float* out, unsigned long long* cyc, int iters, int region_vec) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const f32x4* p = src + (size_t)blockIdx.x * region_vec * (SAME ? 1 : 4) + (SAME ? 0 : w * region_vec) + lane;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int off = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    f32x4 t[U];
#pragma unroll
    for (int u = 0; u < U; u++) { t[u] = p[off]; off += 64; if (off >= region_vec) off = 0; }
#pragma unroll
    for (int u = 0; u < U; u++) acc += t[u];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int U, bool SAME> void run_l1(const f32x4* s, float* d, unsigned long long* c, int region_bytes) {
  const int iters = 4000;
  hipLaunchKernelGGL((l1<U, SAME>), dim3(256), dim3(256), 0, 0, s, d, c, iters, region_bytes / 16);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("global dwordx4, 4 waves/CU, %s region %3d KiB, %d in flight: %.1f B/clk/CU delivered to registers\n",
         SAME ? "shared  " : "per-wave", region_bytes / 1024, U, 4.0 * iters * U * 1024.0 / (double)h);
}
int main() {
  float* d; unsigned long long* c; f32x4* s;
  hipMalloc(&d, 256 * 512 * 4); hipMalloc(&c, 8); hipMalloc(&s, 256u * 4 * 262144); hipMemset(s, 0, 256u * 4 * 262144);
  run_mf<0, 0>(d, c); run_mf<0, 0>(d, c); run_mf<1, 0>(d, c); run_mf<2, 0>(d, c); run_mf<3, 0>(d, c); run_mf<4, 0>(d, c); run_mf<6, 0>(d, c);
  run_mf<0, 1>(d, c); run_mf<2, 1>(d, c); run_mf<4, 1>(d, c); run_mf<5, 1>(d, c); run_mf<6, 1>(d, c); run_mf<7, 1>(d, c); run_mf<8, 1>(d, c); run_mf<12, 1>(d, c);
  run_mf<0, 1, 0, 512>(d, c); run_mf<4, 1, 0, 512>(d, c); run_mf<8, 1, 0, 512>(d, c); run_mf<12, 1, 0, 512>(d, c); run_mf<16, 1, 0, 512>(d, c);
  run_mf<0, 0, 0, 512>(d, c); run_mf<4, 0, 0, 512>(d, c); run_mf<8, 0, 0, 512>(d, c);
  run_mf<4, 1, 1>(d, c); run_mf<6, 1, 1>(d, c); run_mf<4, 1, 2>(d, c); run_mf<6, 1, 2>(d, c); run_mf<2, 0, 1>(d, c); run_mf<3, 0, 1>(d, c);
  run_l1<8, true>(s, d, c, 16384); run_l1<8, true>(s, d, c, 16384); run_l1<16, true>(s, d, c, 16384); run_l1<8, true>(s, d, c, 65536);
  run_l1<8, true>(s, d, c, 262144); run_l1<8, false>(s, d, c, 4096); run_l1<8, false>(s, d, c, 65536); run_l1<16, false>(s, d, c, 262144);
  return 0;
}
