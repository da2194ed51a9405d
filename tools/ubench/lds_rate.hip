// LDS read throughput per CU on gfx950 by width: 4 waves (one per SIMD) read a conflict-free linear image.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(unsigned* out, int iters) {
  __shared__ u32x4 buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 64 * WAVES) buf[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned acc = 0;
  int zero = 0;
  for (int it = 0; it < iters; it++) {
    asm volatile("" : "+s"(zero));
    const u32x4* b = buf + zero + w * 64 * 12;
#pragma unroll
    for (int t = 0; t < 12; t++) {
      if (MODE == 0) { const u32x4 v = b[t * 64 + lane]; acc += v[0] ^ v[3]; }
      if (MODE == 1) { const u32x2 v = reinterpret_cast<const u32x2*>(b)[t * 64 + lane]; acc += v[0] ^ v[1]; }
      if (MODE == 2) { const unsigned v = reinterpret_cast<const unsigned*>(b)[t * 64 + lane]; acc += v; }
    }
  }
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = acc;
}
template <int MODE, int WAVES>
void run(unsigned* o, const char* what, double bytes) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  k<MODE, WAVES><<<256, 64 * WAVES>>>(o, 10);
  (void)hipEventRecord(e0);
  k<MODE, WAVES><<<256, 64 * WAVES>>>(o, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 12 * WAVES;
  printf("%-28s %d waves/CU: %.1f clk per instruction per CU at 2.4 GHz, %.1f B/clk/CU\n", what, WAVES, ms * 1e6 / n * 2.4, bytes / (ms * 1e6 / n * 2.4));
}
int main() {
  unsigned* o; (void)hipMalloc(&o, 256 * 512 * 4);
  run<0, 4>(o, "ds_read_b128", 1024); run<0, 8>(o, "ds_read_b128", 1024);
  run<1, 4>(o, "ds_read_b64", 512);  run<1, 8>(o, "ds_read_b64", 512);
  run<2, 4>(o, "ds_read_b32", 256);  run<2, 8>(o, "ds_read_b32", 256);
  return 0;
}
