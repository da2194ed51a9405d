// L1-hit throughput of raw buffer loads per CU on gfx950, by shape: does an out-of-range lane still cost a slot, and what do
// 12-byte (x3) and 8-byte loads cost?  4 waves per CU (one per SIMD) re-read a 12 KB working set (L1-resident).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* src, unsigned* out, int iters) {
  const int lane = threadIdx.x & 63;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, MODE == 1 ? 768 : 1024, 0x00020000);
  unsigned acc = 0;
  __shared__ u32x4 dmabuf[4 * 12 * 64];
  const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)dmabuf + (threadIdx.x >> 6) * 12 * 1024);
  int zero = 0;
  for (int it = 0; it < iters; it++) {
    asm volatile("" : "+s"(zero));                 // opaque: keeps the loads inside the loop
    const unsigned* srcit = src + zero;
#pragma unroll
    for (int t = 0; t < 12; t++) {
      __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)(srcit + t * 256), 0, MODE == 1 ? 768 : 1024, 0x00020000);
      if (MODE == 0) { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rt, lane * 16, 0, 0); acc += v[0] ^ v[3]; }          // 64 lanes x 16 B
      if (MODE == 1) { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rt, lane < 48 ? lane * 16 : (int)0x80000000, 0, 0); acc += v[0] ^ v[3]; }   // 48 in range + 16 out
      if (MODE == 2) { u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rt, lane * 12, 0, 0); acc += v[0] ^ v[2]; }           // 64 lanes x 12 B
      if (MODE == 3) { u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rt, lane * 8, 0, 0); acc += v[0] ^ v[1]; }            // 64 lanes x 8 B
      if (MODE == 4) {                                                                                                      // LDS-DMA, 64 lanes x 16 B
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(ldsb + t * 1024), "v"(lane * 16), "s"(rt) : "memory");
      }
    }
  }
  (void)r;
  if (MODE == 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += dmabuf[threadIdx.x][0]; }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE>
void run(const unsigned* s, unsigned* o, const char* what, double bytes) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  k<MODE><<<256, 256>>>(s, o, 10);
  (void)hipEventRecord(e0);
  k<MODE><<<256, 256>>>(s, o, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 12 * 4;          // load instructions per CU
  printf("%-34s %.1f ns per instruction per CU (%.1f clk at 2.4 GHz), %.1f useful B/clk/CU\n", what, ms * 1e6 / n, ms * 1e6 / n * 2.4, bytes / (ms * 1e6 / n * 2.4));
}
int main() {
  unsigned *s, *o; (void)hipMalloc(&s, 1 << 16); (void)hipMalloc(&o, 256 * 256 * 4);
  (void)hipMemset(s, 1, 1 << 16);
  run<0>(s, o, "x4, 64 lanes in range", 1024);
  run<1>(s, o, "x4, 48 in range + 16 out of range", 768);
  run<2>(s, o, "x3, 64 lanes", 768);
  run<3>(s, o, "x2, 64 lanes", 512);
  run<4>(s, o, "x4 LDS-DMA, 64 lanes", 1024);
  return 0;
}
