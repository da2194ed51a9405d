// Does s_barrier on gfx950 wait for the wave's outstanding memory operations?  (It must not, for a pipeline that keeps
// loads in flight across per-step barriers.)  One cold global load, then s_barrier, timed with s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* src, unsigned long long* out) {
  unsigned long long t0, t1, t2;
  unsigned v;
  const unsigned* p = src + (size_t)threadIdx.x * 4096 + blockIdx.x * 64;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  asm volatile("s_barrier" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2));
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t0; out[2] = v; }
}
int main() {
  unsigned* s; unsigned long long* o; (void)hipMalloc(&s, 256u * 4096 * 4 + 4096); (void)hipMalloc(&o, 64);
  (void)hipMemset(s, 0, 256u * 4096 * 4 + 4096);
  k<<<1, 256>>>(s, o); (void)hipDeviceSynchronize();
  unsigned long long h[3]; (void)hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
  printf("load issue -> after s_barrier: %llu ticks; -> after vmcnt(0): %llu ticks (s_memtime runs at 100 MHz: 1 tick = ~24 clk)\n", h[0], h[1]);
  return 0;
}
