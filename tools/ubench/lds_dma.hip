// buffer_load_dwordx4 ... lds (LDS-DMA) on gfx950: where do the lanes land, and what do out-of-range lanes do?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned* out) {
  __shared__ u32x4 buf[256];   // 4 KB
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) buf[i] = u32x4{0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 768, 0x00020000);
  const unsigned ldsbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)&buf[64];   // second KB
  const int voff = lane < 48 ? lane * 16 : (int)0x80000000;
  asm volatile("s_mov_b32 m0, %0\n\t"
               "s_nop 0\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
               "s_waitcnt vmcnt(0)"
               : : "s"(ldsbase), "v"(voff), "s"(r) : "memory", "m0");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) { const u32x4 v = buf[i]; for (int e = 0; e < 4; e++) out[i * 4 + e] = v[e]; }
}
int main() {
  unsigned *s, *o; (void)hipMalloc(&s, 4096); (void)hipMalloc(&o, 4096);
  unsigned h[1024]; for (int i = 0; i < 1024; i++) h[i] = i + 1;
  (void)hipMemcpy(s, h, 4096, hipMemcpyHostToDevice);
  k<<<1, 64>>>(s, o); (void)hipDeviceSynchronize();
  unsigned r[1024]; (void)hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
  int linear = 1, oob_zero = 1, oob_untouched = 1, others = 1;
  for (int i = 0; i < 1024; i++) {
    if (i >= 256 && i < 256 + 192) { if (r[i] != (unsigned)(i - 256 + 1)) linear = 0; }
    else if (i >= 256 + 192 && i < 512) { if (r[i] != 0) oob_zero = 0; if (r[i] != 0xdeadbeefu) oob_untouched = 0; }
    else if (r[i] != 0xdeadbeefu) others = 0;
  }
  printf("lanes 0-47 land linearly at M0 + 16 lane: %s; out-of-range lanes 48-63 write zeros: %s / leave LDS untouched: %s; rest untouched: %s\n",
         linear ? "YES" : "NO", oob_zero ? "YES" : "NO", oob_untouched ? "YES" : "NO", others ? "YES" : "NO");
  if (!linear) for (int i = 250; i < 280; i++) printf("%d:%x ", i, r[i]);
  return 0;
}
