// does the raw-buffer range check include soffset?  (gfx950)  expect: lanes < 48 read base+soffset+16*lane, lanes >= 48 get zeros
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 768, 0x00020000);
  const int lane = threadIdx.x;
  for (int t = 0; t < 4; t++) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, t * 768, 0);
    for (int e = 0; e < 4; e++) out[(t * 64 + lane) * 4 + e] = v[e];
  }
}
int main() {
  unsigned *s, *o; hipMalloc(&s, 16384); hipMalloc(&o, 4096);
  unsigned h[4096]; for (int i = 0; i < 4096; i++) h[i] = i + 1;
  hipMemcpy(s, h, 16384, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, o); hipDeviceSynchronize();
  unsigned r[1024]; hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 4; t++) for (int l = 0; l < 64; l++) for (int e = 0; e < 4; e++) {
    unsigned want = l < 48 ? (unsigned)(t * 192 + l * 4 + e + 1) : 0u;
    if (r[(t * 64 + l) * 4 + e] != want) { if (bad < 8) printf("t=%d lane=%d e=%d got %u want %u\n", t, l, e, r[(t * 64 + l) * 4 + e], want); bad++; }
  }
  printf("soffset excluded from range check, OOB lanes zero: %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
  return 0;
}
