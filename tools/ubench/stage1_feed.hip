// microbenchmark: the operand delivery of one stage-1 walk position of the split-f16 SC matcher (6 x v_mfma_f32_16x16x32_f16 + ~24 VALU per
// wave, one wave per SIMD, four waves per workgroup) under different delivery schemes.  cycles per position per wave:
//   NV vector loads (buffer_load_dwordx4, 48 valid lanes; the four waves request the SAME tiles of a 95 KB group image, as the kernel does)
//   NL LDS reads (ds_read_b128)     NWR LDS writes (ds_write_b128, 48 lanes)     BAR: one s_barrier per position
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/stage1_feed.hip -o tools/ubench/stage1_feed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NL, int NWR, int BAR, int NVALU, int DISTINCT>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, const char* buf, int units) {
  extern __shared__ __attribute__((aligned(16))) char sh[];
  for (int i = threadIdx.x; i < 40960; i += blockDim.x) reinterpret_cast<float*>(sh)[i] = i * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)sh;
  const unsigned rd_addr = lds0 + w * 39936 + lane * 16;            // this wave's query image
  const unsigned wr_addr = lane < 48 ? lds0 + 159744 + w * 768 + lane * 16 : 0xffffffffu;
  const int voff = lane < 48 ? lane * 16 : (int)0x80000000;
  float v[16];
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 1e-3f + i;
  u32x4 ld[16];
  for (int i = 0; i < 16; i++) ld[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  f32x4 t[4];
  for (int i = 0; i < 4; i++) t[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int u = 0; u < units; u++) {
    // a fresh 95 KB group image per unit (L2-resident after the first touches: 64 groups cycle through 6 MB); DISTINCT: every wave its own
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(buf) + (size_t)((u * (DISTINCT ? 4 : 1) + (DISTINCT ? w : 0)) & 63) * 95232, 0, 95232, 0x00020000);
#pragma unroll 1
    for (int p2 = 0; p2 < 32; p2 += 2) {
#pragma unroll
      for (int par = 0; par < 2; par++) {          // operands double-buffered by position parity: requested one position ahead of their use
        const int p = p2 + par;
        const int so = (p < 31 ? p : 30) * 3072;
#pragma unroll
        for (int g = 0; g < 6; g++) {
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(t[g & 3]) : "v"(ld[8 * par + (g & 7)]), "v"(ld[8 * par + ((g + 3) & 7)]));
          __builtin_amdgcn_sched_barrier(0);
          if (g < NV) ld[8 * (par ^ 1) + (g & 7)] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so + g * 768, 0);
          if (g >= 6 - NL) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[8 * (par ^ 1) + ((g + 2) & 7)]) : "v"(rd_addr + ((p * 1288 + g * 64) & 0x7ff0)));
          if (g < NWR) asm volatile("ds_write_b128 %0, %1" : : "v"(wr_addr), "v"(ld[8 * par + ((g + 5) & 7)]));
#pragma unroll
          for (int j = 0; j < NVALU / 6; j++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(g * 4 + j) & 15]) : "v"(v[(g * 4 + j + 7) & 15]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NL > 0) asm volatile("s_waitcnt lgkmcnt(0)");     // the asm LDS reads are invisible to hipcc's counters
        if (BAR) asm volatile("s_barrier");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_nop 15\n\ts_nop 15");
  float s = 0;
  for (int i = 0; i < 4; i++) s += t[i][0] + t[i][3];
  for (int i = 0; i < 16; i++) s += v[i];
  for (int i = 0; i < 16; i++) s += __uint_as_float(ld[i][0]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static float* d; static unsigned long long* c; static char* buf;
template <int NV, int NL, int NWR, int BAR, int NVALU, int DISTINCT = 0> void run(const char* what) {
  const int units = 200;
  auto kern = k<NV, NL, NWR, BAR, NVALU, DISTINCT>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 163840, 0, d, c, buf, units);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-64s NV=%d NL=%d NWR=%d BAR=%d VALU=%2d : %6.1f cycles per position\n", what, NV, NL, NWR, BAR, NVALU, (double)h / (units * 32.0));
}
int main() {
  hipMalloc(&d, 256 * 256 * 4); hipMalloc(&c, 8); hipMalloc(&buf, 64 * 95232 + 4096); hipMemset(buf, 0, 64 * 95232 + 4096);
  run<0, 0, 0, 0, 0>("MFMAs only");
  run<0, 0, 0, 0, 24>("MFMAs + 24 VALU");
  run<4, 4, 0, 0, 24>("sc_match_d: 4 vector loads + 4 LDS reads");
  run<4, 2, 0, 0, 24>("sc_match_e: 4 vector loads + 2 LDS reads");
  run<4, 2, 0, 0, 24, 1>("the same, every wave its own group (no L1 sharing)");
  run<4, 0, 0, 0, 24>("4 vector loads only");
  run<2, 4, 0, 0, 24>("roles exchanged: 2 vector loads + 4 LDS reads");
  run<2, 2, 0, 0, 24>("2 vector loads + 2 LDS reads");
  run<1, 6, 1, 1, 24>("DB tiles through LDS: 1 vector load + 1 LDS write + 6 LDS reads + barrier");
  run<1, 6, 1, 0, 24>("the same without the barrier");
  run<1, 6, 0, 0, 24>("the same without the write");
  run<0, 6, 0, 0, 24>("6 LDS reads");
  run<0, 2, 0, 0, 24>("2 LDS reads");
  run<4, 2, 0, 0, 0>("sc_match_e delivery without VALU");
  run<1, 6, 1, 1, 0>("LDS delivery without VALU");
  return 0;
}
