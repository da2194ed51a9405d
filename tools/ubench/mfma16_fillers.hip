// microbenchmark: what does ONE wave per SIMD pay for K filler instructions of a given kind behind each v_mfma_f32_16x16x32_f16 (or
// 32x32x16), with the MFMA result in ArchVGPRs or AccVGPRs?  (the stage-1 / stage-2 situation of sc_match_d / sc_match_e)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma16_fillers.hip -o tools/ubench/mfma16_fillers ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { F_ADD = 0, F_PKADD, F_CVTPK, F_FMAMIX, F_SWAP, F_ACCW, F_DSREAD, F_BUFLOAD, F_SNOP, F_MOV, NKIND };
static const char* NAMES[] = {"v_add_f32", "v_pk_add_f32", "v_cvt_pk_f16_f32", "v_fma_mix_f32", "v_permlane32_swap", "v_accvgpr_write",
                              "ds_read_b128", "buffer_load_b128", "s_nop 0", "v_mov_b32"};

template <int KIND>
__device__ __forceinline__ void filler(float (&v)[16], f32x2 (&p)[8], u32x4 (&ld)[4], int j, unsigned ldsaddr, __amdgpu_buffer_rsrc_t rs, int voff) {
  float& x = v[j & 15];
  float& y = v[(j + 7) & 15];
  if (KIND == F_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  if (KIND == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j & 7]) : "v"(p[(j + 3) & 7]));
  if (KIND == F_CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(v[(j + 3) & 15]));
  if (KIND == F_FMAMIX) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(x) : "v"(y), "v"(v[(j + 3) & 15]));
  if (KIND == F_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  if (KIND == F_ACCW) asm volatile("v_accvgpr_write_b32 a[250], %0" : : "v"(x) : "a250");
  if (KIND == F_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[j & 3]) : "v"(ldsaddr));
  if (KIND == F_BUFLOAD) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld[j & 3]) : "v"(voff), "s"(rs));
  if (KIND == F_SNOP) asm volatile("s_nop 0");
  if (KIND == F_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
}

// BIG = 0: 16x16x32 (4 registers), 1: 32x32x16 (16 registers); ACC = 0: result in ArchVGPRs, 1: AccVGPRs; CHAIN = 1: every MFMA accumulates
// onto its own previous result (as the three split products do), 0: fresh zero accumulator
template <int K, int KIND, int BIG, int ACC, int CHAIN>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, const char* buf, int iters) {
  __shared__ __attribute__((aligned(16))) char sh[8192];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) reinterpret_cast<float*>(sh)[i] = i * 1e-3f;
  __syncthreads();
  const unsigned ldsaddr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)sh + (threadIdx.x & 63) * 16;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(buf), 0, 65536, 0x00020000);
  const int voff = (threadIdx.x & 63) * 16;
  float v[16];
  f32x2 p[8];
  u32x4 ld[4];
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 1e-3f + i;
  for (int i = 0; i < 8; i++) p[i] = f32x2{v[i], v[i + 8]};
  for (int i = 0; i < 4; i++) ld[i] = u32x4{0u, 0u, 0u, 0u};
  u32x4 a = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a;
  f32x4 t4[8];
  f32x16 t16[4];
  for (int i = 0; i < 8; i++) t4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) t16[i][e] = 0.f;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      __builtin_amdgcn_sched_barrier(0);
      if (BIG == 0) {
        if (ACC == 0) { if (CHAIN) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(t4[i]) : "v"(a), "v"(b));
                        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(t4[i]) : "v"(a), "v"(b)); }
        else { if (CHAIN) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(t4[i]) : "v"(a), "v"(b));
               else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&a"(t4[i]) : "v"(a), "v"(b)); }
      } else {
        if (ACC == 0) { if (CHAIN) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(t16[i & 3]) : "v"(a), "v"(b));
                        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(t16[i & 3]) : "v"(a), "v"(b)); }
        else { if (CHAIN) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(t16[i & 3]) : "v"(a), "v"(b));
               else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&a"(t16[i & 3]) : "v"(a), "v"(b)); }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < K; j++) filler<KIND>(v, p, ld, i * K + j, ldsaddr, rs, voff);
      if ((KIND == F_DSREAD || KIND == F_BUFLOAD) && (i & 1) == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(8)");   // keep the queues bounded
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_nop 15\n\ts_nop 15");
  float s = 0;
  for (int i = 0; i < 8; i++) { f32x4 r = t4[i]; if (ACC) asm volatile("" : "+v"(r)); s += r[0] + r[3]; }
  for (int i = 0; i < 4; i++) { f32x16 r = t16[i]; if (ACC) asm volatile("" : "+v"(r)); s += r[0] + r[15]; }
  for (int i = 0; i < 16; i++) s += v[i];
  for (int i = 0; i < 8; i++) s += p[i][0] + p[i][1];
  for (int i = 0; i < 4; i++) s += __uint_as_float(ld[i][0]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static float* d; static unsigned long long* c; static char* buf;
template <int K, int KIND, int BIG, int ACC, int CHAIN> double run() {
  const int iters = 2000;
  hipLaunchKernelGGL((k<K, KIND, BIG, ACC, CHAIN>), dim3(256), dim3(256), 0, 0, d, c, buf, iters);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  return (double)h / (iters * 8.0);
}
template <int KIND, int BIG, int ACC, int CHAIN> void row() {
  printf("%-20s %s dst=%s %s : K=0 %5.1f  K=2 %5.1f  K=4 %5.1f  K=6 %5.1f  cycles per (MFMA + K fillers)\n", NAMES[KIND], BIG ? "32x32x16" : "16x16x32",
         ACC ? "acc " : "arch", CHAIN ? "chain" : "fresh", run<0, KIND, BIG, ACC, CHAIN>(), run<2, KIND, BIG, ACC, CHAIN>(), run<4, KIND, BIG, ACC, CHAIN>(),
         run<6, KIND, BIG, ACC, CHAIN>());
}
template <int BIG, int ACC, int CHAIN> void block() {
  row<F_ADD, BIG, ACC, CHAIN>(); row<F_PKADD, BIG, ACC, CHAIN>(); row<F_CVTPK, BIG, ACC, CHAIN>(); row<F_FMAMIX, BIG, ACC, CHAIN>();
  row<F_SWAP, BIG, ACC, CHAIN>(); row<F_ACCW, BIG, ACC, CHAIN>(); row<F_DSREAD, BIG, ACC, CHAIN>(); row<F_BUFLOAD, BIG, ACC, CHAIN>();
  row<F_SNOP, BIG, ACC, CHAIN>(); row<F_MOV, BIG, ACC, CHAIN>();
}
int main() {
  hipMalloc(&d, 256 * 256 * 4); hipMalloc(&c, 8); hipMalloc(&buf, 65536); hipMemset(buf, 0, 65536);
  block<0, 0, 0>();
  block<0, 1, 0>();
  block<0, 0, 1>();
  block<1, 0, 0>();
  block<1, 1, 0>();
  return 0;
}
