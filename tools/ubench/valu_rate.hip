// issue cost (cycles per instruction, one wave per SIMD, independent operands) of the VALU ops the split-f16 matcher uses
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a, b}, p1 = {b, a}, p2 = {a, a}, p3 = {b, b};
  unsigned long long msk = 0xffffffff00000000ull;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a)); }
    if (OP == 1) { asm volatile("v_cvt_pk_f16_f32 %0, %0, %8\n v_cvt_pk_f16_f32 %1, %1, %8\n v_cvt_pk_f16_f32 %2, %2, %8\n v_cvt_pk_f16_f32 %3, %3, %8\n v_cvt_pk_f16_f32 %4, %4, %8\n v_cvt_pk_f16_f32 %5, %5, %8\n v_cvt_pk_f16_f32 %6, %6, %8\n v_cvt_pk_f16_f32 %7, %7, %8" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a)); }
    if (OP == 2) { asm volatile("v_fma_mixlo_f16 %0, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %1, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %2, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %3, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %4, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %5, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %6, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %7, %8, -1.0, %9 op_sel_hi:[1,0,0]" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a)); }
    if (OP == 3) { asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0), "v"(p1)); }
    if (OP == 4) { asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)); }
    if (OP == 5) { asm volatile("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n v_accvgpr_read_b32 %4, a4\n v_accvgpr_read_b32 %5, a5\n v_accvgpr_read_b32 %6, a6\n v_accvgpr_read_b32 %7, a7" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7"); }
    if (OP == 6) { asm volatile("v_add_f32 %0, %0, |%8|\n v_max3_f32 %1, %1, %8, %9\n v_add_f32 %2, %2, |%8|\n v_max3_f32 %3, %3, %8, %9\n v_add_f32 %4, %4, |%8|\n v_max3_f32 %5, %5, %8, %9\n v_add_f32 %6, %6, |%8|\n v_max3_f32 %7, %7, %8, %9" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a)); }
    if (OP == 7) { asm volatile("v_fma_mix_f32 %0, %8, -1.0, %9 op_sel_hi:[1,0,0] \n v_fma_mix_f32 %1, %8, -1.0, %9 op_sel_hi:[1,0,0] \n v_fma_mix_f32 %2, %8, -1.0, %9 op_sel_hi:[1,0,0] \n v_fma_mix_f32 %3, %8, -1.0, %9 op_sel_hi:[1,0,0] \n v_fma_mix_f32 %4, %8, -1.0, %9 op_sel_hi:[1,0,0] \n v_fma_mix_f32 %5, %8, -1.0, %9 op_sel_hi:[1,0,0] \n v_fma_mix_f32 %6, %8, -1.0, %9 op_sel_hi:[1,0,0] \n v_fma_mix_f32 %7, %8, -1.0, %9 op_sel_hi:[1,0,0]" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a), "s"(msk)); }
    if (OP == 8) { asm volatile("v_cndmask_b32 %0, %0, %8, %10 \n v_cndmask_b32 %1, %1, %8, %10 \n v_cndmask_b32 %2, %2, %8, %10 \n v_cndmask_b32 %3, %3, %8, %10 \n v_cndmask_b32 %4, %4, %8, %10 \n v_cndmask_b32 %5, %5, %8, %10 \n v_cndmask_b32 %6, %6, %8, %10 \n v_cndmask_b32 %7, %7, %8, %10" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a), "s"(msk)); }
    if (OP == 9) { asm volatile("v_cvt_f32_f16 %0, %8 \n v_cvt_f32_f16 %1, %8 \n v_cvt_f32_f16 %2, %8 \n v_cvt_f32_f16 %3, %8 \n v_cvt_f32_f16 %4, %8 \n v_cvt_f32_f16 %5, %8 \n v_cvt_f32_f16 %6, %8 \n v_cvt_f32_f16 %7, %8" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a), "s"(msk)); }
    if (OP == 10) { asm volatile("v_and_b32 %0, %0, %8 \n v_and_b32 %1, %1, %8 \n v_and_b32 %2, %2, %8 \n v_and_b32 %3, %3, %8 \n v_and_b32 %4, %4, %8 \n v_and_b32 %5, %5, %8 \n v_and_b32 %6, %6, %8 \n v_and_b32 %7, %7, %8" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a), "s"(msk)); }
    if (OP == 11) { asm volatile("v_perm_b32 %0, %0, %8, %9 \n v_perm_b32 %1, %1, %8, %9 \n v_perm_b32 %2, %2, %8, %9 \n v_perm_b32 %3, %3, %8, %9 \n v_perm_b32 %4, %4, %8, %9 \n v_perm_b32 %5, %5, %8, %9 \n v_perm_b32 %6, %6, %8, %9 \n v_perm_b32 %7, %7, %8, %9" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a), "s"(msk)); }
    if (OP == 12) { asm volatile("v_mov_b32_dpp %0, %8 row_ror:8 row_mask:0xf bank_mask:0xf \n v_mov_b32_dpp %1, %8 row_ror:8 row_mask:0xf bank_mask:0xf \n v_mov_b32_dpp %2, %8 row_ror:8 row_mask:0xf bank_mask:0xf \n v_mov_b32_dpp %3, %8 row_ror:8 row_mask:0xf bank_mask:0xf \n v_mov_b32_dpp %4, %8 row_ror:8 row_mask:0xf bank_mask:0xf \n v_mov_b32_dpp %5, %8 row_ror:8 row_mask:0xf bank_mask:0xf \n v_mov_b32_dpp %6, %8 row_ror:8 row_mask:0xf bank_mask:0xf \n v_mov_b32_dpp %7, %8 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a), "s"(msk)); }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + p0[0] + p1[0] + p2[1] + p3[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, float* d, unsigned long long* c) {
  const int iters = 4000;
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256), 0, 0, d, c, iters);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-22s %.2f cycles per instruction\n", name, (double)h / (iters * 8.0));
}
int main() {
  float* d; unsigned long long* c; hipMalloc(&d, 256 * 256 * 4); hipMalloc(&c, 8);
  run<0>("v_fma_f32", d, c); run<0>("v_fma_f32", d, c); run<1>("v_cvt_pk_f16_f32", d, c); run<2>("v_fma_mixlo_f16", d, c);
  run<3>("v_pk_fma_f32", d, c); run<4>("v_permlane32_swap", d, c); run<5>("v_accvgpr_read_b32", d, c); run<6>("v_add|abs| / v_max3", d, c);
  run<7>("v_fma_mix_f32", d, c); run<8>("v_cndmask_b32 (sgpr)", d, c); run<9>("v_cvt_f32_f16", d, c); run<10>("v_and_b32", d, c); run<11>("v_perm_b32", d, c); run<12>("v_mov_b32_dpp", d, c);
  return 0;
}
