// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, found by experiment: A[m][k] = 1 + m + 16 k (exact), B[k][n] = delta(k, kk) * (1 + n)
// for each kk, so D[m][n] = (1 + m + 16 kk)(1 + n) identifies (m, n) and confirms k = kk for every (lane, register).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, int kk, int which) {
  const int lane = threadIdx.x;
  // hypothesis: A lane -> (m = lane % 16, k = lane / 16); B lane -> (n = lane % 16, k = lane / 16)
  // which = 0: D[m][n] = A[m][kk] = 1 + m + 16 kk (names the row), which = 1: D[m][n] = B[kk][n] = 1 + n (names the column)
  const double a = which ? 1.0 : 1.0 + (lane % 16) + 16.0 * (lane / 16);
  const double b = (lane / 16 == kk) ? (which ? 1.0 + (lane % 16) : 1.0) : 0.0;
  f64x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; i++) out[lane * 4 + i] = c[i];
}
int main() {
  double* d; hipMalloc(&d, 256 * 8);
  double hm[256], hn[256];
  int ok = 1;
  for (int kk = 0; kk < 4; kk++) {
    probe<<<1, 64>>>(d, kk, 0);
    hipMemcpy(hm, d, sizeof hm, hipMemcpyDeviceToHost);
    probe<<<1, 64>>>(d, kk, 1);
    hipMemcpy(hn, d, sizeof hn, hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; lane++)
      for (int i = 0; i < 4; i++) {
        const int fm = (int)hm[lane * 4 + i] - 1 - 16 * kk, fn = (int)hn[lane * 4 + i] - 1;
        const int em = 4 * i + lane / 16, en = lane % 16;   // found on gfx950: row = 4 reg + lane / 16, col = lane % 16
        if (fm != em || fn != en) { ok = 0; if (kk == 0 && (lane % 16 == 1)) printf("lane %d reg %d: (m %d, n %d), expected (%d, %d)\n", lane, i, fm, fn, em, en); }
      }
  }
  printf(ok ? "layout confirmed: A (m = lane%%16, k = lane/16), B (n = lane%%16, k = lane/16), D reg i -> (row 4 i + lane/16, col lane%%16)\n" : "layout differs\n");
  return !ok;
}
