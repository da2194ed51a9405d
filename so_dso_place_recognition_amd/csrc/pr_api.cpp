// pr_api.cpp — host side of libpr_amd.so: context, packed signature sets and the C ABI declared in
// include/place_recognition.h.  No CPU fallback anywhere: without a usable gfx950 device every call fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/place_recognition.h"
#include "../../include/pr_m2dp_table.h"
#include "kernels.hpp"
#include "hash_order.hpp"
#include "records.hpp"
#include <chrono>
#include <unordered_map>

struct pr_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  bool own_stream = true;        // false: pr_create_on_stream (the caller's stream)
  int nan_policy = PR_NAN_EXCLUDE;
  int warnings = 0;              // PR_WARN_* bits not yet taken
  hipStream_t side = nullptr;    // overlaps the sequential float-average chain with the moments pass
  hipStream_t side2 = nullptr;   // SC generation: odd batches (kernel boundaries of one stream hide behind the other's kernels)
  hipEvent_t ev_b = nullptr;
  char* sc_scratch = nullptr;    // tickets + partial moments / bin grids of the split SC generation, one half per stream
  void* sel_scratch = nullptr;   // slice lists / partial moments of calls with few query rows (fuse_select.hip)
  double* rr_scratch = nullptr;  // candidate scores of pr_rerank_dev (grow-only: a steady-state call allocates nothing and can be graph-captured)
  size_t rr_cap = 0;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_m2[4] = {nullptr, nullptr, nullptr, nullptr};   // M2DP generation: two batches in flight (binning | singular pairs), launch_m2dp_bin_svd
  std::string err;
  int* d_svd_rows = nullptr;     // [1 + M2DP_SVD_ROWS_CAP] rows of the last pr_m2dp_generate* call whose leading singular pair did not converge
  int* d_flags = nullptr;        // [4] deferred bits: [0] zero-norm row at pack time, [1] M2DP singular pair not converged, [2] a query was answered with fp64 row statistics, [3] more flagged queries than one stream-ordered pass resolves
  bool sc_online_h = false;      // PR_SC_ONLINE=h: calls of up to 8 queries through sc_match_h.hip's one-group form (the default until round 5)
  double* d_twiddle = nullptr;   // cos[60], sin[60] of 2*pi*t/60
  float* d_cst = nullptr;        // SC stage-2 constants [31][2][64]
  int32_t* d_margin = nullptr;    // [1] count of margin flags (PR_SC_ARITH_F16)
  int32_t* d_order = nullptr;     // [2 order_cap + 2] order flags of the last re-evaluation | ascending list of the flagged queries | its length (pr_order_resolve*_dev / pr_f16_margin_dev take them)
  size_t order_cap = 0;
  int32_t order_m = -1;           // rows of d_order that are valid, -1: none
  unsigned* rr_tick = nullptr;    // [tick_cap] per-pair tickets of rerank_kernel (zero between launches)
  size_t tick_cap = 0;
  double* res_partial = nullptr;  // [RESOLVE_SLOTS][RESOLVE_NB][4][3] workgroup partials of the fp64-statistics resolution (rerank.hip), allocated on first use
  double* res_exact = nullptr;    // [res_exact_cap][4][3] exact row moments of the flagged queries (single-shard calls)
  size_t res_exact_cap = 0;
  double* xrows = nullptr;        // [min(m, RESOLVE_SLOTS)][4][n_local] exact rows of the flagged queries of one pass (exact_row.hip), grow-only
  size_t xrows_cap = 0;
  double* xqspec = nullptr;       // query spectra of one pass's slots (exact_row.hip), allocated with res_partial
  double* xpart = nullptr;        // the row slices' lists of the exact-row selection, grow-only
  size_t xpart_cap = 0;
  bool xrow_direct = false;       // PR_XROW=direct: exact rows in the reference's own formulation (tests: cross-check of the spectral form)
  void* d_cst_h = nullptr;       // split-f16 stage-2 constants [E|O][half][hi|lo][64 lanes][8 f16] (sc_match_h.hip)
  int sc_kernel = 2;             // split-f16 SC matcher for m > 8: 2 = sc_match_e.hip (default); PR_SC_KERNEL=h selects sc_match_h.hip (0, round 1; always the kernel for m <= 8)
  int sc_mode = PR_SC_ARITH_F16X2;   // PR_SC_ARITH_*: split-f16 MFMA (sc_match_h.hip) | fp32 MFMA (sc_match.hip); PR_SC_MATCH=f32 selects the latter
  double* d_planes = nullptr;    // M2DP xProj[64][3], yProj[64][3]
  int sc_nsplit = 0;             // PR_SC_NSPLIT override (experiments)
  bool force_order = false;      // PR_FORCE_ORDER_FLAGS=1 (tests): every query counts as flagged by the order check, i.e. every query gets fp64 row statistics
  bool sc_binary = true;         // split-f16 arithmetic: a binary intensity channel goes through the single-product kernel with integer rounding (kernels.hpp: ScBin); PR_SC_BINARY=0 / pr_set_sc_binary turn it off
  float sc_bconst = 0.f;         // (u + gamma)(1 + u) + slack of that pass's error bound (create_common)
  int bin_gen = 0;               // sequence number of the binary-path calls: what the single-product pass leaves in d_flags[4] when a pair fails its test
  float sc_pair_scale = 1.f;     // PR_SC_BINARY_PAIR_SCALE (tests): inflates the bound of the per-pair rounding test, which then fails and sends channel 1 to the split-f16 pass behind
  bool timing = false;           // pr_set_kernel_timing: events around the launches of pr_distances_dev
  hipEvent_t ev_t[4] = {nullptr, nullptr, nullptr, nullptr};
  int timing_valid = 0;          // 0: nothing recorded; 1: ev_t[0], ev_t[3] only (one launch); 3: all four (channel 0 | channel 1 split | channel 1 single)
};

struct pr_sigset {
  int type = 0, role = 0;
  int32_t max_sigs = 0, count = 0;
  int groups = 0;                // SC: 8-query or 16-entry groups; M2DP: 32-row tiles (channel stride)
  int sc_mode = 0;               // SC: arithmetic the image was packed for (pr_ctx::sc_mode at creation)
  float* packed = nullptr;
  size_t floats = 0;
  int* bad = nullptr;            // SC: [max_sigs + 1][2], entry [row][c] = 1 << c when channel c of that row has zero norm (NaN row in MATLAB, processSC.m:16,19), else 0; every pack writes all of its rows' entries
  float* binfo = nullptr;        // SC split-f16 sets: [max_sigs + 1][2] {sqrt(ones), 1/sqrt(ones)} of channel 1, then SC_BSTAT_INTS ints of set statistics (kernels.hpp: ScBin)
  int32_t hw = 0;                // rows ever written since the image was last all-zero, and the group count (= channel stride) they were written for
  int packed_groups = -1;        // -1: the image is all-zero
  bool cap_geom = false;         // the channel stride is the CAPACITY's group count whatever the row count (pr_sigset_reserve: what pr_sigset_append needs)
};

static thread_local std::string g_err;   // errors without a context

#define PR_FAIL(ctx, code, ...)                                   \
  do {                                                            \
    char _b[512];                                                 \
    snprintf(_b, sizeof _b, __VA_ARGS__);                         \
    if (ctx) (ctx)->err = _b; else g_err = _b;                    \
    return (code);                                                \
  } while (0)

#define PR_HIP(ctx, call)                                                                     \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) PR_FAIL(ctx, (_e == hipErrorOutOfMemory ? PR_ENOMEM : PR_EHIP),     \
                                  "%s failed: %s", #call, hipGetErrorString(_e));            \
  } while (0)

namespace {

// RAII device scratch for the host-buffer entry points.  An error return may leave work in flight that still touches a buffer - on the
// context's stream or on a side stream it forked: the streams of the context whose call owns the buffer (DevScope, first thing in every
// such call) are idle before it goes.  Only those: a device-wide synchronisation would stall unrelated streams of the process (torch's,
// RCCL's) and is illegal while another thread captures a graph in global mode.
thread_local pr_ctx* g_scope_ctx = nullptr;
struct DevScope {
  pr_ctx* prev;
  explicit DevScope(pr_ctx* c) : prev(g_scope_ctx) { g_scope_ctx = c; }
  ~DevScope() { g_scope_ctx = prev; }
};
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (!p) return;
    if (pr_ctx* c = g_scope_ctx) {
      (void)hipStreamSynchronize(c->stream);
      if (c->side) (void)hipStreamSynchronize(c->side);
      if (c->side2) (void)hipStreamSynchronize(c->side2);
    } else (void)hipDeviceSynchronize();
    (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <typename T> T* as() { return static_cast<T*>(p); }
};

int set_device(pr_ctx* ctx) {
  PR_HIP(ctx, hipSetDevice(ctx->device));
  return PR_OK;
}

size_t sigset_floats(int type, int role, int32_t max_sigs, int* groups, int sc_mode) {
  if (type == PR_TYPE_DELIGHT) { *groups = max_sigs; return (size_t)max_sigs * (4096 + 128) + 16; }   // histograms + empty-bin masks
  if (type == PR_TYPE_SC && sc_mode == PR_SC_ARITH_F16) {   // single-product images (hi halves only), sizes in bytes / 4
    if (role == PR_ROLE_QUERY) { *groups = pr::sc_qgroups8_f16(max_sigs); return (size_t)2 * *groups * pr::SCF_QIMG / 4; }
    *groups = pr::sc_dgroups(max_sigs);
    return (size_t)(2 * *groups + 9) * pr::SCF_DIMG / 4;
  }
  if (type == PR_TYPE_SC && sc_mode == 0) {   // split-f16 images, sizes in bytes / 4
    if (role == PR_ROLE_QUERY) { *groups = pr::sc_qgroups8(max_sigs); return (size_t)2 * *groups * pr::SCH_QIMG / 4; }
    *groups = pr::sc_dgroups(max_sigs);
    return (size_t)(2 * *groups + 9) * pr::SCH_DIMG / 4;   // + all-zero groups: a wave's pipeline runs one unit past the end (4 groups ahead for m <= 8, + its prefetch)
  }
  if (type == PR_TYPE_SC) {
    if (role == PR_ROLE_QUERY) { *groups = pr::sc_qgroups8(max_sigs); return (size_t)2 * *groups * pr::SC_QIMG; }
    *groups = pr::sc_dgroups(max_sigs);
    return (size_t)2 * *groups * pr::SC_DIMG + 16 * pr::SC_DSTEP;  // + zero tail of 8 slots read past the last group (sc_match.hip)
  }
  if (role == PR_ROLE_QUERY) { *groups = pr::m2_qtiles(max_sigs); return (size_t)2 * *groups * pr::M2_TILE; }
  *groups = pr::m2_tiles(max_sigs);
  return (size_t)(2 * *groups + 16) * pr::M2_TILE;                 // + tail tiles read by the last sweep step (8 tiles per step + prefetch)
}

int check_flags(pr_ctx* ctx) {
  int h[4];
  PR_HIP(ctx, hipMemcpyAsync(h, ctx->d_flags, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h[0] || h[1] || h[2] || h[3]) PR_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, sizeof h, ctx->stream));
  if (h[1]) ctx->warnings |= PR_WARN_M2DP_SVD;
  if (h[2]) ctx->warnings |= PR_WARN_ORDER_RESOLVED;
  if (h[3]) ctx->warnings |= PR_WARN_ORDER_UNRESOLVED;
  if (h[0]) {
    if (ctx->nan_policy == PR_NAN_FAIL)
      PR_FAIL(ctx, PR_ENAN, "a signature row has zero L2 norm (MATLAB would produce NaN distances, processSC.m:16,19)");
    ctx->warnings |= PR_WARN_NAN_ROWS;
  }
  return PR_OK;
}

}  // namespace

extern "C" {

const char* pr_version(void) { return "so_dso_place_recognition_amd 0.1 (gfx950)"; }

const char* pr_last_error(const pr_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

static int create_common(int device_id, hipStream_t external, bool use_external, pr_ctx** out) {
  if (!out) PR_FAIL((pr_ctx*)nullptr, PR_EINVAL, "pr_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    PR_FAIL((pr_ctx*)nullptr, PR_EHIP, "pr_create: no HIP device available (%s); this library has no CPU fallback",
            e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev) PR_FAIL((pr_ctx*)nullptr, PR_EINVAL, "pr_create: device %d out of range [0,%d)", device_id, ndev);
  hipDeviceProp_t prop;
  PR_HIP((pr_ctx*)nullptr, hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    PR_FAIL((pr_ctx*)nullptr, PR_EHIP, "pr_create: device %d is %s, this library is built for gfx950 only", device_id, prop.gcnArchName);
  pr_ctx* ctx = new (std::nothrow) pr_ctx;
  if (!ctx) PR_FAIL((pr_ctx*)nullptr, PR_ENOMEM, "pr_create: out of host memory");
  ctx->device = device_id;
  int rc = PR_OK;
  do {
#define TRY(call) if ((call) != hipSuccess) { g_err = std::string(#call) + " failed"; rc = PR_EHIP; break; }
    TRY(hipSetDevice(device_id));
    if (use_external) { ctx->stream = external; ctx->own_stream = false; }
    else TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    {  // (highest priority: the few waves of the average chain are dispatched ahead of the streaming passes they run beside)
      int plo = 0, phi = 0;
      TRY(hipDeviceGetStreamPriorityRange(&plo, &phi));
      TRY(hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, phi));
    }
    TRY(hipStreamCreateWithFlags(&ctx->side2, hipStreamNonBlocking));
    TRY(hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming));
    TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    for (auto& e : ctx->ev_m2) TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    TRY(hipMalloc(&ctx->sel_scratch, pr::select_scratch_bytes()));
    TRY(hipMalloc((void**)&ctx->d_flags, 8 * sizeof(int)));        // [4]: the binary-channel pass's violation flag (kernels.hpp: ScBin), not a deferred bit
    TRY(hipMemset(ctx->d_flags, 0, 8 * sizeof(int)));
    TRY(hipMalloc((void**)&ctx->d_svd_rows, (1 + pr::M2DP_SVD_ROWS_CAP) * sizeof(int)));
    TRY(hipMemset(ctx->d_svd_rows, 0, (1 + pr::M2DP_SVD_ROWS_CAP) * sizeof(int)));
    double tw[120 + 4 * 60 * 8];
    for (int t = 0; t < 60; t++) { tw[t] = std::cos(2.0 * M_PI * t / 60.0); tw[60 + t] = std::sin(2.0 * M_PI * t / 60.0); }
    // exact values at the multiples of 90 degrees
    tw[0] = 1; tw[15] = 0; tw[30] = -1; tw[45] = 0; tw[60] = 0; tw[75] = 1; tw[90] = 0; tw[105] = -1;
    // + the same values regrouped for sc_pack_h_col_kernel: [f block 0..3][sector 0..59][j 0..3]{cos, sin} of f = 4 fb + j
    // (64 contiguous bytes per sector: one scalar load brings the 8 twiddles a workgroup needs for it)
    for (int fb = 0; fb < 4; fb++)
      for (int sct = 0; sct < 60; sct++)
        for (int j = 0; j < 4; j++) {
          const int t = ((4 * fb + j) * sct) % 60;
          tw[120 + ((fb * 60 + sct) * 4 + j) * 2] = tw[t];
          tw[120 + ((fb * 60 + sct) * 4 + j) * 2 + 1] = tw[60 + t];
        }
    TRY(hipMalloc((void**)&ctx->d_twiddle, sizeof tw));
    TRY(hipMemcpy(ctx->d_twiddle, tw, sizeof tw, hipMemcpyHostToDevice));
    TRY(pr::xrow_set_twiddles(tw, tw + 60));
    // stage-2 constants per slot, A operand of v_mfma_f32_32x32x2_f32: lane l -> shift k = l & 31 (k = 31 repeats
    // shift 0, harmless for the max), K index = l >> 5.  Slots (fa, fb) = (0,30), (1,2), ..., (27,28): K = {fa, fb},
    // ce = w_f cos(2 pi f k/60), co = -w_f sin(2 pi f k/60), w_0 = w_30 = 1, else 2.  Slot 15 = frequency 29 alone,
    // applied to the unswapped (Re | Im) tile: ce = (w cos | 0), co = (0 | -w sin).
    std::vector<float> cst((size_t)pr::SC_NSLOT * 128);
    for (int sl = 0; sl < pr::SC_NSLOT; sl++)
      for (int l = 0; l < 64; l++) {
        const int half = l >> 5;
        int k = l & 31; if (k > 30) k = 0;
        const int f = (sl == 0) ? (half ? 30 : 0) : (sl == pr::SC_NSLOT - 1 ? 29 : 2 * sl - 1 + half);
        const int t = (f * k) % 60;
        const double w = (f == 0 || f == 30) ? 1.0 : 2.0;
        float ce = (float)(w * tw[t]), co = (float)(-w * tw[60 + t]);
        if (sl == pr::SC_NSLOT - 1) { if (half) ce = 0.f; else co = 0.f; }
        cst[(size_t)sl * 128 + l] = ce;
        cst[(size_t)sl * 128 + 64 + l] = co;
      }
    TRY(hipMalloc((void**)&ctx->d_cst, cst.size() * sizeof(float)));
    TRY(hipMemcpy(ctx->d_cst, cst.data(), cst.size() * sizeof(float), hipMemcpyHostToDevice));
    // split-f16 stage-2 constants (sc_match_h.hip): A operand of v_mfma_f32_32x32x16_f16, lane l -> shift k = min(l & 31, 30)
    // (row 31 repeats shift 30, harmless for the max), K index = 8*(l >> 5) + e -> frequency f = 16*half + K (f = 31: 0),
    // value = w_f cos(2 pi f k/60) (E) or -w_f sin(2 pi f k/60) (O), scaled by 2^10 and split into hi + lo
    {
      std::vector<_Float16> ch((size_t)8 * 64 * 8);
      for (int eo = 0; eo < 2; eo++)
        for (int half = 0; half < 2; half++)
          for (int l = 0; l < 64; l++)
            for (int e = 0; e < 8; e++) {
              int k = l & 31; if (k > 30) k = 30;
              const int f = 16 * half + 8 * (l >> 5) + e;
              double v = 0.0;
              if (f < pr::SC_NF) {
                const int t = (f * k) % 60;
                const double w = (f == 0 || f == 30) ? 1.0 : 2.0;
                v = (eo == 0 ? w * tw[t] : -w * tw[60 + t]) * 1024.0;
              }
              const _Float16 hi = (_Float16)v;
              const _Float16 lo = (_Float16)(v - (double)hi);
              ch[((((size_t)eo * 2 + half) * 2 + 0) * 64 + l) * 8 + e] = hi;
              ch[((((size_t)eo * 2 + half) * 2 + 1) * 64 + l) * 8 + e] = lo;
            }
      TRY(hipMalloc(&ctx->d_cst_h, ch.size() * sizeof(_Float16)));
      TRY(hipMemcpy(ctx->d_cst_h, ch.data(), ch.size() * sizeof(_Float16), hipMemcpyHostToDevice));
      // error constant of the single-product pass over a binary channel (sc_match_e.hip: sc_bin_fast): u = 2^-11 for the
      // rounding of S_f to f16, gamma = the largest rounding residual of a (cos, -sin) hi pair relative to its weight w_f, 2e-5 for the
      // fp32 accumulations inside the MFMAs (exact 22-bit products; K <= 32 terms, <= 32 x 2^-23 of the absolute sum per stage even if the
      // adder truncates: 3.8e-6 each), the fp32 combination and epilogue and the fp64 DFT; 5e-5 more in case the matrix cores flush
      // subnormal f16 operands to zero (spectrum values below 2^-14 at the 2^8 / 2^7 scaling: at most 4.8e-7 each, sqrt(2 x 1240) x 4.8e-7 =
      // 2.4e-5 per set even if every value were that small)
      double gamma = 0.0;
      for (int k = 0; k <= 30; k++)
        for (int f = 0; f < pr::SC_NF; f++) {
          const int t = (f * k) % 60;
          const double w = (f == 0 || f == 30) ? 1.0 : 2.0;
          const double c = w * tw[t] * 1024.0, sn = -w * tw[60 + t] * 1024.0;
          const double dc = c - (double)(_Float16)c, ds = sn - (double)(_Float16)sn;
          gamma = std::max(gamma, std::sqrt(dc * dc + ds * ds) / (w * 1024.0));
        }
      const double u = 0x1p-11;
      ctx->sc_bconst = (float)(((u + gamma) * (1.0 + u) + 7e-5) * (1.0 + 1e-6));
    }
    // M2DP plane table from the frozen float normals (M2DP/M2DP.cpp:9-30)
    double pl[2][64][3];
    for (int k = 0; k < 64; k++) {
      float nf[3]; memcpy(nf, PR_M2DP_VECN_BITS[k], 12);
      const double n[3] = {nf[0], nf[1], nf[2]};
      const double d = (1.0 * n[0] + 0.0 * n[1]) + 0.0 * n[2];
      const double xa[3] = {1.0, 0.0, 0.0};
      for (int a = 0; a < 3; a++) pl[0][k][a] = xa[a] - d * n[a];
      pl[1][k][0] = n[1] * pl[0][k][2] - n[2] * pl[0][k][1];
      pl[1][k][1] = n[2] * pl[0][k][0] - n[0] * pl[0][k][2];
      pl[1][k][2] = n[0] * pl[0][k][1] - n[1] * pl[0][k][0];
    }
    TRY(hipMalloc((void**)&ctx->d_planes, sizeof pl));
    TRY(hipMemcpy(ctx->d_planes, pl, sizeof pl, hipMemcpyHostToDevice));
#undef TRY
  } while (0);
  if (rc != PR_OK) { pr_destroy(ctx); return rc; }
  if (const char* s = getenv("PR_SC_NSPLIT")) ctx->sc_nsplit = atoi(s);
  if (const char* s = getenv("PR_FORCE_ORDER_FLAGS")) ctx->force_order = atoi(s) != 0;
  if (const char* s = getenv("PR_SC_MATCH")) ctx->sc_mode = (strcmp(s, "f32") == 0) ? PR_SC_ARITH_F32 : (strcmp(s, "f16") == 0) ? PR_SC_ARITH_F16 : PR_SC_ARITH_F16X2;
  if (const char* s = getenv("PR_SC_KERNEL")) ctx->sc_kernel = (strcmp(s, "h") == 0) ? 0 : 2;
  if (const char* s = getenv("PR_SC_BINARY")) ctx->sc_binary = atoi(s) != 0;
  if (const char* s = getenv("PR_SC_BINARY_PAIR_SCALE")) ctx->sc_pair_scale = (float)atof(s);
  if (const char* s = getenv("PR_XROW")) ctx->xrow_direct = strcmp(s, "direct") == 0;
  if (const char* s = getenv("PR_SC_ONLINE")) ctx->sc_online_h = strcmp(s, "h") == 0;
  *out = ctx;
  return PR_OK;
}

int pr_create(int device_id, pr_ctx** out) { return create_common(device_id, nullptr, false, out); }

int pr_create_on_stream(int device_id, void* hip_stream, pr_ctx** out) {
  return create_common(device_id, static_cast<hipStream_t>(hip_stream), true, out);
}

void pr_destroy(pr_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream || !ctx->own_stream) { (void)hipStreamSynchronize(ctx->stream); if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream); }
  if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
  if (ctx->side2) { (void)hipStreamSynchronize(ctx->side2); (void)hipStreamDestroy(ctx->side2); }
  if (ctx->ev_b) (void)hipEventDestroy(ctx->ev_b);
  if (ctx->sc_scratch) (void)hipFree(ctx->sc_scratch);
  if (ctx->rr_scratch) (void)hipFree(ctx->rr_scratch);
  if (ctx->d_order) (void)hipFree(ctx->d_order);
  if (ctx->rr_tick) (void)hipFree(ctx->rr_tick);
  if (ctx->res_partial) (void)hipFree(ctx->res_partial);
  if (ctx->res_exact) (void)hipFree(ctx->res_exact);
  if (ctx->xrows) (void)hipFree(ctx->xrows);
  if (ctx->xqspec) (void)hipFree(ctx->xqspec);
  if (ctx->xpart) (void)hipFree(ctx->xpart);
  if (ctx->sel_scratch) (void)hipFree(ctx->sel_scratch);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  for (auto& e : ctx->ev_m2) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->ev_t) if (e) (void)hipEventDestroy(e);
  if (ctx->d_flags) (void)hipFree(ctx->d_flags);
  if (ctx->d_svd_rows) (void)hipFree(ctx->d_svd_rows);
  if (ctx->d_twiddle) (void)hipFree(ctx->d_twiddle);
  if (ctx->d_cst) (void)hipFree(ctx->d_cst);
  if (ctx->d_cst_h) (void)hipFree(ctx->d_cst_h);
  if (ctx->d_planes) (void)hipFree(ctx->d_planes);
  delete ctx;
}

int pr_set_sc_arith(pr_ctx* ctx, int arith) {
  if (!ctx) return PR_EINVAL;
  if (arith != PR_SC_ARITH_F16X2 && arith != PR_SC_ARITH_F32 && arith != PR_SC_ARITH_F16) PR_FAIL(ctx, PR_EINVAL, "pr_set_sc_arith: unknown arithmetic %d", arith);
  ctx->sc_mode = arith;
  return PR_OK;
}

int pr_get_sc_arith(const pr_ctx* ctx) { return ctx ? ctx->sc_mode : PR_EINVAL; }

int pr_set_nan_policy(pr_ctx* ctx, int policy) {
  if (!ctx) return PR_EINVAL;
  if (policy != PR_NAN_EXCLUDE && policy != PR_NAN_FAIL) PR_FAIL(ctx, PR_EINVAL, "pr_set_nan_policy: unknown policy %d", policy);
  ctx->nan_policy = policy;
  return PR_OK;
}

int pr_get_nan_policy(const pr_ctx* ctx) { return ctx ? ctx->nan_policy : PR_EINVAL; }

int pr_set_exact_statistics(pr_ctx* ctx, int on) {
  if (!ctx) return PR_EINVAL;
  ctx->force_order = on != 0;
  return PR_OK;
}

int pr_take_warnings(pr_ctx* ctx) {
  if (!ctx) return 0;
  if (set_device(ctx) == PR_OK) (void)check_flags(ctx);
  const int w = ctx->warnings;
  ctx->warnings = 0;
  return w;
}

void* pr_stream(pr_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int pr_sync(pr_ctx* ctx) {
  if (!ctx) return PR_EINVAL;
  if (int rc = set_device(ctx)) return rc;
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  PR_HIP(ctx, hipGetLastError());
  return check_flags(ctx);
}


// ------------------------------------------------------------------------------------------- GPU pre-stage (row f1)
// The bucket count libstdc++ picks when the element count crosses each threshold, read off the real container (it
// depends on counts only, never on keys): the table is rehashed to nb[i] buckets when a key is inserted while cnt[i]
// elements exist.
static void probe_bucket_schedule(int kmax, std::vector<int>& cnt, std::vector<int>& nb) {
  std::unordered_map<int, int> m;
  size_t last = m.bucket_count();
  for (int k = 0; k < kmax; k++) {
    m[k] = k;
    if (m.bucket_count() != last) { cnt.push_back(k); nb.push_back((int)m.bucket_count()); last = m.bucket_count(); }
  }
}

int pr_hash_order(const int32_t* keys, int32_t K, int32_t* order) {   // host build of hash_order.hpp (tests; not a hot path)
  if (K < 0 || (K > 0 && (!keys || !order))) return PR_EINVAL;
  std::vector<int> cnt, nb;
  probe_bucket_schedule(K, cnt, nb);
  std::vector<int> next((size_t)K + 1), bkt(nb.empty() ? 1 : (size_t)nb.back());
  pr::hash_order(keys, K, cnt.data(), nb.data(), (int)cnt.size(), next.data(), bkt.data(), order);
  return PR_OK;
}

int pr_pts_preprocess_gpu(pr_ctx* ctx, const char* poses_file, const char* pts_file, const char* incoming_id_file,
                          double lidarRange, int polar, int verbose, pr_clouds** out) {
  if (!ctx) return PR_EINVAL;
  if (!poses_file || !pts_file || !out || !(lidarRange > 0)) PR_FAIL(ctx, PR_EINVAL, "pr_pts_preprocess_gpu: bad arguments");
  if (int rc = set_device(ctx)) return rc;
  std::vector<pr_rec::PoseRec> poses;
  pr_rec::History h;
  pr_rec::read_records(poses_file, pts_file, poses, h);
  FILE* idf = nullptr;
  if (incoming_id_file) {
    idf = fopen(incoming_id_file, "w");
    if (!idf) PR_FAIL(ctx, PR_EIO, "cannot write %s", incoming_id_file);
  }
  const auto t0 = std::chrono::steady_clock::now();
  const int M = (int)poses.size();
  const int64_t T = (int64_t)h.id.size();
  // host bookkeeping of the pose loop (pts_preprocess.h:187-215): resets, the 30 skipped frames, which point is appended
  // at which pose
  std::vector<unsigned char> emit(M ? M : 1, 0);
  std::vector<int> next_reset(M ? M : 1, M), birth((size_t)(T ? T : 1), -1), pose_of;
  std::vector<int64_t> cur_after(M ? M : 1, 0);
  std::vector<double> W((size_t)(M ? M : 1) * 12);
  {
    size_t cursor = 0;
    int since_reset = 0;
    std::vector<int> resets;
    for (int p = 0; p < M; p++) {
      const double* w = poses[p].w;
      memcpy(&W[(size_t)p * 12], w, 12 * sizeof(double));
      if (std::sqrt(w[3] * w[3] + w[7] * w[7] + w[11] * w[11]) < 1.0) {
        if (verbose) printf("\nReset at id: %d\n", poses[p].id);
        since_reset = 0;
        resets.push_back(p);
      }
      while (cursor < (size_t)T && h.id[cursor] <= poses[p].id) birth[cursor++] = p;
      cur_after[p] = (int64_t)cursor;
      if (since_reset < 30) { since_reset++; continue; }
      emit[p] = 1;
      pose_of.push_back(p);
    }
    size_t ri = 0;
    for (int p = 0; p < M; p++) {          // first reset pose strictly after p
      while (ri < resets.size() && resets[ri] <= p) ri++;
      next_reset[p] = ri < resets.size() ? resets[ri] : M;
    }
  }
  const int E = (int)pose_of.size();
  pr_clouds* res = new (std::nothrow) pr_clouds;
  if (!res) { if (idf) fclose(idf); PR_FAIL(ctx, PR_ENOMEM, "out of host memory"); }
  res->offs.assign(1, 0);
  int rc = PR_OK;
  DevScope scope_(ctx);
  DevBuf dxyz, dint, dbirth, dW, demit, dnr, ddeath, dpose, dfa, dcur, doff, dcnt, dlist, dcell, dval, dtv, dtf, dtb, dkeys, dwin,
      dnk, dscnt, dsnb, dnext, dboff, dbkt, dord, dooff, doxyz, doint, dfr;
#define PRE_HIP(call) { hipError_t _e = (call); if (_e != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(_e); \
                        rc = (_e == hipErrorOutOfMemory) ? PR_ENOMEM : PR_EHIP; break; } }
#define UP(buf, vec) { PRE_HIP((buf).alloc((vec).size() * sizeof((vec)[0]))); \
                       PRE_HIP(hipMemcpyAsync((buf).p, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice, ctx->stream)); }
  do {
    if (E == 0 || T == 0) break;           // every emitting pose gets an empty cloud below
    UP(dxyz, h.xyz); UP(dint, h.it); UP(dbirth, birth); UP(dW, W); UP(demit, emit); UP(dnr, next_reset); UP(dpose, pose_of);
    PRE_HIP(ddeath.alloc((size_t)T * 4));
    pr::launch_death(ctx->stream, dxyz.as<double>(), dbirth.as<int>(), T, dW.as<double>(), demit.as<unsigned char>(), dnr.as<int>(),
                     lidarRange, ddeath.as<int>());
    std::vector<int> death((size_t)T);
    PRE_HIP(hipMemcpyAsync(death.data(), ddeath.p, (size_t)T * 4, hipMemcpyDeviceToHost, ctx->stream));
    PRE_HIP(hipStreamSynchronize(ctx->stream));
    // candidate range of every emitting pose: [first alive point, points appended so far); dead is for good, so the lower
    // end only moves forward
    std::vector<int64_t> fa(E), cur(E), off((size_t)E + 1, 0);
    {
      int64_t ptr = 0;
      for (int e = 0; e < E; e++) {
        const int p = pose_of[e];
        cur[e] = cur_after[p];
        while (ptr < cur[e] && death[ptr] <= p) ptr++;
        fa[e] = ptr;
      }
    }
    UP(dfa, fa); UP(dcur, cur);
    PRE_HIP(dcnt.alloc((size_t)E * 4));
    pr::launch_members(ctx->stream, E, ddeath.as<int>(), dpose.as<int>(), dfa.as<int64_t>(), dcur.as<int64_t>(), nullptr, dcnt.as<int>(), nullptr);
    std::vector<int> cnt(E);
    PRE_HIP(hipMemcpyAsync(cnt.data(), dcnt.p, (size_t)E * 4, hipMemcpyDeviceToHost, ctx->stream));
    PRE_HIP(hipStreamSynchronize(ctx->stream));
    for (int e = 0; e < E; e++) off[e + 1] = off[e] + cnt[e];
    const int64_t S = off[E];
    UP(doff, off);
    PRE_HIP(dlist.alloc((size_t)S * 4)); PRE_HIP(dcell.alloc((size_t)S * 4)); PRE_HIP(dval.alloc((size_t)S * 8));
    PRE_HIP(dkeys.alloc((size_t)S * 4)); PRE_HIP(dwin.alloc((size_t)S * 4)); PRE_HIP(dnk.alloc((size_t)E * 4));
    pr::launch_members(ctx->stream, E, ddeath.as<int>(), dpose.as<int>(), dfa.as<int64_t>(), dcur.as<int64_t>(), doff.as<int64_t>(),
                       dcnt.as<int>(), dlist.as<int>());
    // dense (pose, cell) tables, a batch of poses at a time (16 B per cell)
    const int64_t C = pr::prestage_cells(lidarRange, polar);
    int PB = (int)std::max<int64_t>(1, std::min<int64_t>(E, ((int64_t)4 << 30) / (16 * C)));
    PRE_HIP(dtv.alloc((size_t)PB * C * 8)); PRE_HIP(dtf.alloc((size_t)PB * C * 4)); PRE_HIP(dtb.alloc((size_t)PB * C * 4));
    bool fail = false;
    for (int e0 = 0; e0 < E && !fail; e0 += PB) {
      const int e1 = std::min(E, e0 + PB);
      if (hipMemsetAsync(dtv.p, 0xFF, (size_t)(e1 - e0) * C * 8, ctx->stream) != hipSuccess ||
          hipMemsetAsync(dtf.p, 0xFF, (size_t)(e1 - e0) * C * 4, ctx->stream) != hipSuccess ||
          hipMemsetAsync(dtb.p, 0xFF, (size_t)(e1 - e0) * C * 4, ctx->stream) != hipSuccess) { fail = true; break; }
      pr::launch_cells(ctx->stream, dxyz.as<double>(), dlist.as<int>(), doff.as<int64_t>(), dpose.as<int>(), e0, e1, off[e0], off[e1],
                       dW.as<double>(), lidarRange, polar, C, dcell.as<int>(), dval.as<unsigned long long>(),
                       dtv.as<unsigned long long>(), dtf.as<unsigned>(), dtb.as<unsigned>());
      pr::launch_keys(ctx->stream, doff.as<int64_t>(), e0, e1, C, dcell.as<int>(), dlist.as<int>(), dtf.as<unsigned>(), dtb.as<unsigned>(),
                      dkeys.as<int>(), dwin.as<int>(), dnk.as<int>());
    }
    if (fail) { ctx->err = "hipMemsetAsync failed"; rc = PR_EHIP; break; }
    std::vector<int> nk(E);
    PRE_HIP(hipMemcpyAsync(nk.data(), dnk.p, (size_t)E * 4, hipMemcpyDeviceToHost, ctx->stream));
    PRE_HIP(hipStreamSynchronize(ctx->stream));
    PRE_HIP(hipGetLastError());
    int kmax = 0;
    for (int e = 0; e < E; e++) kmax = std::max(kmax, nk[e]);
    std::vector<int> scnt, snb;
    probe_bucket_schedule(kmax, scnt, snb);
    if (scnt.empty()) { scnt.push_back(0); snb.push_back(1); }
    std::vector<int64_t> boff((size_t)E + 1, 0), ooff((size_t)E + 1, 0);
    for (int e = 0; e < E; e++) {
      int nbk = 1;
      for (size_t i = 0; i < scnt.size() && scnt[i] < std::max(nk[e], 1); i++) nbk = snb[i];
      boff[e + 1] = boff[e] + nbk;
      ooff[e + 1] = ooff[e] + nk[e];
    }
    const int64_t TOT = ooff[E];
    UP(dscnt, scnt); UP(dsnb, snb); UP(dboff, boff); UP(dooff, ooff);
    PRE_HIP(dnext.alloc((size_t)S * 4)); PRE_HIP(dord.alloc((size_t)S * 4)); PRE_HIP(dbkt.alloc((size_t)boff[E] * 4));
    PRE_HIP(doxyz.alloc((size_t)TOT * 24)); PRE_HIP(doint.alloc((size_t)TOT * 4)); PRE_HIP(dfr.alloc((size_t)E * 16 * 8));
    pr::launch_order(ctx->stream, E, doff.as<int64_t>(), dnk.as<int>(), dkeys.as<int>(), dscnt.as<int>(), dsnb.as<int>(), (int)scnt.size(),
                     dnext.as<int>(), dboff.as<int64_t>(), dbkt.as<int>(), dord.as<int>());
    pr::launch_gather(ctx->stream, E, TOT, doff.as<int64_t>(), dooff.as<int64_t>(), dpose.as<int>(), dord.as<int>(), dwin.as<int>(),
                      dxyz.as<double>(), dint.as<float>(), dW.as<double>(), lidarRange, doxyz.as<double>(), doint.as<float>(), dfr.as<double>());
    res->xyz.resize((size_t)TOT * 3);
    res->inten.resize((size_t)TOT);
    PRE_HIP(hipMemcpyAsync(res->xyz.data(), doxyz.p, (size_t)TOT * 24, hipMemcpyDeviceToHost, ctx->stream));
    PRE_HIP(hipMemcpyAsync(res->inten.data(), doint.p, (size_t)TOT * 4, hipMemcpyDeviceToHost, ctx->stream));
    PRE_HIP(hipStreamSynchronize(ctx->stream));
    PRE_HIP(hipGetLastError());
    for (int e = 0; e < E; e++) res->offs.push_back(ooff[e + 1]);
    // the clouds and their frames stay in HBM with the host copies (pr_clouds_dev_*): the generators take them from there
    res->d_xyz = doxyz.p; res->d_inten = doint.p; res->d_offs = dooff.p; res->d_frames = dfr.p;
    doxyz.p = doint.p = dooff.p = dfr.p = nullptr;
    res->device = ctx->device;
    res->release = [](pr_clouds* c) {
      (void)hipSetDevice(c->device);
      for (void* p : {c->d_xyz, c->d_inten, c->d_offs, c->d_frames}) if (p) (void)hipFree(p);
      c->d_xyz = c->d_inten = c->d_offs = c->d_frames = nullptr;
    };
  } while (0);
#undef UP
#undef PRE_HIP
  if (rc != PR_OK) { (void)hipStreamSynchronize(ctx->stream); pr_clouds_free(res); if (idf) fclose(idf); return rc; }
  if (res->offs.size() == 1) res->offs.resize((size_t)E + 1, 0);
  for (int e = 0; e < E; e++) {
    res->ids.push_back(poses[pose_of[e]].id);
    if (idf) fprintf(idf, "%d\n", poses[pose_of[e]].id);
  }
  if (idf) fclose(idf);
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  res->avg_ms = E ? 1000.0 * secs / E : NAN;
  res->avg_pts = E ? (double)res->inten.size() / E : NAN;
  if (verbose)
    printf("\ngenerate_spherical_points average time: %gms average points: %g\n", (double)(float)res->avg_ms, (double)(float)res->avg_pts);
  *out = res;
  return PR_OK;
}

// ------------------------------------------------------------------------------------------- signature sets
static int* sigset_bstat(const pr_sigset* s) { return reinterpret_cast<int*>(s->binfo + ((size_t)s->max_sigs + 1) * 2); }
// DELIGHT sets: [capacity][4096] f32 histograms, then [capacity][64 lanes][2] u32 empty-bin masks
static unsigned* delight_masks(const pr_sigset* s) { return reinterpret_cast<unsigned*>(s->packed + (size_t)s->max_sigs * 4096); }
int pr_sigset_create(pr_ctx* ctx, int type, int role, int32_t max_sigs, pr_sigset** out) {
  if (!ctx || !out) return PR_EINVAL;
  if ((type != PR_TYPE_SC && type != PR_TYPE_M2DP && type != PR_TYPE_DELIGHT) || (role != PR_ROLE_QUERY && role != PR_ROLE_DB) || max_sigs < 0)
    PR_FAIL(ctx, PR_EINVAL, "pr_sigset_create: bad type/role/max_sigs");
  // the matchers address one 8/32-row band of the [m][n] distance matrix with 32-bit byte offsets (raw buffer stores):
  // 32 rows x n x 4 B < 2^31 with n = 4 rows per M2DP signature caps a set at 4 M signatures (77 GB of SC signatures)
  if (max_sigs > PR_MAX_SIGS) PR_FAIL(ctx, PR_EINVAL, "pr_sigset_create: max_sigs %d exceeds PR_MAX_SIGS = %d", max_sigs, PR_MAX_SIGS);
  if (int rc = set_device(ctx)) return rc;
  pr_sigset* s = new (std::nothrow) pr_sigset;
  if (!s) PR_FAIL(ctx, PR_ENOMEM, "pr_sigset_create: out of host memory");
  s->type = type; s->role = role; s->max_sigs = max_sigs;
  s->sc_mode = ctx->sc_mode;
  s->floats = sigset_floats(type, role, max_sigs, &s->groups, s->sc_mode);
  hipError_t e = hipMalloc((void**)&s->packed, s->floats * sizeof(float) + 16);
  if (e != hipSuccess) { delete s; PR_FAIL(ctx, PR_ENOMEM, "pr_sigset_create: hipMalloc(%zu B) failed: %s", s->floats * 4, hipGetErrorString(e)); }
  if (type == PR_TYPE_SC) {
    e = hipMalloc((void**)&s->bad, ((size_t)max_sigs + 1) * 2 * sizeof(int));
    if (e != hipSuccess) { (void)hipFree(s->packed); delete s; PR_FAIL(ctx, PR_ENOMEM, "pr_sigset_create: hipMalloc failed: %s", hipGetErrorString(e)); }
    if (s->sc_mode == PR_SC_ARITH_F16X2) {
      const size_t bytes = ((size_t)max_sigs + 1) * 2 * sizeof(float) + (pr::SC_BSTAT_INTS + 6 * ((size_t)max_sigs / 16 + 1)) * sizeof(int);   // + one slot per 16-row block of the pack kernel
      e = hipMalloc((void**)&s->binfo, bytes);
      if (e != hipSuccess) { (void)hipFree(s->packed); (void)hipFree(s->bad); delete s; PR_FAIL(ctx, PR_ENOMEM, "pr_sigset_create: hipMalloc failed: %s", hipGetErrorString(e)); }
      (void)hipMemsetAsync(s->binfo, 0, bytes, ctx->stream);
    }
  }
  // the operand image starts out all-zero: padding rows / groups must be zero (they yield dot = 0 and are masked on store), and a pack
  // only writes the rows it is given - see pr_sigset_pack
  e = hipMemsetAsync(s->packed, 0, s->floats * sizeof(float), ctx->stream);
  if (e != hipSuccess) { (void)hipFree(s->packed); if (s->bad) (void)hipFree(s->bad); if (s->binfo) (void)hipFree(s->binfo); delete s; PR_FAIL(ctx, PR_EHIP, "pr_sigset_create: hipMemsetAsync failed: %s", hipGetErrorString(e)); }
  *out = s;
  return PR_OK;
}

void pr_sigset_destroy(pr_ctx* ctx, pr_sigset* s) {
  if (!s) return;
  if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
  if (s->packed) (void)hipFree(s->packed);
  if (s->bad) (void)hipFree(s->bad);
  if (s->binfo) (void)hipFree(s->binfo);
  delete s;
}

int32_t pr_sigset_count(const pr_sigset* s) { return s ? s->count : 0; }

int pr_sigset_pack(pr_ctx* ctx, pr_sigset* s, const void* sig, int dtype, int where, int32_t n_sigs) {
  if (!ctx || !s) return PR_EINVAL;
  if (n_sigs < 0 || n_sigs > s->max_sigs || (n_sigs > 0 && !sig) || (dtype != PR_F64 && dtype != PR_F32) ||
      (where != PR_HOST && where != PR_DEVICE))
    PR_FAIL(ctx, PR_EINVAL, "pr_sigset_pack: bad arguments (n_sigs=%d, capacity=%d)", n_sigs, s->max_sigs);
  if (int rc = set_device(ctx)) return rc;
  const size_t rows = (s->type == PR_TYPE_SC) ? (size_t)n_sigs : (s->type == PR_TYPE_M2DP ? (size_t)4 * n_sigs : (size_t)16 * n_sigs);
  const size_t cols = (s->type == PR_TYPE_SC) ? PR_SC_SIG_LEN : (s->type == PR_TYPE_M2DP ? PR_M2DP_SIG_LEN : PR_DELIGHT_SIG_LEN);
  const size_t esz = (dtype == PR_F64) ? 8 : 4;
  DevScope scope_(ctx);
  DevBuf stage;
  const void* dsig = sig;
  if (where == PR_HOST && n_sigs > 0) {
    PR_HIP(ctx, stage.alloc(rows * cols * esz));
    PR_HIP(ctx, hipMemcpyAsync(stage.p, sig, rows * cols * esz, hipMemcpyHostToDevice, ctx->stream));
    dsig = stage.p;
  }
  // the channel stride must match the matcher's view of THIS count (not the capacity)
  int groups;
  (void)sigset_floats(s->type, s->role, s->cap_geom ? s->max_sigs : n_sigs, &groups, s->sc_mode);   // (an appendable set: the capacity's, pr_sigset_reserve)
  s->groups = groups;
  // Padding rows / tiles must be zero: they yield dot = 0 and are masked on store.  The image is zeroed at creation and a pack writes
  // only its own rows (the throughput kernels: whole groups, their padding rows as zeros), so a re-pack of at least as many rows in the
  // same geometry - every step of a steady workload - needs no 1.2 GB fill in front of it; anything else starts from zeros again.
  // (DELIGHT sets keep the unconditional fill.)  The zero-norm flags need no clearing either: every pack writes all of its rows' entries.
  const bool keep = s->type != PR_TYPE_DELIGHT && (s->packed_groups < 0 || (s->packed_groups == groups && n_sigs >= s->hw));
  if (!keep) { PR_HIP(ctx, hipMemsetAsync(s->packed, 0, s->floats * sizeof(float), ctx->stream)); s->hw = 0; }
  if (n_sigs > 0) { s->packed_groups = groups; if (n_sigs > s->hw) s->hw = n_sigs; }
  if (s->type == PR_TYPE_SC && (s->sc_mode == PR_SC_ARITH_F16X2 || s->sc_mode == PR_SC_ARITH_F16)) {
    int* bstat = s->binfo ? sigset_bstat(s) : nullptr;
    // every pack starts its statistics afresh (a kernel, not hipMemsetAsync: a 32-byte memset node of a captured call was seen to leave the
    // statistics of the previous replay in place - tests/test_gpu_parity.py::test_match_as_hipgraph_replay)
    if (bstat && n_sigs <= 8) pr::launch_zero_ints(ctx->stream, bstat, pr::SC_BSTAT_INTS);   // (the few-rows kernel adds with atomics; the block kernel's statistics are folded by a kernel of their own)
    pr::launch_sc_pack_h(ctx->stream, dsig, dtype, n_sigs, s->role, s->packed, groups, ctx->d_twiddle, ctx->d_flags, s->bad,
                         s->sc_mode == PR_SC_ARITH_F16, s->binfo, bstat);
  }
  else if (s->type == PR_TYPE_SC)
    pr::launch_sc_pack(ctx->stream, dsig, dtype, n_sigs, s->role, s->packed, groups, ctx->d_twiddle, ctx->d_flags, s->bad);
  else if (s->type == PR_TYPE_M2DP)
    if (s->sc_mode != PR_SC_ARITH_F32) pr::launch_m2dp_pack_h(ctx->stream, dsig, dtype, n_sigs, s->packed, groups);
    else pr::launch_m2dp_pack(ctx->stream, dsig, dtype, n_sigs, s->packed, groups);
  else
    pr::launch_delight_pack(ctx->stream, dsig, dtype, n_sigs, s->packed, delight_masks(s));
  PR_HIP(ctx, hipGetLastError());
  s->count = n_sigs;
  if (stage.p) PR_HIP(ctx, hipStreamSynchronize(ctx->stream));   // staging buffer is freed on return
  return PR_OK;
}

// Appendable sets (SC/test_sc.cpp:40-56 produces one signature per keyframe; run_test.m:57 matches against everything seen so far): the
// operand image of a DB set is [channel][groups][...] with the channel stride = the group count, which pr_sigset_pack takes from the ROW
// COUNT - so one more row can move the second channel.  pr_sigset_reserve fixes the stride at the CAPACITY's group count (the matchers get
// it beside the count); rows are then added in place.
static bool appendable_kind(const pr_sigset* s) {
  return s->role == PR_ROLE_DB && ((s->type == PR_TYPE_SC && (s->sc_mode == PR_SC_ARITH_F16X2 || s->sc_mode == PR_SC_ARITH_F16)) ||
                                   (s->type == PR_TYPE_M2DP && s->sc_mode != PR_SC_ARITH_F32));
}
int pr_sigset_reserve(pr_ctx* ctx, pr_sigset* s) {
  if (!ctx || !s) return PR_EINVAL;
  if (!appendable_kind(s)) PR_FAIL(ctx, PR_EINVAL, "pr_sigset_reserve: DB sets of SC or M2DP signatures in the f16 arithmetics only");
  if (s->cap_geom) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  if (s->packed_groups >= 0) { PR_HIP(ctx, hipMemsetAsync(s->packed, 0, s->floats * sizeof(float), ctx->stream)); s->packed_groups = -1; s->hw = 0; }
  s->cap_geom = true;
  s->count = 0;                                                 // (rows packed in the count's geometry are gone: pack or append again)
  (void)sigset_floats(s->type, s->role, s->max_sigs, &s->groups, s->sc_mode);
  if (s->binfo) pr::launch_zero_ints(ctx->stream, sigset_bstat(s), pr::SC_BSTAT_INTS);
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

int pr_sigset_append(pr_ctx* ctx, pr_sigset* s, const void* sig, int dtype, int where, int32_t n_new) {
  if (!ctx || !s) return PR_EINVAL;
  if (n_new < 0 || (n_new > 0 && !sig) || (dtype != PR_F64 && dtype != PR_F32) || (where != PR_HOST && where != PR_DEVICE))
    PR_FAIL(ctx, PR_EINVAL, "pr_sigset_append: bad arguments (n_new=%d)", n_new);
  if (!appendable_kind(s)) PR_FAIL(ctx, PR_EINVAL, "pr_sigset_append: DB sets of SC or M2DP signatures in the f16 arithmetics only");
  if (!s->cap_geom) {
    if (s->count != 0) PR_FAIL(ctx, PR_EINVAL, "pr_sigset_append: the set holds %d rows in the geometry of that count; pr_sigset_reserve it first (before packing)", s->count);
    if (int rc = pr_sigset_reserve(ctx, s)) return rc;
  }
  if ((int64_t)s->count + n_new > s->max_sigs) PR_FAIL(ctx, PR_EINVAL, "pr_sigset_append: %d + %d rows exceed the capacity %d", s->count, n_new, s->max_sigs);
  if (n_new == 0) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  const size_t rows = s->type == PR_TYPE_SC ? (size_t)n_new : (size_t)4 * n_new, cols = s->type == PR_TYPE_SC ? PR_SC_SIG_LEN : PR_M2DP_SIG_LEN;
  const size_t esz = dtype == PR_F64 ? 8 : 4;
  DevScope scope_(ctx);
  DevBuf stage;
  const void* dsig = sig;
  if (where == PR_HOST) {
    PR_HIP(ctx, stage.alloc(rows * cols * esz));
    PR_HIP(ctx, hipMemcpyAsync(stage.p, sig, rows * cols * esz, hipMemcpyHostToDevice, ctx->stream));
    dsig = stage.p;
  }
  if (s->type == PR_TYPE_SC)       // the row-wise pack kernel: bit for bit the rows a pack of the whole set writes; statistics folded in, not restarted
    pr::launch_sc_pack_h_rows(ctx->stream, dsig, dtype, n_new, s->count, s->packed, s->groups, ctx->d_twiddle, ctx->d_flags, s->bad,
                              s->sc_mode == PR_SC_ARITH_F16, s->binfo, s->binfo ? sigset_bstat(s) : nullptr);
  else
    pr::launch_m2dp_pack_h(ctx->stream, dsig, dtype, n_new, s->packed, s->groups, s->count);
  PR_HIP(ctx, hipGetLastError());
  s->count += n_new;
  s->packed_groups = s->groups;
  if (s->count > s->hw) s->hw = s->count;
  if (stage.p) PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PR_OK;
}

int pr_sigset_image(const pr_sigset* s, const void** image, size_t* bytes, int32_t* channel_stride_groups) {
  if (!s) return PR_EINVAL;
  if (image) *image = s->packed;
  if (bytes) *bytes = s->floats * sizeof(float);
  if (channel_stride_groups) *channel_stride_groups = s->groups;
  return PR_OK;
}

// ------------------------------------------------------------------------------------------- device path
int pr_distances_dev(pr_ctx* ctx, const pr_sigset* q, const pr_sigset* db, float* d_p, float* d_i) {
  if (!ctx || !q || !db || !d_p || (!d_i && q->type != PR_TYPE_DELIGHT)) return PR_EINVAL;
  if (q->type != db->type || q->role != PR_ROLE_QUERY || db->role != PR_ROLE_DB)
    PR_FAIL(ctx, PR_EINVAL, "pr_distances_dev: q must be a QUERY set and db a DB set of the same type");
  if (int rc = set_device(ctx)) return rc;
  if (q->type != PR_TYPE_DELIGHT && q->sc_mode != db->sc_mode)
    PR_FAIL(ctx, PR_EINVAL, "pr_distances_dev: the two sets were packed for different arithmetic modes");
  // an appendable DB set: the channel stride is the capacity's, which only the default matchers take beside the count
  const int dgs = db->cap_geom ? db->groups : 0;
  if (dgs && db->type == PR_TYPE_SC && db->sc_mode == PR_SC_ARITH_F16X2 && (ctx->sc_kernel != 2 || (q->count <= 8 && ctx->sc_online_h)))
    PR_FAIL(ctx, PR_EINVAL, "pr_distances_dev: an appendable (pr_sigset_reserve) SC set needs the default matcher (not PR_SC_KERNEL / PR_SC_ONLINE=h)");
  if (ctx->timing) { ctx->timing_valid = 1; PR_HIP(ctx, hipEventRecord(ctx->ev_t[0], ctx->stream)); }
  if (q->type == PR_TYPE_SC && q->sc_mode == PR_SC_ARITH_F16)
    pr::launch_sc_match_e(ctx->stream, q->packed, q->count, db->packed, db->count, ctx->d_cst_h, d_p, d_i, ctx->sc_nsplit, 1, dgs);
  else if (q->type == PR_TYPE_SC && q->sc_mode == 0 && ctx->sc_kernel == 2 && ctx->sc_binary && q->binfo && db->binfo) {
    ctx->bin_gen = ctx->bin_gen == 0x7fffffff ? 1 : ctx->bin_gen + 1;
    const pr::ScBin bin = {sigset_bstat(q), sigset_bstat(db), q->binfo, db->binfo, ctx->d_flags + 4, ctx->bin_gen, ctx->sc_bconst, ctx->sc_pair_scale, 0, -1};
    pr::launch_sc_match_e_bin(ctx->stream, q->packed, q->count, db->packed, db->count, ctx->d_cst_h, d_p, d_i, ctx->sc_nsplit, bin,
                              ctx->timing ? ctx->ev_t : nullptr, ctx->sc_online_h ? 1 : 0, dgs);
    if (ctx->timing) ctx->timing_valid = 3;
  }
  else if (q->type == PR_TYPE_SC && q->sc_mode == 0 && ctx->sc_kernel == 2 && (q->count > 8 || !ctx->sc_online_h))
    pr::launch_sc_match_e(ctx->stream, q->packed, q->count, db->packed, db->count, ctx->d_cst_h, d_p, d_i, ctx->sc_nsplit, 0, dgs);
  else if (q->type == PR_TYPE_SC && q->sc_mode == 0)
    pr::launch_sc_match_h(ctx->stream, q->packed, q->count, db->packed, db->count, ctx->d_cst_h, d_p, d_i, ctx->sc_nsplit);
  else if (q->type == PR_TYPE_SC)
    pr::launch_sc_match(ctx->stream, q->packed, q->count, db->packed, db->count, ctx->d_cst, d_p, d_i, ctx->sc_nsplit);
  else if (q->type == PR_TYPE_M2DP)
    if (q->sc_mode != PR_SC_ARITH_F32) pr::launch_m2dp_match_h(ctx->stream, q->packed, q->count, db->packed, db->count, d_p, d_i, q->sc_mode == PR_SC_ARITH_F16, dgs);
    else pr::launch_m2dp_match(ctx->stream, q->packed, q->count, db->packed, db->count, d_p, d_i);
  else
    pr::launch_delight_match(ctx->stream, q->packed, q->count, db->packed, delight_masks(db), db->count, d_p);
  if (ctx->timing && ctx->timing_valid == 1) PR_HIP(ctx, hipEventRecord(ctx->ev_t[3], ctx->stream));
  if (q->type == PR_TYPE_SC) pr::launch_nan_fixup(ctx->stream, d_p, d_i, q->count, db->count, q->bad, db->bad);   // processSC.m:16,19
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

int pr_set_sc_binary(pr_ctx* ctx, int on) {
  if (!ctx) return PR_EINVAL;
  ctx->sc_binary = on != 0;
  return PR_OK;
}

int pr_set_kernel_timing(pr_ctx* ctx, int on) {
  if (!ctx) return PR_EINVAL;
  if (int rc = set_device(ctx)) return rc;
  if (on) for (auto& e : ctx->ev_t) if (!e) PR_HIP(ctx, hipEventCreate(&e));
  ctx->timing = on != 0;
  ctx->timing_valid = 0;
  return PR_OK;
}

int pr_last_distance_timing(pr_ctx* ctx, float* ms) {
  if (!ctx || !ms) return PR_EINVAL;
  if (!ctx->timing || !ctx->timing_valid) PR_FAIL(ctx, PR_EINVAL, "pr_last_distance_timing: no timed pr_distances_dev call (pr_set_kernel_timing)");
  if (int rc = set_device(ctx)) return rc;
  PR_HIP(ctx, hipEventSynchronize(ctx->ev_t[3]));
  ms[0] = ms[1] = ms[2] = 0.f;
  if (ctx->timing_valid == 3) {
    PR_HIP(ctx, hipEventElapsedTime(&ms[0], ctx->ev_t[0], ctx->ev_t[1]));
    PR_HIP(ctx, hipEventElapsedTime(&ms[1], ctx->ev_t[1], ctx->ev_t[2]));
    PR_HIP(ctx, hipEventElapsedTime(&ms[2], ctx->ev_t[2], ctx->ev_t[3]));
  } else {
    PR_HIP(ctx, hipEventElapsedTime(&ms[0], ctx->ev_t[0], ctx->ev_t[3]));
  }
  return PR_OK;
}

int pr_sc_binary_state(pr_ctx* ctx, const pr_sigset* q, const pr_sigset* db, int32_t* state) {
  if (!ctx || !q || !db || !state) return PR_EINVAL;
  if (int rc = set_device(ctx)) return rc;
  *state = 0;
  if (!q->binfo || !db->binfo || !ctx->sc_binary) return PR_OK;
  int hq[pr::SC_BSTAT_INTS], hd[pr::SC_BSTAT_INTS];
  PR_HIP(ctx, hipMemcpyAsync(hq, sigset_bstat(q), sizeof hq, hipMemcpyDeviceToHost, ctx->stream));
  PR_HIP(ctx, hipMemcpyAsync(hd, sigset_bstat(db), sizeof hd, hipMemcpyDeviceToHost, ctx->stream));
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  auto fl = [](int b) { float f; memcpy(&f, &b, 4); return f; };
  // the arithmetic of sc_bin_fast (sc_match_e.hip), float for float
  const float eq = std::sqrt(((fl(hq[2]) + fl(hq[3])) + fl(hq[4])) + fl(hq[5])), ed = std::sqrt(((fl(hd[2]) + fl(hd[3])) + fl(hd[4])) + fl(hd[5]));
  const float bound = (eq + ed + eq * ed + ctx->sc_bconst * (1.f + eq) * (1.f + ed)) * 1.001f;
  if (getenv("PR_AMD_VERBOSE"))
    fprintf(stderr, "pr_sc_binary_state: queries {nonbinary %d, max ones %d, eps %.3g} db {nonbinary %d, max ones %d, eps %.3g} bound %.4g x %.1f\n",
            hq[0], hq[1], eq, hd[0], hd[1], ed, bound, std::sqrt((float)hq[1] * (float)hd[1]));
  *state = (hq[0] == 0 && hd[0] == 0 && bound * std::sqrt((float)hq[1] * (float)hd[1]) < 0.72f) ? 1 : 0;
  if (*state) {      // ... and what the last call's rounding tests said (the word lives until the next call's channel-0 launch clears it)
    int viol = 0;
    PR_HIP(ctx, hipMemcpyAsync(&viol, ctx->d_flags + 4, sizeof viol, hipMemcpyDeviceToHost, ctx->stream));
    PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (viol != 0) *state = 2;                 // (the channel-0 launch of every call clears the word)
  }
  return PR_OK;
}

int pr_row_moments_dev(pr_ctx* ctx, const float* d_p, const float* d_i, int32_t m, int32_t n, double* mom) {
  if (!ctx || !d_p || !d_i || !mom || m < 0 || n < 1) return PR_EINVAL;
  if (int rc = set_device(ctx)) return rc;
  pr::launch_row_moments(ctx->stream, d_p, d_i, m, n, mom, ctx->sel_scratch);
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

static int fuse_select_impl(pr_ctx* ctx, const float* d_p, const float* d_i, int32_t m, int32_t n, const double* mom_all,
                            int32_t G, int32_t q_row0, int32_t db_row0, int32_t mask_width, double p_weight, int32_t k,
                            int32_t* idx, float* score, double* score64) {
  if (!ctx || !d_p || (d_i && !mom_all) || !idx || !score || m < 0 || n < 1 || G < 1 || k < 1)
    return PR_EINVAL;
  if (int rc = set_device(ctx)) return rc;
  pr::launch_fuse_select(ctx->stream, d_p, d_i, m, n, mom_all, G, q_row0, db_row0, mask_width, p_weight, k, idx, score, nullptr, nullptr, nullptr, ctx->sel_scratch, score64);
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}
int pr_fuse_select_dev(pr_ctx* ctx, const float* d_p, const float* d_i, int32_t m, int32_t n, const double* mom_all,
                       int32_t G, int32_t q_row0, int32_t db_row0, int32_t mask_width, double p_weight, int32_t k,
                       int32_t* idx, float* score) {
  return fuse_select_impl(ctx, d_p, d_i, m, n, mom_all, G, q_row0, db_row0, mask_width, p_weight, k, idx, score, nullptr);
}
int pr_fuse_select_f64_dev(pr_ctx* ctx, const float* d_p, const float* d_i, int32_t m, int32_t n, const double* mom_all,
                           int32_t G, int32_t q_row0, int32_t db_row0, int32_t mask_width, double p_weight, int32_t k,
                           int32_t* idx, float* score, double* score64) {
  if (!score64) return PR_EINVAL;
  return fuse_select_impl(ctx, d_p, d_i, m, n, mom_all, G, q_row0, db_row0, mask_width, p_weight, k, idx, score, score64);
}

// survivors of the all-pairs selection that the fp64 re-evaluation looks at: k + 8 (fp32-grade passes), k + 56 in the single-product
// arithmetic, whose pass scores only bracket the exact ones to ~0.3 sigma (the interface's cap is 128)
static int rerank_width(int k, int mode) { const int w = k + (mode == PR_SC_ARITH_F16 ? 56 : 8); return w > 128 ? 128 : w; }

// distance error bound of the context's all-pairs arithmetic for the pruning / margin logic of the re-evaluation: 0 selects the
// fp32-grade default (1e-6, generous 64x margin); the single-product f16 arithmetic carries 2e-3 (include/place_recognition.h)
static double pass_eps(const pr_ctx* ctx) { return ctx->sc_mode == PR_SC_ARITH_F16 ? PR_F16_DISTANCE_BOUND : 0.0; }

int pr_f16_margin_dev(pr_ctx* ctx, const double* mom_sc, const double* mom_m2, int32_t m, int32_t G, double p_weight, int32_t k_in,
                      const double* cand_score, int32_t k, const double* score, int32_t* flags, int32_t* count) {
  if (!ctx) return PR_EINVAL;
  if ((!mom_sc && !mom_m2) || !cand_score || !score || !flags || !count || m < 0 || G < 1 || k < 1 || k_in < k || k_in > 128)
    PR_FAIL(ctx, PR_EINVAL, "pr_f16_margin_dev: bad arguments (m=%d, G=%d, k=%d, k_in=%d)", m, G, k, k_in);
  if (int rc = set_device(ctx)) return rc;
  pr::launch_zero_ints(ctx->stream, count, 1);
  pr::launch_margin_check(ctx->stream, mom_sc, mom_m2, G, m, p_weight, k_in, cand_score, k, score, PR_F16_DISTANCE_BOUND, flags, count,
                          ctx->order_m == m ? ctx->d_order : nullptr, PR_F16_SIGMA_REL, PR_F16_DIST_ERR);
  ctx->order_m = -1;
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

// grow-only scratch of the re-evaluation: `need` doubles (the p5 block of a single-shard call), m order flags + the list of flagged queries
// + its count ([3 cap + 2] ints: flags | list | count)
static int rerank_scratch(pr_ctx* ctx, size_t need, int32_t m) {
  if (need > ctx->rr_cap) {
    if (ctx->rr_scratch) { PR_HIP(ctx, hipStreamSynchronize(ctx->stream)); PR_HIP(ctx, hipFree(ctx->rr_scratch)); ctx->rr_scratch = nullptr; ctx->rr_cap = 0; }
    PR_HIP(ctx, hipMalloc((void**)&ctx->rr_scratch, need * sizeof(double)));
    ctx->rr_cap = need;
  }
  if ((size_t)m > ctx->order_cap) {
    if (ctx->d_order) { PR_HIP(ctx, hipStreamSynchronize(ctx->stream)); PR_HIP(ctx, hipFree(ctx->d_order)); ctx->d_order = nullptr; ctx->order_cap = 0; }
    PR_HIP(ctx, hipMalloc((void**)&ctx->d_order, ((size_t)2 * m + 2) * sizeof(int32_t)));
    ctx->order_cap = (size_t)m;
  }
  return PR_OK;
}
// the per-pair tickets of rerank_kernel ([pairs] zeros; the kernel leaves them zero), the work list of a batch ([pairs]) and its length
static int rerank_ticks(pr_ctx* ctx, size_t pairs) {
  if (pairs > ctx->tick_cap) {
    if (ctx->rr_tick) { PR_HIP(ctx, hipStreamSynchronize(ctx->stream)); PR_HIP(ctx, hipFree(ctx->rr_tick)); ctx->rr_tick = nullptr; ctx->tick_cap = 0; }
    PR_HIP(ctx, hipMalloc((void**)&ctx->rr_tick, (2 * pairs + 1) * sizeof(unsigned)));
    PR_HIP(ctx, hipMemsetAsync(ctx->rr_tick, 0, (2 * pairs + 1) * sizeof(unsigned), ctx->stream));
    ctx->tick_cap = pairs;
  }
  return PR_OK;
}
static int32_t* res_list(pr_ctx* ctx) { return ctx->d_order + ctx->order_cap; }
static int32_t* res_cnt(pr_ctx* ctx) { return ctx->d_order + 2 * ctx->order_cap; }
static unsigned* res_tick(pr_ctx* ctx) { return reinterpret_cast<unsigned*>(ctx->res_partial + (size_t)pr::RESOLVE_SLOTS * pr::RESOLVE_NB * 12); }
// scratch of the exact-row resolution: the workgroup partials of one pass, exact [m][4][3] (single-shard calls) ...
static int resolve_scratch_base(pr_ctx* ctx, int32_t m_exact) {
  if (!ctx->res_partial) {                                    // (+ the ticket of the last-workgroup hand-off, zero between launches)
    PR_HIP(ctx, hipMalloc((void**)&ctx->res_partial, ((size_t)pr::RESOLVE_SLOTS * pr::RESOLVE_NB * 12 + 1) * sizeof(double)));
    PR_HIP(ctx, hipMemsetAsync(ctx->res_partial + (size_t)pr::RESOLVE_SLOTS * pr::RESOLVE_NB * 12, 0, sizeof(double), ctx->stream));
  }
  if (!ctx->xqspec) PR_HIP(ctx, hipMalloc((void**)&ctx->xqspec, pr::xrow_qspec_doubles() * sizeof(double)));   // (its own check: a failed allocation is retried, never launched on)
  if ((size_t)m_exact > ctx->res_exact_cap) {
    if (ctx->res_exact) { PR_HIP(ctx, hipStreamSynchronize(ctx->stream)); PR_HIP(ctx, hipFree(ctx->res_exact)); ctx->res_exact = nullptr; ctx->res_exact_cap = 0; }
    PR_HIP(ctx, hipMalloc((void**)&ctx->res_exact, (size_t)m_exact * 12 * sizeof(double)));
    ctx->res_exact_cap = (size_t)m_exact;
  }
  return PR_OK;
}
// ... and the rows of one pass, [slots][4][n_local] doubles: `slots` = the queries the pass can hold (a stream-ordered call cannot know how many
// are flagged: min(m, 64); a call that has read the count back: min(count, 64), and nothing at all when nothing is flagged - the usual case)
static int resolve_scratch_rows(pr_ctx* ctx, int32_t slots, int32_t n_local) {
  const size_t need = (size_t)(slots < pr::RESOLVE_SLOTS ? slots : pr::RESOLVE_SLOTS) * 4 * (size_t)n_local;
  if (need > ctx->xrows_cap) {
    if (ctx->xrows) { PR_HIP(ctx, hipStreamSynchronize(ctx->stream)); PR_HIP(ctx, hipFree(ctx->xrows)); ctx->xrows = nullptr; ctx->xrows_cap = 0; }
    // a DB that grows by a row per call (pr_sigset_append) must not pay a free + malloc (~0.3 ms) per call: an eighth of headroom; the exact
    // size when the device cannot give that much
    size_t cap = need + need / 8;
    if (hipMalloc((void**)&ctx->xrows, cap * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); cap = need; PR_HIP(ctx, hipMalloc((void**)&ctx->xrows, cap * sizeof(double))); }
    ctx->xrows_cap = cap;
  }
  return PR_OK;
}
static int resolve_scratch(pr_ctx* ctx, int32_t m_exact, int32_t m, int32_t n_local) {
  if (int rc = resolve_scratch_base(ctx, m_exact)) return rc;
  return resolve_scratch_rows(ctx, m, n_local);
}

static int select_scratch_x(pr_ctx* ctx, int32_t m, int32_t k) {
  const size_t need = pr::xrow_select_part_doubles(m, k);
  if (need > ctx->xpart_cap) {
    if (ctx->xpart) { PR_HIP(ctx, hipStreamSynchronize(ctx->stream)); PR_HIP(ctx, hipFree(ctx->xpart)); ctx->xpart = nullptr; ctx->xpart_cap = 0; }
    PR_HIP(ctx, hipMalloc((void**)&ctx->xpart, need * sizeof(double)));
    ctx->xpart_cap = need;
  }
  return PR_OK;
}

int pr_rerank_width(const pr_ctx* ctx, int32_t k) { return ctx ? rerank_width(k, ctx->sc_mode) : PR_EINVAL; }

static bool order_consts(const pr_ctx* ctx, double& fl, double& noise) {
  const bool f16 = ctx->sc_mode == PR_SC_ARITH_F16;
  fl = f16 ? PR_F16_SIGMA_REL : PR_F32_SIGMA_REL;
  noise = f16 ? PR_F16_DIST_ERR : PR_F32_DIST_ERR;     // (the largest error of one distance of the pass)
  return f16;
}

int pr_rerank_dev(pr_ctx* ctx, const void* q_sc, const void* db_sc, int sc_dtype, const void* q_m2, const void* db_m2, int m2_dtype,
                  const double* mom_sc, const double* mom_m2, int32_t m, int32_t n_local, int32_t G, int32_t q_row0, int32_t db_row0,
                  int32_t mask_width, double p_weight, int32_t k_in, const int32_t* idx_in, const double* score_in, int32_t k, int32_t* idx,
                  double* score) {
  if (!ctx) return PR_EINVAL;
  const bool sc = q_sc || db_sc, m2 = q_m2 || db_m2;
  if ((!sc && !m2) || (sc && (!q_sc || !db_sc || !mom_sc)) || (m2 && (!q_m2 || !db_m2 || !mom_m2)) || !idx_in || !idx || !score ||
      m < 0 || n_local < 1 || G < 1 || k < 1 || k_in < k || k_in > 128 ||
      (sc && sc_dtype != PR_F64 && sc_dtype != PR_F32) || (m2 && m2_dtype != PR_F64 && m2_dtype != PR_F32))
    PR_FAIL(ctx, PR_EINVAL, "pr_rerank_dev: bad arguments (m=%d, n_local=%d, G=%d, k=%d, k_in=%d)", m, n_local, G, k, k_in);
  if (m == 0) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  // the candidates' scores and exact distances stay in the context (p5 layout) with one order flag per query: pr_order_resolve*_dev /
  // pr_f16_margin_dev take them
  if (int rc = rerank_scratch(ctx, (size_t)5 * m * k_in, m)) return rc;
  if (int rc = rerank_ticks(ctx, (size_t)m * k_in)) return rc;
  double fl, noise;
  order_consts(ctx, fl, noise);
  pr::launch_rerank(ctx->stream, q_sc, db_sc, sc_dtype, q_m2, db_m2, m2_dtype, mom_sc, mom_m2, m, n_local, G, q_row0, db_row0, mask_width,
                    p_weight, k_in, idx_in, ctx->rr_scratch, ctx->rr_tick, ctx->tick_cap, k, idx, score, nullptr, score_in, pass_eps(ctx), fl, noise, ctx->d_order);
  if (ctx->force_order) pr::launch_fill_ints(ctx->stream, ctx->d_order, m, 1);
  ctx->order_m = m;
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

// one pass (RESOLVE_SLOTS list slots from `offset`) of the single-shard resolution behind pr_rerank_dev: the rows + their moments, then the
// selection straight into idx / score (+ the compaction of the flags for calls of more than RESOLVE_SMALL_M queries)
static void resolve_pass(pr_ctx* ctx, const void* q_sc, const void* db_sc, int sc_dtype, const void* q_m2, const void* db_m2, int m2_dtype,
                         double* mom_sc, double* mom_m2, int32_t m, int32_t n, int32_t q_row0, int32_t mask_width, double p_weight, int32_t k,
                         int32_t* idx, double* score, int offset, bool compacted, int* dflags, int last_pass) {
  pr::launch_xrow(ctx->stream, q_sc, db_sc, sc_dtype, q_m2, db_m2, m2_dtype, mom_sc, mom_m2, 1, m, n, ctx->d_order, res_list(ctx), res_cnt(ctx),
                  offset, compacted, ctx->res_partial, ctx->res_exact, ctx->xrows, res_tick(ctx), dflags, last_pass, ctx->xqspec, ctx->d_twiddle,
                  ctx->xrow_direct);
  pr::launch_xrow_select(ctx->stream, ctx->d_order, res_list(ctx), res_cnt(ctx), offset, ctx->res_exact, 1, m, n, q_row0, 0, mask_width, p_weight,
                         q_sc != nullptr, q_m2 != nullptr, k, ctx->xrows, ctx->xpart, nullptr, idx, score, q_sc ? mom_sc : nullptr, q_m2 ? mom_m2 : nullptr);
}

static int resolve_args_ok(pr_ctx* ctx, const char* who, const void* q_sc, const void* db_sc, int sc_dtype, const void* q_m2, const void* db_m2,
                           int m2_dtype, const double* mom_sc, const double* mom_m2, int32_t m, int32_t n, int32_t k, const int32_t* idx,
                           const double* score) {
  const bool sc = q_sc || db_sc, m2 = q_m2 || db_m2;
  if ((!sc && !m2) || (sc && (!q_sc || !db_sc || !mom_sc)) || (m2 && (!q_m2 || !db_m2 || !mom_m2)) || !idx || !score || m < 0 || n < 1 ||
      k < 1 || k > 128 || (sc && sc_dtype != PR_F64 && sc_dtype != PR_F32) || (m2 && m2_dtype != PR_F64 && m2_dtype != PR_F32))
    PR_FAIL(ctx, PR_EINVAL, "%s: bad arguments (m=%d, n=%d, k=%d)", who, m, n, k);
  return PR_OK;
}

int pr_order_resolve_async_dev(pr_ctx* ctx, const void* q_sc, const void* db_sc, int sc_dtype, const void* q_m2, const void* db_m2, int m2_dtype,
                               double* mom_sc, double* mom_m2, int32_t m, int32_t n, int32_t q_row0, int32_t mask_width, double p_weight,
                               int32_t k, int32_t* idx, double* score) {
  if (!ctx) return PR_EINVAL;
  if (int rc = resolve_args_ok(ctx, "pr_order_resolve_async_dev", q_sc, db_sc, sc_dtype, q_m2, db_m2, m2_dtype, mom_sc, mom_m2, m, n, k, idx, score)) return rc;
  if (m == 0 || ctx->order_m != m) { ctx->order_m = -1; return PR_OK; }   // no (fresh) flags of a call of this shape
  if (int rc = set_device(ctx)) return rc;
  if (int rc = resolve_scratch(ctx, m, m, n)) return rc;
  if (int rc = select_scratch_x(ctx, m, k)) return rc;
  ctx->order_m = -1;
  // every pass the call could need, chained on the stream: ceil(m / 64) fixed-grid passes, each of whose kernels leaves at once when the
  // flagged list ends before its offset (the usual case: nothing flagged at all) - a captured graph resolves ALL flagged queries, whatever
  // their number (round 6; until then one pass, and PR_WARN_ORDER_UNRESOLVED beyond 64).  The first pass compacts the flags of a large call.
  const int passes = (m + pr::RESOLVE_SLOTS - 1) / pr::RESOLVE_SLOTS;
  for (int p = 0; p < passes; p++)
    resolve_pass(ctx, q_sc, db_sc, sc_dtype, q_m2, db_m2, m2_dtype, mom_sc, mom_m2, m, n, q_row0, mask_width, p_weight, k, idx, score,
                 p * pr::RESOLVE_SLOTS, p > 0 && m > pr::RESOLVE_SMALL_M, ctx->d_flags, p == passes - 1);
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

int pr_order_resolve_dev(pr_ctx* ctx, const void* q_sc, const void* db_sc, int sc_dtype, const void* q_m2, const void* db_m2, int m2_dtype,
                         double* mom_sc, double* mom_m2, int32_t m, int32_t n, int32_t q_row0, int32_t mask_width, double p_weight, int32_t k,
                         int32_t* idx, double* score, int32_t* resolved) {
  if (!ctx) return PR_EINVAL;
  if (int rc = resolve_args_ok(ctx, "pr_order_resolve_dev", q_sc, db_sc, sc_dtype, q_m2, db_m2, m2_dtype, mom_sc, mom_m2, m, n, k, idx, score)) return rc;
  if (resolved) *resolved = 0;
  if (m == 0 || ctx->order_m != m) { ctx->order_m = -1; return PR_OK; }
  if (int rc = set_device(ctx)) return rc;
  ctx->order_m = -1;
  pr::launch_flag_compact(ctx->stream, ctx->d_order, m, res_list(ctx), res_cnt(ctx));
  int32_t cnt = 0;
  PR_HIP(ctx, hipMemcpyAsync(&cnt, res_cnt(ctx), 4, hipMemcpyDeviceToHost, ctx->stream));
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (cnt > 0) {      // the scratch only when something is flagged, and rows for min(count, 64) queries (4 x n doubles each: 3.2 MB per query at n = 100k)
    if (int rc = resolve_scratch_base(ctx, m)) return rc;
    if (int rc = resolve_scratch_rows(ctx, cnt, n)) return rc;
    if (int rc = select_scratch_x(ctx, m, k)) return rc;
  }
  for (int off = 0; off < cnt; off += pr::RESOLVE_SLOTS)
    resolve_pass(ctx, q_sc, db_sc, sc_dtype, q_m2, db_m2, m2_dtype, mom_sc, mom_m2, m, n, q_row0, mask_width, p_weight, k, idx, score, off, true, nullptr, 0);
  PR_HIP(ctx, hipGetLastError());
  if (cnt > 0) { PR_HIP(ctx, hipStreamSynchronize(ctx->stream)); ctx->warnings |= PR_WARN_ORDER_RESOLVED; }
  if (resolved) *resolved = cnt;
  return PR_OK;
}

int pr_order_flagged_count(pr_ctx* ctx, int32_t m, int32_t* count) {
  if (!ctx || !count) return PR_EINVAL;
  *count = 0;
  if (m <= 0 || ctx->order_m != m) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  pr::launch_flag_compact(ctx->stream, ctx->d_order, m, res_list(ctx), res_cnt(ctx));
  PR_HIP(ctx, hipMemcpyAsync(count, res_cnt(ctx), 4, hipMemcpyDeviceToHost, ctx->stream));
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PR_OK;
}

int pr_rerank_partial_dev(pr_ctx* ctx, const void* q_sc, const void* db_sc, int sc_dtype, const void* q_m2, const void* db_m2, int m2_dtype,
                          const double* mom_sc, const double* mom_m2, int32_t m, int32_t n_local, int32_t G, int32_t q_row0, int32_t db_row0,
                          int32_t mask_width, double p_weight, int32_t k_in, const int32_t* cand_idx, const double* cand_score, int32_t k,
                          double* p5) {
  if (!ctx) return PR_EINVAL;
  const bool sc = q_sc || db_sc, m2 = q_m2 || db_m2;
  if ((!sc && !m2) || (sc && (!q_sc || !db_sc || !mom_sc)) || (m2 && (!q_m2 || !db_m2 || !mom_m2)) || !cand_idx || !p5 ||
      m < 0 || n_local < 1 || G < 1 || k_in < 1 || k_in > 128 || k < 1 || k > k_in ||
      (sc && sc_dtype != PR_F64 && sc_dtype != PR_F32) || (m2 && m2_dtype != PR_F64 && m2_dtype != PR_F32))
    PR_FAIL(ctx, PR_EINVAL, "pr_rerank_partial_dev: bad arguments (m=%d, n_local=%d, G=%d, k_in=%d)", m, n_local, G, k_in);
  if (m == 0) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  if (int rc = rerank_ticks(ctx, (size_t)m * k_in)) return rc;
  pr::launch_rerank_partial(ctx->stream, q_sc, db_sc, sc_dtype, q_m2, db_m2, m2_dtype, mom_sc, mom_m2, m, n_local, G, q_row0, db_row0,
                            mask_width, p_weight, k_in, cand_idx, p5, ctx->rr_tick, ctx->tick_cap, cand_score, k, pass_eps(ctx));
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

int pr_rerank_finish_dev(pr_ctx* ctx, const double* mom_sc, const double* mom_m2, int32_t G_mom, const int32_t* cand_idx, const double* cand_score,
                         const double* p5_all, int32_t G, int32_t m, int32_t k_in, int32_t k, double p_weight, int32_t* idx, double* score) {
  if (!ctx) return PR_EINVAL;
  if ((!mom_sc && !mom_m2) || !cand_idx || !p5_all || !idx || !score || G < 1 || G > 254 || G_mom < 1 || m < 0 || k < 1 || k_in < k || k_in > 128)
    PR_FAIL(ctx, PR_EINVAL, "pr_rerank_finish_dev: bad arguments (G=%d, m=%d, k=%d, k_in=%d)", G, m, k, k_in);
  if (m == 0) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  if (int rc = rerank_scratch(ctx, 0, m)) return rc;
  double fl, noise;
  order_consts(ctx, fl, noise);
  pr::launch_rerank_finish(ctx->stream, cand_idx, p5_all, G, m, k_in, k, idx, score);
  const bool contain = cand_score && ctx->sc_mode != PR_SC_ARITH_F16;      // (the single-product arithmetic: pr_f16_margin_dev)
  pr::launch_order_check(ctx->stream, mom_sc, mom_m2, G_mom, cand_idx, p5_all, G, m, k_in, k, idx, p_weight, fl, noise, ctx->d_order,
                         contain ? cand_score : nullptr, contain ? score : nullptr);
  if (ctx->force_order) pr::launch_fill_ints(ctx->stream, ctx->d_order, m, 1);
  ctx->order_m = m;
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

int pr_order_exact_moments_dev(pr_ctx* ctx, const void* q_sc, const void* db_sc, int sc_dtype, const void* q_m2, const void* db_m2, int m2_dtype,
                               const double* mom_sc, const double* mom_m2, int32_t G_mom, int32_t m, int32_t n_local, int32_t offset, int32_t last_pass,
                               double* exact) {
  if (!ctx) return PR_EINVAL;
  const bool sc = q_sc || db_sc, m2 = q_m2 || db_m2;
  if ((!sc && !m2) || (sc && (!q_sc || !db_sc || !mom_sc)) || (m2 && (!q_m2 || !db_m2 || !mom_m2)) || !exact || m < 0 || n_local < 1 || G_mom < 1 ||
      offset < 0 || (sc && sc_dtype != PR_F64 && sc_dtype != PR_F32) || (m2 && m2_dtype != PR_F64 && m2_dtype != PR_F32))
    PR_FAIL(ctx, PR_EINVAL, "pr_order_exact_moments_dev: bad arguments (m=%d, n_local=%d, G=%d)", m, n_local, G_mom);
  if (m == 0) return PR_OK;
  if (ctx->order_m != m) PR_FAIL(ctx, PR_EINVAL, "pr_order_exact_moments_dev: no flags of a %d-query pr_rerank_finish_dev on this context", m);
  if (int rc = set_device(ctx)) return rc;
  if (int rc = resolve_scratch(ctx, 0, m, n_local)) return rc;
  pr::launch_xrow(ctx->stream, q_sc, db_sc, sc_dtype, q_m2, db_m2, m2_dtype, mom_sc, mom_m2, G_mom, m, n_local, ctx->d_order, res_list(ctx),
                  res_cnt(ctx), offset, false, ctx->res_partial, exact, ctx->xrows, res_tick(ctx), ctx->d_flags, last_pass != 0, ctx->xqspec, ctx->d_twiddle,
                  ctx->xrow_direct);
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

int pr_order_exact_select_dev(pr_ctx* ctx, const double* exact_all, int32_t G, int32_t m, int32_t n_local, int32_t q_row0, int32_t db_row0,
                              int32_t mask_width, double p_weight, int has_sc, int has_m2, int32_t k, int32_t offset, double* sel) {
  if (!ctx) return PR_EINVAL;
  if (!exact_all || !sel || (!has_sc && !has_m2) || G < 1 || G > 64 || m < 0 || n_local < 1 || k < 1 || k > 128 || offset < 0)
    PR_FAIL(ctx, PR_EINVAL, "pr_order_exact_select_dev: bad arguments (G=%d, m=%d, n_local=%d, k=%d; G <= 64, k <= 128)", G, m, n_local, k);
  if (m == 0) return PR_OK;
  if (ctx->order_m != m || !ctx->xrows ||
      ctx->xrows_cap < (size_t)(m < pr::RESOLVE_SLOTS ? m : pr::RESOLVE_SLOTS) * 4 * (size_t)n_local)
    PR_FAIL(ctx, PR_EINVAL, "pr_order_exact_select_dev: no rows of a %d-query pr_order_exact_moments_dev over %d entries on this context", m, n_local);
  if (int rc = set_device(ctx)) return rc;
  if (int rc = select_scratch_x(ctx, m, k)) return rc;
  pr::launch_xrow_select(ctx->stream, ctx->d_order, res_list(ctx), res_cnt(ctx), offset, exact_all, G, m, n_local, q_row0, db_row0, mask_width,
                         p_weight, has_sc, has_m2, k, ctx->xrows, ctx->xpart, sel, nullptr, nullptr, nullptr, nullptr);
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

int pr_order_exact_merge_dev(pr_ctx* ctx, const double* sel_all, int32_t G, int32_t m, int32_t k, int32_t offset, int32_t* idx, double* score) {
  if (!ctx) return PR_EINVAL;
  if (!sel_all || !idx || !score || G < 1 || G > 64 || m < 0 || k < 1 || k > 128 || offset < 0)
    PR_FAIL(ctx, PR_EINVAL, "pr_order_exact_merge_dev: bad arguments (G=%d, m=%d, k=%d; G <= 64, k <= 128)", G, m, k);
  if (m == 0) return PR_OK;
  if (ctx->order_m != m) PR_FAIL(ctx, PR_EINVAL, "pr_order_exact_merge_dev: no flagged-query list of a %d-query call on this context", m);
  if (int rc = set_device(ctx)) return rc;
  pr::launch_xrow_merge(ctx->stream, ctx->d_order, res_list(ctx), res_cnt(ctx), offset, sel_all, G, m, k, idx, score);
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

int pr_widen_scores_dev(pr_ctx* ctx, const float* score32, int64_t count, double* score64) {
  if (!ctx) return PR_EINVAL;
  if (count < 0 || (count > 0 && (!score32 || !score64))) PR_FAIL(ctx, PR_EINVAL, "pr_widen_scores_dev: bad arguments");
  if (int rc = set_device(ctx)) return rc;
  pr::launch_widen(ctx->stream, score32, count, score64);
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

int pr_merge_topk_dev(pr_ctx* ctx, const int32_t* idx_all, const double* score_all, int32_t G, int32_t m, int32_t k, int32_t* idx,
                      double* score) {
  if (!ctx) return PR_EINVAL;
  if (!idx_all || !score_all || !idx || !score || G < 1 || G > 64 || m < 0 || k < 1 || k > 128)
    PR_FAIL(ctx, PR_EINVAL, "pr_merge_topk_dev: bad arguments (G=%d, m=%d, k=%d; G <= 64, k <= 128)", G, m, k);
  if (int rc = set_device(ctx)) return rc;
  pr::launch_merge_topk(ctx->stream, idx_all, score_all, G, m, k, idx, score);
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

// ------------------------------------------------------------------------------------------- host-buffer path


// PR_SC_ARITH_F16, host calls: margin check of the re-evaluated top-k; the flagged queries are matched again in split-f16 against the whole
// DB (whose raw signatures are still on the device) and their rows of dcand / dsc64 overwritten.  hq_*: HOST query signatures, ddb_*: DEVICE
// raw DB signatures (f64), either descriptor type may be absent; mom_*: the f16 pass moments [m][2][3]; cand_sc: its candidate scores.
static int f16_fallback(pr_ctx* ctx, const double* hq_sc, const double* hq_m2, const void* ddb_sc, const void* ddb_m2, int32_t m, int32_t n,
                        int32_t mask_width, double p_weight, int32_t k, const double* mom_sc, const double* mom_m2, int32_t kin,
                        const double* cand_sc, int32_t* dcand, double* dsc64) {
  DevScope scope_(ctx);
  DevBuf dflags, dcount;
  if (dflags.alloc((size_t)m * 4) != hipSuccess || dcount.alloc(4) != hipSuccess) PR_FAIL(ctx, PR_ENOMEM, "out of device memory");
  if (int rc = pr_f16_margin_dev(ctx, mom_sc, mom_m2, m, 1, p_weight, kin, cand_sc, k, dsc64, dflags.as<int32_t>(), dcount.as<int32_t>())) return rc;
  int32_t cnt = 0;
  PR_HIP(ctx, hipMemcpyAsync(&cnt, dcount.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (cnt == 0) return PR_OK;
  std::vector<int32_t> fl(m), F;
  PR_HIP(ctx, hipMemcpy(fl.data(), dflags.p, (size_t)m * 4, hipMemcpyDeviceToHost));
  for (int32_t q = 0; q < m; q++) if (fl[q]) F.push_back(q);
  const int32_t mf = (int32_t)F.size();
  ctx->warnings |= PR_WARN_F16_FALLBACK;
  const int kin2 = rerank_width(k, PR_SC_ARITH_F16X2);
  pr_sigset* ss[4] = {nullptr, nullptr, nullptr, nullptr};   // SC query, SC db, M2DP query, M2DP db
  DevBuf rq[2], d[4], mo[2], didx, dsc, dsw;
  int rc = PR_OK;
  ctx->sc_mode = PR_SC_ARITH_F16X2;
  do {
    const size_t mn = (size_t)mf * n;
    for (int t = 0; t < 2 && rc == PR_OK; t++) {                // t = 0: SC, 1: M2DP
      const double* hq = t ? hq_m2 : hq_sc;
      const void* ddb = t ? ddb_m2 : ddb_sc;
      if (!hq) continue;
      const size_t rowlen = t ? 4 * 384 : 2400;
      std::vector<double> g((size_t)mf * rowlen);
      for (int32_t i = 0; i < mf; i++) memcpy(g.data() + (size_t)i * rowlen, hq + (size_t)F[i] * rowlen, rowlen * 8);
      if (rq[t].alloc(g.size() * 8) != hipSuccess || d[2 * t].alloc(mn * 4) != hipSuccess || d[2 * t + 1].alloc(mn * 4) != hipSuccess ||
          mo[t].alloc((size_t)mf * 6 * 8) != hipSuccess) { ctx->err = "out of device memory (f16 fallback)"; rc = PR_ENOMEM; break; }
      if (hipMemcpy(rq[t].p, g.data(), g.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { ctx->err = "H2D copy failed"; rc = PR_EHIP; break; }
      if ((rc = pr_sigset_create(ctx, t ? PR_TYPE_M2DP : PR_TYPE_SC, PR_ROLE_QUERY, mf, &ss[2 * t])) ||
          (rc = pr_sigset_create(ctx, t ? PR_TYPE_M2DP : PR_TYPE_SC, PR_ROLE_DB, n, &ss[2 * t + 1])) ||
          (rc = pr_sigset_pack(ctx, ss[2 * t], rq[t].p, PR_F64, PR_DEVICE, mf)) || (rc = pr_sigset_pack(ctx, ss[2 * t + 1], ddb, PR_F64, PR_DEVICE, n)) ||
          (rc = pr_distances_dev(ctx, ss[2 * t], ss[2 * t + 1], d[2 * t].as<float>(), d[2 * t + 1].as<float>())) ||
          (rc = pr_row_moments_dev(ctx, d[2 * t].as<float>(), d[2 * t + 1].as<float>(), mf, n, mo[t].as<double>()))) break;
    }
    if (rc) break;
    // the flagged queries in RUNS: a query's own row number only enters through the mask (run_test.m:47-53), so without a mask the whole list
    // is one batch, and with one every maximal run of consecutive row numbers is (a vehicle standing still flags its frames in a row) - one
    // selection, one re-evaluation and one order resolution per run instead of per query (round 6; until then one query at a time, each
    // with its own host synchronisation in pr_order_resolve_dev)
    std::vector<int32_t> run_end;                                // run r = list positions [run_end[r - 1], run_end[r])
    for (int32_t i = 1; i <= mf; i++)
      if (i == mf || (mask_width > 0 && F[i] != F[i - 1] + 1)) run_end.push_back(i);
    int32_t longest = 0;
    for (size_t r = 0, a0 = 0; r < run_end.size(); a0 = run_end[r++]) longest = std::max<int32_t>(longest, run_end[r] - (int32_t)a0);
    DevBuf oidx, osc;                                            // a run's results [len][k] (the list is ascending, not contiguous, without a mask)
    if (didx.alloc((size_t)longest * kin2 * 4) != hipSuccess || dsc.alloc((size_t)longest * kin2 * 4) != hipSuccess ||
        dsw.alloc((size_t)longest * kin2 * 8) != hipSuccess || oidx.alloc((size_t)longest * k * 4) != hipSuccess ||
        osc.alloc((size_t)longest * k * 8) != hipSuccess) { ctx->err = "out of device memory"; rc = PR_ENOMEM; break; }
    const bool both = hq_sc && hq_m2;
    for (size_t r = 0, a0 = 0; r < run_end.size() && rc == PR_OK; a0 = run_end[r++]) {
      const int32_t i0 = (int32_t)a0, len = run_end[r] - i0;
      const size_t o = (size_t)i0 * n;
      const int32_t q0 = mask_width > 0 ? F[i0] : 0;             // (without a mask the row numbers are not read)
      if (both)
        rc = pr_fuse_select2_dev(ctx, d[0].as<float>() + o, d[1].as<float>() + o, d[2].as<float>() + o, d[3].as<float>() + o, len, n,
                                 mo[0].as<double>() + (size_t)i0 * 6, mo[1].as<double>() + (size_t)i0 * 6, 1, q0, 0, mask_width, p_weight, kin2,
                                 didx.as<int32_t>(), dsc.as<float>());
      else {
        const int t = hq_sc ? 0 : 1;
        rc = pr_fuse_select_dev(ctx, d[2 * t].as<float>() + o, d[2 * t + 1].as<float>() + o, len, n, mo[t].as<double>() + (size_t)i0 * 6, 1, q0, 0,
                                mask_width, p_weight, kin2, didx.as<int32_t>(), dsc.as<float>());
      }
      if (rc || (rc = pr_widen_scores_dev(ctx, dsc.as<float>(), (int64_t)len * kin2, dsw.as<double>()))) break;
      rc = pr_rerank_dev(ctx, hq_sc ? rq[0].as<double>() + (size_t)i0 * 2400 : nullptr, hq_sc ? ddb_sc : nullptr, PR_F64,
                         hq_m2 ? rq[1].as<double>() + (size_t)i0 * 4 * 384 : nullptr, hq_m2 ? ddb_m2 : nullptr, PR_F64,
                         hq_sc ? mo[0].as<double>() + (size_t)i0 * 6 : nullptr, hq_m2 ? mo[1].as<double>() + (size_t)i0 * 6 : nullptr, len, n, 1, q0, 0,
                         mask_width, p_weight, kin2, didx.as<int32_t>(), dsw.as<double>(), k, oidx.as<int32_t>(), osc.as<double>());
      if (rc == PR_OK)                                            // (the split pass's own order check: exact row statistics if it fails)
        rc = pr_order_resolve_dev(ctx, hq_sc ? rq[0].as<double>() + (size_t)i0 * 2400 : nullptr, hq_sc ? ddb_sc : nullptr, PR_F64,
                                  hq_m2 ? rq[1].as<double>() + (size_t)i0 * 4 * 384 : nullptr, hq_m2 ? ddb_m2 : nullptr, PR_F64,
                                  hq_sc ? mo[0].as<double>() + (size_t)i0 * 6 : nullptr, hq_m2 ? mo[1].as<double>() + (size_t)i0 * 6 : nullptr, len, n, q0,
                                  mask_width, p_weight, k, oidx.as<int32_t>(), osc.as<double>(), nullptr);
      for (int32_t i = 0; i < len && rc == PR_OK; ) {            // results -> the flagged queries' rows of dcand / dsc64, one copy per stretch of consecutive rows
        int32_t j = i + 1;
        while (j < len && F[i0 + j] == F[i0 + j - 1] + 1) j++;
        if (hipMemcpyAsync(dcand + (size_t)F[i0 + i] * k, oidx.as<int32_t>() + (size_t)i * k, (size_t)(j - i) * k * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(dsc64 + (size_t)F[i0 + i] * k, osc.as<double>() + (size_t)i * k, (size_t)(j - i) * k * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
          ctx->err = "f16 fallback: device copy failed"; rc = PR_EHIP;
        }
        i = j;
      }
    }
    if (rc == PR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "f16 fallback failed"; rc = PR_EHIP; }
  } while (0);
  if (rc != PR_OK) (void)hipStreamSynchronize(ctx->stream);
  for (auto* q : ss) pr_sigset_destroy(ctx, q);
  ctx->sc_mode = PR_SC_ARITH_F16;
  return rc;
}

// score32 / score64: exactly one is non-null
static int distance_host(pr_ctx* ctx, int type, const double* h1, int32_t m, const double* h2, int32_t n,
                         float* out_p, float* out_i, int32_t mask_width, double p_weight, int32_t k, int32_t* idx,
                         float* score32, double* score64, bool want_topk) {
  if (!ctx) return PR_EINVAL;
  if (m < 0 || n < 0 || (m > 0 && !h1) || (n > 0 && !h2)) PR_FAIL(ctx, PR_EINVAL, "bad signature buffers (m=%d, n=%d)", m, n);
  const bool plain = (type == PR_TYPE_DELIGHT);
  if (want_topk && (k < 1 || (!plain && k > 120) || !idx || (!score32 && !score64)))   // (k + 8 candidates of the re-evaluation <= 128; DELIGHT has none)
    PR_FAIL(ctx, PR_EINVAL, "pr_match_topk needs 1 <= k <= 120 (SC, M2DP) and output buffers");
  if (want_topk && n == 0) {   // nothing to match against: every query row is "no candidate"
    for (size_t i = 0; i < (size_t)m * k; i++) { idx[i] = -1; if (score32) score32[i] = NAN; else score64[i] = NAN; }
    return PR_OK;
  }
  if (want_topk && n < 2 && !plain)
    PR_FAIL(ctx, PR_EINVAL, "pr_match_topk needs n >= 2 (N-1 standard deviation)");
  if (m == 0 || n == 0) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  const size_t rows_per = (type == PR_TYPE_SC) ? 1 : (type == PR_TYPE_M2DP ? 4 : 16);
  const size_t cols = (type == PR_TYPE_SC) ? PR_SC_SIG_LEN : (type == PR_TYPE_M2DP ? PR_M2DP_SIG_LEN : PR_DELIGHT_SIG_LEN);
  pr_sigset *q = nullptr, *d = nullptr;
  int rc = pr_sigset_create(ctx, type, PR_ROLE_QUERY, m, &q);
  if (rc == PR_OK) rc = pr_sigset_create(ctx, type, PR_ROLE_DB, n, &d);
  DevScope scope_(ctx);
  DevBuf raw1, raw2, dp, di, mom, didx, dsc, dsc64, dcand;
  do {
    if (rc) break;
    // the raw signatures stay on the device for the fp64 re-evaluation of the selection's survivors
    const size_t b1 = (size_t)m * rows_per * cols * 8, b2 = (size_t)n * rows_per * cols * 8;
    if (raw1.alloc(b1) != hipSuccess || raw2.alloc(b2) != hipSuccess) { ctx->err = "out of device memory for the signatures"; rc = PR_ENOMEM; break; }
    if (hipMemcpyAsync(raw1.p, h1, b1, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(raw2.p, h2, b2, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "H2D copy failed"; rc = PR_EHIP; break; }
    if ((rc = pr_sigset_pack(ctx, q, raw1.p, PR_F64, PR_DEVICE, m))) break;
    if ((rc = pr_sigset_pack(ctx, d, raw2.p, PR_F64, PR_DEVICE, n))) break;
    const size_t mn = (size_t)m * n;
    if (dp.alloc(mn * 4) != hipSuccess || (!plain && di.alloc(mn * 4) != hipSuccess)) { ctx->err = "out of device memory for the m x n distance matrices"; rc = PR_ENOMEM; break; }
    if ((rc = pr_distances_dev(ctx, q, d, dp.as<float>(), plain ? nullptr : di.as<float>()))) break;
    if (want_topk) {
      const int kin = plain ? k : rerank_width(k, ctx->sc_mode);
      if (mom.alloc((size_t)m * 6 * 8) != hipSuccess || didx.alloc((size_t)m * kin * 4) != hipSuccess ||
          dsc.alloc((size_t)m * kin * 4) != hipSuccess || dsc64.alloc((size_t)m * k * 8) != hipSuccess ||
          dcand.alloc((size_t)m * k * 4) != hipSuccess) { ctx->err = "out of device memory"; rc = PR_ENOMEM; break; }
      if (!plain && (rc = pr_row_moments_dev(ctx, dp.as<float>(), di.as<float>(), m, n, mom.as<double>()))) break;
      if ((rc = pr_fuse_select_dev(ctx, dp.as<float>(), plain ? nullptr : di.as<float>(), m, n, mom.as<double>(), 1, 0, 0, mask_width,
                                   p_weight, kin, didx.as<int32_t>(), dsc.as<float>()))) break;
      std::vector<float> tmp;
      if (plain) {   // DELIGHT: a single fp32 chi-square matrix, no fusion (run_test.m:26-36) - nothing to re-evaluate
        if (score64) tmp.resize((size_t)m * k);
        if (hipMemcpyAsync(idx, didx.p, (size_t)m * k * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(score32 ? score32 : tmp.data(), dsc.p, (size_t)m * k * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "D2H copy failed"; rc = PR_EHIP; break; }
        if ((rc = pr_sync(ctx))) break;
        if (score64) for (size_t i = 0; i < tmp.size(); i++) score64[i] = (double)tmp[i];
        break;
      }
      const bool sc = type == PR_TYPE_SC;
      DevBuf dsw;   // the fp32-pass scores as doubles: the re-evaluation skips candidates that cannot reach the top-k
      if (dsw.alloc((size_t)m * kin * 8) != hipSuccess) { ctx->err = "out of device memory"; rc = PR_ENOMEM; break; }
      if ((rc = pr_widen_scores_dev(ctx, dsc.as<float>(), (int64_t)m * kin, dsw.as<double>()))) break;
      if ((rc = pr_rerank_dev(ctx, sc ? raw1.p : nullptr, sc ? raw2.p : nullptr, PR_F64, sc ? nullptr : raw1.p, sc ? nullptr : raw2.p, PR_F64,
                              sc ? mom.as<double>() : nullptr, sc ? nullptr : mom.as<double>(), m, n, 1, 0, 0, mask_width, p_weight, kin,
                              didx.as<int32_t>(), dsw.as<double>(), k, dcand.as<int32_t>(), dsc64.as<double>()))) break;
      std::vector<double> t64;
      if (score32) t64.resize((size_t)m * k);
      // queries whose re-evaluated order hangs on the fp32 pass's sigmas get their row statistics in fp64 (PR_SC_ARITH_F16: the margin
      // check below takes the flags instead and sends them to the split pass, which resolves its own)
      if (ctx->sc_mode != PR_SC_ARITH_F16 &&
          (rc = pr_order_resolve_dev(ctx, sc ? raw1.p : nullptr, sc ? raw2.p : nullptr, PR_F64, sc ? nullptr : raw1.p, sc ? nullptr : raw2.p, PR_F64,
                                     sc ? mom.as<double>() : nullptr, sc ? nullptr : mom.as<double>(), m, n, 0, mask_width, p_weight, k,
                                     dcand.as<int32_t>(), dsc64.as<double>(), nullptr))) break;
      if (ctx->sc_mode == PR_SC_ARITH_F16 &&
          (rc = f16_fallback(ctx, sc ? h1 : nullptr, sc ? nullptr : h1, sc ? raw2.p : nullptr, sc ? nullptr : raw2.p, m, n, mask_width, p_weight, k,
                             sc ? mom.as<double>() : nullptr, sc ? nullptr : mom.as<double>(), kin, dsw.as<double>(), dcand.as<int32_t>(),
                             dsc64.as<double>()))) break;
      if (hipMemcpyAsync(idx, dcand.p, (size_t)m * k * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(score64 ? score64 : t64.data(), dsc64.p, (size_t)m * k * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "D2H copy failed"; rc = PR_EHIP; break; }
      if ((rc = pr_sync(ctx))) break;
      if (score32) for (size_t i = 0; i < t64.size(); i++) score32[i] = (float)t64[i];
      break;
    }
    if (out_p && hipMemcpyAsync(out_p, dp.p, mn * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "D2H copy failed"; rc = PR_EHIP; break; }
    if (out_i && !plain && hipMemcpyAsync(out_i, di.p, mn * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "D2H copy failed"; rc = PR_EHIP; break; }
    rc = pr_sync(ctx);
  } while (0);
  if (rc != PR_OK) (void)hipStreamSynchronize(ctx->stream);
  pr_sigset_destroy(ctx, q);
  pr_sigset_destroy(ctx, d);
  return rc;
}

int pr_sc_distance(pr_ctx* ctx, const double* h1, int32_t m, const double* h2, int32_t n, float* d_struct, float* d_int) {
  return distance_host(ctx, PR_TYPE_SC, h1, m, h2, n, d_struct, d_int, 0, 0, 0, nullptr, nullptr, nullptr, false);
}

int pr_m2dp_distance(pr_ctx* ctx, const double* h1, int32_t m, const double* h2, int32_t n, float* d_cnt, float* d_int) {
  return distance_host(ctx, PR_TYPE_M2DP, h1, m, h2, n, d_cnt, d_int, 0, 0, 0, nullptr, nullptr, nullptr, false);
}

int pr_delight_distance(pr_ctx* ctx, const double* h1, int32_t m, const double* h2, int32_t n, float* dist) {
  return distance_host(ctx, PR_TYPE_DELIGHT, h1, m, h2, n, dist, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, false);
}

int pr_match_topk(pr_ctx* ctx, int type, const double* h1, int32_t m, const double* h2, int32_t n, int32_t mask_width,
                  double p_weight, int32_t k, int32_t* idx, float* score) {
  if (!ctx) return PR_EINVAL;
  if (type != PR_TYPE_SC && type != PR_TYPE_M2DP && type != PR_TYPE_DELIGHT) PR_FAIL(ctx, PR_EINVAL, "pr_match_topk: unknown type %d", type);
  if (!score) PR_FAIL(ctx, PR_EINVAL, "pr_match_topk: score is NULL");
  return distance_host(ctx, type, h1, m, h2, n, nullptr, nullptr, mask_width, p_weight, k, idx, score, nullptr, true);
}

int pr_match_topk_f64(pr_ctx* ctx, int type, const double* h1, int32_t m, const double* h2, int32_t n, int32_t mask_width,
                      double p_weight, int32_t k, int32_t* idx, double* score) {
  if (!ctx) return PR_EINVAL;
  if (type != PR_TYPE_SC && type != PR_TYPE_M2DP && type != PR_TYPE_DELIGHT) PR_FAIL(ctx, PR_EINVAL, "pr_match_topk_f64: unknown type %d", type);
  if (!score) PR_FAIL(ctx, PR_EINVAL, "pr_match_topk_f64: score is NULL");
  return distance_host(ctx, type, h1, m, h2, n, nullptr, nullptr, mask_width, p_weight, k, idx, nullptr, score, true);
}

static int fuse_select2_impl(pr_ctx* ctx, const float* d_p, const float* d_i, const float* e_p, const float* e_i, int32_t m, int32_t n,
                             const double* mom_all, const double* mom2_all, int32_t G, int32_t q_row0, int32_t db_row0,
                             int32_t mask_width, double p_weight, int32_t k, int32_t* idx, float* score, double* score64) {
  if (!ctx || !d_p || !d_i || !e_p || !e_i || !mom_all || !mom2_all || !idx || !score || m < 0 || n < 1 || G < 1 || k < 1) return PR_EINVAL;
  if (int rc = set_device(ctx)) return rc;
  pr::launch_fuse_select(ctx->stream, d_p, d_i, m, n, mom_all, G, q_row0, db_row0, mask_width, p_weight, k, idx, score, e_p, e_i, mom2_all, ctx->sel_scratch, score64);
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}
int pr_fuse_select2_dev(pr_ctx* ctx, const float* d_p, const float* d_i, const float* e_p, const float* e_i, int32_t m, int32_t n,
                        const double* mom_all, const double* mom2_all, int32_t G, int32_t q_row0, int32_t db_row0,
                        int32_t mask_width, double p_weight, int32_t k, int32_t* idx, float* score) {
  return fuse_select2_impl(ctx, d_p, d_i, e_p, e_i, m, n, mom_all, mom2_all, G, q_row0, db_row0, mask_width, p_weight, k, idx, score, nullptr);
}
int pr_fuse_select2_f64_dev(pr_ctx* ctx, const float* d_p, const float* d_i, const float* e_p, const float* e_i, int32_t m, int32_t n,
                            const double* mom_all, const double* mom2_all, int32_t G, int32_t q_row0, int32_t db_row0,
                            int32_t mask_width, double p_weight, int32_t k, int32_t* idx, float* score, double* score64) {
  if (!score64) return PR_EINVAL;
  return fuse_select2_impl(ctx, d_p, d_i, e_p, e_i, m, n, mom_all, mom2_all, G, q_row0, db_row0, mask_width, p_weight, k, idx, score, score64);
}

// BASELINE.json config 5: SC and M2DP signatures of the same places scored together (build-defined)
static int fused_host(pr_ctx* ctx, const double* sc1, const double* m2dp1, int32_t m, const double* sc2, const double* m2dp2,
                      int32_t n, int32_t mask_width, double p_weight, int32_t k, int32_t* idx, float* score32, double* score64) {
  if (!ctx) return PR_EINVAL;
  if (m < 0 || n < 2 || k < 1 || k > 120 || !idx || (!score32 && !score64) || (m > 0 && (!sc1 || !m2dp1)) || !sc2 || !m2dp2)
    PR_FAIL(ctx, PR_EINVAL, "pr_match_topk_fused: bad arguments (m=%d, n=%d, k=%d)", m, n, k);
  if (m == 0) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  pr_sigset* ss[4] = {nullptr, nullptr, nullptr, nullptr};   // SC query, SC db, M2DP query, M2DP db
  DevScope scope_(ctx);
  DevBuf raw[4], d[4], mom[2], didx, dsc, dcand, dsc64, dsw;
  const void* host[4] = {sc1, sc2, m2dp1, m2dp2};
  const size_t bytes[4] = {(size_t)m * 2400 * 8, (size_t)n * 2400 * 8, (size_t)m * 4 * 384 * 8, (size_t)n * 4 * 384 * 8};
  const int kin = rerank_width(k, ctx->sc_mode);
  int rc = PR_OK;
  do {
    if ((rc = pr_sigset_create(ctx, PR_TYPE_SC, PR_ROLE_QUERY, m, &ss[0])) || (rc = pr_sigset_create(ctx, PR_TYPE_SC, PR_ROLE_DB, n, &ss[1])) ||
        (rc = pr_sigset_create(ctx, PR_TYPE_M2DP, PR_ROLE_QUERY, m, &ss[2])) || (rc = pr_sigset_create(ctx, PR_TYPE_M2DP, PR_ROLE_DB, n, &ss[3]))) break;
    bool ok = true;
    for (int i = 0; i < 4; i++)
      ok = ok && raw[i].alloc(bytes[i]) == hipSuccess &&
           hipMemcpyAsync(raw[i].p, host[i], bytes[i], hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    if (!ok) { ctx->err = "out of device memory for the signatures"; rc = PR_ENOMEM; break; }
    if ((rc = pr_sigset_pack(ctx, ss[0], raw[0].p, PR_F64, PR_DEVICE, m)) || (rc = pr_sigset_pack(ctx, ss[1], raw[1].p, PR_F64, PR_DEVICE, n)) ||
        (rc = pr_sigset_pack(ctx, ss[2], raw[2].p, PR_F64, PR_DEVICE, m)) || (rc = pr_sigset_pack(ctx, ss[3], raw[3].p, PR_F64, PR_DEVICE, n))) break;
    const size_t mn = (size_t)m * n;
    for (auto& b : d) ok = ok && b.alloc(mn * 4) == hipSuccess;
    ok = ok && mom[0].alloc((size_t)m * 6 * 8) == hipSuccess && mom[1].alloc((size_t)m * 6 * 8) == hipSuccess &&
         didx.alloc((size_t)m * kin * 4) == hipSuccess && dsc.alloc((size_t)m * kin * 4) == hipSuccess &&
         dcand.alloc((size_t)m * k * 4) == hipSuccess && dsc64.alloc((size_t)m * k * 8) == hipSuccess && dsw.alloc((size_t)m * kin * 8) == hipSuccess;
    if (!ok) { ctx->err = "out of device memory for the four m x n distance matrices"; rc = PR_ENOMEM; break; }
    if ((rc = pr_distances_dev(ctx, ss[0], ss[1], d[0].as<float>(), d[1].as<float>())) ||
        (rc = pr_distances_dev(ctx, ss[2], ss[3], d[2].as<float>(), d[3].as<float>())) ||
        (rc = pr_row_moments_dev(ctx, d[0].as<float>(), d[1].as<float>(), m, n, mom[0].as<double>())) ||
        (rc = pr_row_moments_dev(ctx, d[2].as<float>(), d[3].as<float>(), m, n, mom[1].as<double>())) ||
        (rc = pr_fuse_select2_dev(ctx, d[0].as<float>(), d[1].as<float>(), d[2].as<float>(), d[3].as<float>(), m, n, mom[0].as<double>(),
                                  mom[1].as<double>(), 1, 0, 0, mask_width, p_weight, kin, didx.as<int32_t>(), dsc.as<float>())) ||
        (rc = pr_widen_scores_dev(ctx, dsc.as<float>(), (int64_t)m * kin, dsw.as<double>())) ||
        (rc = pr_rerank_dev(ctx, raw[0].p, raw[1].p, PR_F64, raw[2].p, raw[3].p, PR_F64, mom[0].as<double>(), mom[1].as<double>(), m, n, 1,
                            0, 0, mask_width, p_weight, kin, didx.as<int32_t>(), dsw.as<double>(), k, dcand.as<int32_t>(), dsc64.as<double>()))) break;
    std::vector<double> t64;
    if (score32) t64.resize((size_t)m * k);
    if (ctx->sc_mode != PR_SC_ARITH_F16 &&
        (rc = pr_order_resolve_dev(ctx, raw[0].p, raw[1].p, PR_F64, raw[2].p, raw[3].p, PR_F64, mom[0].as<double>(), mom[1].as<double>(), m, n, 0,
                                   mask_width, p_weight, k, dcand.as<int32_t>(), dsc64.as<double>(), nullptr))) break;
    if (ctx->sc_mode == PR_SC_ARITH_F16 &&
        (rc = f16_fallback(ctx, sc1, m2dp1, raw[1].p, raw[3].p, m, n, mask_width, p_weight, k, mom[0].as<double>(), mom[1].as<double>(), kin,
                           dsw.as<double>(), dcand.as<int32_t>(), dsc64.as<double>()))) break;
    if (hipMemcpyAsync(idx, dcand.p, (size_t)m * k * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(score64 ? score64 : t64.data(), dsc64.p, (size_t)m * k * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "D2H copy failed"; rc = PR_EHIP; break; }
    if ((rc = pr_sync(ctx))) break;
    if (score32) for (size_t i = 0; i < t64.size(); i++) score32[i] = (float)t64[i];
  } while (0);
  if (rc != PR_OK) (void)hipStreamSynchronize(ctx->stream);
  for (auto* q : ss) pr_sigset_destroy(ctx, q);
  return rc;
}

int pr_match_topk_fused(pr_ctx* ctx, const double* sc1, const double* m2dp1, int32_t m, const double* sc2, const double* m2dp2,
                        int32_t n, int32_t mask_width, double p_weight, int32_t k, int32_t* idx, float* score) {
  return fused_host(ctx, sc1, m2dp1, m, sc2, m2dp2, n, mask_width, p_weight, k, idx, score, nullptr);
}

int pr_match_topk_fused_f64(pr_ctx* ctx, const double* sc1, const double* m2dp1, int32_t m, const double* sc2, const double* m2dp2,
                            int32_t n, int32_t mask_width, double p_weight, int32_t k, int32_t* idx, double* score) {
  return fused_host(ctx, sc1, m2dp1, m, sc2, m2dp2, n, mask_width, p_weight, k, idx, nullptr, score);
}

// GIST / BoW (run_test.m:32-35): one distance matrix from raw f64 rows of `cols` columns, no packing, no fusion
static int plain_cols_host(pr_ctx* ctx, int type, const double* h1, int32_t m, const double* h2, int32_t n, int32_t cols,
                           float* out, int32_t mask_width, int32_t k, int32_t* idx, float* score, bool want_topk) {
  if (!ctx) return PR_EINVAL;
  if (type != PR_TYPE_GIST && type != PR_TYPE_BOW) PR_FAIL(ctx, PR_EINVAL, "unknown plain type %d", type);
  if (m < 0 || n < 0 || cols < 1 || (m > 0 && !h1) || (n > 0 && !h2) || m > PR_MAX_SIGS || n > PR_MAX_SIGS)
    PR_FAIL(ctx, PR_EINVAL, "bad signature buffers (m=%d, n=%d, cols=%d)", m, n, cols);
  if (type == PR_TYPE_BOW && (size_t)cols * 16 > 160 * 1024) PR_FAIL(ctx, PR_EINVAL, "BoW rows of %d columns do not fit the LDS", cols);
  if (want_topk && (k < 1 || !idx || !score)) PR_FAIL(ctx, PR_EINVAL, "top-k needs k >= 1 and output buffers");
  if (want_topk && n == 0) { for (size_t i = 0; i < (size_t)m * k; i++) { idx[i] = -1; score[i] = NAN; } return PR_OK; }
  if (m == 0 || n == 0) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  const size_t rows1 = (type == PR_TYPE_BOW ? 2 : 1) * (size_t)m, rows2 = (type == PR_TYPE_BOW ? 2 : 1) * (size_t)n;
  const size_t mn = (size_t)m * n;
  DevScope scope_(ctx);
  DevBuf d1, d2, dd, didx, dsc;
  if (d1.alloc(rows1 * cols * 8) != hipSuccess || d2.alloc(rows2 * cols * 8) != hipSuccess || dd.alloc(mn * 4) != hipSuccess)
    PR_FAIL(ctx, PR_ENOMEM, "out of device memory for %d x %d %s signatures", m, n, type == PR_TYPE_BOW ? "BoW" : "GIST");
  int rc = PR_OK;
  do {
    if (hipMemcpyAsync(d1.p, h1, rows1 * cols * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d2.p, h2, rows2 * cols * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "H2D copy failed"; rc = PR_EHIP; break; }
    if (type == PR_TYPE_GIST) pr::launch_gist_distance(ctx->stream, d1.as<double>(), m, d2.as<double>(), n, cols, dd.as<float>());
    else pr::launch_bow_distance(ctx->stream, d1.as<double>(), m, d2.as<double>(), n, cols, dd.as<float>());
    if (hipGetLastError() != hipSuccess) { ctx->err = "kernel launch failed"; rc = PR_EHIP; break; }
    if (want_topk) {
      if (didx.alloc((size_t)m * k * 4) != hipSuccess || dsc.alloc((size_t)m * k * 4) != hipSuccess) { ctx->err = "out of device memory"; rc = PR_ENOMEM; break; }
      if ((rc = pr_fuse_select_dev(ctx, dd.as<float>(), nullptr, m, n, nullptr, 1, 0, 0, mask_width, 0.0, k, didx.as<int32_t>(), dsc.as<float>()))) break;
      if (hipMemcpyAsync(idx, didx.p, (size_t)m * k * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(score, dsc.p, (size_t)m * k * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "D2H copy failed"; rc = PR_EHIP; break; }
    }
    if (out && hipMemcpyAsync(out, dd.p, mn * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "D2H copy failed"; rc = PR_EHIP; break; }
    rc = pr_sync(ctx);
  } while (0);
  if (rc != PR_OK) (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

int pr_gist_distance(pr_ctx* ctx, const double* h1, int32_t m, const double* h2, int32_t n, int32_t cols, float* dist) {
  return plain_cols_host(ctx, PR_TYPE_GIST, h1, m, h2, n, cols, dist, 0, 0, nullptr, nullptr, false);
}

int pr_bow_distance(pr_ctx* ctx, const double* h1, int32_t m, const double* h2, int32_t n, int32_t cols, float* dist) {
  return plain_cols_host(ctx, PR_TYPE_BOW, h1, m, h2, n, cols, dist, 0, 0, nullptr, nullptr, false);
}

int pr_match_topk_cols(pr_ctx* ctx, int type, const double* h1, int32_t m, const double* h2, int32_t n, int32_t cols,
                       int32_t mask_width, int32_t k, int32_t* idx, float* score) {
  return plain_cols_host(ctx, type, h1, m, h2, n, cols, nullptr, mask_width, k, idx, score, true);
}

// ------------------------------------------------------------------------------------------- generation
static int check_gen_args(pr_ctx* ctx, const void* xyz, const void* inten, const void* offs, int32_t N, double max_rho, const void* out) {
  if (!ctx) return PR_EINVAL;
  if (N < 0 || !offs || (N > 0 && !out) || !(max_rho > 0)) PR_FAIL(ctx, PR_EINVAL, "generate: bad arguments (N=%d, max_rho=%g)", N, max_rho);
  (void)xyz; (void)inten;
  return PR_OK;
}

// frames (moments + eig) on the main stream, the float-average chain on the side stream, joined before binning
static int launch_frames_and_ave(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N,
                                 double* frames, float* ave) {
  PR_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
  PR_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
  pr::launch_ave_chain(ctx->side, inten, offs, N, ave);
  PR_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side));
  pr::launch_cloud_frames(ctx->stream, xyz, offs, N, frames);
  PR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  return PR_OK;
}

int pr_cloud_frames_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double* frames) {
  if (!ctx) return PR_EINVAL;
  if (N < 0 || !offs || (N > 0 && !frames)) PR_FAIL(ctx, PR_EINVAL, "pr_cloud_frames_dev: bad arguments (N=%d)", N);
  if (N == 0) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  if (inten) {            // the float averages beside the moments pass (side stream), into slots 14 / 15 of the frames
    PR_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    PR_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    pr::launch_ave_chain(ctx->side, inten, offs, N, nullptr, frames);
    PR_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side));
  } else {
    PR_HIP(ctx, hipMemset2DAsync(frames + 14, 16 * sizeof(double), 0, 2 * sizeof(double), (size_t)N, ctx->stream));
  }
  pr::launch_cloud_frames(ctx->stream, xyz, offs, N, frames);
  if (inten) PR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  PR_HIP(ctx, hipGetLastError());
  return PR_OK;
}

// the float-average chain (a dependent add per point: ~0.28 ms per 50 000 points, however many clouds) starts on the side stream ...
static int fork_ave(pr_ctx* ctx, const float* inten, const int64_t* offs, int32_t N, float* ave) {
  PR_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
  PR_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
  pr::launch_ave_chain(ctx->side, inten, offs, N, ave);
  PR_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side));
  return PR_OK;
}
// ... and is joined where its result is needed
static int join_ave(pr_ctx* ctx) {
  PR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  return PR_OK;
}
static int launch_ave_only(pr_ctx* ctx, const float* inten, const int64_t* offs, int32_t N, float* ave) {
  if (int rc = fork_ave(ctx, inten, offs, N, ave)) return rc;
  return join_ave(ctx);
}

int pr_sc_generate_frames_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double max_rho,
                              const double* frames, int frames_have_ave, double* out) {
  if (int rc = check_gen_args(ctx, xyz, inten, offs, N, max_rho, out)) return rc;
  if (N == 0) return PR_OK;
  if (!frames) PR_FAIL(ctx, PR_EINVAL, "pr_sc_generate_frames_dev: frames is NULL");
  if (int rc = set_device(ctx)) return rc;
  if (frames_have_ave) {        // everything but the binning pass came with the frames
    pr::launch_sc_bin(ctx->stream, xyz, inten, offs, N, max_rho, frames, nullptr, out, 1);
    PR_HIP(ctx, hipGetLastError());
    PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PR_OK;
  }
  DevScope scope_(ctx);
  DevBuf ave;
  PR_HIP(ctx, ave.alloc((size_t)N * 4));
  // the binning pass writes bin means and does not wait for the average chain beside it; sc_finish applies the averages
  if (int rc = fork_ave(ctx, inten, offs, N, ave.as<float>())) return rc;
  pr::launch_sc_bin(ctx->stream, xyz, inten, offs, N, max_rho, frames, nullptr, out);
  if (int rc = join_ave(ctx)) return rc;
  pr::launch_sc_finish(ctx->stream, ave.as<float>(), N, out);
  PR_HIP(ctx, hipGetLastError());
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PR_OK;
}

int pr_sc_generate_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double max_rho, double* out) {
  if (int rc = check_gen_args(ctx, xyz, inten, offs, N, max_rho, out)) return rc;
  if (N == 0) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  DevScope scope_(ctx);
  DevBuf frames, ave;
  PR_HIP(ctx, frames.alloc((size_t)N * 16 * 8));
  PR_HIP(ctx, ave.alloc((size_t)N * 4));
  // Default: two streaming passes over all clouds at once (moments, then binning).  PR_SC_GEN=batched selects the
  // cache-resident variant below - batches of ~96 MB, W workgroups per cloud, last-arriver merges - which moves half the
  // HBM bytes but measured 6.25 ms against 2.6 ms at 5000 x 50 000 points (74 batches x 2 launches of ~550 workgroups each
  // cannot keep enough loads in flight; tools/experiments/README.md).
  const bool two_pass = !(getenv("PR_SC_GEN") && !strcmp(getenv("PR_SC_GEN"), "batched"));
  const bool cluster = getenv("PR_SC_GEN") && !strcmp(getenv("PR_SC_GEN"), "cluster");
  if (cluster) {   // one HBM pass: clusters of workgroups hold a cloud in registers between the moments and the binning (sc_gen.hip)
    std::vector<int64_t> ho((size_t)N + 1);
    PR_HIP(ctx, hipMemcpyAsync(ho.data(), offs, ((size_t)N + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int64_t pmax = 0;
    for (int c = 0; c < N; c++) pmax = std::max(pmax, ho[c + 1] - ho[c]);
    int ncu = 0;
    PR_HIP(ctx, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
    int CW = 1;
    while ((int64_t)CW * pr::sc_cluster_points_per_workgroup() < pmax) CW *= 2;
    if (CW <= ncu / 8) {
      const int ncl = 8 * ((ncu / 8) / CW);
      const size_t need = pr::sc_cluster_scratch_bytes(ncl, CW);
      DevBuf scr;
      PR_HIP(ctx, scr.alloc(need));
      pr::launch_ave_chain(ctx->stream, inten, offs, N, ave.as<float>());
      int* err = pr::launch_sc_cluster(ctx->stream, xyz, inten, offs, N, max_rho, CW, ncu, scr.as<char>(), frames.as<double>(), out);
      pr::launch_sc_finish(ctx->stream, ave.as<float>(), N, out);
      int herr = 0;
      PR_HIP(ctx, hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream));
      PR_HIP(ctx, hipGetLastError());
      PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (herr) PR_FAIL(ctx, PR_EHIP, "pr_sc_generate_dev: a cluster hand-off timed out");
      return PR_OK;
    }
  }
  if (two_pass) {
    if (int rc = fork_ave(ctx, inten, offs, N, ave.as<float>())) return rc;
    pr::launch_cloud_frames(ctx->stream, xyz, offs, N, frames.as<double>());
    pr::launch_sc_bin(ctx->stream, xyz, inten, offs, N, max_rho, frames.as<double>(), nullptr, out);
    if (int rc = join_ave(ctx)) return rc;
    pr::launch_sc_finish(ctx->stream, ave.as<float>(), N, out);
    PR_HIP(ctx, hipGetLastError());
    PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PR_OK;
  }
  if (!ctx->sc_scratch) {
    PR_HIP(ctx, hipMalloc((void**)&ctx->sc_scratch, pr::sc_generate_scratch_bytes()));
    PR_HIP(ctx, hipMemsetAsync(ctx->sc_scratch, 0, pr::sc_generate_scratch_bytes(), ctx->stream));   // the tickets reset themselves afterwards
  }
  // cloud sizes decide the batches: offs comes back to the host (N + 1 values; the call synchronises at its end anyway)
  std::vector<int64_t> ho((size_t)N + 1);
  PR_HIP(ctx, hipMemcpyAsync(ho.data(), offs, ((size_t)N + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // the float-average chain over all clouds on the side stream; odd batches on side2
  PR_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
  PR_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
  PR_HIP(ctx, hipStreamWaitEvent(ctx->side2, ctx->ev_fork, 0));
  pr::launch_ave_chain(ctx->side, inten, offs, N, ave.as<float>());
  PR_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side));
  // Batches of ~96 MB of points (two of them fit the 256 MB Infinity Cache): the binning pass of a batch re-reads what its
  // moments pass has just pulled through the cache, so HBM sees every point once.
  static const int64_t batch_bytes = getenv("PR_SC_BATCH_MB") ? (int64_t)atol(getenv("PR_SC_BATCH_MB")) << 20 : (int64_t)96 << 20;
  hipStream_t st[2] = {ctx->stream, ctx->side2};
  int nbatch = 0;
  for (int c0 = 0; c0 < N;) {
    int c1 = c0;
    int64_t pts = 0;
    while (c1 < N && (c1 == c0 || (pts + ho[c1 + 1] - ho[c1]) * 28 <= batch_bytes)) { pts += ho[c1 + 1] - ho[c1]; c1++; }
    int nb = c1 - c0;
    int W = (512 + nb - 1) / nb;                       // >= ~512 workgroups per launch
    if (W > pr::SC_MAX_W) W = pr::SC_MAX_W;
    if (nb > 384) W = 1;                                // many small clouds: one workgroup per cloud fills the chip already
    while (W > 1 && (int64_t)nb * W > 768) W--;         // capacity of the partial-grid scratch
    pr::launch_sc_batch(st[nbatch & 1], xyz, inten, offs, c0, c1, W, max_rho, frames.as<double>(),
                        ctx->sc_scratch + (size_t)(nbatch & 1) * pr::SC_SCRATCH_PER_STREAM, out);
    nbatch++;
    c0 = c1;
  }
  PR_HIP(ctx, hipEventRecord(ctx->ev_b, ctx->side2));
  PR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_b, 0));
  PR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  pr::launch_sc_finish(ctx->stream, ave.as<float>(), N, out);
  PR_HIP(ctx, hipGetLastError());
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PR_OK;
}

static int m2dp_generate_impl(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double max_rho,
                              const double* frames_in, int frames_have_ave, double* out) {
  if (int rc = check_gen_args(ctx, xyz, inten, offs, N, max_rho, out)) return rc;
  if (N == 0) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  DevScope scope_(ctx);
  DevBuf frames, ave, mats;
  if (!frames_in) PR_HIP(ctx, frames.alloc((size_t)N * 16 * 8));
  const bool own_ave = !(frames_in && frames_have_ave);
  if (own_ave) PR_HIP(ctx, ave.alloc((size_t)N * 4));
  PR_HIP(ctx, mats.alloc(pr::m2dp_generate_scratch_bytes(N)));
  if (frames_in && own_ave) { if (int rc = launch_ave_only(ctx, inten, offs, N, ave.as<float>())) return rc; }
  else if (!frames_in) { if (int rc = launch_frames_and_ave(ctx, xyz, inten, offs, N, frames.as<double>(), ave.as<float>())) return rc; }
  PR_HIP(ctx, hipMemsetAsync(ctx->d_svd_rows, 0, sizeof(int), ctx->stream));
  pr::launch_m2dp_bin_svd(ctx->stream, xyz, inten, offs, N, max_rho, frames_in ? frames_in : frames.as<double>(), own_ave ? ave.as<float>() : nullptr, ctx->d_planes,
                          mats.as<double>(), out, ctx->d_flags, ctx->d_svd_rows, ctx->side2, ctx->ev_m2, ctx->ev_m2 + 2);
  PR_HIP(ctx, hipGetLastError());
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PR_OK;
}
int pr_m2dp_generate_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double max_rho, double* out) {
  return m2dp_generate_impl(ctx, xyz, inten, offs, N, max_rho, nullptr, 0, out);
}
int pr_m2dp_generate_frames_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double max_rho,
                                const double* frames, int frames_have_ave, double* out) {
  if (ctx && N > 0 && !frames) PR_FAIL(ctx, PR_EINVAL, "pr_m2dp_generate_frames_dev: frames is NULL");
  return m2dp_generate_impl(ctx, xyz, inten, offs, N, max_rho, frames, frames_have_ave, out);
}

int pr_m2dp_svd_rows(pr_ctx* ctx, int32_t* rows, int32_t cap, int32_t* count) {
  if (!ctx || !count || cap < 0 || (cap > 0 && !rows)) return PR_EINVAL;
  if (int rc = set_device(ctx)) return rc;
  std::vector<int> h(1 + pr::M2DP_SVD_ROWS_CAP);
  PR_HIP(ctx, hipMemcpyAsync(h.data(), ctx->d_svd_rows, h.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int nl = h[0] < pr::M2DP_SVD_ROWS_CAP ? h[0] : pr::M2DP_SVD_ROWS_CAP;
  std::sort(h.begin() + 1, h.begin() + 1 + nl);                      // both channels of a row may be listed
  const int nu = (int)(std::unique(h.begin() + 1, h.begin() + 1 + nl) - (h.begin() + 1));
  for (int i = 0; i < nu && i < cap; i++) rows[i] = h[1 + i];
  *count = nu;                                                       // (more than M2DP_SVD_ROWS_CAP pairs: the first ones recorded; the call-wide PR_WARN_M2DP_SVD bit stands for the rest)
  return PR_OK;
}

static int delight_generate_impl(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, const double* frames_in,
                                 double* out);
int pr_delight_generate_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double* out) {
  return delight_generate_impl(ctx, xyz, inten, offs, N, nullptr, out);
}
int pr_delight_generate_frames_dev(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, const double* frames,
                                   double* out) {
  if (ctx && N > 0 && !frames) PR_FAIL(ctx, PR_EINVAL, "pr_delight_generate_frames_dev: frames is NULL");
  return delight_generate_impl(ctx, xyz, inten, offs, N, frames, out);
}
static int delight_generate_impl(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, const double* frames_in,
                                 double* out) {
  if (int rc = check_gen_args(ctx, xyz, inten, offs, N, 1.0, out)) return rc;
  if (N == 0) return PR_OK;
  if (int rc = set_device(ctx)) return rc;
  DevScope scope_(ctx);
  DevBuf frames;
  if (!frames_in) {
    PR_HIP(ctx, frames.alloc((size_t)N * 16 * 8));
    pr::launch_cloud_frames(ctx->stream, xyz, offs, N, frames.as<double>());
  }
  pr::launch_delight_gen(ctx->stream, xyz, inten, offs, N, frames_in ? frames_in : frames.as<double>(), out);
  PR_HIP(ctx, hipGetLastError());
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PR_OK;
}

static int generate_host(pr_ctx* ctx, int type, const double* xyz, const float* inten, const int64_t* offs, int32_t N,
                         double max_rho, double* out) {
  if (int rc = check_gen_args(ctx, xyz, inten, offs, N, max_rho, out)) return rc;
  if (N == 0) return PR_OK;
  for (int c = 0; c < N; c++)
    if (offs[c + 1] < offs[c]) PR_FAIL(ctx, PR_EINVAL, "generate: offs must be non-decreasing (cloud %d)", c);
  if (offs[0] != 0) PR_FAIL(ctx, PR_EINVAL, "generate: offs[0] must be 0");
  const size_t T = (size_t)offs[N];
  if (T > 0 && (!xyz || !inten)) PR_FAIL(ctx, PR_EINVAL, "generate: xyz/inten are NULL");
  if (int rc = set_device(ctx)) return rc;
  const size_t rowlen = (type == PR_TYPE_SC) ? PR_SC_SIG_LEN : (type == PR_TYPE_M2DP ? (size_t)4 * PR_M2DP_SIG_LEN : (size_t)16 * PR_DELIGHT_SIG_LEN);
  DevScope scope_(ctx);
  DevBuf dx, di, dof, dout;
  PR_HIP(ctx, dx.alloc(T * 24));
  PR_HIP(ctx, di.alloc(T * 4));
  PR_HIP(ctx, dof.alloc((size_t)(N + 1) * 8));
  PR_HIP(ctx, dout.alloc((size_t)N * rowlen * 8));
  if (T) {
    PR_HIP(ctx, hipMemcpyAsync(dx.p, xyz, T * 24, hipMemcpyHostToDevice, ctx->stream));
    PR_HIP(ctx, hipMemcpyAsync(di.p, inten, T * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  PR_HIP(ctx, hipMemcpyAsync(dof.p, offs, (size_t)(N + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  int rc = (type == PR_TYPE_SC)
               ? pr_sc_generate_dev(ctx, dx.as<double>(), di.as<float>(), dof.as<int64_t>(), N, max_rho, dout.as<double>())
               : (type == PR_TYPE_M2DP
                      ? pr_m2dp_generate_dev(ctx, dx.as<double>(), di.as<float>(), dof.as<int64_t>(), N, max_rho, dout.as<double>())
                      : pr_delight_generate_dev(ctx, dx.as<double>(), di.as<float>(), dof.as<int64_t>(), N, dout.as<double>()));
  if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  PR_HIP(ctx, hipMemcpyAsync(out, dout.p, (size_t)N * rowlen * 8, hipMemcpyDeviceToHost, ctx->stream));
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PR_OK;
}

int pr_sc_generate(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double max_rho, double* out) {
  return generate_host(ctx, PR_TYPE_SC, xyz, inten, offs, N, max_rho, out);
}

int pr_m2dp_generate(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double max_rho, double* out) {
  return generate_host(ctx, PR_TYPE_M2DP, xyz, inten, offs, N, max_rho, out);
}

int pr_delight_generate(pr_ctx* ctx, const double* xyz, const float* inten, const int64_t* offs, int32_t N, double* out) {
  return generate_host(ctx, PR_TYPE_DELIGHT, xyz, inten, offs, N, 1.0, out);
}

int pr_generate_clouds(pr_ctx* ctx, int type, const pr_clouds* c, double max_rho, double* out) {
  if (!ctx) return PR_EINVAL;
  if (!c || (type != PR_TYPE_SC && type != PR_TYPE_M2DP && type != PR_TYPE_DELIGHT)) PR_FAIL(ctx, PR_EINVAL, "pr_generate_clouds: bad arguments");
  const int32_t N = (int32_t)c->ids.size();
  if (type == PR_TYPE_DELIGHT) max_rho = 1.0;
  if (!c->d_frames || c->device != ctx->device)          // host-made clouds (or another device's): the two-pass path from the host arrays
    return generate_host(ctx, type, c->xyz.data(), c->inten.data(), c->offs.data(), N, max_rho, out);
  if (N > 0 && !out) PR_FAIL(ctx, PR_EINVAL, "pr_generate_clouds: out is NULL");
  if (int rc = set_device(ctx)) return rc;
  const size_t rowlen = (type == PR_TYPE_SC) ? PR_SC_SIG_LEN : (type == PR_TYPE_M2DP ? (size_t)4 * PR_M2DP_SIG_LEN : (size_t)16 * PR_DELIGHT_SIG_LEN);
  DevScope scope_(ctx);
  DevBuf dout;
  PR_HIP(ctx, dout.alloc((size_t)N * rowlen * 8));
  const double* x = static_cast<const double*>(c->d_xyz);
  const float* it = static_cast<const float*>(c->d_inten);
  const int64_t* of = static_cast<const int64_t*>(c->d_offs);
  const double* fr = static_cast<const double*>(c->d_frames);
  int rc = (type == PR_TYPE_SC)     ? pr_sc_generate_frames_dev(ctx, x, it, of, N, max_rho, fr, 1, dout.as<double>())
           : (type == PR_TYPE_M2DP) ? pr_m2dp_generate_frames_dev(ctx, x, it, of, N, max_rho, fr, 1, dout.as<double>())
                                    : pr_delight_generate_frames_dev(ctx, x, it, of, N, fr, dout.as<double>());
  if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  PR_HIP(ctx, hipMemcpyAsync(out, dout.p, (size_t)N * rowlen * 8, hipMemcpyDeviceToHost, ctx->stream));
  PR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PR_OK;
}

}  // extern "C"
