// pr_group.cpp — the database row-sharded over several GPUs of one node inside the C ABI (SURVEY.md §8-b "same calls on a
// pr_group created over device ids {0..G-1}", §8-e).  The reference is single-threaded, single-device MATLAB
// (match_signatures/run_test.m:25-57); this is that computation with hist2 split by rows:
//
//   every device g holds rows [g n/G, (g+1) n/G) of the database (raw f64 + packed operand image) and ALL queries
//   1. per device: pack queries, all-pairs distances, per-row moments of the shard                    pr_distances_dev, pr_row_moments_dev
//   A. all-gather of the moments (48 B per query per rank)                                           ncclAllGather on the compute streams
//   2. per device: fused fp32 score with the statistics of the WHOLE row (Chan combination in rank order), mask on
//      global indices, per-shard top-(k+8)                                                            pr_fuse_select_dev
//   B. all-gather of the per-shard (index, fp32 score) lists; merged on every device into the GLOBAL top-(k+8) candidates -
//      the list an unsharded run selects                                                              ncclAllGather, pr_merge_topk_dev
//   3. per device: fp64 re-evaluation of the candidates inside its own rows (on average (k+8)/G per query: the cost does not
//      grow with G)                                                                                   pr_rerank_partial_dev
//   C. all-gather of the shards' evaluations (p5 blocks: scores + exact channel distances); every device takes each candidate's score
//      from its owner, keeps the k best and checks their order                                        ncclAllGather, pr_rerank_finish_dev
//   4. the queries the re-evaluated candidates cannot answer for certain (order hangs on the fp32 pass's sigmas | the candidate list does
//      not provably hold the top-k: none, as a rule - the count is read back and nothing below runs) in passes of 64:
//      per device the query's exact fp64 distances to every row of the shard and their moments        pr_order_exact_moments_dev
//   D. all-gather of those moments (96 B per query per rank)                                          ncclAllGather
//   5. per device: the shard's k best of the exact row under the statistics of the WHOLE row          pr_order_exact_select_dev
//   E. all-gather of those lists (16 k B per flagged query per rank), merged on every device          ncclAllGather, pr_order_exact_merge_dev
//
// One host thread drives all devices; everything is asynchronous on each context's stream, the two collectives are
// enqueued on those same streams between the kernels (one ncclGroupStart/End per collective, communicators from
// ncclCommInitAll), and the host waits once, at the end.  RCCL is loaded with dlopen at group creation (librccl.so.1 - the
// copy PyTorch has already loaded when the caller is a torch process), so libpr_amd.so itself does not depend on it.
// Device ids may repeat ("virtual shards" on one GPU: how the sharded arithmetic is tested on a one-GPU box); RCCL refuses
// duplicate devices in a communicator, so such a group exchanges by device-to-device copies ordered with events instead.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/place_recognition.h"

namespace {

// phases of a pr_group_match_topk call, timed per shard when pr_group_set_timing is on (include/place_recognition.h: PR_GROUP_PHASES)
enum { PH_PACK = 0, PH_MATCH, PH_MOMENTS, PH_GATHER_A, PH_SELECT, PH_GATHER_B, PH_RERANK, PH_GATHER_C, PH_FINISH, PH_EXACT };

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string& err) {
    if (h) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) { err = std::string("cannot load RCCL: ") + dlerror(); return false; }
#define SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(h, name)); if (!field) { err = "RCCL lacks " name; return false; }
    SYM(CommInitAll, "ncclCommInitAll") SYM(CommDestroy, "ncclCommDestroy") SYM(CommCount, "ncclCommCount") SYM(AllGather, "ncclAllGather")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    return true;
  }
};

struct Shard {
  int device = 0;
  pr_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr, ev_done = nullptr;   // "my slice is ready" / "I have copied every slice" (copy exchange)
  hipEvent_t ph[PR_GROUP_PHASES + 1] = {};      // pr_group_set_timing: events between the phases of a call
  pr_sigset *q = nullptr, *db = nullptr;
  int32_t row0 = 0, rows = 0;                 // this shard's global DB rows
  int32_t cap_rows = 0;                       // rows its buffers are sized for: rows, or rows + the growth reserve of the last shard (pr_group_set_database_growable)
  void *raw_db = nullptr, *raw_q = nullptr;   // f64 signatures (the re-evaluation reads them)
  float *d_p = nullptr, *d_i = nullptr;
  double *mom = nullptr, *mom_all = nullptr, *score = nullptr, *score_all = nullptr, *sc64 = nullptr, *part = nullptr, *dump = nullptr;
  double *p5_all = nullptr, *exact = nullptr, *exact_all = nullptr, *sel = nullptr, *sel_all = nullptr;
  int32_t *idx_in = nullptr, *idx = nullptr, *idx_all = nullptr, *cand = nullptr;
  float* sc32 = nullptr;
};

}  // namespace

struct pr_group {
  int G = 0;
  std::vector<Shard> s;
  bool rccl = false;
  Rccl nc;
  std::vector<ncclComm_t> comms;
  std::string err;
  int type = -1;
  int32_t n = 0, q_cap = 0, k_cap = 0;
  int32_t last_flagged = 0;        // queries of the last pr_group_match_topk that were answered from their exact rows
  bool timing = false;             // pr_group_set_timing
  bool timed = false;              // the last call recorded its phase events
};

static thread_local std::string g_gerr;

#define G_FAIL(g, code, ...)                                      \
  do {                                                            \
    char _b[512];                                                 \
    snprintf(_b, sizeof _b, __VA_ARGS__);                         \
    if (g) (g)->err = _b; else g_gerr = _b;                       \
    return (code);                                                \
  } while (0)
#define G_HIP(g, call)                                                                                  \
  do {                                                                                                  \
    hipError_t _e = (call);                                                                             \
    if (_e != hipSuccess) G_FAIL(g, (_e == hipErrorOutOfMemory ? PR_ENOMEM : PR_EHIP), "%s failed: %s", #call, hipGetErrorString(_e)); \
  } while (0)
#define G_PR(g, sh, call)                                                                               \
  do {                                                                                                  \
    int _rc = (call);                                                                                   \
    if (_rc != PR_OK) G_FAIL(g, _rc, "device %d: %s", (sh).device, pr_last_error((sh).ctx));            \
  } while (0)
#define G_NCCL(g, call)                                                                                 \
  do {                                                                                                  \
    ncclResult_t _r = (call);                                                                           \
    if (_r != ncclSuccess) G_FAIL(g, PR_EHIP, "%s failed: %s", #call, (g)->nc.GetErrorString(_r));      \
  } while (0)

static void free_match_buffers(Shard& sh) {
  (void)hipSetDevice(sh.device);
  for (void* p : {(void*)sh.raw_q, (void*)sh.d_p, (void*)sh.d_i, (void*)sh.mom, (void*)sh.mom_all, (void*)sh.score, (void*)sh.score_all,
                  (void*)sh.idx_in, (void*)sh.idx, (void*)sh.idx_all, (void*)sh.sc32, (void*)sh.sc64, (void*)sh.part, (void*)sh.cand, (void*)sh.dump,
                  (void*)sh.p5_all, (void*)sh.exact, (void*)sh.exact_all, (void*)sh.sel, (void*)sh.sel_all})
    if (p) (void)hipFree(p);
  sh.raw_q = nullptr; sh.d_p = sh.d_i = sh.sc32 = nullptr; sh.mom = sh.mom_all = sh.score = sh.score_all = sh.sc64 = sh.part = sh.dump = nullptr;
  sh.idx_in = sh.idx = sh.idx_all = sh.cand = nullptr;
  sh.p5_all = sh.exact = sh.exact_all = sh.sel = sh.sel_all = nullptr;
  if (sh.q) { pr_sigset_destroy(sh.ctx, sh.q); sh.q = nullptr; }
}

// all-gather of `bytes` per rank: src_g on device g -> dst_h[g * bytes ..] on every device h, on the compute streams
static int exchange(pr_group* g, std::vector<const void*>& src, std::vector<void*>& dst, size_t bytes) {
  const int G = g->G;
  if (g->rccl) {
    G_NCCL(g, g->nc.GroupStart());
    for (int r = 0; r < G; r++) {
      G_HIP(g, hipSetDevice(g->s[r].device));
      G_NCCL(g, g->nc.AllGather(src[r], dst[r], bytes, ncclChar, g->comms[r], g->s[r].stream));
    }
    G_NCCL(g, g->nc.GroupEnd());
    return PR_OK;
  }
  for (int r = 0; r < G; r++) {          // producer side: "my slice is ready"
    G_HIP(g, hipSetDevice(g->s[r].device));
    G_HIP(g, hipEventRecord(g->s[r].ev, g->s[r].stream));
  }
  for (int h = 0; h < G; h++) {
    G_HIP(g, hipSetDevice(g->s[h].device));
    for (int r = 0; r < G; r++) {
      if (r != h) G_HIP(g, hipStreamWaitEvent(g->s[h].stream, g->s[r].ev, 0));
      char* d = static_cast<char*>(dst[h]) + (size_t)r * bytes;
      if (g->s[r].device == g->s[h].device) G_HIP(g, hipMemcpyAsync(d, src[r], bytes, hipMemcpyDeviceToDevice, g->s[h].stream));
      else G_HIP(g, hipMemcpyPeerAsync(d, g->s[h].device, src[r], g->s[r].device, bytes, g->s[h].stream));
    }
    G_HIP(g, hipEventRecord(g->s[h].ev_done, g->s[h].stream));
  }
  // consumer side done -> producers: nothing a shard enqueues after this exchange (the next kernel that writes src, the next exchange's
  // record of `ev`) runs before EVERY shard has copied its slice - the exchange is a full rendezvous of the G streams, like the collective
  // it stands in for, so no later reuse of a source buffer can race with a peer's copy
  for (int r = 0; r < G; r++) {
    G_HIP(g, hipSetDevice(g->s[r].device));
    for (int h = 0; h < G; h++)
      if (h != r) G_HIP(g, hipStreamWaitEvent(g->s[r].stream, g->s[h].ev_done, 0));
  }
  return PR_OK;
}

static int exchange_selftest(pr_group* g) {
  const int G = g->G;
  const size_t sizes[] = {48, 36, 72, 360, 96, 16, 4096};      // moments | candidate indices | their scores | p5 blocks | exact moments | exact lists of a 1-query call, + a page
  const size_t maxb = 4096;
  std::vector<void*> src(G, nullptr), dst(G, nullptr);
  int rc = PR_OK;
  for (int r = 0; r < G && rc == PR_OK; r++) {
    if (hipSetDevice(g->s[r].device) != hipSuccess || hipMalloc(&src[r], maxb) != hipSuccess || hipMalloc(&dst[r], maxb * G) != hipSuccess) {
      g->err = "pr_group_create: self-test allocation failed"; rc = PR_ENOMEM;
    }
  }
  std::vector<unsigned char> host(maxb * G);
  for (size_t bytes : sizes) {
    if (rc != PR_OK) break;
    for (int r = 0; r < G; r++) {
      (void)hipSetDevice(g->s[r].device);
      (void)hipMemsetAsync(src[r], r + 1, bytes, g->s[r].stream);
      (void)hipMemsetAsync(dst[r], 0xEE, bytes * G, g->s[r].stream);
    }
    std::vector<const void*> cs(src.begin(), src.end());
    if ((rc = exchange(g, cs, dst, bytes)) != PR_OK) break;
    for (int h = 0; h < G && rc == PR_OK; h++) {
      (void)hipSetDevice(g->s[h].device);
      if (hipMemcpyAsync(host.data(), dst[h], bytes * G, hipMemcpyDeviceToHost, g->s[h].stream) != hipSuccess ||
          hipStreamSynchronize(g->s[h].stream) != hipSuccess) { g->err = "pr_group_create: self-test copy failed"; rc = PR_EHIP; break; }
      for (int r = 0; r < G && rc == PR_OK; r++)
        for (size_t i = 0; i < bytes; i++)
          if (host[(size_t)r * bytes + i] != (unsigned char)(r + 1)) {
            char b[256];
            snprintf(b, sizeof b, "pr_group_create: exchange self-test failed (%s, %zu B per rank): device %d (rank %d) holds 0x%02x in rank %d's slot, byte %zu",
                     g->rccl ? "RCCL all-gather" : "device copies", bytes, g->s[h].device, h, host[(size_t)r * bytes + i], r, i);
            g->err = b; rc = PR_EHIP;
            break;
          }
    }
  }
  for (int r = 0; r < G; r++) { (void)hipSetDevice(g->s[r].device); (void)hipStreamSynchronize(g->s[r].stream); if (src[r]) (void)hipFree(src[r]); if (dst[r]) (void)hipFree(dst[r]); }
  return rc;
}

extern "C" {

const char* pr_group_last_error(const pr_group* g) { return g ? g->err.c_str() : g_gerr.c_str(); }
int32_t pr_group_size(const pr_group* g) { return g ? g->G : 0; }
int pr_group_uses_rccl(const pr_group* g) { return g && g->rccl; }
int32_t pr_group_last_flagged(const pr_group* g) { return g ? g->last_flagged : 0; }
int pr_group_set_exact_statistics(pr_group* g, int on) {
  if (!g) return PR_EINVAL;
  for (auto& sh : g->s) if (sh.ctx) (void)pr_set_exact_statistics(sh.ctx, on);
  return PR_OK;
}
int32_t pr_group_rccl_ranks(const pr_group* g) {   // asked of the communicator itself, not of the argument list
  if (!g || !g->rccl || g->comms.empty() || !g->comms[0]) return 0;
  int n = 0;
  return g->nc.CommCount(g->comms[0], &n) == ncclSuccess ? n : -1;
}

void pr_group_destroy(pr_group* g) {
  if (!g) return;
  for (auto& sh : g->s) {
    if (!sh.ctx) continue;
    (void)hipSetDevice(sh.device);
    (void)pr_sync(sh.ctx);
  }
  if (g->rccl) for (auto c : g->comms) if (c) (void)g->nc.CommDestroy(c);
  for (auto& sh : g->s) {
    if (!sh.ctx) continue;
    free_match_buffers(sh);
    if (sh.db) pr_sigset_destroy(sh.ctx, sh.db);
    if (sh.raw_db) (void)hipFree(sh.raw_db);
    if (sh.ev) (void)hipEventDestroy(sh.ev);
    if (sh.ev_done) (void)hipEventDestroy(sh.ev_done);
    for (auto& e : sh.ph) if (e) (void)hipEventDestroy(e);
    pr_destroy(sh.ctx);
  }
  delete g;
}

int pr_group_create(const int32_t* device_ids, int32_t G, pr_group** out) {
  if (!out) G_FAIL((pr_group*)nullptr, PR_EINVAL, "pr_group_create: out is NULL");
  *out = nullptr;
  if (!device_ids || G < 1 || G > 64) G_FAIL((pr_group*)nullptr, PR_EINVAL, "pr_group_create: need 1..64 device ids");
  pr_group* g = new (std::nothrow) pr_group;
  if (!g) G_FAIL((pr_group*)nullptr, PR_ENOMEM, "out of host memory");
  g->G = G;
  g->s.resize(G);
  bool distinct = true;
  for (int r = 0; r < G; r++)
    for (int t = 0; t < r; t++) distinct = distinct && device_ids[r] != device_ids[t];
  for (int r = 0; r < G; r++) {
    Shard& sh = g->s[r];
    sh.device = device_ids[r];
    if (pr_create(sh.device, &sh.ctx) != PR_OK) { g_gerr = std::string("pr_group_create: ") + pr_last_error(nullptr); pr_group_destroy(g); return PR_EHIP; }
    sh.stream = static_cast<hipStream_t>(pr_stream(sh.ctx));
    // the group protocol is the split-f16 / fp32 one (k + 8 candidates); the single-product arithmetic needs the margin check and its
    // fallback, which live above pr_group (matcher.py) - a PR_SC_MATCH=f16 environment does not leak in here
    if (pr_get_sc_arith(sh.ctx) == PR_SC_ARITH_F16) (void)pr_set_sc_arith(sh.ctx, PR_SC_ARITH_F16X2);
    if (hipSetDevice(sh.device) != hipSuccess || hipEventCreateWithFlags(&sh.ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sh.ev_done, hipEventDisableTiming) != hipSuccess) {
      g_gerr = "pr_group_create: hipEventCreate failed"; pr_group_destroy(g); return PR_EHIP;
    }
  }
  // RCCL over xGMI when every shard has its own GPU (PR_GROUP_EXCHANGE=rccl|copy overrides: rccl needs distinct devices)
  const char* ex = getenv("PR_GROUP_EXCHANGE");
  bool want_rccl = distinct && G > 1;
  if (ex && !strcmp(ex, "copy")) want_rccl = false;
  if (ex && !strcmp(ex, "rccl")) {
    if (!distinct) { g_gerr = "pr_group_create: PR_GROUP_EXCHANGE=rccl needs distinct devices"; pr_group_destroy(g); return PR_EINVAL; }
    want_rccl = true;
  }
  if (want_rccl) {
    if (!g->nc.load(g_gerr)) { pr_group_destroy(g); return PR_EHIP; }
    g->comms.assign(G, nullptr);
    std::vector<int> devs(device_ids, device_ids + G);
    const ncclResult_t r = g->nc.CommInitAll(g->comms.data(), G, devs.data());
    if (r != ncclSuccess) { g_gerr = std::string("ncclCommInitAll failed: ") + g->nc.GetErrorString(r); g->comms.clear(); pr_group_destroy(g); return PR_EHIP; }
    g->rccl = true;
  } else if (distinct && G > 1) {   // copies between devices: let the copy engines reach the peers directly
    for (int r = 0; r < G; r++) {
      (void)hipSetDevice(device_ids[r]);
      for (int t = 0; t < G; t++) if (t != r) (void)hipDeviceEnablePeerAccess(device_ids[t], 0);
    }
    (void)hipGetLastError();
  }
  // self-test of the exchange, before any real step: every rank's slice must arrive in ITS slot on every device, for every payload size the
  // protocol of a call uses - a topology / RCCL problem fails here with a message instead of inside a step
  if (G > 1) {
    const int rc = exchange_selftest(g);
    if (rc != PR_OK) { g_gerr = g->err; pr_group_destroy(g); return rc; }
  }
  *out = g;
  return PR_OK;
}

int pr_group_set_timing(pr_group* g, int on) {
  if (!g) return PR_EINVAL;
  if (on)
    for (auto& sh : g->s) {
      G_HIP(g, hipSetDevice(sh.device));
      for (auto& e : sh.ph) if (!e) G_HIP(g, hipEventCreate(&e));
    }
  g->timing = on != 0;
  g->timed = false;
  return PR_OK;
}

int pr_group_last_timing(pr_group* g, float* ms, int32_t cap) {
  if (!g || !ms) return PR_EINVAL;
  if (!g->timed) G_FAIL(g, PR_EINVAL, "pr_group_last_timing: no timed pr_group_match_topk call (pr_group_set_timing)");
  if (cap < g->G * PR_GROUP_PHASES) G_FAIL(g, PR_EINVAL, "pr_group_last_timing: ms needs room for %d x %d floats", g->G, PR_GROUP_PHASES);
  for (int r = 0; r < g->G; r++) {
    Shard& sh = g->s[r];
    G_HIP(g, hipSetDevice(sh.device));
    G_HIP(g, hipEventSynchronize(sh.ph[PR_GROUP_PHASES]));
    for (int p = 0; p < PR_GROUP_PHASES; p++) G_HIP(g, hipEventElapsedTime(&ms[r * PR_GROUP_PHASES + p], sh.ph[p], sh.ph[p + 1]));
  }
  return PR_OK;
}

// hist2 of run_test.m:1, host f64: [n][2400] (SC) or [4n][384] (M2DP); rows [g n/G, (g+1) n/G) go to shard g
static int set_database_impl(pr_group* g, int type, const double* h2, int32_t n, int32_t extra) {
  g->type = -1; g->q_cap = 0; g->k_cap = 0;               // nothing usable until this call has succeeded
  if ((type != PR_TYPE_SC && type != PR_TYPE_M2DP) || n < 2 || !h2 || extra < 0) G_FAIL(g, PR_EINVAL, "pr_group_set_database: type must be SC or M2DP, n >= 2, extra >= 0");
  const size_t row_doubles = type == PR_TYPE_SC ? PR_SC_SIG_LEN : (size_t)4 * PR_M2DP_SIG_LEN;
  for (int r = 0; r < g->G; r++) {
    Shard& sh = g->s[r];
    G_HIP(g, hipSetDevice(sh.device));
    if (sh.db) { pr_sigset_destroy(sh.ctx, sh.db); sh.db = nullptr; }
    if (sh.raw_db) { (void)hipFree(sh.raw_db); sh.raw_db = nullptr; }
    free_match_buffers(sh);
    sh.row0 = (int32_t)((int64_t)n * r / g->G);
    sh.rows = (int32_t)((int64_t)n * (r + 1) / g->G) - sh.row0;
    if (sh.rows < 1) G_FAIL(g, PR_EINVAL, "pr_group_set_database: fewer signatures (%d) than shards (%d)", n, g->G);
    // a growable database: the LAST shard (whose rows end the global numbering) gets `extra` rows of room - raw rows, operand image in the
    // layout of that capacity (pr_sigset_reserve), distance matrices - and pr_group_append_database adds rows there in place
    sh.cap_rows = sh.rows + (r == g->G - 1 ? extra : 0);
    const size_t bytes = (size_t)sh.rows * row_doubles * 8;
    G_HIP(g, hipMalloc(&sh.raw_db, (size_t)sh.cap_rows * row_doubles * 8));
    G_HIP(g, hipMemcpyAsync(sh.raw_db, h2 + (size_t)sh.row0 * row_doubles, bytes, hipMemcpyHostToDevice, sh.stream));
    G_PR(g, sh, pr_sigset_create(sh.ctx, type, PR_ROLE_DB, sh.cap_rows, &sh.db));
    if (sh.cap_rows > sh.rows) G_PR(g, sh, pr_sigset_reserve(sh.ctx, sh.db));
    G_PR(g, sh, pr_sigset_pack(sh.ctx, sh.db, sh.raw_db, PR_F64, PR_DEVICE, sh.rows));
  }
  for (auto& sh : g->s) { G_HIP(g, hipSetDevice(sh.device)); G_PR(g, sh, pr_sync(sh.ctx)); }   // h2 may be released by the caller
  g->type = type; g->n = n; g->q_cap = 0; g->k_cap = 0;
  return PR_OK;
}

// run_test.m:26-57 over the sharded database: idx [m][k] global 0-based rows of hist2 (-1: none), score [m][k] f64
static int match_topk_impl(pr_group* g, const double* h1, int32_t m, int32_t mask_width, double p_weight, int32_t k, int32_t* idx,
                           double* score) {
  if (g->type < 0) G_FAIL(g, PR_EINVAL, "pr_group_match_topk: no database (pr_group_set_database)");
  const int G = g->G;
  if (m < 0 || (m > 0 && !h1) || k < 1 || k > 120 || G > 64 || !idx || !score)
    G_FAIL(g, PR_EINVAL, "pr_group_match_topk: bad arguments (m=%d, k=%d; k <= 120, at most 64 shards)", m, k);
  if (m == 0) return PR_OK;
  const int type = g->type;
  const bool sc = type == PR_TYPE_SC;
  const size_t row_doubles = sc ? PR_SC_SIG_LEN : (size_t)4 * PR_M2DP_SIG_LEN;
  const int kin = k + 8 > 128 ? 128 : k + 8;
  if (m > g->q_cap || k > g->k_cap) {                         // (re)allocate the per-device work buffers
    const int32_t qc = m > g->q_cap ? m : g->q_cap, kc = k > g->k_cap ? k : g->k_cap;
    const int kinc = kc + 8 > 128 ? 128 : kc + 8;
    g->q_cap = g->k_cap = 0;                                  // a failed allocation below leaves no buffer that looks usable
    for (auto& sh : g->s) {
      free_match_buffers(sh);
      G_HIP(g, hipSetDevice(sh.device));
      G_PR(g, sh, pr_sigset_create(sh.ctx, type, PR_ROLE_QUERY, qc, &sh.q));
      G_HIP(g, hipMalloc(&sh.raw_q, (size_t)qc * row_doubles * 8));
      G_HIP(g, hipMalloc((void**)&sh.d_p, (size_t)qc * sh.cap_rows * 4));
      G_HIP(g, hipMalloc((void**)&sh.d_i, (size_t)qc * sh.cap_rows * 4));
      G_HIP(g, hipMalloc((void**)&sh.mom, (size_t)qc * 6 * 8));
      G_HIP(g, hipMalloc((void**)&sh.mom_all, (size_t)G * qc * 6 * 8));
      G_HIP(g, hipMalloc((void**)&sh.idx_in, (size_t)qc * kinc * 4));
      G_HIP(g, hipMalloc((void**)&sh.sc32, (size_t)qc * kinc * 4));
      G_HIP(g, hipMalloc((void**)&sh.idx, (size_t)qc * kc * 4));
      G_HIP(g, hipMalloc((void**)&sh.score, (size_t)qc * kc * 8));
      G_HIP(g, hipMalloc((void**)&sh.idx_all, (size_t)G * qc * kinc * 4));
      G_HIP(g, hipMalloc((void**)&sh.score_all, (size_t)G * qc * kinc * 8));
      G_HIP(g, hipMalloc((void**)&sh.sc64, (size_t)qc * kinc * 8));
      G_HIP(g, hipMalloc((void**)&sh.part, (size_t)qc * 5 * kinc * 8));
      G_HIP(g, hipMalloc((void**)&sh.p5_all, (size_t)G * qc * 5 * kinc * 8));
      G_HIP(g, hipMalloc((void**)&sh.exact, (size_t)qc * 12 * 8));
      G_HIP(g, hipMalloc((void**)&sh.exact_all, (size_t)G * qc * 12 * 8));
      G_HIP(g, hipMalloc((void**)&sh.sel, (size_t)64 * 2 * kc * 8));
      G_HIP(g, hipMalloc((void**)&sh.sel_all, (size_t)G * 64 * 2 * kc * 8));
      G_HIP(g, hipMalloc((void**)&sh.dump, (size_t)qc * kinc * 8));
      G_HIP(g, hipMalloc((void**)&sh.cand, (size_t)qc * kinc * 4));
    }
    g->q_cap = qc; g->k_cap = kc;
  }
  g->timed = false;
  auto mark = [&](int phase_end) -> int {                     // event `phase_end` on every shard's stream: the end of phase phase_end - 1
    if (!g->timing) return PR_OK;
    for (auto& sh : g->s) { G_HIP(g, hipSetDevice(sh.device)); G_HIP(g, hipEventRecord(sh.ph[phase_end], sh.stream)); }
    return PR_OK;
  };
  if (int rc = mark(PH_PACK)) return rc;
  // 1. local distances and moments
  for (auto& sh : g->s) {
    G_HIP(g, hipSetDevice(sh.device));
    G_HIP(g, hipMemcpyAsync(sh.raw_q, h1, (size_t)m * row_doubles * 8, hipMemcpyHostToDevice, sh.stream));
    G_PR(g, sh, pr_sigset_pack(sh.ctx, sh.q, sh.raw_q, PR_F64, PR_DEVICE, m));
    if (g->timing) G_HIP(g, hipEventRecord(sh.ph[PH_MATCH], sh.stream));
    G_PR(g, sh, pr_distances_dev(sh.ctx, sh.q, sh.db, sh.d_p, sh.d_i));
    if (g->timing) G_HIP(g, hipEventRecord(sh.ph[PH_MOMENTS], sh.stream));
    G_PR(g, sh, pr_row_moments_dev(sh.ctx, sh.d_p, sh.d_i, m, sh.rows, sh.mom));
  }
  if (int rc = mark(PH_GATHER_A)) return rc;
  std::vector<const void*> src(G);
  std::vector<void*> dst(G);
  // A. moments of every shard on every device
  for (int r = 0; r < G; r++) { src[r] = g->s[r].mom; dst[r] = g->s[r].mom_all; }
  if (int rc = exchange(g, src, dst, (size_t)m * 6 * 8)) return rc;
  if (int rc = mark(PH_SELECT)) return rc;
  // 2. per-shard fp32 selection with the whole row's statistics
  for (auto& sh : g->s) {
    G_HIP(g, hipSetDevice(sh.device));
    G_PR(g, sh, pr_fuse_select_dev(sh.ctx, sh.d_p, sh.d_i, m, sh.rows, sh.mom_all, G, 0, sh.row0, mask_width, p_weight, kin, sh.idx_in, sh.sc32));
    G_PR(g, sh, pr_widen_scores_dev(sh.ctx, sh.sc32, (int64_t)m * kin, sh.sc64));
  }
  if (int rc = mark(PH_GATHER_B)) return rc;
  // B. every shard's candidates on every device, merged into the global top-(k+8) of the fp32 pass
  for (int r = 0; r < G; r++) { src[r] = g->s[r].idx_in; dst[r] = g->s[r].idx_all; }
  if (int rc = exchange(g, src, dst, (size_t)m * kin * 4)) return rc;
  for (int r = 0; r < G; r++) { src[r] = g->s[r].sc64; dst[r] = g->s[r].score_all; }
  if (int rc = exchange(g, src, dst, (size_t)m * kin * 8)) return rc;
  if (int rc = mark(PH_RERANK)) return rc;
  // 3. fp64 re-evaluation of the candidates each shard owns
  for (auto& sh : g->s) {
    G_HIP(g, hipSetDevice(sh.device));
    G_PR(g, sh, pr_merge_topk_dev(sh.ctx, sh.idx_all, sh.score_all, G, m, kin, sh.cand, sh.dump /*the merged fp32 scores, in a buffer of their own: sc64 may still be read by the other shards' copies*/));
    G_PR(g, sh, pr_rerank_partial_dev(sh.ctx, sc ? sh.raw_q : nullptr, sc ? sh.raw_db : nullptr, PR_F64, sc ? nullptr : sh.raw_q,
                                      sc ? nullptr : sh.raw_db, PR_F64, sc ? sh.mom_all : nullptr, sc ? nullptr : sh.mom_all, m, sh.rows, G, 0,
                                      sh.row0, mask_width, p_weight, kin, sh.cand, sh.dump, k, sh.part));
  }
  if (int rc = mark(PH_GATHER_C)) return rc;
  // C. every shard's evaluations (p5 blocks) on every device; every device finishes and checks the order of its result
  for (int r = 0; r < G; r++) { src[r] = g->s[r].part; dst[r] = g->s[r].p5_all; }
  if (int rc = exchange(g, src, dst, (size_t)m * 5 * kin * 8)) return rc;
  if (int rc = mark(PH_FINISH)) return rc;
  // 4. every device finishes (identical inputs: identical results and flags everywhere) ...
  for (auto& sh : g->s) {
    G_HIP(g, hipSetDevice(sh.device));
    G_PR(g, sh, pr_rerank_finish_dev(sh.ctx, sc ? sh.mom_all : nullptr, sc ? nullptr : sh.mom_all, G, sh.cand, sh.dump, sh.p5_all, G, m, kin, k,
                                     p_weight, sh.idx, sh.score));
  }
  if (int rc = mark(PH_EXACT)) return rc;
  // ... D + E. and the flagged queries (order not certain under the fp32 pass's sigmas | candidate list not provably complete: none, as a rule)
  // are answered from their exact rows, 64 per pass: this is a synchronous host call, so the count is read back and every pass runs
  int32_t flagged = 0;
  {
    Shard& s0 = g->s[0];
    G_HIP(g, hipSetDevice(s0.device));
    G_PR(g, s0, pr_order_flagged_count(s0.ctx, m, &flagged));
  }
  g->last_flagged = flagged;
  for (int32_t off = 0; off < flagged; off += 64) {
    for (auto& sh : g->s) {
      G_HIP(g, hipSetDevice(sh.device));
      G_PR(g, sh, pr_order_exact_moments_dev(sh.ctx, sc ? sh.raw_q : nullptr, sc ? sh.raw_db : nullptr, PR_F64, sc ? nullptr : sh.raw_q,
                                             sc ? nullptr : sh.raw_db, PR_F64, sc ? sh.mom_all : nullptr, sc ? nullptr : sh.mom_all, G, m, sh.rows,
                                             off, off + 64 >= flagged, sh.exact));
    }
    for (int r = 0; r < G; r++) { src[r] = g->s[r].exact; dst[r] = g->s[r].exact_all; }
    if (int rc = exchange(g, src, dst, (size_t)m * 12 * 8)) return rc;
    for (auto& sh : g->s) {
      G_HIP(g, hipSetDevice(sh.device));
      G_PR(g, sh, pr_order_exact_select_dev(sh.ctx, sh.exact_all, G, m, sh.rows, 0, sh.row0, mask_width, p_weight, sc ? 1 : 0, sc ? 0 : 1, k, off, sh.sel));
    }
    for (int r = 0; r < G; r++) { src[r] = g->s[r].sel; dst[r] = g->s[r].sel_all; }
    if (int rc = exchange(g, src, dst, (size_t)64 * 2 * k * 8)) return rc;
    for (auto& sh : g->s) {
      G_HIP(g, hipSetDevice(sh.device));
      G_PR(g, sh, pr_order_exact_merge_dev(sh.ctx, sh.sel_all, G, m, k, off, sh.idx, sh.score));
    }
  }
  if (int rc = mark(PR_GROUP_PHASES)) return rc;
  g->timed = g->timing;
  Shard& s0 = g->s[0];
  G_HIP(g, hipSetDevice(s0.device));
  G_HIP(g, hipMemcpyAsync(idx, s0.idx, (size_t)m * k * 4, hipMemcpyDeviceToHost, s0.stream));
  G_HIP(g, hipMemcpyAsync(score, s0.score, (size_t)m * k * 8, hipMemcpyDeviceToHost, s0.stream));
  for (auto& sh : g->s) { G_HIP(g, hipSetDevice(sh.device)); G_PR(g, sh, pr_sync(sh.ctx)); }
  return PR_OK;
}

// a failed call may have work enqueued on some devices: wait for all of them before the error is reported, and keep the message
static int settle(pr_group* g, int rc) {
  if (rc == PR_OK) return rc;
  const std::string keep = g->err;
  for (auto& sh : g->s) if (sh.ctx) { (void)hipSetDevice(sh.device); (void)hipStreamSynchronize(sh.stream); }
  g->err = keep;
  return rc;
}

int pr_group_set_database(pr_group* g, int type, const double* h2, int32_t n) {
  if (!g) return PR_EINVAL;
  return settle(g, set_database_impl(g, type, h2, n, 0));
}

int pr_group_set_database_growable(pr_group* g, int type, const double* h2, int32_t n, int32_t extra_capacity) {
  if (!g) return PR_EINVAL;
  return settle(g, set_database_impl(g, type, h2, n, extra_capacity));
}

// rows [n, n + n_new) of the database: onto the last shard, in place (raw rows + operand rows: pr_sigset_append) - SC/test_sc.cpp:40-56 adds a
// row per keyframe, run_test.m:57 matches the next one against all of them
static int append_database_impl(pr_group* g, const double* h_new, int32_t n_new) {
  if (g->type < 0) G_FAIL(g, PR_EINVAL, "pr_group_append_database: no database (pr_group_set_database_growable)");
  if (n_new < 0 || (n_new > 0 && !h_new)) G_FAIL(g, PR_EINVAL, "pr_group_append_database: bad arguments (n_new=%d)", n_new);
  if (n_new == 0) return PR_OK;
  Shard& sh = g->s[g->G - 1];
  if ((int64_t)sh.rows + n_new > sh.cap_rows)
    G_FAIL(g, PR_EINVAL, "pr_group_append_database: %d + %d rows exceed the last shard's capacity %d (pr_group_set_database_growable's extra_capacity)", sh.rows, n_new, sh.cap_rows);
  const size_t row_doubles = g->type == PR_TYPE_SC ? PR_SC_SIG_LEN : (size_t)4 * PR_M2DP_SIG_LEN;
  G_HIP(g, hipSetDevice(sh.device));
  double* dst = static_cast<double*>(sh.raw_db) + (size_t)sh.rows * row_doubles;
  G_HIP(g, hipMemcpyAsync(dst, h_new, (size_t)n_new * row_doubles * 8, hipMemcpyHostToDevice, sh.stream));
  G_PR(g, sh, pr_sigset_append(sh.ctx, sh.db, dst, PR_F64, PR_DEVICE, n_new));
  G_PR(g, sh, pr_sync(sh.ctx));                               // h_new may be released by the caller
  sh.rows += n_new;
  g->n += n_new;
  return PR_OK;
}

int pr_group_append_database(pr_group* g, const double* h_new, int32_t n_new) {
  if (!g) return PR_EINVAL;
  return settle(g, append_database_impl(g, h_new, n_new));
}

int32_t pr_group_database_rows(const pr_group* g) { return g ? g->n : 0; }

int pr_group_match_topk(pr_group* g, const double* h1, int32_t m, int32_t mask_width, double p_weight, int32_t k, int32_t* idx,
                        double* score) {
  if (!g) return PR_EINVAL;
  return settle(g, match_topk_impl(g, h1, m, mask_width, p_weight, k, idx, score));
}

// PR_WARN_* bits raised on any shard since the last call (zero-norm rows excluded, ...), then cleared
int pr_group_take_warnings(pr_group* g) {
  if (!g) return PR_EINVAL;
  int w = 0;
  for (auto& sh : g->s) if (sh.ctx) { (void)hipSetDevice(sh.device); const int x = pr_take_warnings(sh.ctx); if (x > 0) w |= x; }
  return w;
}

}  // extern "C"
