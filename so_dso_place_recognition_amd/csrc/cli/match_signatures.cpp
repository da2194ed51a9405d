// match_signatures — executable counterpart of match_signatures/run_test.m:25-57 (the reference runs it from MATLAB:
// test_kitti.m:18-28).  Options mirror run_test's arguments:
//   --type sc|m2dp|delight|gist|bow --hist1 F --hist2 F [--mask_width W=0] [--p_weight 2] [--topk K=1] [--one_based 0|1] --out F
//   [--devices 0,1,...]               hist2 row-sharded over these GPUs (pr_group: RCCL all-gathers between the kernels; sc | m2dp)
//   [--gt1 F --gt2 F --loop_diff L]   positions of the signatures (text matrices, one row per signature): the evaluation half of
//                                     run_test (run_test.m:3-22, :58-85) - prints `AUC = ...` and `top_recall = ...`
//   [--exact_statistics 0|1]          every query answered from its exact fp64 row (scores = the reference's doubles to rounding).  Default: 1
//                                     with --gt1 / --gt2 - the sweep ranks the QUERIES by score (run_test.m:58), and two queries whose scores
//                                     agree to 1e-5 must not change places -, else 0
//   [--online 1]                      the causal form of the same match (what a running SLAM front end asks: SC/test_sc.cpp:40-56 produces one row per
//                                     keyframe): row t of hist1 is matched against rows 0 .. t - mask_width of hist1 ITSELF (hist2 is not read; the
//                                     mask of run_test.m:47-53 restricted to the past), then joins the database - resident on the GPU and grown IN
//                                     PLACE (pr_group_set_database_growable / pr_group_append_database), one pr_group_match_topk per keyframe.
//                                     Rows with fewer than two candidates come out as -1 NaN.  sc | m2dp; --devices optional (default: device 0)
// Output: one line per query: K pairs "index score" (0-based indices unless --one_based 1), and the reference's
// console lines `type` / `tm` (ms per query, run_test.m:42-44).  Scores are doubles, as MATLAB holds them.
#include <chrono>
#include <cmath>
#include <vector>

#include "../../../include/place_recognition.h"
#include "cli_common.hpp"

int main(int argc, char** argv) {
  Params prm(argc, argv);
  std::string type, h1f, h2f, outf;
  if (!prm.get("type", type) || !prm.get("hist1", h1f) || !prm.get("hist2", h2f) || !prm.get("out", outf) ||
      (type != "sc" && type != "m2dp" && type != "delight" && type != "gist" && type != "bow")) {
    printf("usage: match_signatures --type sc|m2dp|delight|gist|bow --hist1 F --hist2 F [--mask_width W] [--p_weight 2] [--topk K] [--one_based 0|1] --out F\n");
    return 1;
  }
  const int t = type == "sc" ? PR_TYPE_SC : type == "m2dp" ? PR_TYPE_M2DP : type == "delight" ? PR_TYPE_DELIGHT : type == "gist" ? PR_TYPE_GIST : PR_TYPE_BOW;
  const bool cols_type = t == PR_TYPE_GIST || t == PR_TYPE_BOW;        // no fixed signature length
  const int div = t == PR_TYPE_SC ? 1 : t == PR_TYPE_M2DP ? 4 : t == PR_TYPE_DELIGHT ? 16 : t == PR_TYPE_BOW ? 2 : 1;
  int64_t width = t == PR_TYPE_SC ? PR_SC_SIG_LEN : (t == PR_TYPE_M2DP ? PR_M2DP_SIG_LEN : PR_DELIGHT_SIG_LEN);
  double *h1 = nullptr, *h2 = nullptr;
  int64_t r1, c1, r2, c2;
  auto rd = [](const std::string& f, double** o, int64_t* r, int64_t* c) {
    const bool bin = f.size() > 4 && f.compare(f.size() - 4, 4, ".bin") == 0;
    return bin ? pr_read_signatures_bin(f.c_str(), o, r, c) : pr_read_signatures(f.c_str(), o, r, c);
  };
  if (rd(h1f, &h1, &r1, &c1) != PR_OK || rd(h2f, &h2, &r2, &c2) != PR_OK) {
    fprintf(stderr, "%s\n", pr_host_last_error());
    return 2;
  }
  if (cols_type) width = c1;
  if (c1 != width || c2 != width || r1 % div || r2 % div) { fprintf(stderr, "signature files must be [%d*m x %ld]\n", div, (long)width); return 2; }
  const int32_t m = (int32_t)(r1 / div), n = (int32_t)(r2 / div), k = (int32_t)prm.num("topk", 1);
  std::vector<int32_t> idx((size_t)m * k);
  std::vector<double> score((size_t)m * k);
  std::vector<float> score32;
  pr_ctx* ctx = nullptr;
  std::string devs;
  std::vector<int32_t> dev_ids;
  if (prm.get("devices", devs)) {
    for (size_t p = 0; p < devs.size();) { dev_ids.push_back(atoi(devs.c_str() + p)); p = devs.find(',', p); if (p == std::string::npos) break; p++; }
    if (dev_ids.empty() || (t != PR_TYPE_SC && t != PR_TYPE_M2DP)) { fprintf(stderr, "--devices needs a device list and --type sc|m2dp\n"); return 1; }
  }
  // one context for the single-device call; with --devices the group owns one context per shard and this one is not needed
  if (dev_ids.empty() && pr_create((int)prm.num("device", 0), &ctx) != PR_OK) { fprintf(stderr, "%s\n", pr_last_error(nullptr)); return 3; }
  pr_group* grp = nullptr;
  if (!dev_ids.empty()) {
    if (pr_group_create(dev_ids.data(), (int32_t)dev_ids.size(), &grp) != PR_OK) { fprintf(stderr, "%s\n", pr_group_last_error(nullptr)); return 3; }
    if (pr_group_set_database(grp, t, h2, n) != PR_OK) { fprintf(stderr, "%s\n", pr_group_last_error(grp)); return 4; }
    printf("devices = %d (%s)\n", (int)dev_ids.size(), pr_group_uses_rccl(grp) ? "RCCL" : "copies");
  }
  {
    std::string a, b;
    const bool exact = prm.num("exact_statistics", (prm.get("gt1", a) && prm.get("gt2", b)) ? 1.0 : 0.0) != 0.0;
    if (exact && ctx) (void)pr_set_exact_statistics(ctx, 1);
    if (exact && grp) (void)pr_group_set_exact_statistics(grp, 1);
  }
  const bool online = prm.num("online", 0) != 0;
  if (online && t != PR_TYPE_SC && t != PR_TYPE_M2DP) { fprintf(stderr, "--online needs --type sc|m2dp\n"); return 1; }
  if (online && !grp) {                                                  // the growing database lives in a pr_group (one device unless --devices)
    const int32_t d0 = (int32_t)prm.num("device", 0);
    if (pr_group_create(&d0, 1, &grp) != PR_OK) { fprintf(stderr, "%s\n", pr_group_last_error(nullptr)); return 3; }
  }
  const auto t0 = std::chrono::steady_clock::now();
  int rc;
  if (online) {
    const int32_t mask = (int32_t)prm.num("mask_width", 0), lag = mask > 1 ? mask - 1 : 0;
    const size_t rowd = (size_t)div * (size_t)width;                    // doubles per signature
    int32_t have = 0;                                                   // rows of hist1 in the database: rows 0 .. have - 1
    rc = PR_OK;
    for (int32_t q = 0; q < m && rc == PR_OK; q++) {
      const int32_t want = q - lag;                                     // rows the mask admits for keyframe q: 0 .. q - lag - 1
      if (want >= 2 && have == 0) {                                     // two rows: the first database (N - 1 standard deviation)
        rc = pr_group_set_database_growable(grp, t, h1, 2, m);
        have = 2;
      }
      if (rc == PR_OK && have > 0 && want > have) { rc = pr_group_append_database(grp, h1 + (size_t)have * rowd, want - have); have = want; }
      if (rc != PR_OK) break;
      if (have < 2) { for (int32_t j = 0; j < k; j++) { idx[(size_t)q * k + j] = -1; score[(size_t)q * k + j] = NAN; } continue; }
      rc = pr_group_match_topk(grp, h1 + (size_t)q * rowd, 1, 0, prm.num("p_weight", 2.0), k, idx.data() + (size_t)q * k, score.data() + (size_t)q * k);
    }
    if (rc != PR_OK) { fprintf(stderr, "online match failed (%d): %s\n", rc, pr_group_last_error(grp)); return 4; }
    printf("online = 1 (database rows at the end: %d)\n", (int)pr_group_database_rows(grp));
  } else if (grp) {
    rc = pr_group_match_topk(grp, h1, m, (int32_t)prm.num("mask_width", 0), prm.num("p_weight", 2.0), k, idx.data(), score.data());
    if (rc != PR_OK) { fprintf(stderr, "match failed (%d): %s\n", rc, pr_group_last_error(grp)); return 4; }
  } else if (cols_type) {
    score32.resize(score.size());
    rc = pr_match_topk_cols(ctx, t, h1, m, h2, n, (int32_t)width, (int32_t)prm.num("mask_width", 0), k, idx.data(), score32.data());
    for (size_t i = 0; i < score.size(); i++) score[i] = (double)score32[i];
  } else {
    rc = pr_match_topk_f64(ctx, t, h1, m, h2, n, (int32_t)prm.num("mask_width", 0), prm.num("p_weight", 2.0), k, idx.data(), score.data());
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (rc != PR_OK) { fprintf(stderr, "match failed (%d): %s\n", rc, pr_last_error(ctx)); pr_destroy(ctx); return 4; }
  printf("type = %s\ntm = %g\n", type.c_str(), m ? 1000.0 * secs / m : 0.0);
  const int warn = grp ? pr_group_take_warnings(grp) : pr_take_warnings(ctx);
  if (warn > 0 && (warn & PR_WARN_F16_FALLBACK)) printf("note: some queries were recomputed in split-f16 (PR_SC_ARITH_F16 margin check)\n");
  if (warn > 0 && (warn & PR_WARN_ORDER_RESOLVED)) printf("note: some queries were answered with fp64 row statistics (order check, pr_order_resolve_dev)\n");
  if (warn > 0 && (warn & PR_WARN_NAN_ROWS)) printf("warning: zero-norm signature rows never match (NaN in MATLAB, processSC.m:16,19)\n");
  std::string g1f, g2f;
  if (prm.get("gt1", g1f) && prm.get("gt2", g2f)) {   // run_test.m:3-22, 58-85
    double *g1 = nullptr, *g2 = nullptr;
    int64_t gr1, gc1, gr2, gc2;
    if (rd(g1f, &g1, &gr1, &gc1) != PR_OK || rd(g2f, &g2, &gr2, &gc2) != PR_OK) { fprintf(stderr, "%s\n", pr_host_last_error()); pr_destroy(ctx); return 2; }
    if (gr1 != m || gr2 != n || gc1 != gc2) { fprintf(stderr, "gt files must hold one row per signature (%d / %d rows, same columns)\n", m, n); pr_destroy(ctx); return 2; }
    std::vector<double> v((size_t)m);
    std::vector<int32_t> bi((size_t)m);
    for (int32_t i = 0; i < m; i++) { v[i] = score[(size_t)i * k]; bi[i] = idx[(size_t)i * k]; }
    double auc = 0, tr = 0;
    int32_t nd = 0;
    if (pr_precision_recall(v.data(), bi.data(), m, g1, g2, n, (int32_t)gc1, prm.num("loop_diff", 10.0), (int32_t)prm.num("mask_width", 0), &auc, &tr,
                            nullptr, &nd) != PR_OK) { fprintf(stderr, "%s\n", pr_host_last_error()); pr_destroy(ctx); return 4; }
    printf("AUC = %.9g\ntop_recall = %.9g\nlp_detected = %d\n", auc, tr, nd);
    pr_free(g1); pr_free(g2);
  }
  FILE* f = fopen(outf.c_str(), "w");
  if (!f) { fprintf(stderr, "cannot write %s\n", outf.c_str()); pr_destroy(ctx); return 5; }
  const int base = prm.num("one_based", 0) != 0 ? 1 : 0;
  for (int32_t i = 0; i < m; i++) {
    for (int32_t j = 0; j < k; j++) fprintf(f, "%s%d %.17g", j ? " " : "", idx[(size_t)i * k + j] < 0 ? -1 : idx[(size_t)i * k + j] + base, score[(size_t)i * k + j]);
    fputc('\n', f);
  }
  fclose(f);
  pr_free(h1); pr_free(h2);
  pr_group_destroy(grp);
  pr_destroy(ctx);
  return 0;
}
