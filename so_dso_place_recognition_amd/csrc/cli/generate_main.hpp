// generate_main.hpp — shared body of test_sc / test_m2dp (SC/test_sc.cpp:12-69, M2DP/test_m2dp.cpp:13-89):
// same parameters, same exit code / message when one is missing, same console lines, same output files.
#pragma once
#include <chrono>
#include <vector>

#include "../../../include/place_recognition.h"
#include "cli_common.hpp"

// kind: 0 = SC (test_sc.cpp), 1 = M2DP (test_m2dp.cpp), 2 = DELIGHT (DELIGHT/test_delight.cpp:12-68)
inline int generate_main(int argc, char** argv, int kind) {
  Params prm(argc, argv);
  const bool m2dp = kind == 1, delight = kind == 2;
  const char* out_name = delight ? "delight_file" : (m2dp ? "m2dp_file" : "sc_file");
  std::string poses, pts, outf, idf;
  if (!prm.get("poses_history_file", poses) || !prm.get("pts_history_file", pts) || !prm.get(out_name, outf) ||
      !prm.get("incoming_id_file", idf)) {
    printf("Fail to get params, exit.\n");            // test_sc.cpp:23-24
    return 1;
  }
  const double lidarRange = prm.num("lidarRange", 45.0);   // :27-28
  pr_clouds* clouds = nullptr;
  pr_ctx* ctx = nullptr;
  // PR_CLI_TIMING=1: wall time of each phase on stderr (the console lines of the reference stay as they are)
  const bool timing = getenv("PR_CLI_TIMING") != nullptr;
  auto tp = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    const auto now = std::chrono::steady_clock::now();
    if (timing) fprintf(stderr, "[timing] %-28s %8.3f s\n", what, std::chrono::duration<double>(now - tp).count());
    tp = now;
  };
  int rc = pr_create((int)prm.num("device", 0), &ctx);
  if (rc != PR_OK) { fprintf(stderr, "%s\n", pr_last_error(nullptr)); return 3; }
  lap("create context");
  // pts_preprocess on the device by default (the same clouds bit for bit, 9-20x less wall time end to end);
  // _gpu_prestage:=0 runs the host restatement of utils/pts_preprocess.h instead
  const bool gpu_pre = prm.num("gpu_prestage", 1.0) != 0.0;
  rc = gpu_pre ? pr_pts_preprocess_gpu(ctx, poses.c_str(), pts.c_str(), idf.c_str(), lidarRange, (m2dp || delight) ? 1 : 0, 1, &clouds)
               : pr_pts_preprocess(poses.c_str(), pts.c_str(), idf.c_str(), lidarRange, (m2dp || delight) ? 1 : 0, 1, &clouds);
  if (rc != PR_OK) {
    fprintf(stderr, "pts_preprocess failed: %s\n", gpu_pre ? pr_last_error(ctx) : pr_host_last_error());
    pr_destroy(ctx);
    return 2;
  }
  lap(gpu_pre ? "read + pre-stage (gpu)" : "read + pre-stage (host)");
  const int32_t N = (int32_t)pr_clouds_count(clouds);
  const size_t rows = delight ? (size_t)16 * N : (m2dp ? (size_t)4 * N : (size_t)N);
  const size_t cols = delight ? PR_DELIGHT_SIG_LEN : (m2dp ? PR_M2DP_SIG_LEN : PR_SC_SIG_LEN);
  std::vector<double> sig(rows * cols);
  const auto t0 = std::chrono::steady_clock::now();
  // (clouds the GPU pre-stage left in HBM are binned there with the frames it emitted; host-made clouds are uploaded and take both passes)
  rc = pr_generate_clouds(ctx, delight ? PR_TYPE_DELIGHT : (m2dp ? PR_TYPE_M2DP : PR_TYPE_SC), clouds, lidarRange, sig.data());
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (rc != PR_OK) { fprintf(stderr, "generate failed: %s\n", pr_last_error(ctx)); pr_destroy(ctx); pr_clouds_free(clouds); return 4; }
  lap("generate (incl. copies)");
  if (m2dp && (pr_take_warnings(ctx) & PR_WARN_M2DP_SVD)) {    // rows whose leading singular pair is not unique (may differ from the reference's, N6)
    std::vector<int32_t> rws(1024);
    int32_t cnt = 0;
    if (pr_m2dp_svd_rows(ctx, rws.data(), (int32_t)rws.size(), &cnt) == PR_OK) {
      fprintf(stderr, "warning: leading singular pair not unique in %d row(s):", cnt);
      for (int i = 0; i < cnt && i < (int)rws.size(); i++) fprintf(stderr, " %d", rws[i]);
      fprintf(stderr, "\n");
    }
  }
  printProgress(N ? 1.0 : 0.0);
  printf("\n%s average time: %gms\n", delight ? "DELIGHT" : (m2dp ? "M2DP" : "SC"), N ? 1000.0 * secs / N : 0.0);   // test_sc.cpp:58-61
  const bool bin = outf.size() > 4 && outf.compare(outf.size() - 4, 4, ".bin") == 0;
  rc = bin ? pr_write_signatures_bin(outf.c_str(), sig.data(), (int64_t)rows, (int64_t)cols, PR_F64)
           : pr_write_signatures(outf.c_str(), sig.data(), (int64_t)rows, (int64_t)cols);   // :63-66
  if (rc != PR_OK) fprintf(stderr, "%s\n", pr_host_last_error());
  lap(bin ? "write signatures (.bin)" : "write signatures (text)");
  pr_destroy(ctx);
  pr_clouds_free(clouds);
  return rc == PR_OK ? 0 : 5;
}
