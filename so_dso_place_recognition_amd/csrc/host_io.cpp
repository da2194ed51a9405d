// host_io.cpp — host-side rows a1/a2 of the hot path (CPU in the reference too): PosesPts text records
// (PosesPts.h:5-40), the sliding-window pre-stage of both executables (utils/pts_preprocess.h:17-232) and the
// signature matrix text format (Eigen operator<< at test_sc.cpp:63-66 / test_m2dp.cpp:83-86).
//
// Design (not a transcription): files are slurped once and tokenised in place with strtol/strtod/strtof
// (same acceptance as `ifstream >>`: stop at the first failed extraction); history points live in SoA arrays, the
// per-pose "nearby" set is an index list, camera-frame points of a pose go into a reusable scratch array, and the
// best-point-per-cell filters keep only (cell -> scratch index) in a std::unordered_map<int,int>.  That container is
// kept on purpose: the ORDER of a cloud's points is the libstdc++ hash-map iteration order in the reference
// (pts_preprocess.h:85-89, :124-128) and the float sequential average of SC.cpp:60-64 depends on it;
// iteration order depends only on the key insertion sequence, which is reproduced exactly.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/place_recognition.h"
#include "records.hpp"

namespace {
using namespace pr_rec;

struct Scratch { std::vector<double> p; std::vector<int> src; };   // camera-frame points of the current pose

inline double nrm3(const double* p) { return std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); }

void emit(const std::unordered_map<int, int>& cell, const Scratch& s, const History& h, pr_clouds& out) {
  for (const auto& kv : cell) {
    const double* p = &s.p[3 * (size_t)kv.second];
    out.xyz.push_back(p[0]); out.xyz.push_back(p[1]); out.xyz.push_back(p[2]);
    out.inten.push_back(h.it[s.src[kv.second]]);
  }
}

// pts_preprocess.h:51-94: voxel grid lidarRange/{30,60,30}; keep the smallest camera-y, replace only if strictly smaller
void grid_filter(const Scratch& s, const History& h, double range, pr_clouds& out) {
  const double res[3] = {30, 60, 30};
  double step[3]; int dim[3];
  for (int a = 0; a < 3; a++) {
    const double r = range / res[a];
    step[a] = 1.0 / r;
    dim[a] = (int)(std::floor(2 * range * step[a]) + 1);
  }
  std::unordered_map<int, int> cell;
  const size_t n = s.src.size();
  for (size_t i = 0; i < n; i++) {
    const double* p = &s.p[3 * i];
    const int xi = (int)std::floor((p[0] + range) * step[0]);
    const int yi = (int)std::floor((p[1] + range) * step[1]);
    const int zi = (int)std::floor((p[2] + range) * step[2]);
    const int loc = xi + yi * dim[0] + zi * dim[0] * dim[1];
    auto it = cell.find(loc);
    if (it == cell.end()) cell[loc] = (int)i;
    else if (-s.p[3 * (size_t)it->second + 1] < -p[1]) it->second = (int)i;
  }
  emit(cell, s, h, out);
}

// pts_preprocess.h:96-133: 1 deg x 1 deg azimuth/elevation cells; keep the closest point, strict
void polar_filter(const Scratch& s, const History& h, pr_clouds& out) {
  const double res = 1.0 / 180.0 * M_PI;
  const double inv = 1.0 / res;
  const int azi_bins = (int)(std::floor(2 * M_PI * inv) + 1);
  std::unordered_map<int, int> cell;
  const size_t n = s.src.size();
  for (size_t i = 0; i < n; i++) {
    const double* p = &s.p[3 * i];
    const double xz = std::sqrt(p[0] * p[0] + p[2] * p[2]);
    const int azi = (int)std::floor((std::atan2(p[2], p[0]) + M_PI) * inv);
    const int ele = (int)std::floor((std::atan2(p[1], xz) + M_PI / 2) * inv);
    const int loc = azi + ele * azi_bins;
    auto it = cell.find(loc);
    if (it == cell.end()) cell[loc] = (int)i;
    else if (nrm3(&s.p[3 * (size_t)it->second]) > nrm3(p)) it->second = (int)i;
  }
  emit(cell, s, h, out);
}

thread_local std::string g_io_err;

// "%g"-style text of a double exactly as `os << double` with the default precision 6
// (std::to_chars with a precision is specified as printf("%.6g") in the "C" locale and is 2.7x faster than snprintf)
inline int fmt_g(char* b, size_t n, double v) { return (int)(std::to_chars(b, b + n, v, std::chars_format::general, 6).ptr - b); }

}  // namespace

extern "C" {

const char* pr_host_last_error(void) { return g_io_err.c_str(); }

// Replaces pts_preprocess(...) (utils/pts_preprocess.h:169-232).  verbose != 0 prints the reference's console lines.
int pr_pts_preprocess(const char* poses_file, const char* pts_file, const char* incoming_id_file, double lidarRange,
                      int polar, int verbose, pr_clouds** out) {
  if (!poses_file || !pts_file || !out || !(lidarRange > 0)) { g_io_err = "pr_pts_preprocess: bad arguments"; return PR_EINVAL; }
  std::vector<PoseRec> poses;
  History h;
  read_records(poses_file, pts_file, poses, h);
  FILE* idf = nullptr;
  if (incoming_id_file) {
    idf = fopen(incoming_id_file, "w");
    if (!idf) { g_io_err = std::string("cannot write ") + incoming_id_file; return PR_EIO; }
  }
  pr_clouds* res = new pr_clouds;
  res->offs.push_back(0);
  std::vector<int> nearby, keep;
  Scratch s;
  size_t cursor = 0;
  int since_reset = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (const PoseRec& ps : poses) {
    const double* w = ps.w;
    if (std::sqrt(w[3] * w[3] + w[7] * w[7] + w[11] * w[11]) < 1.0) {               // :189-193
      if (verbose) printf("\nReset at id: %d\n", ps.id);
      since_reset = 0;
      nearby.clear();
    }
    while (cursor < h.id.size() && h.id[cursor] <= ps.id) nearby.push_back((int)cursor++);   // :196-200
    if (since_reset < 30) { since_reset++; continue; }                             // INIT_FRAME :203-206
    s.p.clear(); s.src.clear(); keep.clear();
    for (int idx : nearby) {                                                        // :140-149
      const double* g = &h.xyz[3 * (size_t)idx];
      double l[3];
      for (int r = 0; r < 3; r++) l[r] = ((w[4 * r] * g[0] + w[4 * r + 1] * g[1]) + w[4 * r + 2] * g[2]) + w[4 * r + 3] * 1.0;
      if (nrm3(l) < lidarRange) {
        s.p.push_back(l[0]); s.p.push_back(l[1]); s.p.push_back(l[2]);
        s.src.push_back(idx);
        keep.push_back(idx);
      }
    }
    const size_t before = res->inten.size();
    if (polar) polar_filter(s, h, *res); else grid_filter(s, h, lidarRange, *res);
    if (verbose) {
      printf("\rFrame count: %d, Pts (Total: %lu, Sphere: %lu, Filtered: %lu)", ps.id, (unsigned long)nearby.size(),
             (unsigned long)s.src.size(), (unsigned long)(res->inten.size() - before));   // :160-163
      fflush(stdout);
    }
    nearby.swap(keep);                                                              // :166
    res->offs.push_back((int64_t)res->inten.size());
    res->ids.push_back(ps.id);
    if (idf) fprintf(idf, "%d\n", ps.id);                                            // :215
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const size_t N = res->ids.size();
  res->avg_ms = N ? 1000.0 * secs / N : NAN;
  res->avg_pts = N ? (double)res->inten.size() / N : NAN;
  if (verbose)
    printf("\ngenerate_spherical_points average time: %gms average points: %g\n", (double)(float)res->avg_ms,
           (double)(float)res->avg_pts);                                            // :221-225
  if (idf) fclose(idf);
  *out = res;
  return PR_OK;
}

int64_t pr_clouds_count(const pr_clouds* c) { return c ? (int64_t)c->ids.size() : 0; }
const int64_t* pr_clouds_offs(const pr_clouds* c) { return c->offs.data(); }
const double* pr_clouds_xyz(const pr_clouds* c) { return c->xyz.data(); }
const float* pr_clouds_inten(const pr_clouds* c) { return c->inten.data(); }
const int32_t* pr_clouds_ids(const pr_clouds* c) { return c->ids.data(); }
double pr_clouds_avg_ms(const pr_clouds* c) { return c ? c->avg_ms : 0.0; }
double pr_clouds_avg_pts(const pr_clouds* c) { return c ? c->avg_pts : 0.0; }
const double* pr_clouds_dev_xyz(const pr_clouds* c) { return c ? static_cast<const double*>(c->d_xyz) : nullptr; }
const float* pr_clouds_dev_inten(const pr_clouds* c) { return c ? static_cast<const float*>(c->d_inten) : nullptr; }
const int64_t* pr_clouds_dev_offs(const pr_clouds* c) { return c ? static_cast<const int64_t*>(c->d_offs) : nullptr; }
const double* pr_clouds_dev_frames(const pr_clouds* c) { return c ? static_cast<const double*>(c->d_frames) : nullptr; }
void pr_clouds_free(pr_clouds* c) {
  if (c && c->release) c->release(c);
  delete c;
}

// Text format of `ofstream << Eigen::MatrixXd` (test_sc.cpp:63-66): default IOFormat = stream precision (6 significant
// digits), " " between coefficients, "\n" between rows, no trailing newline, every coefficient right-aligned to the
// width of the widest one in the whole matrix.
int pr_write_signatures(const char* path, const double* sig, int64_t rows, int64_t cols) {
  if (!path || (!sig && rows * cols > 0) || rows < 0 || cols < 0) { g_io_err = "pr_write_signatures: bad arguments"; return PR_EINVAL; }
  FILE* f = fopen(path, "w");
  if (!f) { g_io_err = std::string("cannot write ") + path; return PR_EIO; }
  // Eigen pads every coefficient to the width of the widest one (IOFormat default, Core/IO.h): two passes.  Both are
  // snprintf-bound (130 ns per number: 30 s for a 100k x 2400 DB), so rows are formatted by host_threads() threads in
  // blocks and the blocks written in order.
  const unsigned T = host_threads((size_t)rows * (size_t)cols / 65536);
  std::vector<int> wmax(T, 0);
  run_threads(T, [&](unsigned t) {
    char b[64];
    int w = 0;
    const int64_t n = rows * cols, i0 = n * t / T, i1 = n * (t + 1) / T;
    for (int64_t i = i0; i < i1; i++) { const int k = fmt_g(b, sizeof b, sig[i]); if (k > w) w = k; }
    wmax[t] = w;
  });
  const int width = *std::max_element(wmax.begin(), wmax.end());
  const int64_t block = std::max<int64_t>(1, (int64_t)(((size_t)64 << 20) / ((size_t)(width + 1) * (size_t)std::max<int64_t>(cols, 1)))) * T;
  std::vector<std::string> part(T);
  bool ok = true;
  for (int64_t r0 = 0; r0 < rows && ok; r0 += block) {
    const int64_t nr = std::min(block, rows - r0);
    run_threads(T, [&](unsigned t) {
      char b[64];
      std::string& o = part[t];
      o.clear();
      for (int64_t r = r0 + nr * t / T; r < r0 + nr * (t + 1) / T; r++) {
        if (r) o.push_back('\n');
        for (int64_t c = 0; c < cols; c++) {
          if (c) o.push_back(' ');
          const int k = fmt_g(b, sizeof b, sig[r * cols + c]);
          o.append((size_t)(width - k), ' ');
          o.append(b, (size_t)k);
        }
      }
    });
    for (unsigned t = 0; t < T && ok; t++) ok = part[t].empty() || fwrite(part[t].data(), 1, part[t].size(), f) == part[t].size();
  }
  ok = (fclose(f) == 0) && ok;
  if (!ok) { g_io_err = std::string("short write to ") + path; return PR_EIO; }
  return PR_OK;
}

// Whitespace-tolerant reader of the same files (MATLAB `load`, test_kitti.m:26): rows = lines, cols from the first line.
// Numbers are whatever strtod takes (`nan`, `inf` included - Eigen prints them and `load` reads them); reading stops at
// the first thing that is not a number.  The buffer is cut at line ends and parsed by host_threads() threads; a chunk
// that stops early sends the whole file to the one-thread loop, which then stops where it always did.
int pr_read_signatures(const char* path, double** out, int64_t* rows, int64_t* cols) {
  if (!path || !out || !rows || !cols) { g_io_err = "pr_read_signatures: bad arguments"; return PR_EINVAL; }
  std::string buf;
  if (!slurp(path, buf)) { g_io_err = std::string("cannot read ") + path; return PR_EIO; }
  int64_t nc = 0;
  {
    const char* p = buf.c_str();
    while (*p == '\n' || *p == '\r') p++;               // leading blank lines
    while (true) {
      while (*p == ' ' || *p == '\t' || *p == '\r') p++;
      if (!*p || *p == '\n') break;
      char* e; strtod(p, &e);
      if (e == p) break;
      nc++; p = e;
    }
  }
  const size_t n = buf.size();
  const unsigned T = host_threads(n >> 20);
  std::vector<size_t> cut(T + 1, n);
  cut[0] = 0;
  for (unsigned t = 1; t < T; t++) {
    size_t q = n / T * t;
    while (q < n && buf[q] != '\n') q++;
    cut[t] = q;                                          // the '\n' itself starts the next chunk: strtod skips it
  }
  std::vector<std::vector<double>> part(T);
  std::vector<char> stopped(T, 0);
  auto parse = [&](size_t b0, size_t b1, std::vector<double>& v) {   // true = consumed the whole range
    const char* p = buf.c_str() + b0;
    const char* end = buf.c_str() + b1;
    v.reserve((b1 - b0) / 8 + 16);
    while (true) {
      while (p < end && is_space(*p)) p++;
      if (p >= end) return true;
      char* e;
      const double x = strtod(p, &e);
      if (e == p) return false;
      if (e > end) return false;                         // cannot happen when the cut is a line end; be safe
      v.push_back(x);
      p = e;
    }
  };
  run_threads(T, [&](unsigned t) { stopped[t] = !parse(cut[t], cut[t + 1], part[t]); });
  bool any = false;
  for (unsigned t = 0; t + 1 < T; t++) any = any || stopped[t];
  if (any) {                                             // junk before the last chunk: only what precedes it counts
    for (auto& v : part) v.clear();
    parse(0, n, part[0]);
  }
  size_t tot = 0;
  for (auto& v : part) tot += v.size();
  if (nc == 0 || tot % (size_t)nc) { g_io_err = std::string("ragged signature file ") + path; return PR_EIO; }
  *cols = nc; *rows = (int64_t)(tot / (size_t)nc);
  *out = (double*)malloc(tot * sizeof(double) + 8);
  if (!*out) { g_io_err = "out of memory"; return PR_ENOMEM; }
  size_t o = 0;
  for (auto& v : part) { if (!v.empty()) memcpy(*out + o, v.data(), v.size() * sizeof(double)); o += v.size(); }
  return PR_OK;
}

void pr_free(void* p) { free(p); }

// Binary side-car of the signature text files: 32-byte header {magic "PRSIG1\0\0", u32 dtype
// (0 = f64, 1 = f32), u32 reserved, u64 rows, u64 cols} followed by the row-major matrix; mem-mappable, loads at
// NVMe speed where the 6-significant-digit text of a 100k x 2400 DB is 2 GB of strtod.
namespace {
struct BinHeader { char magic[8]; uint32_t dtype, reserved; uint64_t rows, cols; };
}
int pr_write_signatures_bin(const char* path, const double* sig, int64_t rows, int64_t cols, int dtype) {
  if (!path || (!sig && rows * cols > 0) || rows < 0 || cols < 0 || (dtype != PR_F64 && dtype != PR_F32)) {
    g_io_err = "pr_write_signatures_bin: bad arguments"; return PR_EINVAL;
  }
  FILE* f = fopen(path, "wb");
  if (!f) { g_io_err = std::string("cannot write ") + path; return PR_EIO; }
  BinHeader h; memset(&h, 0, sizeof h);
  memcpy(h.magic, "PRSIG1", 6); h.dtype = (uint32_t)dtype; h.rows = (uint64_t)rows; h.cols = (uint64_t)cols;
  bool ok = fwrite(&h, sizeof h, 1, f) == 1;
  const size_t n = (size_t)rows * (size_t)cols;
  if (dtype == PR_F64) ok = ok && (n == 0 || fwrite(sig, 8, n, f) == n);
  else {
    std::vector<float> tmp(1 << 16);
    for (size_t i = 0; i < n && ok; i += tmp.size()) {
      const size_t c = std::min(tmp.size(), n - i);
      for (size_t k = 0; k < c; k++) tmp[k] = (float)sig[i + k];
      ok = fwrite(tmp.data(), 4, c, f) == c;
    }
  }
  fclose(f);
  if (!ok) { g_io_err = std::string("short write to ") + path; return PR_EIO; }
  return PR_OK;
}

int pr_read_signatures_bin(const char* path, double** out, int64_t* rows, int64_t* cols) {
  if (!path || !out || !rows || !cols) { g_io_err = "pr_read_signatures_bin: bad arguments"; return PR_EINVAL; }
  FILE* f = fopen(path, "rb");
  if (!f) { g_io_err = std::string("cannot read ") + path; return PR_EIO; }
  BinHeader h;
  if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "PRSIG1", 6) != 0 || h.dtype > 1) {
    fclose(f); g_io_err = std::string("not a PRSIG1 file: ") + path; return PR_EIO;
  }
  const size_t n = (size_t)h.rows * (size_t)h.cols;
  double* buf = (double*)malloc(n * sizeof(double) + 8);
  if (!buf) { fclose(f); g_io_err = "out of memory"; return PR_ENOMEM; }
  bool ok = true;
  if (h.dtype == PR_F64) ok = (n == 0 || fread(buf, 8, n, f) == n);
  else {
    std::vector<float> tmp(1 << 16);
    for (size_t i = 0; i < n && ok; i += tmp.size()) {
      const size_t c = std::min(tmp.size(), n - i);
      ok = fread(tmp.data(), 4, c, f) == c;
      for (size_t k = 0; k < c && ok; k++) buf[i + k] = tmp[k];
    }
  }
  fclose(f);
  if (!ok) { free(buf); g_io_err = std::string("truncated file ") + path; return PR_EIO; }
  *out = buf; *rows = (int64_t)h.rows; *cols = (int64_t)h.cols;
  return PR_OK;
}

// PosesPts.h:12-24 / :36-39 writers (used by tests and the synthetic-input tool)
int pr_write_poses(const char* path, const int32_t* ids, const double* w2c /*[n][12]*/, int64_t n) {
  FILE* f = fopen(path, "w");
  if (!f) { g_io_err = std::string("cannot write ") + path; return PR_EIO; }
  for (int64_t i = 0; i < n; i++) {
    fprintf(f, "%d ", ids[i]);
    for (int k = 0; k < 12; k++) fprintf(f, "%g ", w2c[i * 12 + k]);
    fputc('\n', f);
  }
  fclose(f);
  return PR_OK;
}

int pr_write_points(const char* path, const int32_t* ids, const double* xyz, const float* inten, int64_t n) {
  FILE* f = fopen(path, "w");
  if (!f) { g_io_err = std::string("cannot write ") + path; return PR_EIO; }
  for (int64_t i = 0; i < n; i++)
    fprintf(f, "%d %g %g %g %g\n", ids[i], xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], (double)inten[i]);
  fclose(f);
  return PR_OK;
}

// run_test.m:3-22 + :58-85: ground-truth loop pairs, precision / recall sweep over the queries sorted by their best score,
// top recall at 100 % precision, AUC = trapz(recall, precision).  gt1 [m][cols], gt2 [n][cols] positions of the signatures
// (test_kitti.m:23-25: columns 4, 8, 12 of gt.txt; test_robotcar.m:31-36: gps.txt), diff_idx 0-based.
int pr_precision_recall(const double* diff_v, const int32_t* diff_idx, int32_t m, const double* gt1, const double* gt2, int32_t n,
                        int32_t cols, double loop_diff, int32_t mask_width, double* auc, double* top_recall, int32_t* lp_detected,
                        int32_t* n_detected) {
  if (m < 0 || n < 0 || cols < 1 || (m > 0 && (!diff_v || !diff_idx || !gt1)) || (n > 0 && !gt2) || !auc || !top_recall) {
    g_io_err = "pr_precision_recall: bad arguments";
    return PR_EINVAL;
  }
  const double thr = loop_diff * loop_diff;
  auto d2 = [&](int a, int b) {
    double s = 0.0;
    for (int c = 0; c < cols; c++) { const double t = gt1[(size_t)a * cols + c] - gt2[(size_t)b * cols + c]; s += t * t; }
    return s;
  };
  // ground-truth pairs (:3-22): for every i the closest j with |i - j| >= mask_width (first minimum), kept when closer than loop_diff
  int64_t L = 0;
  for (int i = 0; i < m; i++) {
    double best = INFINITY;
    bool any = false;
    for (int j = 0; j < n; j++) {
      if (std::abs(i - j) < mask_width) continue;
      const double d = d2(i, j);
      if (d < best) { best = d; any = true; }
    }
    if (any && best < thr) L++;
  }
  const int64_t total_lp = L == 0 ? 0 : (L < 2 ? 2 : L);     // MATLAB length() of an L x 2 matrix (:22)
  std::vector<int> rank((size_t)m);
  for (int i = 0; i < m; i++) rank[i] = i;
  std::stable_sort(rank.begin(), rank.end(), [&](int a, int b) {    // [~, diff_rank] = sort(diff_v): ascending, NaN last, stable
    const double x = diff_v[a], y = diff_v[b];
    if (std::isnan(x) || std::isnan(y)) return !std::isnan(x) && std::isnan(y);
    return x < y;
  });
  int64_t tp = 0, fp = 0;
  int top_count = 0;
  double tr = 0.0, area = 0.0, pprev = 0.0, rprev = 0.0;
  for (int i = 0; i < m; i++) {
    // a query without a finite candidate (every distance NaN: zero-norm signature; or everything masked) comes back as index -1: MATLAB's
    // min over an all-NaN / all-Inf row returns index 1 (:57), and the sweep pairs the query with gt2(1,:)
    const int a = rank[i], b = diff_idx[a] < 0 ? 0 : diff_idx[a];
    if (b < n && d2(a, b) < thr) tp++; else fp++;
    const double p = (double)tp / (double)(tp + fp);
    const double r = (double)tp / (double)total_lp;              // :77 as MATLAB divides: no ground-truth pair -> 0/0 = NaN (x/0 = Inf)
    if (p == 1.0) { top_count = i + 1; tr = r; }
    if (i > 0) area += (r - rprev) * (p + pprev) / 2.0;      // trapz(recall, precision) (:85)
    pprev = p; rprev = r;
  }
  *auc = area;
  *top_recall = tr;
  if (n_detected) *n_detected = top_count;
  if (lp_detected)
    for (int i = 0; i < top_count; i++) { lp_detected[2 * i] = rank[i]; lp_detected[2 * i + 1] = diff_idx[rank[i]] < 0 ? 0 : diff_idx[rank[i]]; }
  return PR_OK;
}

}  // extern "C"
