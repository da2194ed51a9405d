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
inline int fmt_g(char* b, size_t n, double v) { return snprintf(b, n, "%g", v); }

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
void pr_clouds_free(pr_clouds* c) { delete c; }

// Text format of `ofstream << Eigen::MatrixXd` (test_sc.cpp:63-66): default IOFormat = stream precision (6 significant
// digits), " " between coefficients, "\n" between rows, no trailing newline, every coefficient right-aligned to the
// width of the widest one in the whole matrix.
int pr_write_signatures(const char* path, const double* sig, int64_t rows, int64_t cols) {
  if (!path || (!sig && rows * cols > 0) || rows < 0 || cols < 0) { g_io_err = "pr_write_signatures: bad arguments"; return PR_EINVAL; }
  FILE* f = fopen(path, "w");
  if (!f) { g_io_err = std::string("cannot write ") + path; return PR_EIO; }
  char b[64];
  int width = 0;
  for (int64_t i = 0; i < rows * cols; i++) { const int n = fmt_g(b, sizeof b, sig[i]); if (n > width) width = n; }
  std::string line;
  for (int64_t r = 0; r < rows; r++) {
    line.clear();
    if (r) line.push_back('\n');
    for (int64_t c = 0; c < cols; c++) {
      if (c) line.push_back(' ');
      const int n = fmt_g(b, sizeof b, sig[r * cols + c]);
      line.append((size_t)(width - n), ' ');
      line.append(b, (size_t)n);
    }
    fwrite(line.data(), 1, line.size(), f);
  }
  fclose(f);
  return PR_OK;
}

// Whitespace-tolerant reader of the same files (MATLAB `load`, test_kitti.m:26): rows = lines, cols from the first line.
int pr_read_signatures(const char* path, double** out, int64_t* rows, int64_t* cols) {
  if (!path || !out || !rows || !cols) { g_io_err = "pr_read_signatures: bad arguments"; return PR_EINVAL; }
  std::string buf;
  if (!slurp(path, buf)) { g_io_err = std::string("cannot read ") + path; return PR_EIO; }
  int64_t nc = 0;
  {
    const char* p = buf.c_str();
    while (*p && *p != '\n') {
      char* e; strtod(p, &e);
      if (e == p) { p++; continue; }
      nc++; p = e;
    }
  }
  std::vector<double> v;
  Cursor c{buf.c_str()};
  double x;
  while (c.next_double(x)) v.push_back(x);
  if (nc == 0 || v.size() % (size_t)nc) { g_io_err = std::string("ragged signature file ") + path; return PR_EIO; }
  *cols = nc; *rows = (int64_t)(v.size() / (size_t)nc);
  *out = (double*)malloc(v.size() * sizeof(double) + 8);
  if (!*out) { g_io_err = "out of memory"; return PR_ENOMEM; }
  memcpy(*out, v.data(), v.size() * sizeof(double));
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

}  // extern "C"
