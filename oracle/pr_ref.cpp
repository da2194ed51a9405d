// oracle/pr_ref.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A plain fp64 restatement of the reference's place_recognition hot path
// (generate_signatures + match_signatures), function by function, each citing the
// reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; the product (libpr_amd.so) never links or calls it.
//
// PARITY STATUS: "parity unpinned" for signatures / distances.  The reference cannot be built
// here (needs Eigen3 + roscpp; the matcher is MATLAB) and ships no golden vectors; the only
// reference-pinned known-answer test is incoming_id_file.txt == f(poses_history_file.txt)
// (pts_preprocess.h:187-215), which tests/test_oracle_kat.py checks against the committed
// fixtures.  Third-party arithmetic restated here (not vendored in /root/reference):
//   Eigen3 (unpinned, find_package(Eigen3) CMakeLists.txt:7): SelfAdjointEigenSolver -> own cyclic
//     Jacobi 3x3 (eigenvalues ascending), JacobiSVD -> own one-sided Jacobi (Hestenes);
//   MATLAB normalize/min (run_test.m:40,57): z-score with N-1 std, first index on ties;
//   libstdc++ unordered_map iteration order (pts_preprocess.h:85,124): same container used.
// Declared sign conventions: eigenvectors get their largest-|component| positive,
// v2 is then flipped so det=+1; singular pair scaled so that sum(u1) >= 0.
//
// Build: see oracle/Makefile  (g++ -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "../include/pr_m2dp_table.h"

#define PR_REF_OK 0
#define PR_REF_EINVAL (-1)
#define PR_REF_ENAN (-5)

namespace {

// ---------------------------------------------------------------- records (PosesPts.h:5-40)
struct Pose { int id; double w[12]; };            // w2c row-major 3x4
struct Pt { int id; double p[3]; float it; };

// pts_preprocess.h:17-49 — token-wise parse, stop at first failed extraction; a short pose line
// still pushes the pose (:28-34).
void read_poses_pts(const char* poses_file, const char* pts_file, std::vector<Pose>& poses,
                    std::vector<Pt>& pts) {
  std::ifstream pf(poses_file);
  while (true) {
    Pose ps; std::memset(&ps, 0, sizeof ps);
    if (!(pf >> ps.id)) break;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 4; j++) {
        if (!(pf >> ps.w[i * 4 + j])) break;
      }
    }
    poses.push_back(ps);
  }
  std::ifstream qf(pts_file);
  while (true) {
    Pt p;
    if (!(qf >> p.id >> p.p[0] >> p.p[1] >> p.p[2] >> p.it)) break;
    pts.push_back(p);
  }
}

struct Local { int src; double p[3]; };  // index into the history + camera-frame point

// pts_preprocess.h:51-94 (grid filter: keep the smallest camera-y per voxel, strict replace)
void filter_grid(const std::vector<Local>& in, const std::vector<Pt>& hist, double lidar_range,
                 std::vector<double>& oxyz, std::vector<float>& oint) {
  const double resolution[3] = {30, 60, 30};  // RES_GRID, 2*RES_GRID, RES_GRID (:13,:156-157)
  double res_xyz[3], steps[3]; int voxel[3];
  for (int a = 0; a < 3; a++) {
    res_xyz[a] = lidar_range / resolution[a];
    steps[a] = 1.0 / res_xyz[a];
    voxel[a] = static_cast<int>(std::floor(2 * lidar_range * steps[a]) + 1);
  }
  const int loc_step[3] = {1, voxel[0], voxel[0] * voxel[1]};
  std::unordered_map<int, std::pair<int, std::array<double, 3>>> cell;
  for (size_t idx = 0; idx < in.size(); idx++) {
    const double* pt = in[idx].p;
    int xi = static_cast<int>(std::floor((pt[0] + lidar_range) * steps[0]));
    int yi = static_cast<int>(std::floor((pt[1] + lidar_range) * steps[1]));
    int zi = static_cast<int>(std::floor((pt[2] + lidar_range) * steps[2]));
    int loc = xi * loc_step[0] + yi * loc_step[1] + zi * loc_step[2];
    if (cell.find(loc) == cell.end() || -cell[loc].second[1] < -pt[1]) {
      cell[loc] = {static_cast<int>(idx), {pt[0], pt[1], pt[2]}};
    }
  }
  for (auto& kv : cell) {  // libstdc++ iteration order == the reference's output order (:85-89)
    const auto& v = kv.second;
    oxyz.push_back(v.second[0]); oxyz.push_back(v.second[1]); oxyz.push_back(v.second[2]);
    oint.push_back(hist[in[v.first].src].it);
  }
}

inline double norm3(const double* p) { return std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); }

// pts_preprocess.h:96-133 (polar filter: 1 deg x 1 deg cells, keep the closest point, strict)
void filter_polar(const std::vector<Local>& in, const std::vector<Pt>& hist,
                  std::vector<double>& oxyz, std::vector<float>& oint) {
  const double res = 1.0 / 180.0 * M_PI;  // RES_POLAR (:15)
  const double azi_res_inv = 1.0 / res, ele_res_inv = 1.0 / res;
  const int azi_bins = static_cast<int>(std::floor(2 * M_PI * azi_res_inv) + 1);
  std::unordered_map<int, std::pair<int, std::array<double, 3>>> cell;
  for (int idx = 0; idx < (int)in.size(); idx++) {
    const double* pt = in[idx].p;
    double xz = std::sqrt(pt[0] * pt[0] + pt[2] * pt[2]);
    int azi = static_cast<int>(std::floor((std::atan2(pt[2], pt[0]) + M_PI) * azi_res_inv));
    int ele = static_cast<int>(std::floor((std::atan2(pt[1], xz) + M_PI / 2) * ele_res_inv));
    int loc = azi + ele * azi_bins;
    if (cell.find(loc) == cell.end() || norm3(cell[loc].second.data()) > norm3(pt)) {
      cell[loc] = {idx, {pt[0], pt[1], pt[2]}};
    }
  }
  for (auto& kv : cell) {
    const auto& v = kv.second;
    oxyz.push_back(v.second[0]); oxyz.push_back(v.second[1]); oxyz.push_back(v.second[2]);
    oint.push_back(hist[in[v.first].src].it);
  }
}

// ---------------------------------------------------------------- 3x3 symmetric eigen (a3)
// Stand-in for Eigen::SelfAdjointEigenSolver (pts_align.h:31): cyclic Jacobi, ascending order,
// canonical signs (N3).
void eig3_sym(const double c[6] /*xx xy xz yy yz zz*/, double evec[9] /*col-major: v0|v1|v2*/,
              double eval[3]) {
  double a[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    double dia = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off == 0.0 || off <= 1e-36 * dia) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (a[p][q] == 0.0) continue;
        double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < 3; k++) {  // A <- A*J
          double akp = a[k][p], akq = a[k][q];
          a[k][p] = cs * akp - sn * akq; a[k][q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < 3; k++) {  // A <- J^T*A
          double apk = a[p][k], aqk = a[q][k];
          a[p][k] = cs * apk - sn * aqk; a[q][k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = cs * vkp - sn * vkq; v[k][q] = sn * vkp + cs * vkq;
        }
      }
  }
  int order[3] = {0, 1, 2};
  std::sort(order, order + 3, [&](int x, int y) { return a[x][x] < a[y][y]; });
  for (int j = 0; j < 3; j++) {
    eval[j] = a[order[j]][order[j]];
    for (int k = 0; k < 3; k++) evec[j * 3 + k] = v[k][order[j]];
  }
  for (int j = 0; j < 2; j++) {  // canonical sign: largest-|component| positive
    double* e = evec + 3 * j; int im = 0;
    for (int k = 1; k < 3; k++) if (std::fabs(e[k]) > std::fabs(e[im])) im = k;
    if (e[im] < 0) for (int k = 0; k < 3; k++) e[k] = -e[k];
  }
  {  // v2 := sign such that det(v0,v1,v2) = +1
    double* e0 = evec; double* e1 = evec + 3; double* e2 = evec + 6;
    double cx = e0[1] * e1[2] - e0[2] * e1[1], cy = e0[2] * e1[0] - e0[0] * e1[2],
           cz = e0[0] * e1[1] - e0[1] * e1[0];
    if (cx * e2[0] + cy * e2[1] + cz * e2[2] < 0) for (int k = 0; k < 3; k++) e2[k] = -e2[k];
  }
}

// pts_align.h:7-46
void align_pca(const double* xyz, int64_t P, double* out, double* evec_out) {
  double mx = 0, my = 0, mz = 0;
  for (int64_t i = 0; i < P; i++) { mx += xyz[3 * i]; my += xyz[3 * i + 1]; mz += xyz[3 * i + 2]; }
  mx /= P; my /= P; mz /= P;                                                     // :10-18
  std::vector<double> c(3 * (size_t)P);
  double cov[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t i = 0; i < P; i++) {                                             // :21-30
    double x = xyz[3 * i] - mx, y = xyz[3 * i + 1] - my, z = xyz[3 * i + 2] - mz;
    c[3 * i] = x; c[3 * i + 1] = y; c[3 * i + 2] = z;
    cov[0] += x * x; cov[1] += x * y; cov[2] += x * z; cov[3] += y * y; cov[4] += y * z; cov[5] += z * z;
  }
  double evec[9], eval[3];
  eig3_sym(cov, evec, eval);                                                     // :31-34
  for (int64_t i = 0; i < P; i++)                                                // :37-39
    for (int j = 0; j < 3; j++)
      out[3 * i + j] = (c[3 * i] * evec[3 * j] + c[3 * i + 1] * evec[3 * j + 1]) + c[3 * i + 2] * evec[3 * j + 2];
  if (evec_out) std::memcpy(evec_out, evec, sizeof evec);
}

// SC/SC.cpp:60-64, M2DP/M2DP.cpp:77-81 — FLOAT sequential mean over all points (H2)
float ave_intensity_f32(const float* inten, int64_t P) {
  float ave = 0;
  for (int64_t i = 0; i < P; i++) ave += inten[i];
  return ave / P;
}

// SC/SC.cpp:12-76 on already-aligned points (alignment is SC.cpp:17)
void sc_signature_aligned(const double* al, const float* inten, int64_t P, double max_rho, double* out) {
  const int NS = 60, NR = 20, SZ = NS * NR;
  const double S_res_inv = NS / (2.0 * M_PI), R_res_inv = NR / max_rho;          // SC.cpp:5-8
  std::vector<double> cnt(SZ, 0.0), lo(SZ, 0.0), hi(SZ, 0.0), sum(SZ, 0.0);
  for (int64_t i = 0; i < P; i++) {                                              // :29-57
    double yp = al[3 * i + 1], zp = al[3 * i + 2];
    int si = static_cast<int>(std::floor((std::atan2(zp, yp) + M_PI) * S_res_inv));
    int ri = static_cast<int>(std::floor(std::sqrt(yp * yp + zp * zp) * R_res_inv));
    int idx = si * NR + ri;
    if (idx >= SZ) continue;                                                     // :42-44 (aliasing kept, H3)
    if (idx < 0) continue;  // unreachable for finite input (si,ri >= 0); guards the array only
    if (cnt[idx] == 0) { sum[idx] = inten[i]; lo[idx] = al[3 * i]; hi[idx] = al[3 * i]; }
    else { sum[idx] += double(inten[i]); lo[idx] = std::min(lo[idx], al[3 * i]); hi[idx] = std::max(hi[idx], al[3 * i]); }
    cnt[idx]++;
  }
  float ave = ave_intensity_f32(inten, P);                                       // :60-64
  for (int b = 0; b < SZ; b++) {                                                 // :67-75
    double iv = sum[b];
    if (cnt[b]) { iv = iv / cnt[b]; iv = iv > ave ? 1 : 0; }
    out[b] = hi[b] - lo[b];
    out[SZ + b] = iv;
  }
}

// M2DP/M2DP.cpp:4-34 — plane table from the frozen float normals
struct Planes { double x[64][3], y[64][3]; };
const Planes& planes() {
  static Planes pl; static bool init = false;
  if (!init) {
    for (int k = 0; k < 64; k++) {
      float nf[3]; std::memcpy(nf, PR_M2DP_VECN_BITS[k], 12);
      double n[3] = {nf[0], nf[1], nf[2]};
      double d = (1.0 * n[0] + 0.0 * n[1]) + 0.0 * n[2];                         // xAxis . vecN (:21)
      double xa[3] = {1, 0, 0};
      for (int a = 0; a < 3; a++) pl.x[k][a] = xa[a] - d * n[a];                  // :22
      const double* xp = pl.x[k];
      pl.y[k][0] = n[1] * xp[2] - n[2] * xp[1];                                   // :25 cross
      pl.y[k][1] = n[2] * xp[0] - n[0] * xp[2];
      pl.y[k][2] = n[0] * xp[1] - n[1] * xp[0];
    }
    init = true;
  }
  return pl;
}

// M2DP/M2DP.cpp:47-91 — the two 64x128 matrices (row-major [plane][ring*16+sector]) of one variant
void m2dp_matrices(const double* pts, const float* inten, int64_t P, double max_rho,
                   double* count /*64*128*/, double* imat /*64*128*/) {
  const Planes& pl = planes();
  const double S_res_inv = 16 / (2.0 * M_PI), R_res_inv = 8 / max_rho;           // :32-33
  std::fill(count, count + 64 * 128, 0.0);
  std::fill(imat, imat + 64 * 128, 0.0);
  for (int k = 0; k < 64; k++) {
    const double* xP = pl.x[k]; const double* yP = pl.y[k];
    for (int64_t i = 0; i < P; i++) {
      const double* p = pts + 3 * i;
      double xp = xP[0] * p[0] + (xP[1] * p[1] + xP[2] * p[2]);                   // :56 (no +0 seed, N5/H4)
      double yp = yP[0] * p[0] + (yP[1] * p[1] + yP[2] * p[2]);                   // :57
      int si = static_cast<int>(std::floor((std::atan2(yp, xp) + M_PI) * S_res_inv));
      int ri = static_cast<int>(std::floor(std::sqrt(xp * xp + yp * yp) * R_res_inv));
      int idx_sr = ri * 16 + si;
      if (idx_sr >= 128 || idx_sr < 0) continue;                                 // :66-68
      count[k * 128 + idx_sr]++;
      imat[k * 128 + idx_sr] += inten[i];                                        // :71 (f32 -> f64)
    }
  }
  float ave = ave_intensity_f32(inten, P);                                       // :77-81
  for (int e = 0; e < 64 * 128; e++)                                             // :84-91
    if (count[e]) { imat[e] = imat[e] / count[e]; imat[e] = imat[e] > ave ? 1 : 0; }
}

// Stand-in for Eigen::JacobiSVD(...).matrixU().col(0) / matrixV().col(0) (M2DP.cpp:94-103):
// one-sided Jacobi on B = A^T (128x64); A = V_B * S * U_B^T.  Returns [u1(64) | v1(128)].
void top_singular_pair(const double* A /*64x128 row-major*/, double* out /*192*/) {
  const int R = 64, C = 128;
  std::vector<double> W((size_t)C * R), V((size_t)R * R, 0.0);  // W[:,j] = row j of A (column-major, len C)
  for (int j = 0; j < R; j++) { for (int i = 0; i < C; i++) W[(size_t)j * C + i] = A[j * C + i]; V[(size_t)j * R + j] = 1.0; }
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < R - 1; p++)
      for (int q = p + 1; q < R; q++) {
        double* wp = &W[(size_t)p * C]; double* wq = &W[(size_t)q * C];
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < C; i++) { al += wp[i] * wp[i]; be += wq[i] * wq[i]; ga += wp[i] * wq[i]; }
        if (ga == 0.0 || std::fabs(ga) <= 1e-15 * std::sqrt(al * be)) continue;
        rotated = true;
        double zeta = (be - al) / (2.0 * ga);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < C; i++) { double a = wp[i], b = wq[i]; wp[i] = cs * a - sn * b; wq[i] = sn * a + cs * b; }
        double* vp = &V[(size_t)p * R]; double* vq = &V[(size_t)q * R];
        for (int i = 0; i < R; i++) { double a = vp[i], b = vq[i]; vp[i] = cs * a - sn * b; vq[i] = sn * a + cs * b; }
      }
    if (!rotated) break;
  }
  int jm = 0; double sm = -1;
  for (int j = 0; j < R; j++) {
    double s = 0; for (int i = 0; i < C; i++) s += W[(size_t)j * C + i] * W[(size_t)j * C + i];
    if (s > sm) { sm = s; jm = j; }
  }
  double sigma = std::sqrt(sm);
  if (!(sigma > 0)) {  // zero matrix: JacobiSVD returns identity factors -> first basis vectors
    std::fill(out, out + 192, 0.0); out[0] = 1.0; out[64] = 1.0; return;
  }
  double su = 0;
  for (int i = 0; i < R; i++) { out[i] = V[(size_t)jm * R + i]; su += out[i]; }
  for (int i = 0; i < C; i++) out[64 + i] = W[(size_t)jm * C + i] / sigma;
  if (su < 0) for (int i = 0; i < 192; i++) out[i] = -out[i];                    // N6: sum(u1) >= 0
}

// processSC.m:12-34 for one channel.  h1: m x 1200, h2: n x 1200 (bin = sector*20 + ring).
int sc_process(const double* h1, int m, const double* h2, int n, double* res) {
  const int L = 1200;
  std::vector<double> a((size_t)m * L), b((size_t)n * L);
  int bad = 0;
  for (int i = 0; i < m; i++) {                                                  // :15-17
    double s = 0; for (int c = 0; c < L; c++) s += h1[(size_t)i * L + c] * h1[(size_t)i * L + c];
    double nr = std::sqrt(s); if (!(nr > 0)) bad = 1;
    for (int c = 0; c < L; c++) a[(size_t)i * L + c] = h1[(size_t)i * L + c] / nr;
  }
  for (int j = 0; j < n; j++) {                                                  // :18-20
    double s = 0; for (int c = 0; c < L; c++) s += h2[(size_t)j * L + c] * h2[(size_t)j * L + c];
    double nr = std::sqrt(s); if (!(nr > 0)) bad = 1;
    for (int c = 0; c < L; c++) b[(size_t)j * L + c] = h2[(size_t)j * L + c] / nr;
  }
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < m; i++) {                                                  // :22-33
    // sigT[c][v] = variant v of query i at column c.  The 120 dot products advance together over c, so the
    // loop vectorises ACROSS variants while every individual sum keeps the sequential c = 0..1199 order.
    std::vector<double> sigT((size_t)L * 120);
    const double* row = &a[(size_t)i * L];
    for (int k0 = 0; k0 < 60; k0++)                                              // permute_sc :37-45 (0-based)
      for (int c = 0; c < 60; c++) {
        int sf = (k0 + c) % 60, sm = ((k0 - c) % 60 + 60) % 60;
        for (int r = 0; r < 20; r++) {
          sigT[(size_t)(c * 20 + r) * 120 + 2 * k0] = row[sf * 20 + r];
          sigT[(size_t)(c * 20 + r) * 120 + 2 * k0 + 1] = row[sm * 20 + r];
        }
      }
    for (int j = 0; j < n; j++) {
      const double* d = &b[(size_t)j * L];
      double dot[120];
      for (int v = 0; v < 120; v++) dot[v] = 0.0;
      for (int c = 0; c < L; c++) {
        const double dc = d[c];
        const double* s = &sigT[(size_t)c * 120];
        for (int v = 0; v < 120; v++) dot[v] += s[v] * dc;
      }
      double best = std::numeric_limits<double>::quiet_NaN();                    // MATLAB min skips NaN
      for (int v = 0; v < 120; v++) {
        double diff = (1 - dot[v]) / 2;                                          // :30
        if (std::isnan(best) || diff < best) best = diff;                        // :31 (min)
      }
      res[(size_t)i * n + j] = best;
    }
  }
  return bad ? PR_REF_ENAN : PR_REF_OK;
}

// processM2DP.m:12-22 for one channel.  h1: 4m x 192, h2: 4n x 192 (rows NOT re-normalised).
void m2dp_process(const double* h1, int m, const double* h2, int n, double* res) {
  const int L = 192;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double best = std::numeric_limits<double>::infinity();
      for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) {
          const double* x = h1 + (size_t)(4 * i + a) * L; const double* y = h2 + (size_t)(4 * j + b) * L;
          double dot = 0; for (int c = 0; c < L; c++) dot += x[c] * y[c];
          double diff = (1 - dot) / 2;                                           // :15
          if (diff < best) best = diff;                                          // :19
        }
      res[(size_t)i * n + j] = best;
    }
}

// run_test.m:38-41 + :47-53 + :57 generalised to top-k (k=1 == reference).  MATLAB normalize(.,2):
// (x-mean)/std with N-1.
void fuse_topk(const double* dp, const double* di, int m, int n, int mask_width, double p_weight, int k,
               int32_t* idx, double* score) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < m; i++) {
    const double* a = dp + (size_t)i * n; const double* b = di + (size_t)i * n;
    // MATLAB normalize(.,2) = (x - mean)/std with N-1, both computed with 'omitnan' [from memory: normalize.m]: NaN
    // distances (a zero-norm signature, processSC.m:16,19) stay NaN themselves but do not poison the row
    double ma = 0, mb = 0; int na = 0, nb = 0;
    for (int j = 0; j < n; j++) { if (!std::isnan(a[j])) { ma += a[j]; na++; } if (!std::isnan(b[j])) { mb += b[j]; nb++; } }
    ma /= na; mb /= nb;
    double va = 0, vb = 0;
    for (int j = 0; j < n; j++) {
      if (!std::isnan(a[j])) va += (a[j] - ma) * (a[j] - ma);
      if (!std::isnan(b[j])) vb += (b[j] - mb) * (b[j] - mb);
    }
    double sa = std::sqrt(va / (na - 1)), sb = std::sqrt(vb / (nb - 1));
    std::vector<double> f(n);
    for (int j = 0; j < n; j++) {
      f[j] = p_weight * ((a[j] - ma) / sa) + (b[j] - mb) / sb;                    // :40
      if (std::abs(i - j) < mask_width) f[j] = std::numeric_limits<double>::infinity();  // :47-53
    }
    std::vector<char> used(n, 0);
    for (int t = 0; t < k; t++) {                                                // :57, first index on ties
      int bj = -1; double bv = 0;
      for (int j = 0; j < n; j++) {
        if (used[j] || std::isnan(f[j])) continue;
        if (bj < 0 || f[j] < bv) { bj = j; bv = f[j]; }
      }
      if (bj < 0) { idx[(size_t)i * k + t] = -1; score[(size_t)i * k + t] = std::numeric_limits<double>::quiet_NaN(); }
      else { used[bj] = 1; idx[(size_t)i * k + t] = bj; score[(size_t)i * k + t] = bv; }
    }
  }
}

// DELIGHT/DELIGHT.cpp:8-24 on one cloud (alignment inside, :12-13): 16 x 256 intensity histograms
void delight_signature(const double* xyz, const float* inten, int64_t P, double* out /*16*256*/) {
  std::fill(out, out + 16 * 256, 0.0);
  if (P <= 0) return;
  std::vector<double> al(3 * (size_t)P);
  align_pca(xyz, P, al.data(), nullptr);
  for (int64_t i = 0; i < P; i++) {
    const double* p = &al[3 * i];
    float x = p[0], y = p[1], z = p[2];                                           // :17-19 (double -> float)
    float d = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);                // :20 norm() then float
    float clr = inten[i];
    int hist = 8 * (d > 10.0) + 4 * (z > 0) + 2 * (y > 0) + 1 * (x > 0);          // :23, RADIUS 10.0
    int bin = int(clr);                                                           // :24 truncation
    if (bin < 0 || bin >= 256) continue;   // out of range is undefined behaviour in the reference; dropped here
    out[hist * 256 + bin]++;
  }
}

// processDELIGHT.m:1-38: chi-square over non-empty bins, min over the 4 octant permutations
const int DELIGHT_MUT[4][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
                                {5, 4, 7, 6, 1, 0, 3, 2, 13, 12, 15, 14, 9, 8, 11, 10},
                                {6, 7, 4, 5, 2, 3, 0, 1, 14, 15, 12, 13, 10, 11, 8, 9},
                                {3, 2, 1, 0, 7, 6, 5, 4, 11, 10, 9, 8, 15, 14, 13, 12}};
void delight_process(const double* h1, int m, const double* h2, int n, double* res) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      const double* A = h1 + (size_t)i * 4096; const double* B = h2 + (size_t)j * 4096;
      double best = std::numeric_limits<double>::infinity();
      for (int k = 0; k < 4; k++) {
        double ts = 0; double tc = 0;
        for (int c = 0; c < 256; c++)          // Ak = A(:) is column-major: the order of the terms of the sum
          for (int r = 0; r < 16; r++) {
            double a = A[r * 256 + c], b = B[DELIGHT_MUT[k][r] * 256 + c], sum = a + b;
            if (sum > 0) { ts += 2 * (a - b) * (a - b) / sum; tc += 1; }
          }
        ts = ts / tc;
        if (best > ts) best = ts;                                                 // :31-33 (NaN never wins)
      }
      res[(size_t)i * n + j] = best;
    }
}

}  // namespace

// =================================================================== C ABI of the oracle
extern "C" {

struct pr_ref_clouds {
  std::vector<int64_t> offs; std::vector<double> xyz; std::vector<float> inten; std::vector<int> ids;
};

// pts_preprocess.h:169-232.  Writes incoming_id_file (one id per line, std::endl) when non-NULL.
int pr_ref_pts_preprocess(const char* poses_file, const char* pts_file, const char* incoming_id_file,
                          double lidarRange, int polar_filter, pr_ref_clouds** out) {
  if (!poses_file || !pts_file || !out) return PR_REF_EINVAL;
  std::vector<Pose> poses; std::vector<Pt> hist;
  read_poses_pts(poses_file, pts_file, poses, hist);
  std::ofstream idf; if (incoming_id_file) idf.open(incoming_id_file);
  auto* res = new pr_ref_clouds; res->offs.push_back(0);
  std::vector<int> nearby;  // indices into hist
  size_t pts_idx = 0; int frame_from_reset = 0;
  for (const Pose& ps : poses) {
    const double* w = ps.w;
    if (std::sqrt(w[3] * w[3] + w[7] * w[7] + w[11] * w[11]) < 1.0) {             // :189-193
      frame_from_reset = 0; nearby.clear();
    }
    while (pts_idx < hist.size() && hist[pts_idx].id <= ps.id) { nearby.push_back((int)pts_idx); pts_idx++; }  // :196-200
    if (frame_from_reset < 30) { frame_from_reset++; continue; }                 // :203-206 INIT_FRAME
    std::vector<Local> raw; std::vector<int> keep;                               // :135-167
    for (int h : nearby) {
      const double* p = hist[h].p;
      Local l; l.src = h;
      for (int r = 0; r < 3; r++)
        l.p[r] = ((w[4 * r] * p[0] + w[4 * r + 1] * p[1]) + w[4 * r + 2] * p[2]) + w[4 * r + 3] * 1.0;  // :141-142
      if (norm3(l.p) < lidarRange) { raw.push_back(l); keep.push_back(h); }      // :144-148
    }
    if (polar_filter) filter_polar(raw, hist, res->xyz, res->inten);
    else filter_grid(raw, hist, lidarRange, res->xyz, res->inten);
    nearby.swap(keep);                                                           // :166
    res->offs.push_back((int64_t)res->inten.size());
    res->ids.push_back(ps.id);
    if (incoming_id_file) idf << ps.id << std::endl;                             // :215
  }
  *out = res;
  return PR_REF_OK;
}
int64_t pr_ref_clouds_count(const pr_ref_clouds* c) { return (int64_t)c->offs.size() - 1; }
const int64_t* pr_ref_clouds_offs(const pr_ref_clouds* c) { return c->offs.data(); }
const double* pr_ref_clouds_xyz(const pr_ref_clouds* c) { return c->xyz.data(); }
const float* pr_ref_clouds_inten(const pr_ref_clouds* c) { return c->inten.data(); }
const int* pr_ref_clouds_ids(const pr_ref_clouds* c) { return c->ids.data(); }
void pr_ref_clouds_free(pr_ref_clouds* c) { delete c; }

void pr_ref_align_pca(const double* xyz, int64_t P, double* out_xyz, double* evec9) { align_pca(xyz, P, out_xyz, evec9); }

float pr_ref_ave_intensity(const float* inten, int64_t P) { return ave_intensity_f32(inten, P); }

void pr_ref_m2dp_plane_table(double* xproj /*64*3*/, double* yproj /*64*3*/) {
  const Planes& pl = planes();
  std::memcpy(xproj, pl.x, sizeof pl.x); std::memcpy(yproj, pl.y, sizeof pl.y);
}

// SC/test_sc.cpp:40-56: row i = [structure(1200) | intensity(1200)]
int pr_ref_sc_generate(const double* xyz, const float* inten, const int64_t* offs, int32_t N, double max_rho, double* out) {
  if (!offs || !out || N < 0) return PR_REF_EINVAL;
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < N; c++) {
    int64_t P = offs[c + 1] - offs[c];
    std::vector<double> al(3 * (size_t)std::max<int64_t>(P, 1));
    if (P > 0) align_pca(xyz + 3 * offs[c], P, al.data(), nullptr);
    sc_signature_aligned(al.data(), inten + offs[c], P, max_rho, out + (size_t)c * 2400);
  }
  return PR_REF_OK;
}

// Exposes the intermediate matrices of one (dx,dy) variant of one ALREADY-ALIGNED cloud (tests).
void pr_ref_m2dp_matrices(const double* aligned, const float* inten, int64_t P, double max_rho, int dx, int dy,
                          double* count, double* imat) {
  std::vector<double> v(3 * (size_t)P);
  for (int64_t i = 0; i < P; i++) {                                              // test_m2dp.cpp:52-56
    v[3 * i] = dx * aligned[3 * i]; v[3 * i + 1] = dy * aligned[3 * i + 1]; v[3 * i + 2] = (dx * dy) * aligned[3 * i + 2];
  }
  m2dp_matrices(v.data(), inten, P, max_rho, count, imat);
}

void pr_ref_top_singular_pair(const double* A, double* out192) { top_singular_pair(A, out192); }

// M2DP/test_m2dp.cpp:41-68: rows 4c..4c+3 = variants (-,-),(-,+),(+,-),(+,+); row = [cnt U1|V1 | int U1|V1]
int pr_ref_m2dp_generate(const double* xyz, const float* inten, const int64_t* offs, int32_t N, double max_rho, double* out) {
  if (!offs || !out || N < 0) return PR_REF_EINVAL;
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < N; c++) {
    int64_t P = offs[c + 1] - offs[c];
    std::vector<double> al(3 * (size_t)std::max<int64_t>(P, 1)), cm(64 * 128), im(64 * 128);
    if (P > 0) align_pca(xyz + 3 * offs[c], P, al.data(), nullptr);
    int sub = 0;
    for (int dx = -1; dx < 2; dx += 2)
      for (int dy = -1; dy < 2; dy += 2) {
        pr_ref_m2dp_matrices(al.data(), inten + offs[c], P, max_rho, dx, dy, cm.data(), im.data());
        double* row = out + ((size_t)4 * c + sub) * 384;
        top_singular_pair(cm.data(), row);
        top_singular_pair(im.data(), row + 192);
        sub++;
      }
  }
  return PR_REF_OK;
}

// processSC.m:1-10.  h1: m x 2400, h2: n x 2400 -> two m x n (fp64).  PR_REF_ENAN if a row has zero norm
// (MATLAB would produce NaN rows); the distances are still written (NaN where MATLAB has NaN).
int pr_ref_sc_distance(const double* h1, int32_t m, const double* h2, int32_t n, double* d_struct, double* d_int) {
  if (!h1 || !h2 || m < 0 || n < 0) return PR_REF_EINVAL;
  std::vector<double> a((size_t)m * 1200), b((size_t)n * 1200);
  int rc = PR_REF_OK;
  for (int ch = 0; ch < 2; ch++) {
    for (int i = 0; i < m; i++) std::memcpy(&a[(size_t)i * 1200], h1 + (size_t)i * 2400 + ch * 1200, 1200 * 8);
    for (int j = 0; j < n; j++) std::memcpy(&b[(size_t)j * 1200], h2 + (size_t)j * 2400 + ch * 1200, 1200 * 8);
    double* dst = ch ? d_int : d_struct;
    if (dst) { int r = sc_process(a.data(), m, b.data(), n, dst); if (r) rc = r; }
  }
  return rc;
}

// processM2DP.m:1-10.  h1: 4m x 384, h2: 4n x 384.
int pr_ref_m2dp_distance(const double* h1, int32_t m, const double* h2, int32_t n, double* d_cnt, double* d_int) {
  if (!h1 || !h2 || m < 0 || n < 0) return PR_REF_EINVAL;
  std::vector<double> a((size_t)4 * m * 192), b((size_t)4 * n * 192);
  for (int ch = 0; ch < 2; ch++) {
    for (int i = 0; i < 4 * m; i++) std::memcpy(&a[(size_t)i * 192], h1 + (size_t)i * 384 + ch * 192, 192 * 8);
    for (int j = 0; j < 4 * n; j++) std::memcpy(&b[(size_t)j * 192], h2 + (size_t)j * 384 + ch * 192, 192 * 8);
    double* dst = ch ? d_int : d_cnt;
    if (dst) m2dp_process(a.data(), m, b.data(), n, dst);
  }
  return PR_REF_OK;
}

// DELIGHT/test_delight.cpp:41-56: rows 16c..16c+15 = the 16 histograms of cloud c
int pr_ref_delight_generate(const double* xyz, const float* inten, const int64_t* offs, int32_t N, double* out) {
  if (!offs || !out || N < 0) return PR_REF_EINVAL;
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < N; c++)
    delight_signature(xyz + 3 * offs[c], inten + offs[c], offs[c + 1] - offs[c], out + (size_t)c * 4096);
  return PR_REF_OK;
}

int pr_ref_delight_distance(const double* h1, int32_t m, const double* h2, int32_t n, double* dist) {
  if (!h1 || !h2 || !dist || m < 0 || n < 0) return PR_REF_EINVAL;
  delight_process(h1, m, h2, n, dist);
  return PR_REF_OK;
}

// processGIST.m:1-10 - squared Euclidean distance of every row pair
int pr_ref_gist_distance(const double* h1, int32_t m, const double* h2, int32_t n, int32_t cols, double* dist) {
  if (!h1 || !h2 || !dist || m < 0 || n < 0 || cols < 0) return PR_REF_EINVAL;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double s = 0.0;
      for (int c = 0; c < cols; c++) { const double d = h1[(size_t)i * cols + c] - h2[(size_t)j * cols + c]; s += d * d; }   // :7
      dist[(size_t)i * n + j] = s;
    }
  return PR_REF_OK;
}

// processBoW.m:1-38 - rows alternate (word ids | word weights), both padded with -1 (test_bow.cpp:147-162); DBoW2's L1 score
// by a merge of the two sorted id lists.  The loop guards are `ii < length(list)` with 1-based ii (processBoW.m:23): the
// LAST column is never looked at - kept.
int pr_ref_bow_distance(const double* h1, int32_t m, const double* h2, int32_t n, int32_t cols, double* dist) {
  if (!h1 || !h2 || !dist || m < 0 || n < 0 || cols < 0) return PR_REF_EINVAL;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      const double *i1 = h1 + (size_t)(2 * i) * cols, *v1 = i1 + cols, *i2 = h2 + (size_t)(2 * j) * cols, *v2 = i2 + cols;
      int a = 1, b = 1;                                  // 1-based like the reference
      double score = 0.0;
      while (a < cols && i1[a - 1] > -1 && b < cols && i2[b - 1] > -1) {                                       // :23
        if (i1[a - 1] == i2[b - 1]) { score = score + std::fabs(v1[a - 1] - v2[b - 1]) - std::fabs(v1[a - 1]) - std::fabs(v2[b - 1]); a++; b++; }   // :25
        else if (i1[a - 1] < i2[b - 1]) a++;
        else b++;
      }
      dist[(size_t)i * n + j] = 1.0 - (-score / 2.0);                                                          // :37, :14
    }
  return PR_REF_OK;
}

// run_test.m:47-57 without the z-score fusion (types other than m2dp / sc, run_test.m:26-41)
// iteration order of the real std::unordered_map after inserting the K keys in this order (what the reference's
// filterPoints / filterPointsPolar emit: pts_preprocess.h:85-89, :124-128); order[t] = index of the t-th element
int pr_ref_unordered_order(const int32_t* keys, int32_t K, int32_t* order) {
  std::unordered_map<int, int> m;
  for (int k = 0; k < K; k++) m[keys[k]] = k;
  int t = 0;
  for (const auto& kv : m) order[t++] = kv.second;
  return t == K ? 0 : -1;
}

int pr_ref_select_topk(const double* d, int32_t m, int32_t n, int32_t mask_width, int32_t k, int32_t* idx, double* score) {
  if (!d || m < 0 || n < 1 || k < 1) return PR_REF_EINVAL;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < m; i++) {
    std::vector<double> f(d + (size_t)i * n, d + (size_t)(i + 1) * n);
    for (int j = 0; j < n; j++) if (std::abs(i - j) < mask_width) f[j] = std::numeric_limits<double>::infinity();
    std::vector<char> used(n, 0);
    for (int t = 0; t < k; t++) {
      int bj = -1; double bv = 0;
      for (int j = 0; j < n; j++) {
        if (used[j] || std::isnan(f[j])) continue;
        if (bj < 0 || f[j] < bv) { bj = j; bv = f[j]; }
      }
      if (bj < 0) { idx[(size_t)i * k + t] = -1; score[(size_t)i * k + t] = std::numeric_limits<double>::quiet_NaN(); }
      else { used[bj] = 1; idx[(size_t)i * k + t] = bj; score[(size_t)i * k + t] = bv; }
    }
  }
  return PR_REF_OK;
}

int pr_ref_fuse_topk(const double* d_p, const double* d_i, int32_t m, int32_t n, int32_t mask_width, double p_weight,
                     int32_t k, int32_t* idx, double* score) {
  if (!d_p || !d_i || m < 0 || n < 2 || k < 1) return PR_REF_EINVAL;
  fuse_topk(d_p, d_i, m, n, mask_width, p_weight, k, idx, score);
  return PR_REF_OK;
}

// run_test.m:26-57 end to end.  type 0 = SC (m x 2400 / n x 2400), 1 = M2DP (4m x 384 / 4n x 384).
int pr_ref_match_topk(int type, const double* h1, int32_t m, const double* h2, int32_t n, int32_t mask_width,
                      double p_weight, int32_t k, int32_t* idx, double* score) {
  if (type == 2) {   // DELIGHT: h1 [16m][256], h2 [16n][256]
    if (n < 1 || m < 0 || k < 1) return PR_REF_EINVAL;
    std::vector<double> d((size_t)m * n);
    delight_process(h1, m, h2, n, d.data());
    return pr_ref_select_topk(d.data(), m, n, mask_width, k, idx, score);
  }
  if (n < 2 || m < 0 || k < 1) return PR_REF_EINVAL;
  std::vector<double> dp((size_t)m * n), di((size_t)m * n);
  int rc = type == 0 ? pr_ref_sc_distance(h1, m, h2, n, dp.data(), di.data())
                     : pr_ref_m2dp_distance(h1, m, h2, n, dp.data(), di.data());
  if (rc && rc != PR_REF_ENAN) return rc;              // ENAN: the NaN rows / columns are in dp / di, as in MATLAB
  const int rc2 = pr_ref_fuse_topk(dp.data(), di.data(), m, n, mask_width, p_weight, k, idx, score);
  return rc2 ? rc2 : rc;
}

// BASELINE.json config 5 ("fused SC + M2DP scoring"), which has NO reference counterpart (run_test.m handles one type per
// run): the build-defined score of SURVEY.md §6 - fused = [p z(sc_struct) + z(sc_int)] + [p z(m2dp_count) + z(m2dp_int)],
// every z a MATLAB normalize(.,2) row z-score (N-1) as in run_test.m:40 - then mask and row minimum as run_test.m:47-57.
int pr_ref_match_topk_fused(const double* sc1, const double* m2dp1, int32_t m, const double* sc2, const double* m2dp2, int32_t n,
                            int32_t mask_width, double p_weight, int32_t k, int32_t* idx, double* score) {
  if (n < 2 || m < 0 || k < 1) return PR_REF_EINVAL;
  const size_t mn = (size_t)m * n;
  std::vector<double> d[4];
  for (auto& v : d) v.resize(mn);
  int rc = pr_ref_sc_distance(sc1, m, sc2, n, d[0].data(), d[1].data());
  if (!rc || rc == PR_REF_ENAN) { const int r2 = pr_ref_m2dp_distance(m2dp1, m, m2dp2, n, d[2].data(), d[3].data()); if (r2) rc = r2; }
  if (rc && rc != PR_REF_ENAN) return rc;
  std::vector<double> f(mn, 0.0);
  for (int c = 0; c < 4; c++)
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; i++) {
      const double* a = d[c].data() + (size_t)i * n;
      double mu = 0, va = 0; int cnt = 0;
      for (int j = 0; j < n; j++) if (!std::isnan(a[j])) { mu += a[j]; cnt++; }
      mu /= cnt;
      for (int j = 0; j < n; j++) if (!std::isnan(a[j])) va += (a[j] - mu) * (a[j] - mu);
      const double sd = std::sqrt(va / (cnt - 1)), wt = (c % 2 == 0) ? p_weight : 1.0;
      for (int j = 0; j < n; j++) f[(size_t)i * n + j] += wt * ((a[j] - mu) / sd);
    }
  return pr_ref_select_topk(f.data(), m, n, mask_width, k, idx, score);
}

// run_test.m:3-22 (ground-truth loop pairs) and :58-85 (precision / recall sweep, top recall at 100 % precision, trapz AUC, lp_detected),
// line for line.  Input = what run_test.m:57 leaves: diff_v [m] (NaN / +Inf allowed) and diff_idx [m], 0-based; an index of -1 (the build's
// "no finite candidate") is read as 0: MATLAB's min over an all-NaN / all-Inf row returns index 1.  gt1 [m][cols], gt2 [n][cols].
// Outputs (any may be NULL): auc, top_recall, lp_gt [m][2] + n_gt (0-based pairs of :19), lp_detected [m][2] + n_detected (:85),
// precision [m], recall [m].  Divisions are MATLAB's (IEEE): total_lp = 0 gives recall NaN (0/0) or Inf.
int pr_ref_precision_recall(const double* diff_v, const int32_t* diff_idx, int32_t m, const double* gt1, const double* gt2, int32_t n,
                            int32_t cols, double loop_diff, int32_t mask_width, double* auc, double* top_recall, int32_t* lp_gt,
                            int32_t* n_gt, int32_t* lp_detected, int32_t* n_detected, double* precision_out, double* recall_out) {
  if (m < 0 || n < 0 || cols < 1 || (m > 0 && (!diff_v || !diff_idx || !gt1)) || (n > 0 && !gt2)) return PR_REF_EINVAL;
  const double inf = std::numeric_limits<double>::infinity();
  auto sq = [&](int a, int b) {                                   // diff = gt1(a,:) - gt2(b,:); diff = diff*diff'  (:10-11, :70-71)
    double s = 0.0;
    for (int c = 0; c < cols; c++) { const double t = gt1[(size_t)a * cols + c] - gt2[(size_t)b * cols + c]; s += t * t; }
    return s;
  };
  // :3-22
  std::vector<std::array<int, 2>> lp;
  for (int i = 0; i < m; i++) {                                   // :4  for i=1:size(gt1,1)
    double min_diff = inf;                                        // :5
    int min_j = -1;                                               // :6
    for (int j = 0; j < n; j++) {                                 // :7
      if (std::abs(i - j) < mask_width) continue;                 // :8-10
      const double diff = sq(i, j);                               // :11-12
      if (min_diff > diff) { min_diff = diff; min_j = j; }        // :13-16
    }
    if (min_diff < loop_diff * loop_diff) lp.push_back({i, min_j});   // :18-20
  }
  const size_t L = lp.size();
  const double total_lp = L == 0 ? 0.0 : (double)std::max<size_t>(L, 2);   // :22 length() of an L x 2 matrix = max(L, 2); of [] = 0
  if (n_gt) *n_gt = (int32_t)L;
  if (lp_gt) for (size_t i = 0; i < L; i++) { lp_gt[2 * i] = lp[i][0]; lp_gt[2 * i + 1] = lp[i][1]; }
  // :58 [~, diff_rank] = sort(diff_v): ascending, stable, NaN after +Inf
  std::vector<int> diff_rank((size_t)m);
  for (int i = 0; i < m; i++) diff_rank[i] = i;
  std::stable_sort(diff_rank.begin(), diff_rank.end(), [&](int a, int b) {
    const double x = diff_v[a], y = diff_v[b];
    if (std::isnan(x) || std::isnan(y)) return !std::isnan(x) && std::isnan(y);
    return x < y;
  });
  // :60-83
  double tp = 0, fp = 0;                                          // :60-61 (MATLAB doubles)
  std::vector<double> precision((size_t)m, 0.0), recall((size_t)m, 0.0);   // :62-63
  double tr = 0.0;                                                // :64
  int top_count = 0;                                              // :65
  for (int i = 0; i < m; i++) {                                   // :66
    const int a = diff_rank[i];                                   // :67
    const int b = diff_idx[a] < 0 ? 0 : diff_idx[a];              // :68 (min's index 1 for an all-NaN / all-Inf row)
    const double diff = sq(a, b);                                 // :69-70
    if (diff < loop_diff * loop_diff) tp = tp + 1; else fp = fp + 1;   // :71-75
    precision[i] = tp / (tp + fp);                                // :76
    recall[i] = tp / total_lp;                                    // :77
    if (precision[i] == 1) { top_count = i + 1; tr = recall[i]; } // :79-82
  }
  double area = 0.0;                                              // :84 trapz(recall, precision)
  for (int i = 0; i + 1 < m; i++) area += (recall[i + 1] - recall[i]) * (precision[i] + precision[i + 1]) / 2.0;
  if (auc) *auc = area;
  if (top_recall) *top_recall = tr;
  if (n_detected) *n_detected = top_count;                        // :85 lp_detected = [diff_rank(1:top_count)', diff_idx(diff_rank(1:top_count))']
  if (lp_detected)
    for (int i = 0; i < top_count; i++) { lp_detected[2 * i] = diff_rank[i]; lp_detected[2 * i + 1] = diff_idx[diff_rank[i]] < 0 ? 0 : diff_idx[diff_rank[i]]; }
  if (precision_out) std::copy(precision.begin(), precision.end(), precision_out);
  if (recall_out) std::copy(recall.begin(), recall.end(), recall_out);
  return PR_REF_OK;
}

}  // extern "C"
