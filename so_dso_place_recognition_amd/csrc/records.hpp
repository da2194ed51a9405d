// records.hpp — PosesPts text records (PosesPts.h:5-40, reader utils/pts_preprocess.h:17-49) and the cloud container shared by
// the host pre-stage (host_io.cpp) and the GPU pre-stage (pr_api.cpp / prestage.hip).  Internal to libpr_amd.so.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

struct pr_clouds {
  std::vector<int64_t> offs;
  std::vector<double> xyz;
  std::vector<float> inten;
  std::vector<int32_t> ids;
  double avg_ms = 0, avg_pts = 0;
  // pr_pts_preprocess_gpu leaves the emitted clouds and their PCA frames in HBM too (NULL otherwise), for pr_*_generate_frames_dev
  void* d_xyz = nullptr;
  void* d_inten = nullptr;
  void* d_offs = nullptr;
  void* d_frames = nullptr;
  int device = -1;
  void (*release)(pr_clouds*) = nullptr;    // frees the device copies (set by whoever made them)
};

namespace pr_rec {

inline bool slurp(const char* path, std::string& buf) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf.resize(n > 0 ? (size_t)n : 0);
  size_t got = n > 0 ? fread(&buf[0], 1, (size_t)n, f) : 0;
  fclose(f);
  buf.resize(got);
  return true;
}

// Token grammar of `istream >> int/double/float` in libstdc++ (bits/locale_facets.tcc: _M_extract_int / _M_extract_float,
// "C" locale): the stream collects [sign] digits [. digits] [e [sign] digits] itself and only then converts, so text that
// strtod would take but the grammar does not ("inf", "nan", "0x10", "1e", a value that overflows) FAILS the extraction in
// the reference - and a failed extraction ends the file (pts_preprocess.h:36).  Returns the token length (0 = none).
inline size_t float_token(const char* q) {
  const char* p = q;
  if (*p == '+' || *p == '-') p++;
  bool mant = false, dec = false, sci = false;
  for (;; p++) {
    const char c = *p;
    if (c >= '0' && c <= '9') mant = true;
    else if (c == '.' && !dec && !sci) dec = true;
    else if ((c == 'e' || c == 'E') && !sci && mant) {
      sci = true;
      if (p[1] == '+' || p[1] == '-') p++;
    } else break;
  }
  return (size_t)(p - q);
}

inline bool is_space(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

struct Cursor {
  const char* p;
  bool next_int(int& v) {
    char* e;
    const long x = strtol(p, &e, 10);
    if (e == p || x > 2147483647L || x < -2147483647L - 1) return false;
    v = (int)x; p = e;
    return true;
  }
  template <typename T, typename F>
  bool next_fp(T& v, F conv) {
    while (is_space(*p)) p++;
    const size_t n = float_token(p);
    if (!n) return false;
    char* e;
    v = conv(p, &e);
    if (e != p + n) {   // strtod stopped short ("1e", "-") or ran past the stream's token ("0x10"): convert the token alone
      const std::string t(p, n);
      v = conv(t.c_str(), &e);
      if (e != t.c_str() + n) return false;
    }
    if (v > std::numeric_limits<T>::max() || v < -std::numeric_limits<T>::max()) return false;   // overflow sets failbit
    p += n;
    return true;
  }
  bool next_double(double& v) { return next_fp(v, strtod); }
  bool next_float(float& v) { return next_fp(v, strtof); }
};

// Host threads for the text formats: PR_PARSE_THREADS overrides (1 = everything on the calling thread), at most 32, and no
// more than `units` (callers pass the amount of work in units that are worth a thread each).
inline unsigned host_threads(size_t units) {
  unsigned T = std::thread::hardware_concurrency();
  if (const char* e = getenv("PR_PARSE_THREADS")) T = (unsigned)atoi(e);
  if (T > 32) T = 32;
  if (T > units) T = (unsigned)units;
  return T < 1 ? 1 : T;
}

template <typename F>
inline void run_threads(unsigned T, F&& body) {
  std::vector<std::thread> th;
  for (unsigned t = 1; t < T; t++) th.emplace_back([&body, t] { body(t); });
  body(0);
  for (auto& x : th) x.join();
}

struct PoseRec { int id; double w[12]; };
struct History { std::vector<int> id; std::vector<double> xyz; std::vector<float> it; };

// Points file in parallel ("fast parser for the legacy text"): once the window / filter work runs on the
// GPU the single-threaded strtod loop is what is left of the pre-stage (1.6 s of 1.8 s at 9 M points).  The file is cut at line
// ends into one chunk per host thread; a chunk is accepted only if every non-blank line holds exactly the 5 tokens of one
// record (`id x y z intensity`, PosesPts.h:37) - then token-stream extraction (`ifstream >>`, pts_preprocess.h:36-47) and
// line-wise parsing are the same thing.  Anything else (short line, junk, a failed extraction) returns false and the caller
// parses the whole file sequentially with the reference's stop-at-first-failure semantics.
inline bool parse_points_parallel(const std::string& buf, History& h) {
  const size_t n = buf.size();
  const unsigned T = host_threads(n);
  size_t min_bytes = (size_t)4 << 20;   // below this the thread start-up costs more than the parse
  if (const char* e = getenv("PR_PARSE_MIN_BYTES")) min_bytes = (size_t)atoll(e);
  if (T < 2 || n < min_bytes || n < T) return false;
  std::vector<size_t> cut(T + 1, n);
  cut[0] = 0;
  for (unsigned t = 1; t < T; t++) {
    size_t p = n / T * t;
    while (p < n && buf[p] != '\n') p++;
    cut[t] = p < n ? p + 1 : n;
  }
  std::vector<History> part(T);
  std::vector<char> bad(T, 0);
  auto work = [&](unsigned t) {
    const char* p = buf.c_str() + cut[t];
    const char* end = buf.c_str() + cut[t + 1];
    History& o = part[t];
    const size_t guess = (size_t)(end - p) / 40 + 16;
    o.id.reserve(guess); o.xyz.reserve(3 * guess); o.it.reserve(guess);
    while (p < end) {
      const char* eol = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
      if (!eol) eol = end;
      const char* q = p;
      while (q < eol && is_space(*q)) q++;
      if (q < eol) {
        Cursor c{q};
        int id;
        double v[3];
        float it;
        bool ok = c.next_int(id) && c.p <= eol;
        for (int k = 0; k < 3 && ok; k++) ok = c.next_double(v[k]) && c.p <= eol;
        ok = ok && c.next_float(it) && c.p <= eol;
        if (!ok) { bad[t] = 1; return; }
        for (q = c.p; q < eol; q++)
          if (!is_space(*q)) { bad[t] = 1; return; }
        o.id.push_back(id);
        o.xyz.push_back(v[0]); o.xyz.push_back(v[1]); o.xyz.push_back(v[2]);
        o.it.push_back(it);
      }
      p = eol + 1;
    }
  };
  run_threads(T, work);
  for (unsigned t = 0; t < T; t++)
    if (bad[t]) return false;
  size_t tot = 0;
  for (unsigned t = 0; t < T; t++) tot += part[t].id.size();
  h.id.reserve(tot); h.xyz.reserve(3 * tot); h.it.reserve(tot);
  for (unsigned t = 0; t < T; t++) {
    h.id.insert(h.id.end(), part[t].id.begin(), part[t].id.end());
    h.xyz.insert(h.xyz.end(), part[t].xyz.begin(), part[t].xyz.end());
    h.it.insert(h.it.end(), part[t].it.begin(), part[t].it.end());
  }
  return true;
}

// pts_preprocess.h:17-49
inline void read_records(const char* poses_file, const char* pts_file, std::vector<PoseRec>& poses, History& h) {
  std::string buf;
  if (slurp(poses_file, buf)) {
    Cursor c{buf.c_str()};
    while (true) {
      PoseRec r;
      memset(&r, 0, sizeof r);
      if (!c.next_int(r.id)) break;
      bool ok = true;
      for (int k = 0; k < 12 && ok; k++) ok = c.next_double(r.w[k]);   // a short line still yields a pose (:28-34)
      poses.push_back(r);
      if (!ok) {   // the stream is now in a failed state in the reference: every later extraction fails
        break;
      }
    }
  }
  if (slurp(pts_file, buf) && !parse_points_parallel(buf, h)) {
    h.id.clear(); h.xyz.clear(); h.it.clear();
    Cursor c{buf.c_str()};
    while (true) {
      int id; double x, y, z; float it;
      if (!c.next_int(id) || !c.next_double(x) || !c.next_double(y) || !c.next_double(z) || !c.next_float(it)) break;
      h.id.push_back(id);
      h.xyz.push_back(x); h.xyz.push_back(y); h.xyz.push_back(z);
      h.it.push_back(it);
    }
  }
}

}  // namespace pr_rec
