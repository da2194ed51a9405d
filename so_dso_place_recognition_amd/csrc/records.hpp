// records.hpp — PosesPts text records (PosesPts.h:5-40, reader utils/pts_preprocess.h:17-49) and the cloud container shared by
// the host pre-stage (host_io.cpp) and the GPU pre-stage (pr_api.cpp / prestage.hip).  Internal to libpr_amd.so.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct pr_clouds {
  std::vector<int64_t> offs;
  std::vector<double> xyz;
  std::vector<float> inten;
  std::vector<int32_t> ids;
  double avg_ms = 0, avg_pts = 0;
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

struct Cursor {
  const char* p;
  bool next_int(int& v) { char* e; long x = strtol(p, &e, 10); if (e == p) return false; v = (int)x; p = e; return true; }
  bool next_double(double& v) { char* e; v = strtod(p, &e); if (e == p) return false; p = e; return true; }
  bool next_float(float& v) { char* e; v = strtof(p, &e); if (e == p) return false; p = e; return true; }
};

struct PoseRec { int id; double w[12]; };
struct History { std::vector<int> id; std::vector<double> xyz; std::vector<float> it; };

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
  if (slurp(pts_file, buf)) {
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
