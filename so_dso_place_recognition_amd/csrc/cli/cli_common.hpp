// cli_common.hpp — parameter handling of the drop-in executables.  The reference's executables are ROS nodes whose
// "CLI" is a set of private parameters (test_sc.cpp:17-28, launch/lidar.launch:15-21), settable on argv as
// `_name:=value` (consumed by ros::init).  Both that form and `--name value` / `--name=value` are accepted.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

struct Params {
  std::map<std::string, std::string> kv;
  Params(int argc, char** argv) {
    for (int i = 1; i < argc; i++) {
      std::string a = argv[i];
      size_t p;
      if (a.size() > 1 && a[0] == '_' && (p = a.find(":=")) != std::string::npos) kv[a.substr(1, p - 1)] = a.substr(p + 2);
      else if (a.rfind("--", 0) == 0) {
        if ((p = a.find('=')) != std::string::npos) kv[a.substr(2, p - 2)] = a.substr(p + 1);
        else if (i + 1 < argc) kv[a.substr(2)] = argv[++i];
        else kv[a.substr(2)] = "";
      }
    }
  }
  bool get(const char* name, std::string& out) const {
    auto it = kv.find(name);
    if (it == kv.end()) return false;
    out = it->second;
    return true;
  }
  double num(const char* name, double def) const {
    auto it = kv.find(name);
    return it == kv.end() ? def : atof(it->second.c_str());
  }
};

// utils/print_progress.h:8-14
inline void printProgress(double percentage) {
  static const char* bar = "||||||||||||||||||||||||||||||||||||||||||||||||||||||||||||";
  const int val = (int)(percentage * 100), lpad = (int)(percentage * 60), rpad = 60 - lpad;
  printf("\r%3d%% [%.*s%*s]", val, lpad, bar, rpad, "");
  fflush(stdout);
}
