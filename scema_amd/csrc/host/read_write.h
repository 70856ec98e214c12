// read_write.h -- tensor text files and the replica json of the hot path
// (reference headers/read_write.h:72-244; one "%.16g"-style value per line).
#pragma once
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <string>

#include "tensors.h"

namespace scema {

inline bool file_exists(const std::string &file) {
  struct stat buf;
  return stat(file.c_str(), &buf) == 0;
}

inline bool read_tensor(const char *filename, double &t) {
  std::ifstream f(filename);
  if (!f.is_open()) { std::cout << "Unable to open" << filename << " to read it" << std::endl; return false; }
  std::string line;
  if (std::getline(f, line)) t = std::strtod(line.c_str(), nullptr);
  return true;
}
inline bool read_tensor(const char *filename, Tensor1 &t) {
  std::ifstream f(filename);
  if (!f.is_open()) { std::cout << "Unable to open" << filename << " to read it" << std::endl; return false; }
  std::string line;
  for (int k = 0; k < 3; k++)
    if (std::getline(f, line)) t[k] = std::strtod(line.c_str(), nullptr);
  return true;
}
// order 00,01,02,11,12,22 (reference read_write.h:136-142)
inline bool read_tensor(const char *filename, SymmetricTensor2 &t) {
  std::ifstream f(filename);
  if (!f.is_open()) { std::cout << "Unable to open" << filename << " to read it" << std::endl; return false; }
  std::string line;
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++)
      if (std::getline(f, line)) t(k, l) = std::strtod(line.c_str(), nullptr);
  return true;
}
// 36 lines (reference read_write.h:159-167)
inline bool read_tensor(const char *filename, SymmetricTensor4 &t) {
  std::ifstream f(filename);
  if (!f.is_open()) { std::cout << "Unable to open" << filename << " to read it..." << std::endl; return false; }
  std::string line;
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++)
      for (int m = 0; m < 3; m++)
        for (int n = m; n < 3; n++)
          if (std::getline(f, line)) t(k, l, m, n) = std::strtod(line.c_str(), nullptr);
  return true;
}
inline bool write_tensor(const char *filename, const double &t) {
  std::ofstream f(filename);
  if (!f.is_open()) { std::cout << "Unable to open" << filename << " to write in it" << std::endl; return false; }
  f << std::setprecision(16) << t << std::endl;
  return true;
}
inline bool write_tensor(const char *filename, const Tensor1 &t) {
  std::ofstream f(filename);
  if (!f.is_open()) { std::cout << "Unable to open" << filename << " to write in it" << std::endl; return false; }
  for (int k = 0; k < 3; k++) f << std::setprecision(16) << t[k] << std::endl;
  return true;
}
inline bool write_tensor(const char *filename, const SymmetricTensor2 &t) {
  std::ofstream f(filename);
  if (!f.is_open()) { std::cout << "Unable to open" << filename << " to write in it" << std::endl; return false; }
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++) f << std::setprecision(16) << t(k, l) << std::endl;
  return true;
}
inline bool write_tensor(const char *filename, const SymmetricTensor4 &t) {
  std::ofstream f(filename);
  if (!f.is_open()) { std::cout << "Unable to open" << filename << " to write in it" << std::endl; return false; }
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++)
      for (int m = 0; m < 3; m++)
        for (int n = m; n < 3; n++) f << std::setprecision(16) << t(k, l, m, n) << std::endl;
  return true;
}

// Minimal JSON reader for <mat>_<r>.json (the reference walks a Boost property tree,
// stmd_sync.h:306-345): flattens scalars to "a.b.c" -> text.
class FlatJson {
 public:
  bool parse_file(const std::string &path) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    s_ = ss.str();
    p_ = 0;
    kv_.clear();
    skip();
    if (p_ >= s_.size() || s_[p_] != '{') return false;
    return object("");
  }
  bool has(const std::string &k) const { return kv_.count(k) != 0; }
  std::string get(const std::string &k, const std::string &dflt = "") const {
    auto it = kv_.find(k);
    return it == kv_.end() ? dflt : it->second;
  }

 private:
  std::string s_;
  size_t p_ = 0;
  std::map<std::string, std::string> kv_;
  void skip() { while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\n' || s_[p_] == '\t' || s_[p_] == '\r')) p_++; }
  bool str(std::string &out) {
    if (s_[p_] != '"') return false;
    p_++;
    out.clear();
    while (p_ < s_.size() && s_[p_] != '"') {
      if (s_[p_] == '\\' && p_ + 1 < s_.size()) p_++;
      out.push_back(s_[p_++]);
    }
    if (p_ >= s_.size()) return false;
    p_++;
    return true;
  }
  bool value(const std::string &key) {
    skip();
    if (p_ >= s_.size()) return false;
    if (s_[p_] == '{') return object(key);
    if (s_[p_] == '[') {
      p_++;
      int idx = 0;
      skip();
      if (s_[p_] == ']') { p_++; return true; }
      for (;;) {
        if (!value(key + "." + std::to_string(idx++))) return false;
        skip();
        if (s_[p_] == ',') { p_++; continue; }
        if (s_[p_] == ']') { p_++; return true; }
        return false;
      }
    }
    if (s_[p_] == '"') {
      std::string v;
      if (!str(v)) return false;
      kv_[key] = v;
      return true;
    }
    size_t b = p_;
    while (p_ < s_.size() && s_[p_] != ',' && s_[p_] != '}' && s_[p_] != ']' && s_[p_] != ' ' && s_[p_] != '\n') p_++;
    kv_[key] = s_.substr(b, p_ - b);
    return true;
  }
  bool object(const std::string &prefix) {
    p_++;  // {
    skip();
    if (s_[p_] == '}') { p_++; return true; }
    for (;;) {
      skip();
      std::string k;
      if (!str(k)) return false;
      skip();
      if (s_[p_] != ':') return false;
      p_++;
      if (!value(prefix.empty() ? k : prefix + "." + k)) return false;
      skip();
      if (s_[p_] == ',') { p_++; continue; }
      if (s_[p_] == '}') { p_++; return true; }
      return false;
    }
  }
};

}  // namespace scema
