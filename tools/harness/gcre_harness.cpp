// gcre_harness.cpp -- native stand-alone driver of the MI355X path-join scorer, the counterpart of the reference's
// test/harness.cpp (main: harness.cpp:21-184): reads the same text dump of all ProcessPaths inputs (format: test/test.cpp
// 23-118), takes the same flags, runs the same six-join sequence -- through the C ABI of
// include/gcre_hip.h (gcre_process_paths / gcre_process_paths_devices in libgcre_hip.so, loaded at run time) -- and prints
// the level-4 block in the reference's layout (harness.cpp:160-173) plus, with -a, every level.
//
//   g++ -O2 -std=c++17 -Iinclude tools/harness/gcre_harness.cpp -ldl -o tools/harness/gcre_harness
//   tools/harness/gcre_harness -f dump.txt [-p perms] [-m method1|method2] [-l length] [-k top_k] [-t devices] [-r repeats] [-a]
//
// Flags as harness.cpp:42-63: -p permutations (default 10), -m method (default method2), -l path length (default: the
// file's), -k top_k (default 12), -r repeats of the whole sequence (the reference repeats the level-4 join; timing only).
// -t: the reference's thread count; here the number of GPUs of the node to use (0 or absent: one device).
// There is no CPU fallback: without a gfx950 device the library reports GCRE_ERR_DEVICE and the program exits 2.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "gcre_hip.h"

namespace {

std::string payload(const std::string& line) {   // everything up to the first space is a free label (test.cpp:23-30)
  const size_t p = line.find(' ');
  return p == std::string::npos ? std::string() : line.substr(p + 1);
}

std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> out;
  std::string tok;
  std::istringstream is(s);
  while (std::getline(is, tok, sep))
    if (!tok.empty()) out.push_back(tok);
  return out;
}

struct Level {
  std::vector<int32_t> src, trg, count, signs;
  std::vector<int64_t> location;
};

template <typename T>
struct Matrix {
  std::vector<T> v;
  int64_t rows = 0, cols = 0;
};

template <typename T>
Matrix<T> parse_matrix(const std::string& text) {   // rows space-separated, columns comma-separated (test.cpp:76-118)
  Matrix<T> m;
  for (const std::string& row : split(text, ' ')) {
    const auto cells = split(row, ',');
    if (m.rows == 0) m.cols = (int64_t)cells.size();
    if ((int64_t)cells.size() != m.cols) { std::fprintf(stderr, "ragged matrix row\n"); std::exit(1); }
    for (const std::string& c : cells) m.v.push_back((T)std::strtod(c.c_str(), nullptr));
    m.rows++;
  }
  return m;
}

}  // namespace

int main(int argc, char** argv) {
  std::string file, method = "method2", libpath;
  int perms = 10, length = -1, top_k = 12, devices = 0, repeat = 1;
  bool all = false;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(1); } return argv[++i]; };
    if (a == "-f") file = next();
    else if (a == "-p") perms = std::atoi(next());
    else if (a == "-m") method = next();
    else if (a == "-l") length = std::atoi(next());
    else if (a == "-k") top_k = std::atoi(next());
    else if (a == "-t") devices = std::atoi(next());
    else if (a == "-r") repeat = std::max(1, std::atoi(next()));
    else if (a == "-a") all = true;
    else if (a == "--lib") libpath = next();
    else if (a == "-h" || a == "--help") {
      std::printf("usage: gcre_harness -f dump.txt [-p perms] [-m method1|method2] [-l length] [-k top_k] [-t devices] [-r repeats] [-a] [--lib libgcre_hip.so]\n");
      return 0;
    }
  }
  if (file.empty()) { std::fprintf(stderr, "gcre_harness: -f <dump> is required\n"); return 1; }

  // ---- the dump (harness.cpp:38-107) ----
  std::ifstream in(file);
  if (!in) { std::fprintf(stderr, "cannot open %s\n", file.c_str()); return 1; }
  std::vector<std::string> lines;
  for (std::string l; std::getline(in, l);)
    if (l.find_first_not_of(" \t\r\n") != std::string::npos) lines.push_back(payload(l));
  if (lines.size() < 3 + 12 + 4 + 4) { std::fprintf(stderr, "%s: truncated dump (%zu records)\n", file.c_str(), lines.size()); return 1; }
  size_t at = 0;
  const int file_len = std::atoi(lines[at++].c_str());
  const int n_cases = std::atoi(lines[at++].c_str()), n_ctrls = std::atoi(lines[at++].c_str());
  Level lv[6];
  for (int k = 0; k < 6; k++) {
    for (const std::string& tok : split(lines[at++], ' ')) {   // src:trg:count:location (test.cpp:32-58)
      const auto f = split(tok, ':');
      if (f.size() != 4) { std::fprintf(stderr, "bad uid record '%s'\n", tok.c_str()); return 1; }
      lv[k].src.push_back(std::atoi(f[0].c_str()));
      lv[k].trg.push_back(std::atoi(f[1].c_str()));
      lv[k].count.push_back(std::atoi(f[2].c_str()));
      lv[k].location.push_back(std::atoll(f[3].c_str()));
    }
    for (const std::string& tok : split(lines[at++], ' ')) lv[k].signs.push_back(std::atoi(tok.c_str()));
  }
  std::vector<int32_t> idx[4];
  for (int k = 0; k < 4; k++)
    for (const std::string& tok : split(lines[at++], ' ')) idx[k].push_back(std::atoi(tok.c_str()));
  const Matrix<int32_t> data1 = parse_matrix<int32_t>(lines[at++]), data2 = parse_matrix<int32_t>(lines[at++]);
  const Matrix<int32_t> pm = parse_matrix<int32_t>(lines[at++]);
  const Matrix<double> table = parse_matrix<double>(lines[at++]);
  if (length < 0) length = file_len;

  // ---- the library ----
  if (libpath.empty()) {
    const char* e = std::getenv("GCRE_HIP_LIB");
    libpath = e ? e : "libgcre_hip.so";
  }
  void* h = dlopen(libpath.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) { std::fprintf(stderr, "cannot load %s: %s\n", libpath.c_str(), dlerror()); return 2; }
  auto pp = (int (*)(int, int, int, int, int, const int*, int, const gcre_pp_input*, gcre_result[5], char*, size_t))dlsym(h, "gcre_process_paths_devices");
  auto rfree = (void (*)(gcre_result*))dlsym(h, "gcre_result_free");
  if (!pp || !rfree) { std::fprintf(stderr, "%s lacks the C ABI of include/gcre_hip.h\n", libpath.c_str()); return 2; }

  gcre_pp_input inp;
  std::memset(&inp, 0, sizeof inp);
  for (int k = 0; k < 6; k++) {
    inp.level[k].uid_count = lv[k].count.data();
    inp.level[k].uid_location = lv[k].location.data();
    inp.level[k].n_uids = (int64_t)lv[k].count.size();
    inp.level[k].signs = lv[k].signs.data();
    inp.level[k].n_signs = (int64_t)lv[k].signs.size();
  }
  for (int k = 0; k < 4; k++) {
    inp.data_inds[k] = idx[k].data();
    inp.n_data_inds[k] = (int64_t)idx[k].size();
  }
  inp.data1 = data1.v.data();
  inp.data1_rows = data1.rows;
  inp.data2 = data2.v.data();
  inp.data2_rows = data2.rows;
  inp.value_table = table.v.data();
  inp.vt_rows = (int)table.rows;
  inp.vt_cols = (int)table.cols;
  inp.perm_cases = pm.rows ? pm.v.data() : nullptr;
  inp.perm_rows = (int)pm.rows;
  inp.path_length = length;

  const int m = method == "method1" ? 1 : 2;   // anything else is method 2 (gcre.h:125-133)
  gcre_result res[5];
  char err[512] = "";
  double best_ms = 1e300;
  for (int r = 0; r < repeat; r++) {
    if (r) for (auto& x : res) rfree(&x);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = pp(m, n_cases, n_ctrls, perms, top_k, nullptr, devices > 0 ? devices : 1, &inp, res, err, sizeof err);
    best_ms = std::min(best_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (rc != GCRE_OK) { std::fprintf(stderr, "gcre_harness: %s (code %d)\n", err, rc); return 2; }
  }
  const int W = (n_cases + n_ctrls + 63) / 64;
  auto block = [&](int level, const gcre_result& r) {
    std::printf("\n################################\n");
    std::printf("   length : %d  width: %d  iters: %d  thread: %d\n", level, W, perms, devices);
    std::printf("  results : %d |", r.n);
    for (int k = 0; k < r.n; k++) std::printf(" %f[%d:%d]", r.scores[k], r.src[k], r.trg[k]);
    std::printf("\n    perms :");
    for (int k = 0; k < std::min(24, r.n_perm); k++) std::printf(" %0.2f", (double)r.null_max[k]);
    std::printf("\n################################\n");
  };
  if (all) {
    for (int l = 1; l <= 5; l++)
      if (res[l - 1].n >= 0) block(l, res[l - 1]);
  } else if (length >= 4 && res[3].n >= 0) {
    block(length, res[3]);   // the reference prints the level-4 join only (harness.cpp:160-173)
  }
  std::fprintf(stderr, "[gcre_harness] %d repeat(s), best %.2f ms per ProcessPaths sequence\n", repeat, best_ms);
  for (auto& x : res) rfree(&x);
  std::printf("done\n");
  return 0;
}
