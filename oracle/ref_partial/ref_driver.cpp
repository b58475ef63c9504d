/*
 * ref_driver.cpp -- PARTIAL reference build, used only to generate golden vectors for tests/golden/.
 *
 * The reference's scoring code is header-only and compiles as it lies under /root/reference:
 *     src/methods.h     JoinMethod1/2::score_permute, merge_scores, compute_value_table_max, keep_score
 *     src/gcre_paths.h  PathSet (load / select / set)
 *     src/gcre.h        UidRelSet::need_flip, JoinExec declaration
 *     src/gcre_types.h  Score ordering, check_* assertions
 *     test/test.cpp     the harness text-dump parser
 * The one translation unit that cannot be built here is src/join_base.cpp (it includes <Rcpp.h>, absent from this
 * image, and no stand-in header is written).  This driver therefore supplies ITS OWN definitions of the JoinExec
 * members that file would define (constructor, setValueTable, setPermutedCases, createMethod, createPathSet, join,
 * format_result) -- a restatement, flagged as such -- and calls the reference's score_permute / merge_scores for
 * every join.  What the goldens pin is consequently the reference's scoring kernels, path-set packing, sign rule
 * and heap semantics; the driver-side members are pinned only by SURVEY.md Appendix B (tests/golden/appendix_b_*).
 *
 * Built by oracle/ref_partial/Makefile into oracle/_ref/ (git-ignored); never shipped, never linked by the product.
 */
#include "gcre.h"
#include "methods.h"
#include "test.h"

#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <string>

using namespace std;

// ---- driver-side JoinExec members (own restatement of what src/join_base.cpp defines) ----------------------------
static int words_for(int patients) { return (patients + 63) / 64; }

JoinExec::JoinExec(string method_name, int num_cases, int num_ctrls, int iters)
    : method(to_method(method_name)), num_cases(num_cases), num_ctrls(num_ctrls),
      width_ul(words_for(num_cases + num_ctrls)), iterations(iters), iters_requested(iters) {
  check_true(num_cases > 0 && num_ctrls > 0 && iters >= 0);
  case_mask = new uint64_t[width_ul]();
  for (int c = 0; c < num_cases; c++) case_mask[c >> 6] |= bit_one_ul << (c & 63);
}

void JoinExec::setValueTable(const vec2d_d& table) {
  const size_t dim = (size_t)num_cases + num_ctrls + 1;
  value_table.assign(dim, vec_d(dim, -1.0));
  for (size_t r = 0; r < table.size() && r < dim; r++)
    for (size_t c = 0; c < table[r].size() && c < dim; c++) value_table[r][c] = table[r][c];
}

void JoinExec::setPermutedCases(const vec2d_i& rows) {
  perm_case_mask = new uint64_t[(size_t)iterations * width_ul + 1]();
  const size_t have = rows.size();
  for (int r = 0; r < iterations; r++) {
    const vec_i& row = rows[(size_t)r < have ? (size_t)r : (size_t)r % have];
    check_equal((size_t)num_cases + num_ctrls, row.size());
    for (int k = 0; k < width_ul; k++) {
      uint64_t flipped = 0;
      for (int b = 0; b < 64 && k * 64 + b < (int)row.size(); b++)
        if (row[k * 64 + b] != 1) flipped |= bit_one_ul << b;
      perm_case_mask[(size_t)k * iterations + r] = case_mask[k] ^ flipped;
    }
  }
}

TJoinMethod JoinExec::createMethod(const UidRelSet& uids, float* p_perm_scores) const {
  if (method == Method::method1) return TJoinMethod(new JoinMethod1(this, uids, p_perm_scores));
  return TJoinMethod(new JoinMethod2(this, uids, p_perm_scores));
}

TPathSet JoinExec::createPathSet(st_pathset_size size) const {
  return TPathSet(new PathSet(size, width_ul, width_ul * (int)method));
}

joined_res JoinExec::format_result() const {
  while (scores.size() > (size_t)top_k) scores.pop();
  joined_res res;
  res.permuted_scores.assign(perm_scores, perm_scores + iters_requested);
  for (; !scores.empty(); scores.pop()) res.scores.push_back(scores.top());
  return res;
}

joined_res JoinExec::join(const UidRelSet& uids, const PathSet& paths0, const PathSet& paths1, PathSet& paths_res) const {
  while (!scores.empty()) scores.pop();
  scores.push(Score());
  check_equal(uids.size(), paths0.size);
  check_true(paths_res.size == 0 || paths_res.size == uids.count_total_paths());
  vector<float> global_null((size_t)iterations + 1, 0.0f), local_null((size_t)iterations + 1, 0.0f);
  perm_scores = global_null.data();
  const bool keep = paths_res.size != 0;
  (void)local_null;
  atomic<size_t> next(0);
  mutex mu;
  auto worker = [&]() {
    vector<float> mine((size_t)iterations + 1, 0.0f);
    TJoinMethod m = createMethod(uids, mine.data());
    vector<uint64_t> row(paths_res.vlen ? paths_res.vlen : 1);
    size_t idx;
    while ((idx = next.fetch_add(1)) < uids.size()) {
      const uid_ref& u = uids[(int)idx];
      st_path_count out = u.path_idx;
      for (int j = 0; j < u.count; j++, out++) {
        const st_pathset_size loc = u.location + j;
        if (keep) fill(row.begin(), row.end(), 0);
        m->score_permute((int)idx, (int)loc, paths0[(st_pathset_size)idx], paths1[loc], row.data(), keep);   // REFERENCE
        if (keep) paths_res.set((st_pathset_size)out, row.data());
      }
    }
    lock_guard<mutex> lock(mu);
    m->merge_scores();                                                                                        // REFERENCE
  };
  if (nthreads <= 0) {
    worker();
  } else {
    vector<thread> pool;
    for (int t = 0; t < nthreads; t++) pool.emplace_back(worker);
    for (auto& th : pool) th.join();
  }
  return format_result();
}

// ---- dump -----------------------------------------------------------------------------------------------------------
static uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static uint64_t hash_rows(const PathSet& ps) {
  uint64_t h = 1469598103934665603ull;
  for (st_pathset_size r = 0; r < ps.size; r++)
    for (int k = 0; k < ps.vlen; k++) { h ^= ps[r][k]; h *= 1099511628211ull; }
  return h;
}

static void dump(const char* name, const joined_res& res, const PathSet* kept, bool last) {
  printf("  \"%s\": {\"scores\": [", name);
  for (size_t k = 0; k < res.scores.size(); k++) printf("%s\"%016" PRIx64 "\"", k ? ", " : "", bits(res.scores[k].score));
  printf("], \"src\": [");
  for (size_t k = 0; k < res.scores.size(); k++) printf("%s%d", k ? ", " : "", res.scores[k].src);
  printf("], \"trg\": [");
  for (size_t k = 0; k < res.scores.size(); k++) printf("%s%d", k ? ", " : "", res.scores[k].trg);
  printf("], \"cases\": [");
  for (size_t k = 0; k < res.scores.size(); k++) printf("%s%d", k ? ", " : "", res.scores[k].cases);
  printf("], \"ctrls\": [");
  for (size_t k = 0; k < res.scores.size(); k++) printf("%s%d", k ? ", " : "", res.scores[k].ctrls);
  printf("], \"null\": [");
  for (size_t k = 0; k < res.permuted_scores.size(); k++)
    printf("%s\"%08x\"", k ? ", " : "", bits((float)res.permuted_scores[k]));
  printf("]");
  if (kept) printf(", \"kept_rows\": %u, \"kept_hash\": \"%016" PRIx64 "\"", (unsigned)kept->size, hash_rows(*kept));
  printf("}%s\n", last ? "" : ",");
}

static UidRelSet with_idx(int path_length, vector<uid_ref> uids, vector<int> signs) {
  st_path_count at = 0;
  for (auto& u : uids) { u.path_idx = at; at += u.count; }
  return UidRelSet(path_length, uids, signs);
}

// ---- timing mode: one join on binary operands (written by bench.py), threads as given ------------------------------
static bool read_exact(FILE* f, void* p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }

static int bench_main(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); return 2; }
  char magic[8];
  int32_t h[8];
  int64_t d[5];
  if (!read_exact(f, magic, 8) || memcmp(magic, "GCREBIN1", 8) || !read_exact(f, h, sizeof h) || !read_exact(f, d, sizeof d)) return 2;
  const int method = h[0], n_cases = h[1], n_ctrls = h[2], K = h[3], W = h[4], top_k = h[5], plen = h[6], nthreads = h[7];
  const int64_t n_uids = d[0], n_rows1 = d[1], n_signs = d[2], t_rows = d[3], t_cols = d[4];
  JoinExec exec(method == 1 ? "method1" : "method2", n_cases, n_ctrls, K);
  exec.top_k = top_k;
  exec.nthreads = nthreads;
  if (exec.width_ul != W) { fprintf(stderr, "width mismatch\n"); return 2; }
  vector<int32_t> cnt(n_uids), sg(n_signs);
  vector<int64_t> loc(n_uids);
  if (!read_exact(f, cnt.data(), cnt.size() * 4) || !read_exact(f, loc.data(), loc.size() * 8) || !read_exact(f, sg.data(), sg.size() * 4)) return 2;
  vector<uid_ref> uv(n_uids);
  for (int64_t i = 0; i < n_uids; i++) { uv[i].src = (int)i; uv[i].trg = 0; uv[i].count = cnt[i]; uv[i].location = (st_pathset_size)loc[i]; }
  UidRelSet uids = with_idx(plen, uv, vector<int>(sg.begin(), sg.end()));
  const int vlen = W * method;
  auto p0 = exec.createPathSet((st_pathset_size)n_uids);
  auto p1 = exec.createPathSet((st_pathset_size)n_rows1);
  vector<uint64_t> row(vlen);
  for (int64_t r = 0; r < n_uids; r++) { if (!read_exact(f, row.data(), (size_t)vlen * 8)) return 2; p0->set((st_pathset_size)r, row.data()); }
  for (int64_t r = 0; r < n_rows1; r++) { if (!read_exact(f, row.data(), (size_t)vlen * 8)) return 2; p1->set((st_pathset_size)r, row.data()); }
  vector<uint64_t> masks((size_t)K * W);
  if (!read_exact(f, masks.data(), masks.size() * 8)) return 2;
  {
    // rebuild the K x n "label kept" rows setPermutedCases expects from the packed masks (bit = case under the permutation)
    vec2d_i rows(K, vec_i(n_cases + n_ctrls));
    for (int r = 0; r < K; r++)
      for (int c = 0; c < n_cases + n_ctrls; c++) {
        const int is_case = (int)((masks[(size_t)r * W + c / 64] >> (c % 64)) & 1);
        rows[r][c] = (is_case == (c < n_cases ? 1 : 0)) ? 1 : 0;
      }
    if (K > 0) exec.setPermutedCases(rows);
  }
  vec2d_d table(t_rows, vec_d(t_cols));
  for (auto& tr : table) if (!read_exact(f, tr.data(), (size_t)t_cols * 8)) return 2;
  fclose(f);
  exec.setValueTable(table);
  auto none = exec.createPathSet(0);
  const auto t0 = chrono::steady_clock::now();
  joined_res res = exec.join(uids, *p0, *p1, *none);
  const double secs = chrono::duration<double>(chrono::steady_clock::now() - t0).count();
  printf("{\"seconds\": %.6f, \"paths\": %llu, \"threads\": %d, \"best\": \"%016" PRIx64 "\", \"null\": [", secs,
         (unsigned long long)uids.count_total_paths(), nthreads, res.scores.empty() ? 0 : bits(res.scores.back().score));
  for (size_t k = 0; k < res.permuted_scores.size(); k++) printf("%s\"%08x\"", k ? ", " : "", bits((float)res.permuted_scores[k]));
  printf("]}\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 2 && !strcmp(argv[1], "--selftest")) { printf("ok\n"); return 0; }
  if (argc == 3 && !strcmp(argv[1], "--bench")) return bench_main(argv[2]);
  if (argc < 6) { fprintf(stderr, "usage: ref_driver <dump.txt> <method1|method2> <iterations> <top_k> <path_length>\n"); return 2; }
  ifstream f(argv[1]);
  const string method = argv[2];
  const int iters = atoi(argv[3]), top_k = atoi(argv[4]), L = atoi(argv[5]);
  test::read_line(f);
  const int num_cases = stoi(test::read_line(f)), num_ctrls = stoi(test::read_line(f));
  JoinExec exec(method, num_cases, num_ctrls, iters);
  exec.top_k = top_k;
  exec.nthreads = 0;
  vector<UidRelSet> lv;
  const int plen[6] = {1, 1, 2, 3, 4, 5};
  for (int i = 0; i < 6; i++) { auto u = test::read_uids(f); auto s = test::read_ints(f); lv.push_back(with_idx(plen[i], u, s)); }
  auto idx0 = test::read_ints(f), idx1 = test::read_ints(f), idx2 = test::read_ints(f), idx3 = test::read_ints(f);
  auto data1 = test::read_data(f), data2 = test::read_data(f), perms = test::read_data(f);
  auto table = test::read_vals(f);
  exec.setValueTable(table);
  if (iters > 0) exec.setPermutedCases(perms);
  auto parsed1 = exec.createPathSet(data1.size());
  parsed1->load(data1);                                  // REFERENCE PathSet::load
  auto zero_set = exec.createPathSet(0);
  TPathSet paths1, paths2, paths3;
  printf("{\n");
  if (L >= 1) {
    paths1 = exec.createPathSet(lv[0].count_total_paths());
    auto z1 = exec.createPathSet(idx0.size());
    auto in1 = parsed1->select(idx0);                    // REFERENCE PathSet::select
    auto r0 = exec.join(lv[0], *z1, *in1, *paths1);
    dump("lst1a", r0, paths1.get(), false);
    auto z2 = exec.createPathSet(idx1.size());
    auto parsed2 = exec.createPathSet(data2.size());
    parsed2->load(data2);
    auto in2 = parsed2->select(idx1);
    dump("lst1", exec.join(lv[1], *z2, *in2, *zero_set), nullptr, L == 1);
  }
  if (L >= 2) {
    paths2 = exec.createPathSet(lv[2].count_total_paths());
    auto in = parsed1->select(idx2);
    dump("lst2", exec.join(lv[2], *paths1, *in, *paths2), paths2.get(), L == 2);
  }
  if (L >= 3) {
    paths3 = exec.createPathSet(lv[3].count_total_paths());
    auto in = parsed1->select(idx3);
    dump("lst3", exec.join(lv[3], *paths2, *in, *paths3), paths3.get(), L == 3);
  }
  if (L >= 4) dump("lst4", exec.join(lv[4], *paths3, *paths2, *zero_set), nullptr, L == 4);
  if (L >= 5) dump("lst5", exec.join(lv[5], *paths3, *paths3, *zero_set), nullptr, true);
  printf("}\n");
  return 0;
}
