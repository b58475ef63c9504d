/*
 * gcre_oracle.cpp -- CPU restatement of the reference join engine.  TEST INFRASTRUCTURE ONLY
 * (see gcre_oracle.h for who may load it and for the parity status).
 *
 * Written from the behaviour documented in SURVEY.md §8/App. A; each function names the
 * reference lines it follows.  Nothing here is shared with the HIP product path.
 */
#include "gcre_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

namespace {

struct Entry {            // mirrors Score, src/gcre_types.h:32-43
  double score;
  int src, trg, cases, ctrls;
  int64_t path;           // joined-path ordinal, only used by the canonical order
};

// reversed comparison -> std::priority_queue becomes a min-heap on score (gcre_types.h:42)
struct MinHeapLess {
  bool operator()(const Entry& a, const Entry& b) const { return a.score > b.score; }
};
using Heap = std::priority_queue<Entry, std::vector<Entry>, MinHeapLess>;

inline Entry sentinel() {  // Score() default, gcre_types.h:34-38
  return Entry{-std::numeric_limits<double>::infinity(), -1, -1, 0, 0, -1};
}

}  // namespace

struct gcre_o_ctx {
  int method, n_cases, n_ctrls, n, iters, width, vlen;
  int tdim;                          // n + 1
  std::vector<uint64_t> case_mask;   // [width]
  std::vector<uint64_t> perm_mask;   // [width][iters], word-major / perm-minor (join_base.cpp:109)
  std::vector<double> vt;            // [(n+1)^2], padded with -1 (join_base.cpp:72)
  std::vector<double> vtmax;         // max(vt[r][c], vt[c][r]) (methods.h:110-118)
  // Past 16k patients the padded square is gigabytes (20 GB at 50,000 patients, twice for method 2: SURVEY App. C): the
  // supplied nrow x ncol table is kept as it came and the padding / the symmetrisation are evaluated per access --
  // the same cells, the same values.
  bool lazy = false;
  int t_rows = 0, t_cols = 0;
  std::vector<double> raw;
  double at(size_t r, size_t q) const {
    if (!lazy) return vt[r * size_t(tdim) + q];
    return (r < size_t(t_rows) && q < size_t(t_cols)) ? raw[r * size_t(t_cols) + q] : -1.0;
  }
  double at_max(size_t r, size_t q) const {
    if (!lazy) return vtmax[r * size_t(tdim) + q];
    return std::max(at(r, q), at(q, r));
  }
  bool have_table = false, have_perms = false;
};

extern "C" {

gcre_o_ctx* gcre_o_create(int method, int n_cases, int n_ctrls, int iters) {
  // check_true(num_cases > 0 && num_ctrls > 0 && iters >= 0) -- join_base.cpp:47
  if ((method != 1 && method != 2) || n_cases <= 0 || n_ctrls <= 0 || iters < 0) return nullptr;
  auto* c = new gcre_o_ctx();
  c->method = method;
  c->n_cases = n_cases;
  c->n_ctrls = n_ctrls;
  c->n = n_cases + n_ctrls;
  c->iters = iters;
  c->width = (c->n + 63) / 64;
  c->vlen = c->width * method;
  c->tdim = c->n + 1;
  // cases are patient columns 0..n_cases-1 -- join_base.cpp:50-54
  c->case_mask.assign(c->width, 0);
  for (int k = 0; k < n_cases; k++) c->case_mask[k / 64] |= uint64_t(1) << (k % 64);
  c->perm_mask.assign(size_t(c->width) * iters, 0);
  return c;
}

void gcre_o_destroy(gcre_o_ctx* ctx) { delete ctx; }

int gcre_o_width(const gcre_o_ctx* ctx) { return ctx->width; }
int gcre_o_vlen(const gcre_o_ctx* ctx) { return ctx->vlen; }

int gcre_o_set_value_table(gcre_o_ctx* c, const double* tbl, int nrow, int ncol) {
  // (n+1)x(n+1), cells outside the supplied table are -1 -- join_base.cpp:67-78
  const int T = c->tdim;
  c->lazy = T > 16384;
  if (c->lazy) {
    c->t_rows = std::min(T, nrow);
    c->t_cols = std::min(T, ncol);
    c->raw.resize(size_t(c->t_rows) * c->t_cols);
    for (int r = 0; r < c->t_rows; r++)
      for (int q = 0; q < c->t_cols; q++) c->raw[size_t(r) * c->t_cols + q] = tbl[size_t(r) * ncol + q];
    c->vt.clear();
    c->vtmax.clear();
    c->have_table = true;
    return 0;
  }
  c->vt.assign(size_t(T) * T, -1.0);
  for (int r = 0; r < std::min(T, nrow); r++)
    for (int q = 0; q < std::min(T, ncol); q++) c->vt[size_t(r) * T + q] = tbl[size_t(r) * ncol + q];
  if (c->method == 2) {
    // compute_value_table_max -- methods.h:110-118 (the reference redoes this per worker per join)
    c->vtmax.resize(c->vt.size());
    for (int r = 0; r < T; r++)
      for (int q = 0; q < T; q++)
        c->vtmax[size_t(r) * T + q] = std::max(c->vt[size_t(r) * T + q], c->vt[size_t(q) * T + r]);
  }
  c->have_table = true;
  return 0;
}

int gcre_o_set_perm_cases(gcre_o_ctx* c, const int* perms, int nrow, int ncol) {
  const int K = c->iters, W = c->width;
  std::fill(c->perm_mask.begin(), c->perm_mask.end(), 0);
  if (K > 0 && nrow <= 0) return -1;            // reference divides by zero here (join_base.cpp:119)
  if (nrow > 0 && ncol != c->n) return -1;      // check_equal -- join_base.cpp:101
  const int rows = std::min(K, nrow);           // surplus rows are dropped -- join_base.cpp:89-90,97
  std::vector<uint64_t> flipped(W);
  for (int r = 0; r < rows; r++) {
    std::fill(flipped.begin(), flipped.end(), 0);
    for (int q = 0; q < ncol; q++)              // anything but 1 means "label flipped" -- :103-104
      if (perms[size_t(r) * ncol + q] != 1) flipped[q / 64] |= uint64_t(1) << (q % 64);
    for (int k = 0; k < W; k++) c->perm_mask[size_t(k) * K + r] = c->case_mask[k] ^ flipped[k];  // :108-109
  }
  for (int r = rows; r < K; r++) {              // too few rows: reuse cyclically -- :116-123
    const int s = r % nrow;
    for (int k = 0; k < W; k++) c->perm_mask[size_t(k) * K + r] = c->perm_mask[size_t(k) * K + s];
  }
  c->have_perms = true;
  return 0;
}

int gcre_o_set_perm_masks(gcre_o_ctx* c, const uint64_t* masks, int nrow) {
  // the masks setPermutedCases would derive, supplied packed: [nrow][width]
  const int K = c->iters, W = c->width;
  std::fill(c->perm_mask.begin(), c->perm_mask.end(), 0);
  if (K > 0 && nrow <= 0) return -1;
  for (int r = 0; r < K; r++) {
    const int s = r % nrow;
    for (int k = 0; k < W; k++) c->perm_mask[size_t(k) * K + r] = masks[size_t(s) * W + k];
  }
  c->have_perms = true;
  return 0;
}

int gcre_o_get_perm_mask(const gcre_o_ctx* c, int r, uint64_t* out) {
  if (r < 0 || r >= c->iters) return -2;
  for (int k = 0; k < c->width; k++) out[k] = c->perm_mask[size_t(k) * c->iters + r];
  return 0;
}

int gcre_o_pack_dense(const gcre_o_ctx* c, const int* data, int nrow, int ncol, uint64_t* out) {
  // bits always land in the first `width` words of the row (the "pos" half) -- gcre_paths.h:56-70
  // check_index(data[r].size(), width_ul*64) -- gcre_paths.h:63: every column must fit the mask words
  if (ncol > c->width * 64) return -2;
  std::memset(out, 0, sizeof(uint64_t) * size_t(nrow) * c->vlen);
  for (int r = 0; r < nrow; r++)
    for (int q = 0; q < ncol; q++)
      if (data[size_t(r) * ncol + q] != 0) out[size_t(r) * c->vlen + q / 64] |= uint64_t(1) << (q % 64);
  return 0;
}

}  // extern "C"

namespace {

struct Worker {
  const gcre_o_ctx* c;
  int path_length;
  const int* signs;
  int top_k;
  Heap heap;
  std::vector<float> null_max;       // thread-private, starts at 0 -- methods.h:17-18
  std::vector<uint32_t> cnt;         // per-permutation counts (stack VLA in the reference)

  Worker(const gcre_o_ctx* c_, int pl, const int* s, int k)
      : c(c_), path_length(pl), signs(s), top_k(k), null_max(c_->iters, 0.0f), cnt(size_t(c_->iters) * c_->method) {
    heap.push(sentinel());           // methods.h:14
  }

  // UidRelSet::need_flip -- gcre.h:71-81.  true = keep path1's halves as they are.
  bool keep_halves(int64_t idx, int64_t loc) const {
    int sign;
    if (path_length > 3) sign = signs[idx];
    else if (path_length < 3) sign = signs[loc];
    else sign = (signs[idx] + signs[loc] == 0) ? -1 : 1;
    return sign == 1;
  }

  void offer(double score, int64_t idx, int64_t loc, int cases, int ctrls, int64_t path) {
    // strict '>' against the current minimum, then trim -- methods.h:91-94 / 255-263
    if (score > heap.top().score) heap.push(Entry{score, int(idx), int(loc), cases, ctrls, path});
    while (heap.size() > size_t(top_k)) heap.pop();
  }

  // JoinMethod1::score_permute -- methods.h:58-105
  void score_m1(int64_t idx, int64_t loc, const uint64_t* p0, const uint64_t* p1, uint64_t* res, int64_t path,
                double* all_scores, int* all_cases, int* all_ctrls) {
    const int W = c->width, K = c->iters, T = c->tdim;
    uint32_t* pc = cnt.data();
    std::memset(pc, 0, sizeof(uint32_t) * K);
    int cases = 0, ctrls = 0;
    for (int k = 0; k < W; k++) {
      const uint64_t joined = p0[k] | p1[k];
      if (joined == 0) continue;                                  // :75
      cases += __builtin_popcountll(joined & c->case_mask[k]);     // :77
      ctrls += __builtin_popcountll(joined & ~c->case_mask[k]);    // :78
      const uint64_t* m = c->perm_mask.data() + size_t(k) * K;
      for (int r = 0; r < K; r++) pc[r] += __builtin_popcountll(joined & m[r]);   // :81-82
      if (res) res[k] = joined;                                   // :84-85
    }
    const double score = c->at(size_t(cases), size_t(ctrls));         // :90
    offer(score, idx, loc, cases, ctrls, path);
    if (all_scores) { all_scores[path] = score; all_cases[path] = cases; all_ctrls[path] = ctrls; }
    const int total = cases + ctrls;
    float* nm = null_max.data();
    for (int r = 0; r < K; r++) {                                  // :96-103
      const double p = c->at(size_t(pc[r]), size_t(total - int(pc[r])));
      if (p > double(nm[r])) nm[r] = float(p);
    }
  }

  // JoinMethod2::score_permute + keep_score -- methods.h:130-232, 253-264
  void score_m2(int64_t idx, int64_t loc, const uint64_t* p0, const uint64_t* p1, uint64_t* res, int64_t path,
                double* all_scores, int* all_cases, int* all_ctrls) {
    const int W = c->width, K = c->iters, T = c->tdim;
    const bool keep = keep_halves(idx, loc);
    const uint64_t* pos0 = p0;
    const uint64_t* neg0 = p0 + W;
    const uint64_t* pos1 = keep ? p1 : p1 + W;                     // :140-142
    const uint64_t* neg1 = keep ? p1 + W : p1;
    uint32_t* pa = cnt.data();        // popc(pos & mask_r)
    uint32_t* pb = cnt.data() + K;    // popc(neg & mask_r)
    std::memset(pa, 0, sizeof(uint32_t) * 2 * K);
    uint32_t case_pos = 0, case_neg = 0, ctrl_pos = 0, ctrl_neg = 0, total_pos = 0, total_neg = 0;
    for (int k = 0; k < W; k++) {
      const uint64_t bp = pos0[k] | pos1[k];                       // :164-165
      const uint64_t bn = neg0[k] | neg1[k];
      if (bp == 0 && bn == 0) continue;                            // :167
      const uint64_t cm = c->case_mask[k];
      total_pos += __builtin_popcountll(bp);                       // :180-185
      total_neg += __builtin_popcountll(bn);
      case_pos += __builtin_popcountll(bp & cm);
      case_neg += __builtin_popcountll(bn & ~cm);
      ctrl_pos += __builtin_popcountll(bn & cm);
      ctrl_neg += __builtin_popcountll(bp & ~cm);
      const uint64_t* m = c->perm_mask.data() + size_t(k) * K;
      if (bp != 0) for (int r = 0; r < K; r++) pa[r] += __builtin_popcountll(bp & m[r]);   // :188-193
      if (bn != 0) for (int r = 0; r < K; r++) pb[r] += __builtin_popcountll(bn & m[r]);   // :197-202
      if (res) { res[k] = bp; res[W + k] = bn; }                   // :205-208
    }
    // observed score uses vt, the null uses vtmax -- methods.h:255 vs :227
    const double score = c->at(size_t(case_pos), size_t(ctrl_neg)) + c->at(size_t(case_neg), size_t(ctrl_pos));
    const int cases = int(case_pos + case_neg), ctrls = int(ctrl_pos + ctrl_neg);   // :256-257
    offer(score, idx, loc, cases, ctrls, path);
    if (all_scores) { all_scores[path] = score; all_cases[path] = cases; all_ctrls[path] = ctrls; }
    float* nm = null_max.data();
    for (int r = 0; r < K; r++) {                                  // :220-230
      const int a = int(pa[r]), b = int(pb[r]);
      const int perm_case_neg = int(total_neg) - b;
      const int perm_ctrl_neg = int(total_pos) - a;
      const double p = c->at_max(size_t(a), size_t(perm_ctrl_neg)) + c->at_max(size_t(perm_case_neg), size_t(b));
      if (p > double(nm[r])) nm[r] = float(p);
    }
  }
};

}  // namespace

extern "C" int gcre_o_join(gcre_o_ctx* c, int path_length,
                           const int* uid_count, const int64_t* uid_location, int64_t n_uids,
                           const int* signs, int64_t n_signs,
                           const uint64_t* paths0, int64_t n0,
                           const uint64_t* paths1, int64_t n1,
                           uint64_t* paths_res,
                           int top_k, int nthreads, int order_mode,
                           double* out_scores, int* out_src, int* out_trg, int* out_cases, int* out_ctrls, int* out_n,
                           float* out_null,
                           double* all_scores, int* all_cases, int* all_ctrls) {
  (void)n_signs;
  if (!c || !c->have_table || (c->iters > 0 && !c->have_perms) || top_k < 1) return -1;
  const int vlen = c->vlen, K = c->iters;
  // checks of join_base.cpp:196-200
  if (n_uids != n0) return -1;
  std::vector<int64_t> path_idx(n_uids + 1, 0);   // uid_ref.path_idx: prefix sum of count (wrapper.cpp:128-130)
  for (int64_t i = 0; i < n_uids; i++) {
    const int cnt = uid_count[i];
    if (cnt > 0) {
      const int64_t last = uid_location[i] + cnt - 1;
      if (uid_location[i] < 0 || last >= n1) return -2;
    }
    path_idx[i + 1] = path_idx[i] + std::max(cnt, 0);
  }

  Heap global;
  global.push(sentinel());                         // join_base.cpp:192-194
  std::vector<float> global_null(K, 0.0f);         // :203-206
  std::atomic<int64_t> next(0);
  std::mutex mu;

  auto work = [&]() {
    Worker w(c, path_length, signs, top_k);
    std::vector<uint64_t> row(vlen);
    int64_t idx;
    while ((idx = next.fetch_add(1)) < n_uids) {   // dynamic uid scheduling -- :230
      const int cnt = uid_count[idx];
      if (cnt <= 0) continue;                      // :236
      const uint64_t* p0 = paths0 + size_t(idx) * vlen;
      int64_t path = path_idx[idx];
      for (int64_t loc = uid_location[idx]; loc < uid_location[idx] + cnt; loc++, path++) {   // :242
        uint64_t* res = nullptr;
        if (paths_res) {                           // zeroed, then only nonzero words written -- :243-249
          res = row.data();
          std::memset(res, 0, sizeof(uint64_t) * vlen);
        }
        const uint64_t* p1 = paths1 + size_t(loc) * vlen;
        if (c->method == 1) w.score_m1(idx, loc, p0, p1, res, path, all_scores, all_cases, all_ctrls);
        else w.score_m2(idx, loc, p0, p1, res, path, all_scores, all_cases, all_ctrls);
        if (paths_res) std::memcpy(paths_res + size_t(path) * vlen, res, sizeof(uint64_t) * vlen);
      }
    }
    // merge_scores under the lock -- join_base.cpp:257-258, methods.h:25-39
    std::lock_guard<std::mutex> lock(mu);
    while (!w.heap.empty()) {
      const Entry e = w.heap.top();
      if (e.score > global.top().score) global.push(e);
      w.heap.pop();
    }
    for (int r = 0; r < K; r++)
      if (w.null_max[r] > global_null[r]) global_null[r] = w.null_max[r];
  };

  if (std::max(0, nthreads) == 0) {                // inline on the caller -- join_base.cpp:170-171
    work();
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++) pool.emplace_back(work);
    for (auto& th : pool) th.join();
  }

  // format_result -- join_base.cpp:138-154
  while (global.size() > size_t(top_k)) global.pop();
  std::vector<Entry> out;
  while (!global.empty()) { out.push_back(global.top()); global.pop(); }   // ascending

  if (order_mode == 1) {
    // canonical tie rule: same multiset of score values, ties resolved towards the
    // smaller joined-path ordinal.  Needs the full score list, so all_scores must be supplied.
    if (!all_scores) return -1;
    const int64_t P = path_idx[n_uids];
    std::vector<Entry> cand;
    cand.push_back(sentinel());
    for (int64_t i = 0; i < n_uids; i++)
      for (int j = 0; j < std::max(uid_count[i], 0); j++) {
        const int64_t p = path_idx[i] + j;
        if (all_scores[p] > -std::numeric_limits<double>::infinity())   // strict '>' vs the sentinel; NaN fails too
          cand.push_back(Entry{all_scores[p], int(i), int(uid_location[i] + j), all_cases[p], all_ctrls[p], p});
      }
    (void)P;
    auto better = [](const Entry& a, const Entry& b) {
      if (a.score != b.score) return a.score > b.score;
      // the sentinel (path -1) only ever ties with nothing: real entries are > -inf
      return a.path < b.path;
    };
    const size_t keep = std::min(cand.size(), size_t(top_k));
    std::partial_sort(cand.begin(), cand.begin() + keep, cand.end(), better);
    out.assign(cand.begin(), cand.begin() + keep);
    std::reverse(out.begin(), out.end());          // ascending, best last
  }

  *out_n = int(out.size());
  for (size_t i = 0; i < out.size(); i++) {
    out_scores[i] = out[i].score;
    out_src[i] = out[i].src;
    out_trg[i] = out[i].trg;
    out_cases[i] = out[i].cases;
    out_ctrls[i] = out[i].ctrls;
  }
  for (int r = 0; r < K; r++) out_null[r] = global_null[r];
  return 0;
}
