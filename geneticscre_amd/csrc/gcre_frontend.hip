// gcre_frontend.hip -- native versions of the input builders that sit either side of the hot path in the
// reference's R code (SURVEY.md §8f "next" rows):
//   * gcre_build_levels          the per-level join tables GWASPA builds (R/ProcessPaths.R:206-256,
//                                getUidsCountsLocations R/PathMethods.R:133-152, getRels3 src/wrapper.cpp:18-48)
//   * gcre_values_table          getValuesTable, R/Utils.R:137-159
//   * gcre_generate_perm_masks   getRandIndicesMat + getCaseORControl + setPermutedCases fused on the device
//                                (R/Utils.R:22-46, 246-262; src/join_base.cpp:85-125)
// None of this is on the scored path; it removes the K x n integer matrix (4 GB at BASELINE configs[3]) and the
// O(n*m^2) R table builder from the caller's side.
#include "../../include/gcre_hip.h"
#include "gcre_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace gcre {

// Permutation r keeps a uniformly random n_cases-subset of the patients as cases (what a uniform permutation of the
// labels induces, Utils.R:22-46 + 246-262); with strata the number of cases inside every stratum is preserved
// (Utils.R:8-13).  Selection sampling (Knuth, Algorithm S) per stratum: patient c becomes a case with probability
// need/remaining.  Random numbers are a pure function of (seed, r, c) -- splitmix64 finaliser -- so the CPU
// restatement in the tests reproduces every mask bit.
__host__ __device__ inline uint64_t gcre_mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__global__ void k_generate_masks(uint64_t seed, int K, int n, int n_strata, const int32_t* stratum, const uint32_t* cases_in,
                                 const uint32_t* size_of, uint32_t* work, int W32p, int Kpad, uint32_t* masks) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= K) return;
  uint32_t* need = work + (size_t)r * n_strata * 2;
  uint32_t* rem = need + n_strata;
  for (int s = 0; s < n_strata; s++) { need[s] = cases_in[s]; rem[s] = size_of[s]; }
  const uint64_t base = gcre_mix64(seed ^ (0x51ed270b7f3c9a1dull * (uint64_t)(r + 1)));
  uint32_t word = 0;
  for (int c = 0; c < n; c++) {
    const int s = stratum ? stratum[c] : 0;
    const uint64_t u = gcre_mix64(base + (uint64_t)c);
    // u * rem >> 64 is uniform on [0, rem) up to 2^-64
    const uint32_t pick = (uint32_t)(((unsigned __int128)u * rem[s]) >> 64);
    if (pick < need[s]) {
      word |= 1u << (c & 31);
      need[s]--;
    }
    rem[s]--;
    if ((c & 31) == 31 || c == n - 1) {
      masks[(size_t)(c >> 5) * Kpad + r] = word;
      word = 0;
    }
  }
  for (int k = (n + 31) >> 5; k < W32p; k++) masks[(size_t)k * Kpad + r] = 0;
}

hipError_t launch_generate_masks(uint64_t seed, int K, int n, int n_strata, const int32_t* stratum, const uint32_t* cases_in,
                                 const uint32_t* size_of, uint32_t* work, int W32p, int Kpad, uint32_t* masks,
                                 hipStream_t stream) {
  if (K == 0) return hipSuccess;
  hipLaunchKernelGGL(k_generate_masks, dim3((K + 63) / 64), dim3(64), 0, stream, seed, K, n, n_strata, stratum, cases_in,
                     size_of, work, W32p, Kpad, masks);
  return hipGetLastError();
}

// The caller's value table (nrow x ncol doubles, row- or column-major) -> the device's diagonal-major triangles:
// dvt[t(t+1)/2 + i] = table[i][t-i], -1 outside the table or beyond n patients (the reference pads its (n+1)^2 copy
// with -1, join_base.cpp:67-78); t32 = the same cell rounded to f32 with everything that can never win clamped to +0
// (method 1, methods.h:96-103); dmax = max(table[i][t-i], table[t-i][i]) with std::max semantics (method 2,
// compute_value_table_max, methods.h:110-118).  One block row per diagonal.
__global__ __launch_bounds__(256) void k_table_to_diag(const double* table, int nrow, int ncol, int col_major, int n, int TD,
                                                       double* dvt, float* t32, double* dmax) {
  const int t = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= TD || i > t) return;
  auto VT = [&](int r, int q) -> double {
    if (r > n || q > n || r >= nrow || q >= ncol) return -1.0;
    return col_major ? table[(size_t)q * nrow + r] : table[(size_t)r * ncol + q];
  };
  const size_t at = ((size_t)t * ((size_t)t + 1)) / 2 + (size_t)i;
  const double a = VT(i, t - i);
  dvt[at] = a;
  if (t32) {
    const float f = (float)a;
    t32[at] = (f > 0.0f) ? f : 0.0f;
  }
  if (dmax) {
    const double b = VT(t - i, i);
    dmax[at] = (a < b) ? b : a;
  }
}

hipError_t launch_table_to_diag(const double* table, int nrow, int ncol, int col_major, int n, int TD, double* dvt,
                                float* t32, double* dmax, hipStream_t stream) {
  const dim3 grid((unsigned)((TD + 255) / 256), (unsigned)TD);
  hipLaunchKernelGGL(k_table_to_diag, grid, dim3(256), 0, stream, table, nrow, ncol, col_major, n, TD, dvt, t32, dmax);
  return hipGetLastError();
}


// ---- R's stats::dhyper as nmath publishes it (R 3.x - 4.3: dhyper.c, dbinom.c, stirlerr.c, bd0.c; C. Loader, "Fast and
// accurate computation of binomial probabilities", 2000), restated for the integer arguments getValuesTable passes
// (R/Utils.R:144).  give_log = FALSE throughout.  Evaluation order follows the published expressions term by term: the
// result is meant to be the double R returns, not merely close to it.  No fused multiply-add (R's builds have none on
// x86-64's baseline).  Host only.
#pragma clang fp contract(off)
static double r_stirlerr(double n) {
  // log(n!) - log(sqrt(2 pi n) (n/e)^n); exact values for n = 0 .. 15 (stirlerr.c's sferr_halves at the integers)
  static const double sferr[16] = {
      0.0,
      0.0810614667953272582196702, 0.0413406959554092940938221, 0.02767792568499833914878929,
      0.02079067210376509311152277, 0.01664469118982119216319487, 0.01387612882307074799874573,
      0.01189670994589177009505572, 0.010411265261972096497478567, 0.009255462182712732917728637,
      0.008330563433362871256469318, 0.007573675487951840794972024, 0.006942840107209529865664152,
      0.006408994188004207068439631, 0.005951370112758847735624416, 0.005554733551962801371038690};
  constexpr double S0 = 0.083333333333333333333;          // 1/12
  constexpr double S1 = 0.00277777777777777777778;        // 1/360
  constexpr double S2 = 0.00079365079365079365079365;     // 1/1260
  constexpr double S3 = 0.000595238095238095238095238;    // 1/1680
  constexpr double S4 = 0.0008417508417508417508417508;   // 1/1188
  if (n <= 15.0) return sferr[(int)n];
  const double nn = n * n;
  if (n > 500) return (S0 - S1 / nn) / n;
  if (n > 80) return (S0 - (S1 - S2 / nn) / nn) / n;
  if (n > 35) return (S0 - (S1 - (S2 - S3 / nn) / nn) / nn) / n;
  return (S0 - (S1 - (S2 - (S3 - S4 / nn) / nn) / nn) / nn) / n;
}

static double r_bd0(double x, double np) {
  // x log(x/np) + np - x, by its Taylor series where x is close to np (bd0.c)
  if (std::fabs(x - np) < 0.1 * (x + np)) {
    double v = (x - np) / (x + np);
    double s = (x - np) * v;
    if (std::fabs(s) < std::numeric_limits<double>::min()) return s;
    double ej = 2 * x * v;
    v = v * v;
    for (int j = 1; j < 1000; j++) {
      ej *= v;
      const double s1 = s + ej / ((j << 1) + 1);
      if (s1 == s) return s1;
      s = s1;
    }
  }
  return x * std::log(x / np) + np - x;
}

static double r_dbinom_raw(double x, double n, double p, double q) {
  constexpr double kLn2Pi = 1.837877066409345483560659472811;   // M_LN_2PI
  if (p == 0) return x == 0 ? 1.0 : 0.0;
  if (q == 0) return x == n ? 1.0 : 0.0;
  if (x == 0) {
    if (n == 0) return 1.0;
    const double lc = (p < 0.1) ? -r_bd0(n, n * q) - n * p : n * std::log(q);
    return std::exp(lc);
  }
  if (x == n) {
    const double lc = (q < 0.1) ? -r_bd0(n, n * p) - n * q : n * std::log(p);
    return std::exp(lc);
  }
  if (x < 0 || x > n) return 0.0;
  const double lc = r_stirlerr(n) - r_stirlerr(x) - r_stirlerr(n - x) - r_bd0(x, n * p) - r_bd0(n - x, n * q);
  const double lf = kLn2Pi + std::log(x) + std::log1p(-x / n);
  return std::exp(lc - 0.5 * lf);
}

// dhyper(x, r, b, n): x white balls among n drawn from r white + b black (dhyper.c)
double r_dhyper(double x, double r, double b, double n) {
  if (x < 0) return 0.0;
  if (n < x || r < x || n - x > b) return 0.0;
  if (n == 0) return x == 0 ? 1.0 : 0.0;
  const double p = n / (r + b);
  const double q = (r + b - n) / (r + b);
  const double p1 = r_dbinom_raw(x, r, p, q);
  const double p2 = r_dbinom_raw(n - x, b, p, q);
  const double p3 = r_dbinom_raw(n, r + b, p, q);
  return p1 * p2 / p3;
}

}  // namespace gcre

namespace {

template <typename T>
T* dup(const std::vector<T>& v) {
  T* p = (T*)std::malloc(std::max<size_t>(v.size(), 1) * sizeof(T));
  if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

void fill_level(gcre_level_table* lt, int plen, const std::vector<int32_t>& src, const std::vector<int32_t>& trg,
                const std::vector<int32_t>& count, const std::vector<int64_t>& location, const std::vector<int32_t>& signs) {
  lt->path_length = plen;
  lt->n_uids = (int64_t)trg.size();
  lt->src = dup(src);
  lt->trg = dup(trg);
  lt->count = dup(count);
  lt->location = dup(location);
  lt->n_signs = (int64_t)signs.size();
  lt->signs = dup(signs);
  int64_t t = 0;
  for (int32_t c : count) t += std::max(c, 0);
  lt->total_paths = t;
}

}  // namespace

extern "C" {

int gcre_build_levels(int32_t n_genes, const int32_t* src, const int32_t* trg, const int32_t* sign, int64_t n_edges,
                      gcre_levels* out) {
  if (!out || n_genes < 0 || n_edges < 0 || (n_edges > 0 && (!src || !trg || !sign))) return GCRE_ERR_ARG;
  std::memset(out, 0, sizeof *out);
  // relations must arrive sorted by (src, trg), unique, without self loops (ProcessPaths.R:145-149, 210)
  for (int64_t e = 0; e < n_edges; e++) {
    if (src[e] < 0 || src[e] >= n_genes || trg[e] < 0 || trg[e] >= n_genes) return GCRE_ERR_RANGE;
    if (e > 0 && (src[e] < src[e - 1] || (src[e] == src[e - 1] && trg[e] <= trg[e - 1]))) return GCRE_ERR_ARG;
  }
  // run-length index of the sorted source column: first edge and out-degree per gene (getUidsCountsLocations)
  std::vector<int64_t> first((size_t)n_genes, -1);
  std::vector<int32_t> deg((size_t)n_genes, 0);
  for (int64_t e = 0; e < n_edges; e++) {
    if (deg[(size_t)src[e]]++ == 0) first[(size_t)src[e]] = e;
  }
  std::vector<int32_t> genes((size_t)n_genes), ones((size_t)n_genes, 1);
  std::vector<int64_t> self((size_t)n_genes);
  for (int32_t g = 0; g < n_genes; g++) { genes[(size_t)g] = g; self[(size_t)g] = g; }
  // level 1a: every gene joined with its own data row (ProcessPaths.R:214-218)
  fill_level(&out->level[0], 1, genes, genes, ones, self, ones);
  out->data_inds[0] = dup(genes);
  out->n_data_inds[0] = n_genes;
  // level 1b: genes that are the source of a relation (Ents2, ProcessPaths.R:153-155, 220-224)
  std::vector<int32_t> genes2;
  for (int32_t g = 0; g < n_genes; g++) if (deg[(size_t)g] > 0) genes2.push_back(g);
  {
    std::vector<int32_t> one2(genes2.size(), 1), idx2(genes2.size());
    std::vector<int64_t> loc2(genes2.size());
    for (size_t i = 0; i < genes2.size(); i++) { idx2[i] = (int32_t)i; loc2[i] = (int64_t)i; }
    fill_level(&out->level[1], 1, genes2, genes2, one2, loc2, one2);
    out->data_inds[1] = dup(idx2);
    out->n_data_inds[1] = (int64_t)idx2.size();
  }
  std::vector<int32_t> esrc(src, src + n_edges), etrg(trg, trg + n_edges), esign(sign, sign + n_edges);
  // level 2: gene g joined with the data of each of its targets (ProcessPaths.R:226-230); no relations -> (0, -1)
  {
    std::vector<int32_t> cnt((size_t)n_genes);
    std::vector<int64_t> loc((size_t)n_genes);
    for (int32_t g = 0; g < n_genes; g++) { cnt[(size_t)g] = deg[(size_t)g]; loc[(size_t)g] = first[(size_t)g]; }
    fill_level(&out->level[2], 2, genes, genes, cnt, loc, esign);
    out->data_inds[2] = dup(etrg);
    out->n_data_inds[2] = n_edges;
  }
  // level 3: edge a->b joined with the data of each target of b (ProcessPaths.R:232-236)
  std::vector<int32_t> c3((size_t)n_edges);
  std::vector<int64_t> l3((size_t)n_edges);
  int64_t p3 = 0;
  for (int64_t e = 0; e < n_edges; e++) {
    c3[(size_t)e] = deg[(size_t)trg[e]];
    l3[(size_t)e] = first[(size_t)trg[e]];
    p3 += c3[(size_t)e];
  }
  fill_level(&out->level[3], 3, esrc, etrg, c3, l3, esign);
  out->data_inds[3] = dup(etrg);
  out->n_data_inds[3] = n_edges;
  // Rels3 (getRels3, wrapper.cpp:18-48): one row per 2-edge walk a->b->c, grouped by the first edge
  std::vector<int32_t> r_src((size_t)p3), r_trg((size_t)p3), r_sign((size_t)p3), r_trg2((size_t)p3), r_sign2((size_t)p3),
      third((size_t)p3);
  {
    int64_t o = 0;
    for (int64_t e = 0; e < n_edges; e++)
      for (int64_t j = l3[(size_t)e]; j < l3[(size_t)e] + c3[(size_t)e]; j++, o++) {
        r_src[(size_t)o] = src[e];
        r_trg[(size_t)o] = trg[e];
        r_sign[(size_t)o] = sign[e];
        r_trg2[(size_t)o] = trg[j];
        r_sign2[(size_t)o] = sign[j];
        // sign of the third gene relative to a (+) first gene (ProcessPaths.R:243-245)
        third[(size_t)o] = (sign[e] * sign[j] == -1) ? -1 : 1;
      }
  }
  out->n_rels3 = p3;
  out->r3_src = dup(r_src);
  out->r3_trg = dup(r_trg);
  out->r3_sign = dup(r_sign);
  out->r3_trg2 = dup(r_trg2);
  out->r3_sign2 = dup(r_sign2);
  // level 4: 3-path a->b->c joined with the stored 2-paths c->d (ProcessPaths.R:247-250)
  {
    std::vector<int32_t> cnt((size_t)p3);
    std::vector<int64_t> loc((size_t)p3);
    for (int64_t i = 0; i < p3; i++) { cnt[(size_t)i] = deg[(size_t)r_trg2[(size_t)i]]; loc[(size_t)i] = first[(size_t)r_trg2[(size_t)i]]; }
    fill_level(&out->level[4], 4, r_src, r_trg2, cnt, loc, third);
  }
  // level 5: 3-path a->b->c joined with the stored 3-paths c->d->e: Rels3 rows whose first gene is c, contiguous
  // because Rels3 inherits the (src, trg) order (ProcessPaths.R:253-256)
  {
    std::vector<int64_t> first3((size_t)n_genes, -1);
    std::vector<int32_t> deg3((size_t)n_genes, 0);
    for (int64_t i = 0; i < p3; i++)
      if (deg3[(size_t)r_src[(size_t)i]]++ == 0) first3[(size_t)r_src[(size_t)i]] = i;
    std::vector<int32_t> cnt((size_t)p3);
    std::vector<int64_t> loc((size_t)p3);
    for (int64_t i = 0; i < p3; i++) { cnt[(size_t)i] = deg3[(size_t)r_trg2[(size_t)i]]; loc[(size_t)i] = first3[(size_t)r_trg2[(size_t)i]]; }
    fill_level(&out->level[5], 5, r_src, r_trg2, cnt, loc, third);
  }
  return GCRE_OK;
}

void gcre_levels_free(gcre_levels* lv) {
  if (!lv) return;
  for (auto& l : lv->level) {
    std::free(l.src); std::free(l.trg); std::free(l.count); std::free(l.location); std::free(l.signs);
  }
  for (auto* p : lv->data_inds) std::free(p);
  std::free(lv->r3_src); std::free(lv->r3_trg); std::free(lv->r3_sign); std::free(lv->r3_trg2); std::free(lv->r3_sign2);
  std::memset(lv, 0, sizeof *lv);
}

static double vt_work(int n_cases, int n_ctrls) {
  const int n = n_cases + n_ctrls;
  double work_est = 0;
  for (int i = 0; i <= n; i++) {
    const double m = (double)(std::min(i, n_cases) - std::max(0, i - n_ctrls) + 1);
    work_est += m * m;
  }
  return work_est;
}
static double vt_exact_work_limit() {
  double exact_work = 2.5e11;
  if (const char* e = std::getenv("GCRE_VT_EXACT_WORK")) exact_work = std::atof(e);   // tests: 0 forces the prefix-sum form
  return exact_work;
}

int gcre_values_table_exact_order(int n_cases, int n_ctrls) {
  if (n_cases < 0 || n_ctrls < 0) return GCRE_ERR_ARG;
  return vt_work(n_cases, n_ctrls) <= vt_exact_work_limit() ? 1 : 0;
}

int gcre_values_table(int n_cases, int n_ctrls, double* out) {
  // getValuesTable (Utils.R:137-159): out[x][i-x] = -log(two-sided hypergeometric p of x cases among i carriers);
  // two-sided p = sum(prob[prob <= prob_x]) with R's EXACT `<=` on the doubles stats::dhyper returns (:153);
  // infinities -> max finite + 1 (:156).  dhyper is restated as R's nmath publishes it (gcre::r_dhyper), and the sum is
  // R's: a long double accumulator over the diagonal in index order, rounded to double once (R's rsum).  So the table
  // is the one R builds wherever libm's log / log1p / exp agree -- R is not in this image, the row stays unpinned, but
  // no rule of ours (the 1e-12 tie slack of the earlier builder) separates it from R's any more.
  // Cost: R's sapply is O(m^2) per diagonal; so is the index-order sum.  Up to 2.5e11 inner steps (n ~ 14,000
  // patients: ~15 s on 16 threads) every cell is summed exactly like R does; beyond that (configs[4]'s 50,000 patients,
  // where R itself would need days) the qualifying outcomes are summed in ascending order of probability through a
  // long double prefix sum: the same set of outcomes (exact `<=`), the same 64-bit accumulator, another order of
  // additions -- the rounded double can differ in its last bit for a cell in a hundred.
  if (n_cases < 0 || n_ctrls < 0 || !out) return GCRE_ERR_ARG;
  const int n = n_cases + n_ctrls;
  const size_t cols = (size_t)n_ctrls + 1;
  const bool exact_order = vt_work(n_cases, n_ctrls) <= vt_exact_work_limit();   // (gcre_values_table_exact_order says which)
  int T = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (n < 256) T = 1;
  std::vector<double> tmax((size_t)T, -std::numeric_limits<double>::infinity());
  auto work = [&](int t) {
    std::vector<double> prob, sorted;
    std::vector<long double> csum;
    double max_finite = -std::numeric_limits<double>::infinity();
    // diagonals dealt round the threads: the long ones (the middle of the table) alternate between them
    for (int i = t; i <= n; i += T) {
      const int lo = std::max(0, i - n_ctrls), hi = std::min(i, n_cases);
      const size_t m = (size_t)(hi - lo + 1);
      prob.resize(m);
      for (size_t k = 0; k < m; k++) prob[k] = gcre::r_dhyper((double)(lo + (int)k), (double)n_cases, (double)n_ctrls, (double)i);
      if (!exact_order) {
        sorted = prob;
        std::sort(sorted.begin(), sorted.end());
        csum.resize(m);
        long double run = 0.0L;
        for (size_t k = 0; k < m; k++) { run += (long double)sorted[k]; csum[k] = run; }
      }
      for (size_t k = 0; k < m; k++) {
        const double x = prob[k];
        double p_two;
        if (exact_order) {
          long double acc = 0.0L;
          for (size_t j = 0; j < m; j++) acc += (prob[j] <= x) ? (long double)prob[j] : 0.0L;   // + 0 leaves acc as it is
          p_two = (double)acc;
        } else {
          const size_t upto = (size_t)(std::upper_bound(sorted.begin(), sorted.end(), x) - sorted.begin());
          p_two = (double)csum[upto - 1];
        }
        const double v = -std::log(p_two);
        out[(size_t)(lo + (int)k) * cols + (size_t)(i - lo - (int)k)] = v;
        if (std::isfinite(v)) max_finite = std::max(max_finite, v);
      }
    }
    tmax[(size_t)t] = max_finite;
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
  }
  double max_finite = -std::numeric_limits<double>::infinity();
  for (double v : tmax) max_finite = std::max(max_finite, v);
  for (size_t k = 0; k < ((size_t)n_cases + 1) * cols; k++)
    if (!std::isfinite(out[k])) out[k] = max_finite + 1.0;
  return GCRE_OK;
}

uint64_t gcre_mix64(uint64_t z) { return gcre::gcre_mix64(z); }

}  // extern "C"
