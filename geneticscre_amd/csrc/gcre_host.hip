// gcre_host.hip -- host side of libgcre_hip.so: the C ABI of include/gcre_hip.h on top of the gfx950
// kernels in gcre_kernels.hip.  Owns device memory, the stream, the join driver (JoinExec::join,
// reference src/join_base.cpp:189-264), the top-k merge (merge_scores / format_result, methods.h:25-39,
// join_base.cpp:138-154) and the ProcessPaths sequence (src/wrapper.cpp:216-276).
//
// There is no CPU fallback: without a gfx950 device gcre_create fails with GCRE_ERR_DEVICE.
#include "../../include/gcre_hip.h"
#include "gcre_kernels.h"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen'ed where several devices are used (RcclApi)

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <string>
#include <unordered_map>
#include <thread>
#include <functional>
#include <deque>
#include <vector>

using namespace gcre;

namespace {

thread_local std::string g_create_error;

// GCRE_POISON (diagnostics): count-plane buffers handed to a set start as 0x5A bytes instead of whatever was there, so
// that a read of rows nobody wrote shows in the results of a fresh process too (planes are data, never indices)
inline bool poison_fresh_planes() {
  static const bool on = std::getenv("GCRE_POISON") != nullptr && std::atoi(std::getenv("GCRE_POISON")) != 0;
  return on;
}
inline void poison_planes(void* p, size_t bytes) {
  if (!poison_fresh_planes() || !p) return;
  (void)hipDeviceSynchronize();
  (void)hipMemset(p, 0x5A, bytes);
  (void)hipDeviceSynchronize();
}

template <typename T>
struct DevBuf {   // grow-only device scratch
  T* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  // grow and keep the first `keep` elements
  hipError_t grow_keep(size_t n, size_t keep, hipStream_t stream) {
    if (n <= cap) return hipSuccess;
    T* q = nullptr;
    hipError_t e = hipMalloc((void**)&q, n * sizeof(T));
    if (e != hipSuccess) return e;
    if (p && keep) {
      e = hipMemcpyAsync(q, p, std::min(keep, cap) * sizeof(T), hipMemcpyDeviceToDevice, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    if (p) (void)hipFree(p);
    p = q;
    cap = n;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct Candidate {
  double score;
  int64_t path;   // absolute joined-path ordinal
  int32_t src, trg, cases, ctrls;
};

// top-k of a chunk: indices chosen by the radix select, their keys / counts / rows gathered and copied out
struct Winners {
  std::vector<uint32_t> sel, cases, ctrls, r0, r1;
  std::vector<uint64_t> key;
  uint32_t n = 0;
};

// What the inspector of one chunk of a join left behind -- expanded row numbers, statistics, score keys, lists, flags
// and the chunk's top-k winners.  None of it depends on the permutation masks: with the inspection cache on
// (gcre_set_inspect_cache) the buffers belong to the join index instead of the context's scratch, and the next
// permutation window of the same join starts at the null kernel.
struct ChunkInsp {
  int64_t cb = -1, n = 0, s0 = 0, s1 = 0;
  int64_t padded = 0;         // rows / totals are zero up to here (whole path tiles of the dense kernel)
  bool inspected = false;     // rows / statistics / keys (and kept rows) are those of this chunk
  bool with_lists = false;    // ... written by the inclusion-exclusion inspector: lists, rowz, linfo
  bool in_recipe = false;     // ... into the kept set's recipe (not into the buffers below)
  bool flags_valid = false;   // host copy of the inspector's flag block
  bool win_valid = false;     // top-k winners
  uint32_t flags[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  Winners win;
  DevBuf<uint32_t> row0, row1, tot, cases, ctrls, dcnt, dlist, rowz, linfo, lover, dover;
  DevBuf<uint64_t> key;
  void release() {
    for (auto* b : {&row0, &row1, &tot, &cases, &ctrls, &dcnt, &dlist, &rowz, &linfo, &lover, &dover}) b->release();
    key.release();
    inspected = with_lists = flags_valid = win_valid = false;
  }
};

struct InspKey {
  uint64_t p0_id = 0, p0_ver = 0, p1_id = 0, p1_ver = 0, red_id = 0, red_ver = 0, res_id = 0, obs_epoch = 0;
  int64_t sb = 0, se = 0, keep_begin = 0, keep_end = 0, chunk_paths = 0;
  int keep_mode = 0, top_k = 0, null_kernel = 0;
  bool operator==(const InspKey& o) const {
    return p0_id == o.p0_id && p0_ver == o.p0_ver && p1_id == o.p1_id && p1_ver == o.p1_ver && red_id == o.red_id &&
           red_ver == o.red_ver && res_id == o.res_id && obs_epoch == o.obs_epoch && sb == o.sb && se == o.se &&
           keep_begin == o.keep_begin && keep_end == o.keep_end && chunk_paths == o.chunk_paths && keep_mode == o.keep_mode &&
           top_k == o.top_k && null_kernel == o.null_kernel;
  }
};

inline double key_to_score(uint64_t k) {
  const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double d;
  std::memcpy(&d, &b, sizeof d);
  return d;
}

inline size_t tri(size_t t) { return t * (t + 1) / 2; }

// Dense int matrix -> packed bit rows ON THE HOST, before anything crosses PCIe (round 4).  What R hands the shim is hundreds
// of MB of 4-byte ints that carry one bit each (340 MB of genotypes twice and 200 MB of permutation labels at configs[2]):
// pageable host memory goes up at ~10 GB/s, so uploading the ints and packing them on the device cost the one-shot call more
// than its kernels.  Packed here they are 1/32 of that.  out[r][w] (stride `stride` words) gets bit b of word w set iff
// pred(value of (r, 64 w + b), 64 w + b).  Threads own whole word columns (column-major input: 64 consecutive columns are
// streamed once each, the word column is accumulated in a buffer that stays in cache) or row blocks (row-major input).
template <typename Pred>
void pack_bits_host(const int32_t* data, int64_t nrow, int ncol, bool col_major, uint64_t* out, size_t stride, Pred pred) {
  const int W = (ncol + 63) / 64;
  if (nrow <= 0 || W <= 0) return;
  unsigned T = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = std::getenv("GCRE_PACK_THREADS")) T = (unsigned)std::max(1, std::atoi(e));
  if ((double)nrow * ncol < 4e6) T = 1;
  auto work = [&](unsigned t) {
    if (col_major) {
      std::vector<uint64_t> acc((size_t)nrow);
      for (int w = (int)t; w < W; w += (int)T) {
        std::fill(acc.begin(), acc.end(), 0);
        const int q1 = std::min(ncol, (w + 1) * 64);
        for (int q = w * 64; q < q1; q++) {
          const int32_t* col = data + (size_t)q * (size_t)nrow;
          const uint64_t bit = uint64_t(1) << (q & 63);
          for (int64_t r = 0; r < nrow; r++) acc[(size_t)r] |= pred(col[r], q) ? bit : 0;
        }
        for (int64_t r = 0; r < nrow; r++) out[(size_t)r * stride + (size_t)w] = acc[(size_t)r];
      }
    } else {
      const int64_t r0 = nrow * t / T, r1 = nrow * (t + 1) / T;
      for (int64_t r = r0; r < r1; r++) {
        const int32_t* row = data + (size_t)r * (size_t)ncol;
        for (int w = 0; w < W; w++) {
          uint64_t v = 0;
          const int q1 = std::min(ncol, (w + 1) * 64);
          for (int q = w * 64; q < q1; q++) v |= pred(row[q], q) ? uint64_t(1) << (q & 63) : 0;
          out[(size_t)r * stride + (size_t)w] = v;
        }
      }
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < T; t++) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
}

// GCRE_HOST_TIMING=1: wall time of the host-side steps around the kernels (stderr)
struct HostTimer {
  const char* what;
  std::chrono::steady_clock::time_point t0;
  explicit HostTimer(const char* w) : what(w), t0(std::chrono::steady_clock::now()) {}
  ~HostTimer() {
    static const bool on = std::getenv("GCRE_HOST_TIMING") != nullptr;
    if (on) std::fprintf(stderr, "[host] %-28s %8.2f ms\n", what,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
};

}  // namespace

// RCCL, loaded on first use (gcre_process_paths_devices with several distinct devices, gcre_rccl_selftest): the library
// does not link librccl, so a one-GPU user -- the R drop-in's default -- never needs it.  north_star: "RCCL all-reduce over
// xGMI of the per-permutation null maxima": ncclAllReduce(ncclMax) on each device's stream, in place on the device, for
// the thresholds shared inside a join and for the per-level merge; the host hub below stays the fallback (RCCL missing,
// a device listed twice) and the place where the device threads meet under a deadline before every collective.
std::atomic<int64_t> g_rccl_collectives{0};   // RCCL collectives issued by this process (gcre_rccl_collectives)

struct RcclApi {
  void* so = nullptr;
  decltype(&ncclCommInitAll) comm_init_all = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  bool ok = false;
  static RcclApi& get() {
    static RcclApi api = [] {
      RcclApi a;
      const char* off = std::getenv("GCRE_RCCL");
      if (off && std::strcmp(off, "0") == 0) return a;   // GCRE_RCCL=0: host hub only
      for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        a.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (a.so) break;
      }
      if (!a.so) return a;
      a.comm_init_all = (decltype(a.comm_init_all))dlsym(a.so, "ncclCommInitAll");
      a.comm_destroy = (decltype(a.comm_destroy))dlsym(a.so, "ncclCommDestroy");
      a.all_reduce = (decltype(a.all_reduce))dlsym(a.so, "ncclAllReduce");
      a.error_string = (decltype(a.error_string))dlsym(a.so, "ncclGetErrorString");
      a.ok = a.comm_init_all && a.comm_destroy && a.all_reduce && a.error_string;
      return a;
    }();
    return api;
  }
};

// gcre_process_paths_devices: the device threads of one call meet here to MAX-merge their running null maxima during a
// join (gcre_join_opts.exchange, served inside the library).  K floats per call: the host does the reduction.  With RCCL
// the data stays on the devices and only the meeting (`meet`) happens here.
struct ExchangeHub {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t gen = 0;
  bool failed = false, timed_out = false;
  double timeout_s = 300.0;
  std::vector<float> acc, result;
  uint64_t round_tag = 0;
  // tag = (level, permutation window, ordinal of the exchange inside the join): every device of a round must bring the same
  // one -- a device that skipped or repeated an exchange would otherwise MAX another level's maxima into the thresholds
  int reduce(std::vector<float>& mine, uint64_t tag) {   // in: this device's maxima; out: the MAX over all devices
    std::unique_lock<std::mutex> lk(m);
    if (failed) return 1;
    if (arrived == 0) {
      acc = mine;
      round_tag = tag;
    } else {
      if (acc.size() != mine.size() || tag != round_tag) { failed = true; cv.notify_all(); return 1; }
      for (size_t i = 0; i < mine.size(); i++) acc[i] = std::max(acc[i], mine[i]);
    }
    if (++arrived == n) {
      result.swap(acc);
      arrived = 0;
      gen++;
      cv.notify_all();
    } else {
      // a device that never arrives (its thread died, it took another road through the join) must not hold the others for
      // ever: past the deadline the round -- and with it the call -- fails (GCRE_HUB_TIMEOUT_S, default 300 s: a join of
      // configs[4] on a shared GPU takes seconds)
      const uint64_t g = gen;
      if (!cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return gen != g || failed; })) {
        failed = true;
        timed_out = true;
        cv.notify_all();
        return 1;
      }
      if (gen == g) return 1;   // somebody failed before this round completed
    }
    mine = result;
    return 0;
  }
  // every device arrives with the same tag or the round fails; no data (the collective that follows moves it)
  int meet(uint64_t tag) {
    std::vector<float> none;
    return reduce(none, tag);
  }
  void fail() {
    std::lock_guard<std::mutex> lk(m);
    failed = true;
    cv.notify_all();
  }
};

namespace { struct JoinPlan; }

struct gcre_ctx {
  Geometry g{};
  int device = 0;
  int top_k = 12;   // JoinExec::top_k, gcre.h:120
  hipStream_t stream = nullptr;
  // the top-k selection of a chunk only reads the keys its inspector wrote: it runs beside the warm-up slice and the
  // null kernel on a stream of its own (GCRE_SELECT_STREAM=0: on the main stream, as in round 1)
  hipStream_t sel_stream = nullptr;
  hipEvent_t ev_sel = nullptr, ev_sel_done = nullptr;
  bool sel_async = true;
  // Inspect-ahead (gcre_join_ahead, round 4).  The inspector of the NEXT join of a sequence only needs what the current
  // join's inspector wrote (kept rows, recipe) -- not its permutation kernel -- so it runs on a stream of its own while
  // that kernel is in flight, into the next join's inspection cache; the next join then starts at its null kernel.
  hipStream_t insp_stream = nullptr;
  hipEvent_t ev_insp_done = nullptr;    // recorded on insp_stream when an ahead inspection has queued all its work
  hipEvent_t ev_insp_main = nullptr;    // recorded on the main stream behind the last inspector that ran there
  hipEvent_t ev_tail = nullptr;         // behind a join's own result copies, before the chain it launches
  uint32_t* d_max_tot_b = nullptr;      // the flag block of ahead inspections (the null kernel in flight owns d_max_tot)
  std::vector<JoinPlan>* ahead = nullptr;   // the registered later joins of the sequence, consumed by the next join call
  bool ahead_closed = false;            // a join that cannot run ahead was offered: nothing behind it is registered either
  bool ahead_on = true;                 // GCRE_AHEAD=0 turns gcre_join_ahead into a no-op.  The caller decides which joins to register:
                                        // measured, the chain is worth 8 % on configs[1] (host gaps between small joins) and
                                        // 0.5 % on configs[2] -- there the next level's inspector and this level's permutation
                                        // kernel each fill the GPU (26.1 + 9.9 ms of kernel time inside 31.1 ms instead of
                                        // 23.0 + 6.0 one after the other)
  SelectState h_sel{};               // where the digit passes' state lands (outlives any one chunk: the copy is asynchronous)
  std::string err;
  int last_code = GCRE_OK;
  bool quiet = false;
  bool have_table = false, have_perms = false;
  int64_t chunk_paths = int64_t(1) << 25;
  int null_blocks_per_cu = 12;

  // resident inputs
  uint64_t* d_case_mask = nullptr;   // [Wp]
  uint32_t* d_masks = nullptr;       // [W32p][Kpad]
  float* d_t32 = nullptr;            // method 1 null table
  double* d_dvt = nullptr;           // observed-score table
  double* d_dmax = nullptr;          // method 2 null table (vtmax)
  uint32_t* d_null = nullptr;        // [Kpad]
  uint32_t* d_mt = nullptr;          // transposed masks for the sparse kernel [nkt][64*Wp + 1][64]
  int ieq_batch = 0;                 // GCRE_IEQ_BATCH: quads per ticket of the quad kernel (0: twice ie_batch)
  int ie_quad = 1;                   // GCRE_IE_QUAD=0: the pruned method-1 launches stay on k_null_ie_m1 (cross-check)
  int ie_warm_items = 4;             // (segment, tile) items per wave of the warm-up launch (GCRE_IE_WARM_ITEMS)
  int exchange_tail = 0;             // slices of the pruned launch that are equal steps at its end (GCRE_EXCHANGE_TAIL; -1: half of them; 0: doubling slices only)
  int ie_warm_segs = 1024;           // least number of segments in the warm-up slice (GCRE_IE_WARM; 2048 until round 3: the filter's second look made early thresholds matter less)
  int ie_small_join_tiles = 8;       // GCRE_IE_SJT (tuning)
  int ie_batch = 2;                  // segments per ticket (GCRE_IE_BATCH)
  uint32_t* d_queue = nullptr;       // ticket counters of the pruned kernels' work queues (8 x 16 words)
  uint32_t* d_max_tot = nullptr;     // 8 words: largest carrier total of the chunk, "reduced operand is wrong", overlap lists,
                                     // looked-up tiles, entries reserved in the long-list area
  uint32_t* d_ladder = nullptr;      // method 1: pruning ladder of the null table [kLadderLevels][TD]
  uint32_t g00_rows = 0xffffffffu;   // method 2: vtmax[0][0] in ladder rows, rounded up (IeArgs::g00_rows)
  int null_kernel = 0;               // 0 auto, 1 dense, 2 sparse, 3 ie (GCRE_NULL_KERNEL)
  int sparse_waves_per_cu = 32;
  int ie_prune = 1;                  // GCRE_IE_PRUNE=0 looks every count up (diagnostics)
  uint64_t mask_epoch = 0;           // bumped whenever the permutation masks change: count planes are per epoch
  uint64_t obs_epoch = 0;            // bumped whenever the value table changes: observed scores (keys, winners) are per epoch
  bool insp_cache = false;           // gcre_set_inspect_cache: a join's inspector output stays with its join index
  ExchangeHub* hub = nullptr;        // set by gcre_process_paths_devices for the duration of a call
  ncclComm_t comm = nullptr;         // ... and this device's RCCL communicator when the devices are distinct and RCCL loads
  int64_t rccl_calls = 0;            // collectives this context issued during the call (diagnostics, tests)
  DevBuf<float> d_hub_null;          // the maxima this device hands to the hub
  int hub_level = 0, hub_round = 0;  // what the next exchange of this device is: part of the hub's round tag
  // permutation window [win_k0, win_k0 + win_K): what a join scores.  The whole range by default; gcre_set_perm_window
  // narrows it so that the count planes of the kept sets (one per 2048-permutation tile) fit in device memory
  int win_k0 = 0, win_K = 0;
  int win_K_nominal = 0;   // the largest window since the masks were set: whether a kept set leaves with planes or with a recipe
                           // is decided for THAT size, so that a short last window does not flip the decision (and hipMalloc
                           // gigabytes of planes for one window: 120-460 ms on a fresh context)

  // per-join scratch
  DevBuf<uint32_t> d_row0, d_row1, d_tot, d_cases, d_ctrls, d_sel, d_small, d_chunk, d_rec_segs;
  DevBuf<uint64_t> d_key, d_wkey, d_doff, d_scan, d_excess, d_excess_b;
  DevBuf<uint32_t> d_dcnt, d_dlist, d_rowz, d_linfo, d_lover, d_dover;
  DevBuf<uint32_t> d_wcases, d_wctrls, d_wrow0, d_wrow1;

  // count-plane buffers of freed path sets, kept for the next set that needs one (hipMalloc of tens of GB costs
  // ~40 ms per GB on this platform, hipFree nothing)
  struct PlaneBuf { uint32_t* p; size_t bytes; };
  std::vector<PlaneBuf> plane_pool;
  // live path sets by id: a recipe names its operands by id + version, never by pointer alone
  std::unordered_map<uint64_t, const gcre_pathset*> live_sets;
  std::vector<gcre_uids*> live_uids;   // join indices created on this context (gcre_destroy releases what is still alive)
  uint64_t next_set_id = 0;
  size_t planes_out_max = (size_t)8 << 30;   // kept sets (method 1) whose planes are larger keep a recipe only

  gcre_profile prof{};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_null, ev_stats;
  std::vector<hipEvent_t> ev_pool;
};

// The context's per-join scratch and a chunk's cached buffers trade places while the chunk is worked on: every kernel
// argument keeps naming c->d_*; what the inspector writes ends up owned by the join index.
struct InspSwap {
  gcre_ctx* c;
  ChunkInsp* e;
  InspSwap(gcre_ctx* c_, ChunkInsp* e_) : c(c_), e(e_) { swap(); }
  ~InspSwap() { swap(); }
  InspSwap(const InspSwap&) = delete;
  InspSwap& operator=(const InspSwap&) = delete;
  void swap() {
    if (!e) return;
    std::swap(c->d_row0, e->row0);
    std::swap(c->d_row1, e->row1);
    std::swap(c->d_tot, e->tot);
    std::swap(c->d_cases, e->cases);
    std::swap(c->d_ctrls, e->ctrls);
    std::swap(c->d_key, e->key);
    std::swap(c->d_dcnt, e->dcnt);
    std::swap(c->d_dlist, e->dlist);
    std::swap(c->d_rowz, e->rowz);
    std::swap(c->d_linfo, e->linfo);
    std::swap(c->d_lover, e->lover);
    std::swap(c->d_dover, e->dover);
  }
};

// How a kept path set was made: row r = row row0[r] of set A | row rowz[r] of set Z, with the producing
// join's list (the overlap of the two rows, or what Z adds) as its inspector left it.  Enough to rebuild the count
// planes of any row for any permutation tile from the planes of A and Z, so the set's own planes (3 KB per row and
// tile) need not be stored, written or read.  Independent of the masks.
struct gcre_recipe {
  uint64_t a_id = 0, z_id = 0;     // the operands: path-set ids and the versions of their rows
  uint64_t a_ver = 0, z_ver = 0;
  DevBuf<uint32_t> row0, rowz, linfo, lover, slot, over, tot;   // tot: carriers of every kept row
  uint32_t max_len = 0;            // longest list (padded) the producing join's inspector wrote
  bool valid = false;
  void release() {
    for (auto* b : {&row0, &rowz, &linfo, &lover, &slot, &over, &tot}) b->release();
    valid = false;
  }
};

struct gcre_pathset {
  gcre_ctx* ctx;
  int64_t nrows;
  uint64_t* d_rows;   // max(nrows,1) x S words
  uint64_t id = 0;                 // never reused inside a context
  mutable uint64_t version = 0;    // bumped when the rows are rewritten
  mutable gcre_recipe* rec = nullptr;
  // CSR bit lists for the sparse kernel, built on first use and dropped whenever the rows are rewritten
  mutable uint64_t* d_loff = nullptr;
  mutable uint32_t* d_lidx = nullptr;
  mutable std::vector<uint64_t> h_loff;   // host copy of the offsets (sizes the work of a sparse launch)
  mutable uint32_t max_bits = 0;          // longest list (entries incl. padding): bounds the carriers of any row
  mutable bool max_known = false;
  mutable int one_sided = -1;             // method 2: every row has an empty half (-1: not looked at; from h_loff)
  // count planes for the inclusion-exclusion kernel: [tile][row*M+h][groups][64][4] dwords, valid for one mask epoch
  mutable uint32_t* d_planes = nullptr;
  mutable int plane_groups = 0;
  mutable size_t planes_bytes = 0;   // capacity of d_planes
  mutable uint64_t planes_epoch = 0;
  mutable bool planes_valid = false;
  mutable int64_t planes_lo = 0, planes_hi = 0;   // rows whose planes are valid (a multi-device join fills a range)
  mutable bool planes_wanted = false;   // a later join had to rebuild this set's planes from its bit lists: next time the
                                        // join that writes its rows leaves the planes too, whatever their size
};

// UidRelSet (src/gcre.h:49-90) resident on the device: prefix sums of count, locations, signs
struct gcre_uids {
  gcre_ctx* ctx;
  int path_length;
  int64_t n_uids;
  int64_t n_signs;
  int64_t total;      // count_total_paths()
  int64_t max_loc;    // largest paths1 row referenced, -1 if none
  int64_t max_idx;    // largest uid row with count > 0, -1 if none
  int64_t* d_path_idx;
  int64_t* d_location;
  int32_t* d_signs;
  std::vector<int64_t> h_path_idx;   // host copy, for building the sparse kernel's segment tables
  mutable std::vector<int64_t> h_nonempty;   // prefix count of the uids with count > 0 (built on first use)
  std::vector<int64_t> h_location;   // host copy: segments are ordered by the paths1 rows they join (L2 reuse of their planes)
  struct SegCache {
    int64_t first, count, score_b, score_e, plane_b, plane_e;
    int64_t nsegs, nscored;
    SparseSeg* d_segs;
    // quad table of the pruned method-1 kernel (gcre_ieq.hip), built on first use for one warm-up length: runs of up to
    // four consecutive segments that join the same paths1 rows, none straddling `q_warm` or `nscored`
    std::vector<SparseSeg> h_segs;
    int64_t q_warm = -1, nquads = 0, quad_begin = 0;
    uint32_t* d_quads = nullptr;
  };
  mutable std::vector<SegCache> seg_cache;
  // A segment table (and its quads) built ahead of the join that will ask for it, by a helper thread that touches nothing
  // but the host copies of the join index (gcre_process_paths: the last level's tables while the first levels run)
  struct Prefetch {
    int64_t first = 0, count = 0, score_b = 0, score_e = 0, plane_b = 0, plane_e = 0;
    std::vector<SparseSeg> segs;
    int64_t nscored = 0;
    std::vector<uint32_t> quads;
    int64_t q_warm = -1, quad_begin = 0;
    bool ready = false, quads_ready = false, quads_for_table = false;
    std::thread th;
  };
  mutable std::unique_ptr<Prefetch> prefetch;
  // inspection cache (gcre_set_inspect_cache): the inspector output of the last join that ran on this index, per chunk,
  // valid while the operands' rows, the kept set, the shard and the observed-score inputs are the same
  mutable std::deque<ChunkInsp> insp;
  mutable InspKey insp_key;
  mutable bool insp_valid = false;       // the join completed: every chunk entry describes it
  mutable bool insp_hinted = false;      // ... with the reduced operand standing (the hint was not broken)
  mutable uint64_t insp_res_ver = 0;     // version of the kept set's rows as that join left them
  // A join on this index whose permutation kernels were LAUNCHED ahead (gcre_join_ahead chain): they write into the index's
  // own maxima, the winners are the inspection cache's; the join call that comes for it only waits, copies and merges
  struct Launched {
    bool active = false;
    InspKey key;
    uint64_t res_ver = 0, mask_epoch = 0;
    int win_k0 = 0, win_K = 0;
    std::vector<Candidate> cands;
    DevBuf<uint32_t> d_null;              // Kpad running maxima + one word: the launch's look-up counter
    hipEvent_t done = nullptr;            // behind the last kernel of the launch (main stream)
    gcre_profile prof{};                  // what the ahead inspection and the launch accumulated for this join
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_null, ev_stats;
  };
  mutable Launched launch;
  // optional hint (gcre_uids_set_reduced): paths0[idx] | paths1[loc] == paths0[idx] | red[red_index[loc]] for every
  // joined path; checked on the device for every join, ignored when it does not hold
  const gcre_pathset* red = nullptr;
  uint64_t red_id = 0;             // the set's id: a freed operand is noticed, not dereferenced
  int32_t* d_red_index = nullptr;
  int64_t n_red_index = 0;
  // distinct (location, count) ranges of the uids (all uids with the same pivot gene share one): built on first use
  mutable int64_t n_ranges = -1;
  mutable int32_t* d_range_of = nullptr;    // uid -> range
  mutable int64_t n_pairs = 0;              // (range, paths1 row) pairs = sum of the range lengths
  mutable int32_t* d_pair_range = nullptr;
  mutable int64_t* d_pair_loc = nullptr;
};

namespace {

int fail(gcre_ctx* c, int code, const std::string& msg) {
  if (c) {
    c->err = msg;
    c->last_code = code;
  } else {
    g_create_error = msg;
  }
  return code;
}

#define HIP_TRY(ctx, expr)                                                                            \
  do {                                                                                                \
    hipError_t e__ = (expr);                                                                          \
    if (e__ != hipSuccess)                                                                            \
      return fail((ctx), GCRE_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));        \
  } while (0)

hipEvent_t get_event(gcre_ctx* c) {
  if (!c->ev_pool.empty()) {
    hipEvent_t e = c->ev_pool.back();
    c->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

double drain_events(gcre_ctx* c, std::vector<std::pair<hipEvent_t, hipEvent_t>>& v) {
  double ms = 0;
  for (auto& pr : v) {
    float t = 0;
    if (hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) ms += t;
    c->ev_pool.push_back(pr.first);
    c->ev_pool.push_back(pr.second);
  }
  v.clear();
  return ms;
}

gcre_pathset* new_pathset(gcre_ctx* c, int64_t nrows, bool zero) {
  HostTimer ht("new_pathset");
  if (nrows < 0) {
    fail(c, GCRE_ERR_ARG, "negative path-set size");
    return nullptr;
  }
  // rows are addressed by 31-bit row numbers and 32-bit offsets in 32-byte units inside the null kernel
  const int64_t units = nrows * (int64_t)(c->g.S / 4);
  if (nrows >= (int64_t(1) << 31) || units >= (int64_t(1) << 32)) {
    fail(c, GCRE_ERR_RANGE, "path set too large for 32-bit row addressing");
    return nullptr;
  }
  auto* ps = new gcre_pathset{};
  ps->ctx = c;
  ps->nrows = nrows;
  ps->id = ++c->next_set_id;
  const size_t bytes = (size_t)std::max<int64_t>(nrows, 1) * c->g.S * sizeof(uint64_t);
  hipError_t me = hipMalloc((void**)&ps->d_rows, bytes);
  if (me != hipSuccess && !c->plane_pool.empty()) {   // pooled plane buffers are only a convenience
    (void)hipGetLastError();
    for (auto& pb : c->plane_pool) (void)hipFree(pb.p);
    c->plane_pool.clear();
    me = hipMalloc((void**)&ps->d_rows, bytes);
  }
  if (me != hipSuccess) {
    fail(c, GCRE_ERR_DEVICE, "hipMalloc failed for a path set of " + std::to_string(bytes) + " bytes");
    delete ps;
    return nullptr;
  }
  if (zero || nrows == 0) {
    if (hipMemsetAsync(ps->d_rows, 0, bytes, c->stream) != hipSuccess) {
      fail(c, GCRE_ERR_DEVICE, "hipMemsetAsync failed");
      (void)hipFree(ps->d_rows);
      delete ps;
      return nullptr;
    }
  }
  c->live_sets[ps->id] = ps;
  return ps;
}

// The sparse kernel needs counts that fit 16 bits (64*Wp < 65535 patients); GCRE_NULL_KERNEL=dense turns it off.
bool sparse_enabled(const gcre_ctx* c) {
  return c->null_kernel != 1 && c->g.K > 0 && 64 * c->g.Wp < 65535;
}

int build_transposed_masks(gcre_ctx* c) {
  if (!sparse_enabled(c)) return GCRE_OK;
  const Geometry& g = c->g;
  const int nkt = (g.K + kSparseTile - 1) / kSparseTile;
  const uint32_t mt_rows = (uint32_t)(64 * g.Wp + 1);
  const size_t bytes = (size_t)nkt * mt_rows * 64 * 4;
  if (!c->d_mt) HIP_TRY(c, hipMalloc((void**)&c->d_mt, bytes));
  HIP_TRY(c, hipMemsetAsync(c->d_mt, 0, bytes, c->stream));
  HIP_TRY(c, launch_build_mt(c->d_masks, 2 * g.Wp, g.Kpad, nkt, mt_rows, c->d_mt, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->mask_epoch++;   // every count plane built so far belongs to the old masks
  c->win_k0 = 0;
  c->win_K = g.K;
  c->win_K_nominal = 0;
  return GCRE_OK;
}

void drop_planes(const gcre_pathset* ps) {
  if (ps->d_planes) {
    gcre_ctx* c = ps->ctx;
    if (c) {
      c->plane_pool.push_back({ps->d_planes, ps->planes_bytes});
      if (c->plane_pool.size() > 8) {   // keep the eight largest
        size_t small = 0;
        for (size_t k = 1; k < c->plane_pool.size(); k++)
          if (c->plane_pool[k].bytes < c->plane_pool[small].bytes) small = k;
        (void)hipFree(c->plane_pool[small].p);
        c->plane_pool.erase(c->plane_pool.begin() + (long)small);
      }
    } else {
      (void)hipFree(ps->d_planes);
    }
  }
  ps->d_planes = nullptr;
  ps->planes_bytes = 0;
  ps->plane_groups = 0;
  ps->planes_valid = false;
}

void drop_lists(const gcre_pathset* ps) {
  if (ps->d_loff) (void)hipFree(ps->d_loff);
  if (ps->d_lidx) (void)hipFree(ps->d_lidx);
  ps->d_loff = nullptr;
  ps->d_lidx = nullptr;
  ps->h_loff.clear();
  ps->max_bits = 0;
  ps->max_known = false;
  ps->one_sided = -1;
  drop_planes(ps);
}

// CSR bit lists of a path set (method 1: one list per row): offsets on the host by prefix sum, entries on the device
int ensure_lists(gcre_ctx* c, const gcre_pathset* ps) {
  if (ps->d_loff) return GCRE_OK;
  HostTimer ht("ensure_lists");
  const Geometry& g = c->g;
  // method 2: every row is two half-rows (+)/(-) of W32p dwords, stored back to back -> 2*nrows lists
  const int64_t n = ps->nrows * g.method;
  const int hs32 = 2 * g.Wp;   // dwords per half-row == stride between half-rows
  std::vector<uint32_t> cnt((size_t)std::max<int64_t>(n, 1), 0);
  uint32_t* d_cnt = nullptr;
  HIP_TRY(c, hipMalloc((void**)&d_cnt, cnt.size() * 4));
  hipError_t e = launch_row_bits((const uint32_t*)ps->d_rows, n, hs32, hs32, d_cnt, c->stream);
  if (e == hipSuccess && n > 0) e = hipMemcpyAsync(cnt.data(), d_cnt, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_cnt);
  if (e != hipSuccess) return fail(c, GCRE_ERR_DEVICE, std::string("row lists: ") + hipGetErrorString(e));
  std::vector<uint64_t> off((size_t)n + 1, 0);
  uint32_t longest = 0;
  for (int64_t r = 0; r < n; r++) {
    off[(size_t)r + 1] = off[(size_t)r] + cnt[(size_t)r];
    longest = std::max(longest, cnt[(size_t)r]);
  }
  const uint64_t total = off[(size_t)n];
  HIP_TRY(c, hipMalloc((void**)&ps->d_loff, off.size() * 8));
  HIP_TRY(c, hipMalloc((void**)&ps->d_lidx, (size_t)(total + 16) * 4));
  e = hipMemcpyAsync(ps->d_loff, off.data(), off.size() * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess)
    e = launch_row_fill((const uint32_t*)ps->d_rows, n, hs32, hs32, ps->d_loff, (uint32_t)(64 * g.Wp) << 8,
                        ps->d_lidx, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // off is a local
  if (e != hipSuccess) {
    drop_lists(ps);
    return fail(c, GCRE_ERR_DEVICE, std::string("row lists: ") + hipGetErrorString(e));
  }
  ps->h_loff = std::move(off);
  ps->max_bits = longest;
  ps->max_known = true;
  return GCRE_OK;
}

// method 2: does every row of the set have an empty half?  (genes: a row's carriers sit in one half.)  From the host copy of
// the list offsets, once per set of lists.
bool one_sided(const gcre_pathset* ps) {
  if (!ps->d_loff || ps->h_loff.size() != (size_t)ps->nrows * 2 + 1) return false;
  if (ps->one_sided < 0) {
    int yes = 1;
    for (int64_t r = 0; r < ps->nrows && yes; r++)
      if (ps->h_loff[(size_t)2 * r + 1] > ps->h_loff[(size_t)2 * r] && ps->h_loff[(size_t)2 * r + 2] > ps->h_loff[(size_t)2 * r + 1]) yes = 0;
    ps->one_sided = yes;
  }
  return ps->one_sided == 1;
}

// largest number of carriers a row of the set can have (from its lists, or from the join that produced it)
uint32_t row_max(const gcre_ctx* c, const gcre_pathset* ps) {
  return ps->max_known ? ps->max_bits : (uint32_t)(64 * c->g.Wp);
}

int plane_groups_for(uint32_t max_count) {
  int bits = 1;
  while (bits < 16 && (max_count >> bits) != 0) bits++;
  return std::max(2, (bits + 3) / 4);
}

// planes of rows [lo, hi) are there for the current masks
bool planes_cover(const gcre_ctx* c, const gcre_pathset* ps, int64_t lo, int64_t hi) {
  return ps->d_planes && ps->planes_valid && ps->planes_epoch == c->mask_epoch && (hi <= lo || (ps->planes_lo <= lo && hi <= ps->planes_hi));
}
bool planes_current(const gcre_ctx* c, const gcre_pathset* ps) { return planes_cover(c, ps, 0, ps->nrows); }

size_t plane_bytes(const gcre_ctx* c, int64_t nrows, int groups) {
  const size_t nkt = (size_t)((c->win_K + kSparseTile - 1) / kSparseTile);
  return (size_t)std::max<int64_t>(nrows, 1) * c->g.method * nkt * (size_t)groups * 1024;
}

// room for `bytes` more on the device, leaving a quarter of what is free for the scratch of the join itself
// `bytes` fit into three quarters of the free device memory (counting `reclaimable` bytes as free)
bool planes_fit(size_t bytes, size_t reclaimable = 0) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
  free_b += reclaimable;
  return bytes < free_b - free_b / 4;
}

// the operands of a set's recipe, if the recipe is complete and both operands still hold the rows it was made from
bool recipe_operands(const gcre_ctx* c, const gcre_pathset* ps, const gcre_pathset** a, const gcre_pathset** z) {
  if (!ps->rec || !ps->rec->valid) return false;
  auto ia = c->live_sets.find(ps->rec->a_id), iz = c->live_sets.find(ps->rec->z_id);
  if (ia == c->live_sets.end() || iz == c->live_sets.end()) return false;
  if (ia->second->version != ps->rec->a_ver || iz->second->version != ps->rec->z_ver) return false;
  *a = ia->second;
  *z = iz->second;
  return true;
}

// allocate (not fill) the plane array of a set; false when it does not fit
bool alloc_planes(gcre_ctx* c, const gcre_pathset* ps, int groups) {
  HostTimer ht("alloc_planes");
  const size_t bytes = plane_bytes(c, ps->nrows, groups);
  if (ps->d_planes && ps->planes_bytes >= bytes) {   // its own buffer is large enough (another window, other groups)
    ps->plane_groups = groups;
    ps->planes_valid = false;
    poison_planes(ps->d_planes, ps->planes_bytes);
    return true;
  }
  drop_planes(ps);
  // smallest pooled buffer that is large enough
  int best = -1;
  for (size_t k = 0; k < c->plane_pool.size(); k++)
    if (c->plane_pool[k].bytes >= bytes && (best < 0 || c->plane_pool[k].bytes < c->plane_pool[(size_t)best].bytes)) best = (int)k;
  if (best >= 0) {
    ps->d_planes = c->plane_pool[(size_t)best].p;
    ps->planes_bytes = c->plane_pool[(size_t)best].bytes;
    c->plane_pool.erase(c->plane_pool.begin() + best);
  } else {
    if (!planes_fit(bytes)) {
      // make room -- pooled buffers that are too small are of no use to this set -- but only when that helps: a set
      // that cannot fit anyway (it will run on its recipe or on bit lists) must not cost the others their buffers
      size_t pooled = 0;
      for (const auto& pb : c->plane_pool) pooled += pb.bytes;
      if (!planes_fit(bytes, pooled)) return false;
      for (auto& pb : c->plane_pool) (void)hipFree(pb.p);
      c->plane_pool.clear();
      if (!planes_fit(bytes)) return false;
    }
    if (hipMalloc((void**)&ps->d_planes, bytes) != hipSuccess) {
      ps->d_planes = nullptr;
      (void)hipGetLastError();
      return false;
    }
    ps->planes_bytes = bytes;
  }
  poison_planes(ps->d_planes, ps->planes_bytes);
  ps->plane_groups = groups;
  return true;
}

// count planes of a path set from its bit lists (sets that were not produced by a kept join on this epoch).
// Returns GCRE_OK with planes_current() false when the planes do not fit in device memory.
int ensure_planes(gcre_ctx* c, const gcre_pathset* ps) {
  if (planes_current(c, ps)) return GCRE_OK;
  if (int rc = ensure_lists(c, ps)) return rc;
  const int groups = plane_groups_for(ps->max_bits);
  if (!alloc_planes(c, ps, groups)) return GCRE_OK;
  const Geometry& g = c->g;
  const int nkt = (c->win_K + kSparseTile - 1) / kSparseTile;
  const uint32_t* w_mt = c->d_mt + (size_t)(c->win_k0 / kSparseTile) * (size_t)(64 * g.Wp + 1) * 64;
  HIP_TRY(c, launch_build_planes(w_mt, (uint32_t)(64 * g.Wp + 1), nkt, ps->d_loff, ps->d_lidx, ps->nrows * g.method,
                                 groups, ps->d_planes, c->stream));
  ps->planes_epoch = c->mask_epoch;
  ps->planes_valid = true;
  ps->planes_lo = 0;
  ps->planes_hi = ps->nrows;
  return GCRE_OK;
}

// Segment table of the joined paths [first, first+count): runs of paths that share their paths0 row, at most
// kSparseSegMax long, none straddling the scored range [score_b, score_e).  The scored segments come first
// (*nscored of them); of the others only the paths inside [plane_b, plane_e) are listed (the rest need no count
// planes).  Cached per uids object (the join index is resident input; repeated joins reuse it).
// The host half of sparse_segments: the table itself (no device call: a helper thread may build it ahead of the join that
// needs it, gcre_uids::prefetch_tables).
void build_segments_host(const gcre_uids& u, int64_t first, int64_t count, int64_t score_b, int64_t score_e, int64_t plane_b,
                         int64_t plane_e, std::vector<SparseSeg>& segs, int64_t* nscored_out) {
  const auto& pi = u.h_path_idx;
  const int64_t end = first + count;
  // uids whose joined paths reach into [first, end)
  const int64_t i_lo = std::max<int64_t>(0, (int64_t)(std::upper_bound(pi.begin(), pi.end(), first) - pi.begin()) - 1);
  const int64_t i_hi = std::min<int64_t>(u.n_uids, (int64_t)(std::lower_bound(pi.begin(), pi.end(), end) - pi.begin()));
  // Millions of uids: the table is built by a few host threads, each over a contiguous piece of the uid range with
  // about the same number of joined paths (part 0: scored segments, part 1: the others; key = the paths1 row the SEGMENT
  // joins first: all uids with one pivot gene join the same rows, so equal keys = the same added rows, which is what the
  // quad kernel shares), and put together by a stable counting sort on the keys -- a noticeable part of a one-shot
  // gcre_process_paths call when done by one thread with a comparison sort.
  int T = 1;
  if (i_hi - i_lo > 200000) T = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  struct Local {
    std::vector<SparseSeg> part[2];
    std::vector<uint32_t> key[2];
    bool dense = true;
    uint32_t max_key = 0;
  };
  std::vector<Local> loc((size_t)T);
  std::vector<int64_t> cut((size_t)T + 1, i_hi);
  cut[0] = i_lo;
  for (int t = 1; t < T; t++) {
    const int64_t want = first + count * t / T;
    cut[(size_t)t] = std::min(i_hi, std::max(cut[(size_t)t - 1], (int64_t)(std::lower_bound(pi.begin() + i_lo, pi.begin() + i_hi, want) - pi.begin())));
  }
  auto build = [&](int t) {
    Local& L = loc[(size_t)t];
    const size_t guess = (size_t)(cut[(size_t)t + 1] - cut[(size_t)t]) + (size_t)(count / T / kSparseSegMax) + 16;
    L.part[score_e > score_b ? 0 : 1].reserve(guess);
    L.key[score_e > score_b ? 0 : 1].reserve(guess);
    for (int64_t i = cut[(size_t)t]; i < cut[(size_t)t + 1]; i++) {
      const int64_t lo = std::max(pi[(size_t)i], first), hi = std::min(pi[(size_t)i + 1], end);
      if (hi <= lo) continue;
      const int64_t l = u.h_location[(size_t)i];
      if (l < 0 || l > 0xfffffff0ll) L.dense = false;
      L.max_key = std::max(L.max_key, (uint32_t)l);
      // the uid's paths before, inside and after the scored range
      const int64_t pc[4] = {lo, std::min(std::max(score_b, lo), hi), std::min(std::max(score_e, lo), hi), hi};
      for (int piece = 0; piece < 3; piece++) {
        const int dst = (piece == 1) ? 0 : 1;
        int64_t a0 = pc[piece], a1 = pc[piece + 1];
        if (piece != 1) {   // planes only: clip to the rows that want them
          a0 = std::max(a0, plane_b);
          a1 = std::min(a1, plane_e);
        }
        for (int64_t a = a0; a < a1; a += kSparseSegMax) {
          L.part[dst].push_back(SparseSeg{(uint32_t)i, (uint32_t)(a - first), (uint32_t)std::min<int64_t>(kSparseSegMax, a1 - a)});
          const int64_t k = l + (a - pi[(size_t)i]);
          if (k > 0xfffffff0ll) L.dense = false;
          L.key[dst].push_back((uint32_t)k);
          L.max_key = std::max(L.max_key, (uint32_t)k);
        }
      }
    }
  };
  auto run_all = [&](const std::function<void(int)>& fn) {
    if (T == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(fn, t);
    fn(0);
    for (auto& x : th) x.join();
  };
  run_all(build);
  size_t n_part[2] = {0, 0};
  bool dense = true;
  uint32_t max_key = 0;
  for (const Local& L : loc) {
    n_part[0] += L.part[0].size();
    n_part[1] += L.part[1].size();
    dense = dense && L.dense;
    max_key = std::max(max_key, L.max_key);
  }
  const size_t n_scored = n_part[0];
  // Segments that join the same paths1 rows (all uids with the same pivot gene share `location`) run next to each
  // other: the waves of an XCD walk a contiguous window of this table, so the planes of those rows stay in its L2.
  // Any order gives the same maxima.
  segs.assign(n_part[0] + n_part[1], SparseSeg{});
  size_t base = 0;
  for (int q = 0; q < 2; q++) {
    const size_t nq = n_part[q];
    if (nq == 0) continue;
    const size_t nkeys = (size_t)max_key + 1;
    if (dense && (uint64_t)nkeys * (uint64_t)T <= ((uint64_t)64 << 20) && (uint64_t)max_key < (uint64_t)64 * nq + 1024) {
      // stable counting sort: per-thread histograms, then for every key the threads' slices one after the other
      std::vector<std::vector<uint32_t>> start((size_t)T);
      run_all([&](int t) {
        start[(size_t)t].assign(nkeys, 0);
        for (uint32_t k : loc[(size_t)t].key[q]) start[(size_t)t][k]++;
      });
      uint32_t run = 0;
      for (size_t k = 0; k < nkeys; k++)
        for (int t = 0; t < T; t++) {
          const uint32_t h = start[(size_t)t][k];
          start[(size_t)t][k] = run;
          run += h;
        }
      run_all([&](int t) {
        const Local& L = loc[(size_t)t];
        std::vector<uint32_t>& st = start[(size_t)t];
        for (size_t e = 0; e < L.part[q].size(); e++) segs[base + st[L.key[q][e]]++] = L.part[q][e];
      });
    } else {
      std::vector<SparseSeg> all;
      all.reserve(nq);
      for (const Local& L : loc) all.insert(all.end(), L.part[q].begin(), L.part[q].end());
      auto key_of = [&](const SparseSeg& x) { return u.h_location[x.row0] + ((int64_t)x.first + first - pi[x.row0]); };
      std::stable_sort(all.begin(), all.end(), [&](const SparseSeg& x, const SparseSeg& y) { return key_of(x) < key_of(y); });
      std::copy(all.begin(), all.end(), segs.begin() + (std::ptrdiff_t)base);
    }
    base += nq;
  }
  *nscored_out = (int64_t)n_scored;
}

// quads of a segment table (gcre_ieq.hip): runs of up to ieq_quad_segs() consecutive segments with the same first joined row and
// length, none straddling n_warm or nscored.  Host only.
void build_quads_host(const gcre_uids& u, const std::vector<SparseSeg>& segs, int64_t first, int64_t nscored, int64_t n_warm,
                      std::vector<uint32_t>& quads, int64_t* quad_begin_out) {
  const auto& pi = u.h_path_idx;
  const int64_t n = (int64_t)segs.size();
  quads.clear();
  int64_t quad_begin = -1;
  const int64_t qmax = ieq_quad_segs();
  quads.reserve((size_t)n / (size_t)std::max<int64_t>(1, (qmax + 1) / 2) + 16);
  auto key_of = [&](const SparseSeg& x) { return u.h_location[x.row0] + ((int64_t)x.first + first - pi[x.row0]); };
  int64_t i = 0;
  while (i < n) {
    if (i >= n_warm && quad_begin < 0) quad_begin = (int64_t)quads.size();
    const int64_t stop = i < n_warm ? n_warm : (i < nscored ? nscored : n);
    const int64_t k0 = key_of(segs[(size_t)i]);
    const uint32_t n0 = segs[(size_t)i].n;
    int64_t j = i + 1;
    while (j < stop && j < i + qmax && segs[(size_t)j].n == n0 && key_of(segs[(size_t)j]) == k0) j++;
    quads.push_back((uint32_t)i | ((uint32_t)(j - i - 1) << 30));
    i = j;
  }
  if (quad_begin < 0) quad_begin = (int64_t)quads.size();
  *quad_begin_out = quad_begin;
}

// segments of the warm-up slice of a pruned launch (run_join): the first n_warm scored segments are scored without pruning
int64_t warm_segments(const gcre_ctx* c, int64_t nseg_scored, int nkt) {
  if (c->ie_warm_segs == 0) return 0;   // GCRE_IE_WARM=0 (tests): everything through the pruned kernel, thresholds from 0
  const int64_t warm_min = std::max<int64_t>(256, (int64_t)c->ie_warm_segs * 8 / std::max(nkt, 8));
  return std::min<int64_t>(nseg_scored, std::min<int64_t>(std::max<int64_t>(warm_min, nseg_scored / 1024),
                                                          std::max<int64_t>(256, nseg_scored / 8)));
}

int sparse_segments(gcre_ctx* c, const gcre_uids& u, int64_t first, int64_t count, int64_t score_b, int64_t score_e,
                    int64_t plane_b, int64_t plane_e, const SparseSeg** d_out, int64_t* nsegs, int64_t* nscored,
                    gcre_uids::SegCache** entry = nullptr) {
  plane_b = std::max(plane_b, first);
  plane_e = std::min(plane_e, first + count);
  for (auto& sc : u.seg_cache)
    if (sc.first == first && sc.count == count && sc.score_b == score_b && sc.score_e == score_e && sc.plane_b == plane_b &&
        sc.plane_e == plane_e) {
      *d_out = sc.d_segs;
      *nsegs = sc.nsegs;
      *nscored = sc.nscored;
      if (entry) *entry = &sc;
      return GCRE_OK;
    }
  std::vector<SparseSeg> segs;
  int64_t n_scored_i = 0;
  gcre_uids::Prefetch* pf = u.prefetch.get();
  if (pf && pf->th.joinable()) {   // a table being built ahead: wait for it, whichever table this call wants
    HostTimer hw("prefetched tables: wait");
    pf->th.join();
  }
  if (pf && pf->ready && pf->first == first && pf->count == count && pf->score_b == score_b && pf->score_e == score_e &&
      pf->plane_b == plane_b && pf->plane_e == plane_e) {
    segs.swap(pf->segs);
    n_scored_i = pf->nscored;
    pf->ready = false;
    pf->quads_for_table = true;   // its quads belong to the table made now
  } else {
    HostTimer ht("sparse_segments");
    build_segments_host(u, first, count, score_b, score_e, plane_b, plane_e, segs, &n_scored_i);
  }
  const size_t n_scored = (size_t)n_scored_i;
  SparseSeg* d = nullptr;
  HIP_TRY(c, hipMalloc((void**)&d, std::max<size_t>(segs.size(), 1) * sizeof(SparseSeg)));
  if (!segs.empty())
    HIP_TRY(c, hipMemcpyAsync(d, segs.data(), segs.size() * sizeof(SparseSeg), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));   // segs is a local
  if (u.seg_cache.size() >= 64) {                // bound the cache: drop the oldest table
    (void)hipFree(u.seg_cache.front().d_segs);
    if (u.seg_cache.front().d_quads) (void)hipFree(u.seg_cache.front().d_quads);
    u.seg_cache.erase(u.seg_cache.begin());
  }
  u.seg_cache.push_back({first, count, score_b, score_e, plane_b, plane_e, (int64_t)segs.size(), (int64_t)n_scored, d});
  if (c->g.method == 1 && c->ie_quad) u.seg_cache.back().h_segs = std::move(segs);
  if (entry) *entry = &u.seg_cache.back();
  *d_out = d;
  *nsegs = u.seg_cache.back().nsegs;
  *nscored = (int64_t)n_scored;
  return GCRE_OK;
}

// Quad table of a cached segment table (see gcre_ieq.hip): greedy runs of up to ieq_quad_segs() consecutive segments with the same
// first joined paths1 row and the same length -- they join the same rows --, cut at `n_warm` (the pruned launch starts
// there) and at the end of the scored segments.  quad_begin = the first quad of the pruned launch.
int ensure_quads(gcre_ctx* c, const gcre_uids& u, gcre_uids::SegCache& sc, int64_t n_warm) {
  if (sc.q_warm == n_warm && sc.d_quads) return GCRE_OK;
  const int64_t n = (int64_t)sc.h_segs.size();
  if (n != sc.nsegs || n >= ((int64_t)1 << 30)) return GCRE_ERR_ARG;   // no host copy (or too many segments): the caller falls back
  std::vector<uint32_t> quads;
  int64_t quad_begin = 0;
  gcre_uids::Prefetch* pf = u.prefetch.get();
  if (pf && pf->quads_ready && pf->quads_for_table && pf->q_warm == n_warm && pf->first == sc.first && pf->count == sc.count &&
      pf->score_b == sc.score_b && pf->score_e == sc.score_e && pf->plane_b == sc.plane_b && pf->plane_e == sc.plane_e) {
    quads.swap(pf->quads);   // built ahead with the table
    quad_begin = pf->quad_begin;
    pf->quads_ready = false;
  } else {
    HostTimer ht("ensure_quads");
    build_quads_host(u, sc.h_segs, sc.first, sc.nscored, n_warm, quads, &quad_begin);
  }
  if (sc.d_quads) (void)hipFree(sc.d_quads);
  sc.d_quads = nullptr;
  HIP_TRY(c, hipMalloc((void**)&sc.d_quads, std::max<size_t>(quads.size(), 1) * 4));
  if (!quads.empty()) HIP_TRY(c, hipMemcpyAsync(sc.d_quads, quads.data(), quads.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));   // quads is a local
  sc.q_warm = n_warm;
  sc.nquads = (int64_t)quads.size();
  sc.quad_begin = quad_begin;
  return GCRE_OK;
}

// ---- top-k selection of one scored chunk: indices of the best min(k, valid) keys, ties cut in index order ----
// Two halves, so that a caller with a read-back of its own (the inspector's flags) can fold the state's into it:
// select_begin queues the eight digit passes (one read-back of the state they leave, gcre_kernels.hip) and the copy
// of that state into *hs; select_finish -- after a stream synchronisation -- queues the collection of the winners.
int select_begin(gcre_ctx* c, int64_t first, int64_t count, int k, SelectState* hs, hipStream_t sst) {
  *hs = SelectState{};
  if (count == 0 || k <= 0) return GCRE_OK;
  HIP_TRY(c, c->d_small.reserve(512));
  HIP_TRY(c, c->d_sel.reserve((size_t)k + 64));
  SelectState* d_state = (SelectState*)(c->d_small.p + 264);
  HIP_TRY(c, launch_radix_select(c->d_key.p + first, count, std::min<int64_t>(k, count), c->d_small.p, d_state, sst));
  HIP_TRY(c, hipMemcpyAsync(hs, d_state, sizeof *hs, hipMemcpyDeviceToHost, sst));
  return GCRE_OK;
}

int select_finish(gcre_ctx* c, int64_t first, int64_t count, int k, const SelectState& hs, uint32_t* n_selected, hipStream_t sst) {
  *n_selected = 0;
  if (count == 0 || k <= 0) return GCRE_OK;
  uint32_t* d_counter = c->d_small.p + 256;
  const uint64_t* key = c->d_key.p + first;   // selected indices are relative to `first`
  const uint64_t prefix = hs.prefix;
  const int64_t need = hs.need, greater = hs.greater;
  const uint32_t eq_count = hs.eq_count;
  const uint64_t T = prefix;   // the need-th largest key overall
  HIP_TRY(c, hipMemsetAsync(d_counter, 0, sizeof(uint32_t), sst));
  uint32_t nsel = 0;
  if (T == 0) {
    // fewer scorable paths than k: everything with a real score is selected
    HIP_TRY(c, launch_collect_gt(key, count, 0, c->d_sel.p, d_counter, (uint32_t)k, sst));
    nsel = (uint32_t)greater;
  } else if ((int64_t)eq_count == need) {
    // every path that ties with the threshold is wanted: no cut needed
    HIP_TRY(c, launch_collect_gt(key, count, T - 1, c->d_sel.p, d_counter, (uint32_t)k, sst));
    nsel = (uint32_t)(greater + need);
  } else {
    // ties at the threshold exceed the remaining slots: keep the `need` smallest ordinals
    HIP_TRY(c, launch_collect_gt(key, count, T, c->d_sel.p, d_counter, (uint32_t)k, sst));
    const int64_t chunks = (count + 1023) / 1024;
    HIP_TRY(c, c->d_chunk.reserve((size_t)chunks));
    HIP_TRY(c, launch_eq_count(key, count, T, c->d_chunk.p, sst));
    std::vector<uint32_t> cnt((size_t)chunks);
    HIP_TRY(c, hipMemcpyAsync(cnt.data(), c->d_chunk.p, cnt.size() * 4, hipMemcpyDeviceToHost, sst));
    HIP_TRY(c, hipStreamSynchronize(sst));
    uint32_t run = 0;
    for (auto& v : cnt) {
      const uint32_t t = v;
      v = run;
      run += t;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_chunk.p, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice, sst));
    HIP_TRY(c, launch_eq_collect(key, count, T, c->d_chunk.p, (uint32_t)need, c->d_sel.p + greater, sst));
    HIP_TRY(c, hipStreamSynchronize(sst));   // cnt must outlive the copy
    nsel = (uint32_t)(greater + need);
  }
  *n_selected = nsel;
  return GCRE_OK;
}

int select_chunk(gcre_ctx* c, int64_t first, int64_t count, int k, uint32_t* n_selected, hipStream_t sst) {
  SelectState hs{};
  if (int rc = select_begin(c, first, count, k, &hs, sst)) return rc;
  HIP_TRY(c, hipStreamSynchronize(sst));
  return select_finish(c, first, count, k, hs, n_selected, sst);
}

struct JoinPlan {
  const gcre_uids* u;
  const gcre_pathset* p0;
  const gcre_pathset* p1;
  gcre_pathset* res;
  bool sharded;
  int64_t shard_begin, shard_end;
  void* d_null_out;
  bool keep_ranged = false;        // rows outside [keep_begin, keep_end) and the shard are not produced at all
  bool planes_ranged = false;      // every row is produced, count planes only for [keep_begin, keep_end) and the shard
  int64_t keep_begin = 0, keep_end = 0;
  // thresholds shared with the other devices during the join (gcre_join_opts.exchange)
  int exchanges = 0;
  int (*exchange)(void*, void*, int32_t, int32_t) = nullptr;
  void* exchange_user = nullptr;
  void take(const gcre_join_opts* o) {
    if (!o) return;
    if (o->keep_ranged) {
      keep_ranged = o->keep_ranged == 1;
      planes_ranged = o->keep_ranged == 2;
      keep_begin = o->keep_begin;
      keep_end = o->keep_end;
    }
    if (o->exchange && o->exchanges > 0 && o->d_null_out) {
      exchanges = o->exchanges;
      exchange = o->exchange;
      exchange_user = o->exchange_user;
    }
  }
};

void drop_ahead(gcre_ctx* c) {
  delete c->ahead;
  c->ahead = nullptr;
}

// a launch nobody came for (another window, other operands, the pass ended): wait for its kernels, forget it
void drop_launch(const gcre_uids* u) {
  if (u->launch.active && u->launch.done) (void)hipEventSynchronize(u->launch.done);
  u->launch.active = false;
  u->launch.cands.clear();
  if (u->ctx) {
    for (auto& pr : u->launch.ev_null) { u->ctx->ev_pool.push_back(pr.first); u->ctx->ev_pool.push_back(pr.second); }
    for (auto& pr : u->launch.ev_stats) { u->ctx->ev_pool.push_back(pr.first); u->ctx->ev_pool.push_back(pr.second); }
  }
  u->launch.ev_null.clear();
  u->launch.ev_stats.clear();
  u->launch.prof = gcre_profile{};
}

void free_uids(gcre_uids* u) {
  if (!u) return;
  if (u->ctx && u->ctx->ahead)   // a registered later join names this index
    for (const JoinPlan& a : *u->ctx->ahead)
      if (a.u == u) { drop_ahead(u->ctx); break; }
  drop_launch(u);
  u->launch.d_null.release();
  if (u->launch.done) (void)hipEventDestroy(u->launch.done);
  if (u->ctx && u->ctx->stream) (void)hipStreamSynchronize(u->ctx->stream);
  if (u->ctx && u->ctx->insp_stream) (void)hipStreamSynchronize(u->ctx->insp_stream);
  if (u->ctx) {
    auto& v = u->ctx->live_uids;
    v.erase(std::remove(v.begin(), v.end(), u), v.end());
  }
  for (void* p : {(void*)u->d_path_idx, (void*)u->d_location, (void*)u->d_signs, (void*)u->d_red_index,
                  (void*)u->d_range_of, (void*)u->d_pair_range, (void*)u->d_pair_loc})
    if (p) (void)hipFree(p);
  for (auto& sc : u->seg_cache) {
    if (sc.d_segs) (void)hipFree(sc.d_segs);
    if (sc.d_quads) (void)hipFree(sc.d_quads);
  }
  for (auto& ci : u->insp) ci.release();
  if (u->prefetch && u->prefetch->th.joinable()) u->prefetch->th.join();
  delete u;
}

// validation that does not need the path sets (join_base.cpp:198-200 checks the rest in run_join)
gcre_uids* make_uids(gcre_ctx* c, int path_length, const int32_t* uid_count, const int64_t* uid_location,
                     int64_t n_uids, const int32_t* signs, int64_t n_signs) {
  HostTimer ht("make_uids");
  auto* u = new gcre_uids{};
  u->ctx = c;
  u->path_length = path_length;
  u->n_uids = n_uids;
  u->n_signs = n_signs;
  u->max_loc = -1;
  u->max_idx = -1;
  std::vector<int64_t>& path_idx = u->h_path_idx;
  path_idx.assign((size_t)n_uids + 1, 0);
  for (int64_t i = 0; i < n_uids; i++) {
    const int cnt = uid_count[i];
    if (cnt > 0) {
      const int64_t loc = uid_location[i];
      if (loc < 0) {
        fail(c, GCRE_ERR_RANGE, "assertion: uid location out of range");
        delete u;
        return nullptr;
      }
      u->max_loc = std::max(u->max_loc, loc + cnt - 1);
      u->max_idx = i;
    }
    path_idx[(size_t)i + 1] = path_idx[(size_t)i] + std::max(cnt, 0);   // uid_ref.path_idx, wrapper.cpp:128-130
  }
  u->total = path_idx[(size_t)n_uids];
  u->h_location.assign(uid_location, uid_location + n_uids);
  hipError_t e = hipMalloc((void**)&u->d_path_idx, path_idx.size() * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&u->d_location, (size_t)std::max<int64_t>(n_uids, 1) * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&u->d_signs, (size_t)std::max<int64_t>(n_signs, 1) * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(u->d_path_idx, path_idx.data(), path_idx.size() * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && n_uids > 0)
    e = hipMemcpyAsync(u->d_location, uid_location, (size_t)n_uids * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && n_signs > 0)
    e = hipMemcpyAsync(u->d_signs, signs, (size_t)n_signs * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // path_idx is a local
  if (e != hipSuccess) {
    fail(c, GCRE_ERR_DEVICE, std::string("uids upload: ") + hipGetErrorString(e));
    free_uids(u);
    return nullptr;
  }
  c->live_uids.push_back(u);
  return u;
}

// distinct (location, count) ranges of a join index, uid -> range (see k_range_union)
int ensure_ranges(gcre_ctx* c, const gcre_uids& u) {
  if (u.n_ranges >= 0) return GCRE_OK;
  HostTimer ht("ensure_ranges");
  const auto& pi = u.h_path_idx;
  std::vector<int32_t> range_of((size_t)std::max<int64_t>(u.n_uids, 1), 0);
  std::vector<int64_t> loc;
  std::vector<int32_t> cnt;
  // uids with the same pivot gene share (location, count): one slot per location remembers the first range seen
  // there (direct-address table; a hash map when the locations are too spread out for one); a second count at the same
  // location simply gets ranges of its own (sharing is an optimisation, correctness does not depend on it)
  const bool direct = u.max_loc + 1 <= 8 * u.n_uids + ((int64_t)1 << 22);
  std::vector<int32_t> by_loc(direct ? (size_t)std::max<int64_t>(u.max_loc + 1, 1) : 1, -1);
  std::unordered_map<int64_t, int32_t> by_loc_map;
  for (int64_t i = 0; i < u.n_uids; i++) {
    const int64_t n = pi[(size_t)i + 1] - pi[(size_t)i];
    if (n <= 0) continue;   // never joined: its range is never looked at
    const int64_t l = u.h_location[(size_t)i];
    int32_t r = -1;
    if (direct) {
      r = by_loc[(size_t)l];
    } else {
      auto it = by_loc_map.find(l);
      if (it != by_loc_map.end()) r = it->second;
    }
    if (r < 0 || cnt[(size_t)r] != (int32_t)n) {
      const int32_t fresh = (int32_t)loc.size();
      loc.push_back(l);
      cnt.push_back((int32_t)n);
      if (r < 0) {
        if (direct) by_loc[(size_t)l] = fresh;
        else by_loc_map.emplace(l, fresh);
      }
      r = fresh;
    }
    range_of[(size_t)i] = r;
  }
  const size_t R = loc.size();
  std::vector<int32_t> pair_range;
  std::vector<int64_t> pair_loc;
  for (size_t r = 0; r < R; r++)
    for (int32_t t = 0; t < cnt[r]; t++) {
      pair_range.push_back((int32_t)r);
      pair_loc.push_back(loc[r] + t);
    }
  const size_t T = pair_range.size();
  HIP_TRY(c, hipMalloc((void**)&u.d_range_of, range_of.size() * 4));
  HIP_TRY(c, hipMalloc((void**)&u.d_pair_range, std::max<size_t>(T, 1) * 4));
  HIP_TRY(c, hipMalloc((void**)&u.d_pair_loc, std::max<size_t>(T, 1) * 8));
  HIP_TRY(c, hipMemcpyAsync(u.d_range_of, range_of.data(), range_of.size() * 4, hipMemcpyHostToDevice, c->stream));
  if (T) {
    HIP_TRY(c, hipMemcpyAsync(u.d_pair_range, pair_range.data(), T * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(u.d_pair_loc, pair_loc.data(), T * 8, hipMemcpyHostToDevice, c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));   // the vectors are locals
  u.n_pairs = (int64_t)T;
  u.n_ranges = (int64_t)R;
  return GCRE_OK;
}

// ---- merge: global heap starts with the {-inf,-1,-1,0,0} sentinel (join_base.cpp:192-194); the best
// top_k of {sentinel} U candidates survive, reported in ascending order (format_result :140-151).
// Ties: smaller joined-path ordinal wins (DESIGN.md "Ties").
void merge_candidates(std::vector<Candidate>& cands, int top_k, gcre_result* out) {
  std::sort(cands.begin(), cands.end(), [](const Candidate& a, const Candidate& b) {
    if (a.score != b.score) return a.score > b.score;
    return a.path < b.path;
  });
  size_t keepn = std::min(cands.size(), (size_t)top_k);
  const bool with_sentinel = cands.size() < (size_t)top_k;
  const size_t n_out = keepn + (with_sentinel ? 1 : 0);
  out->n = (int32_t)n_out;
  out->scores = (double*)std::calloc(std::max<size_t>(n_out, 1), sizeof(double));
  out->src = (int32_t*)std::calloc(std::max<size_t>(n_out, 1), sizeof(int32_t));
  out->trg = (int32_t*)std::calloc(std::max<size_t>(n_out, 1), sizeof(int32_t));
  out->cases = (int32_t*)std::calloc(std::max<size_t>(n_out, 1), sizeof(int32_t));
  out->ctrls = (int32_t*)std::calloc(std::max<size_t>(n_out, 1), sizeof(int32_t));
  size_t o = 0;
  if (with_sentinel) {
    out->scores[o] = -std::numeric_limits<double>::infinity();
    out->src[o] = -1;
    out->trg[o] = -1;
    o++;
  }
  for (size_t i = keepn; i-- > 0;) {   // ascending: worst kept first
    out->scores[o] = cands[i].score;
    out->src[o] = cands[i].src;
    out->trg[o] = cands[i].trg;
    out->cases[o] = cands[i].cases;
    out->ctrls[o] = cands[i].ctrls;
    o++;
  }
}

// The tail of a join whose kernels were launched ahead (run_join, kLaunch): wait for them, copy the maxima out of the join
// index's own array, merge the winners its inspection cached.
int finish_launched(gcre_ctx* c, const JoinPlan& jp, gcre_result* out) {
  const auto t_begin = std::chrono::steady_clock::now();
  const gcre_uids& u = *jp.u;
  gcre_uids::Launched& L = u.launch;
  const int K = c->win_K;
  const int Kpad = ((K + kPermTileMax - 1) / kPermTileMax) * kPermTileMax;
  HIP_TRY(c, hipEventSynchronize(L.done));
  L.active = false;
  out->n_perm = K;
  out->null_max = (float*)std::calloc((size_t)std::max(K, 1), sizeof(float));
  uint32_t lookups = 0;
  if (K > 0) {
    // (on the inspection stream, idle by now: the main stream holds the later levels' kernels, which this join's caller
    // need not wait for)
    hipStream_t cs = c->insp_stream;
    HIP_TRY(c, hipMemcpyAsync(out->null_max, L.d_null.p, (size_t)K * 4, hipMemcpyDeviceToHost, cs));
    HIP_TRY(c, hipMemcpyAsync(&lookups, L.d_null.p + Kpad, 4, hipMemcpyDeviceToHost, cs));
    if (jp.d_null_out) HIP_TRY(c, hipMemcpyAsync(jp.d_null_out, L.d_null.p, (size_t)K * 4, hipMemcpyDeviceToDevice, cs));
    HIP_TRY(c, hipStreamSynchronize(cs));
  }
  std::vector<Candidate> cands = std::move(L.cands);
  L.cands.clear();
  merge_candidates(cands, c->top_k, out);
  c->prof = L.prof;
  L.prof = gcre_profile{};
  c->prof.null_kernel_ms = drain_events(c, L.ev_null);
  c->prof.stats_kernel_ms = drain_events(c, L.ev_stats);
  c->prof.ie_lookup_tiles += lookups;
  c->prof.scores = c->prof.paths * (int64_t)K;
  c->prof.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return GCRE_OK;
}

// inspect_only (gcre_join_ahead): the mask-independent half of the join -- expansion, inspector, kept rows, recipe, flags, top-k
// winners -- on the inspection stream, into the join index's inspection cache; nothing of the permutation kernel's (no maxima,
// no count planes, no results).  The join proper then replays it and starts at its null kernel.
// kLaunch (the chain of gcre_join_ahead): the join's permutation kernels are launched -- on the main stream, into the join
// index's own maxima, from the inspection that ran ahead -- and nothing is waited for; the call that comes for the join finds
// it launched and only finishes it (wait, copy the maxima, merge the cached winners).
enum JoinMode { kFull = 0, kInspect = 1, kLaunch = 2 };

int run_join(gcre_ctx* c, const JoinPlan& jp, gcre_result* out, JoinMode mode = kFull) {
  const bool inspect_only = mode == kInspect, launch_only = mode == kLaunch;
  // the join sees the permutation window (all permutations unless gcre_set_perm_window narrowed it): K, the slice of
  // the transposed masks, of the masks (row stride stays the full Kpad) and of the maxima
  Geometry g = c->g;
  g.K = c->win_K;
  g.Kpad = ((g.K + kPermTileMax - 1) / kPermTileMax) * kPermTileMax;
  const int Kstride = c->g.Kpad;
  uint32_t* w_null = c->d_null ? c->d_null + c->win_k0 : nullptr;
  const uint32_t* const w_masks = c->d_masks ? c->d_masks + c->win_k0 : nullptr;
  const uint32_t* const w_mt = c->d_mt ? c->d_mt + (size_t)(c->win_k0 / kSparseTile) * (size_t)(64 * g.Wp + 1) * 64 : nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  if (out) std::memset(out, 0, sizeof *out);
  // the registered next join (gcre_join_ahead) belongs to THIS call: taken here, inspected below once this join's own
  // kernels are in flight, dropped on every road out
  std::unique_ptr<std::vector<JoinPlan>> ahead_plan;
  if (mode == kFull) {
    ahead_plan.reset(c->ahead);
    c->ahead = nullptr;
    c->ahead_closed = false;
  }
  uint32_t* const flagblk = inspect_only ? c->d_max_tot_b : c->d_max_tot;
  if (!c->have_table) return fail(c, GCRE_ERR_ASSERT, "value table not set");
  if (g.K > 0 && !c->have_perms) return fail(c, GCRE_ERR_ASSERT, "permuted cases not set");
  if (c->top_k < 1) return fail(c, GCRE_ERR_ARG, "top_k must be >= 1");
  if (!jp.u || jp.u->ctx != c || !jp.p0 || !jp.p1 || jp.p0->ctx != c || jp.p1->ctx != c || (jp.res && jp.res->ctx != c))
    return fail(c, GCRE_ERR_ARG, "uids / path set do not belong to this context");
  const gcre_uids& u = *jp.u;

  // ---- the checks of JoinExec::join, join_base.cpp:196-200 ----
  if (u.n_uids != jp.p0->nrows) return fail(c, GCRE_ERR_ASSERT, "assertion: uids.size() != paths0.size");
  if (u.max_loc >= jp.p1->nrows) return fail(c, GCRE_ERR_RANGE, "assertion: uid location out of range");
  const int64_t P = u.total;
  const bool keep = jp.res != nullptr && jp.res->nrows != 0;   // keep_paths = paths_res.size != 0, join_base.cpp:217
  if (jp.res && jp.res->nrows != 0 && jp.res->nrows != P)
    return fail(c, GCRE_ERR_ASSERT, "assertion: paths_res.size != total paths");
  // ---- inspection cache: has this very join (same operand rows, kept set, shard, table) run on this index before? ----
  InspKey ikey;
  bool replay = false;
  if (c->insp_cache) {
    ikey.p0_id = jp.p0->id; ikey.p0_ver = jp.p0->version;
    ikey.p1_id = jp.p1->id; ikey.p1_ver = jp.p1->version;
    if (u.red) {
      auto it = c->live_sets.find(u.red_id);
      if (it != c->live_sets.end() && it->second == u.red) { ikey.red_id = u.red_id; ikey.red_ver = u.red->version; }
    }
    ikey.res_id = jp.res ? jp.res->id : 0;
    ikey.obs_epoch = c->obs_epoch;
    ikey.sb = jp.sharded ? jp.shard_begin : 0;
    ikey.se = jp.sharded ? jp.shard_end : -1;
    ikey.keep_mode = (keep ? 1 : 0) | (jp.keep_ranged ? 2 : 0) | (jp.planes_ranged ? 4 : 0);
    ikey.keep_begin = (jp.keep_ranged || jp.planes_ranged) ? jp.keep_begin : 0;
    ikey.keep_end = (jp.keep_ranged || jp.planes_ranged) ? jp.keep_end : 0;
    ikey.chunk_paths = c->chunk_paths;
    ikey.top_k = c->top_k;
    ikey.null_kernel = c->null_kernel;
    replay = u.insp_valid && u.insp_key == ikey && (!keep || jp.res->version == u.insp_res_ver);
    // not launched: the join's own call runs it whole
    if (launch_only && (!replay || jp.exchange || g.K <= 0)) return GCRE_OK;
    if (mode == kFull && u.launch.active) {
      // this join was launched ahead: if it is still the same join (operands, kept set, shard, table, window, masks) only its
      // results are left to collect; otherwise its kernels are waited for and forgotten
      const gcre_uids::Launched& L = u.launch;
      if (replay && L.key == ikey && L.win_k0 == c->win_k0 && L.win_K == c->win_K && L.mask_epoch == c->mask_epoch &&
          (!keep || jp.res->version == L.res_ver) && !jp.exchange)
        return finish_launched(c, jp, out);
      drop_launch(&u);
    }
    if (inspect_only) drop_launch(&u);          // (a stale launch of an earlier pass)
    if (inspect_only && replay) return GCRE_OK;   // already inspected (a later permutation window, kept inspections)
    if (!replay) {
      for (auto& ci : u.insp) {   // the buffers stay and serve the new chunks in turn
        ci.inspected = ci.with_lists = ci.flags_valid = ci.win_valid = false;
        ci.cb = -1;
      }
      u.insp_key = ikey;
    }
    u.insp_valid = false;   // until this call completes
  } else if (mode != kFull) {
    return GCRE_OK;   // no inspection cache, nothing to inspect or launch ahead
  } else if (u.insp_valid || !u.insp.empty()) {
    for (auto& ci : u.insp) ci.release();
    u.insp.clear();
    u.insp_valid = false;
  }
  if (keep && !replay) jp.res->version++;
  if (keep && !replay) {   // its rows are about to be rewritten: lists go, the plane buffer stays allocated for the new rows
    uint32_t* planes = jp.res->d_planes;
    const int groups = jp.res->plane_groups;
    const size_t pbytes = jp.res->planes_bytes;
    jp.res->d_planes = nullptr;
    drop_lists(jp.res);
    jp.res->d_planes = planes;
    jp.res->plane_groups = groups;
    jp.res->planes_bytes = pbytes;
  }
  if (g.method == 2 && P > 0) {
    // need_flip reads signs[idx] and/or signs[loc] (gcre.h:71-81); the reference would read out of bounds
    int64_t need_signs = 0;
    if (u.path_length > 3) need_signs = u.max_idx + 1;
    else if (u.path_length < 3) need_signs = u.max_loc + 1;
    else need_signs = std::max(u.max_idx, u.max_loc) + 1;
    if (u.n_signs < need_signs) return fail(c, GCRE_ERR_RANGE, "signs vector shorter than the rows it is indexed by");
  }

  int64_t sb = jp.shard_begin, se = jp.shard_end;
  if (!jp.sharded) { sb = 0; se = P; }
  sb = std::max<int64_t>(0, std::min(sb, P));
  se = std::max(sb, std::min(se, P));

  hipStream_t st = inspect_only ? c->insp_stream : c->stream;
  const int Kpad = g.Kpad;
  // what an ahead inspection and an ahead launch cost is booked on the join they work for, not on the join they run beside
  struct ProfSwap {
    gcre_ctx* c;
    gcre_uids::Launched* L;
    ProfSwap(gcre_ctx* c_, gcre_uids::Launched* L_) : c(c_), L(L_) { swap(); }
    ~ProfSwap() { swap(); }
    void swap() {
      if (!L) return;
      std::swap(c->prof, L->prof);
      std::swap(c->ev_null, L->ev_null);
      std::swap(c->ev_stats, L->ev_stats);
    }
  } prof_swap(c, mode != kFull ? &u.launch : nullptr);
  if (inspect_only) {
    // behind the last inspector that ran on the main stream (its kept rows and recipe are this one's operands)
    HIP_TRY(c, hipStreamWaitEvent(st, c->ev_insp_main, 0));
  } else {
    // behind an inspection that ran ahead on its own stream (a no-op when there was none)
    HIP_TRY(c, hipStreamWaitEvent(st, c->ev_insp_done, 0));
    if (launch_only) {   // the launch's own maxima (+ its look-up counter): the context's belong to the join that is being finished
      HIP_TRY(c, u.launch.d_null.reserve((size_t)Kpad + 64));
      w_null = u.launch.d_null.p;
      if (!u.launch.done && hipEventCreateWithFlags(&u.launch.done, hipEventDisableTiming) != hipSuccess)
        return fail(c, GCRE_ERR_DEVICE, "hipEventCreate failed");
      HIP_TRY(c, hipMemsetAsync(w_null, 0, ((size_t)Kpad + 1) * 4, st));
    } else if (Kpad > 0) {
      HIP_TRY(c, hipMemsetAsync(w_null, 0, (size_t)Kpad * 4, st));
    }
  }
  // The chain of gcre_join_ahead: once this join's own work is queued, every registered later join is inspected (on the
  // inspection stream) and launched (on the main stream) in turn -- the big inspector of the last level then runs beside the
  // small permutation kernels of the levels before it instead of after them
  auto run_chain = [&]() -> int {
    if (!ahead_plan) return GCRE_OK;
    std::unique_ptr<std::vector<JoinPlan>> chain = std::move(ahead_plan);
    int budget = 1 << 30;   // GCRE_AHEAD_MAX (diagnostics): how many of the registered joins run ahead
    if (const char* e = std::getenv("GCRE_AHEAD_MAX")) budget = std::atoi(e);
    for (const JoinPlan& a : *chain) {
      if (budget-- <= 0) break;
      if (std::getenv("GCRE_AHEAD_INSPECT_ONLY")) {   // (diagnostics: inspections ahead, no launches)
        if (int rc = run_join(c, a, nullptr, kInspect)) return rc;
        continue;
      }
      if (int rc = run_join(c, a, nullptr, kInspect)) return rc;
      if (int rc = run_join(c, a, nullptr, kLaunch)) return rc;
      // a join that was not launched (its inspection did not validate: a broken hint) has not written what the joins
      // behind it read: they run whole, in their own calls
      if (!a.u->launch.active) break;
    }
    return GCRE_OK;
  };

  const NullConfig cfg = null_config(g.method, g.K);
  // ---- thresholds shared across devices: the maxima so far go out, the merged ones come back (gcre_join_opts.exchange) ----
  int exchanges_done = 0;
  auto exchange_now = [&]() -> int {
    if (!jp.exchange || exchanges_done >= jp.exchanges) return GCRE_OK;
    exchanges_done++;
    if (g.K <= 0) return GCRE_OK;
    HIP_TRY(c, hipMemcpyAsync(jp.d_null_out, w_null, (size_t)g.K * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (jp.exchange(jp.exchange_user, jp.d_null_out, c->win_k0, c->win_k0 + g.K) != 0)
      return fail(c, GCRE_ERR_DEVICE, "the caller's threshold exchange failed");
    // merged maxima are >= this shard's: a plain copy back (non-negative floats order like their bit patterns)
    HIP_TRY(c, hipMemcpyAsync(w_null, jp.d_null_out, (size_t)g.K * 4, hipMemcpyDeviceToDevice, st));
    return GCRE_OK;
  };
  std::vector<Candidate> cands;
  if (mode == kFull) c->prof = gcre_profile{};
  double select_ms = 0, select_wait_ms = 0;
  bool keep_planes_done = false;
  int64_t keep_planes_lo = 0, keep_planes_hi = 0;
  uint32_t keep_max_tot = 0;

  if (P > 0) {
    // segments: rows outside the shard are only materialised (when kept); the shard [sb, se) inside a segment is scored.
    // Kept rows next to the shard share its launches (one inspector pass, one flag read-back, one kernel that scores
    // the shard's path runs and only writes count planes / recipe entries for the others).
    struct Seg { int64_t b, e, sb, se; };
    std::vector<Seg> segs;
    // kept rows outside the scored shard: all of them, or only [keep_begin, keep_end) (gcre_join_opts.keep_ranged)
    int64_t kb = 0, ke = P;
    if (jp.keep_ranged) {
      kb = std::max<int64_t>(0, std::min(jp.keep_begin, P));
      ke = std::max(kb, std::min(jp.keep_end, P));
    }
    // rows of `res` that get count planes: every produced row, or (gcre_join_opts.keep_ranged == 2) the hull of
    // [keep_begin, keep_end) and the shard
    int64_t pl_b = std::min(kb, se > sb ? sb : kb), pl_e = std::max(ke, se > sb ? se : ke);
    if (jp.planes_ranged) {
      const int64_t qb = std::max<int64_t>(0, std::min(jp.keep_begin, P)), qe = std::max(qb, std::min(jp.keep_end, P));
      pl_b = qe > qb ? qb : (se > sb ? sb : 0);
      pl_e = qe > qb ? qe : (se > sb ? se : 0);
      if (se > sb) {
        pl_b = std::min(pl_b, sb);
        pl_e = std::max(pl_e, se);
      }
    }
    if (!keep || kb >= ke) {
      if (se > sb) segs.push_back({sb, se, sb, se});
    } else if (se <= sb) {
      segs.push_back({kb, ke, kb, kb});
    } else if (kb <= se && sb <= ke) {   // overlapping or adjacent: one segment
      segs.push_back({std::min(kb, sb), std::max(ke, se), sb, se});
    } else if (ke < sb) {
      segs.push_back({kb, ke, kb, kb});
      segs.push_back({sb, se, sb, se});
    } else {
      segs.push_back({sb, se, sb, se});
      segs.push_back({kb, ke, kb, kb});
    }

    const int64_t tile = cfg.path_tile;
    const int64_t chunk_cap = std::max<int64_t>(tile, (c->chunk_paths / tile) * tile);
    const size_t cap = (size_t)std::min<int64_t>(chunk_cap, ((P + tile - 1) / tile) * tile) + (size_t)tile;
    bool hint_broke_late = false;

    // uids (rows of paths0) with at least one joined path inside [first, first+count)
    auto uids_in = [&](int64_t first, int64_t count) {
      const auto& pi = u.h_path_idx;
      int64_t lo = std::upper_bound(pi.begin(), pi.end(), first) - pi.begin() - 1;
      int64_t hi = std::lower_bound(pi.begin(), pi.end(), first + count) - pi.begin();   // first uid starting at/after the end
      if (u.h_nonempty.empty()) {   // prefix count of the uids that join anything: once per join index
        u.h_nonempty.assign((size_t)u.n_uids + 1, 0);
        for (int64_t i = 0; i < u.n_uids; i++)
          u.h_nonempty[(size_t)i + 1] = u.h_nonempty[(size_t)i] + (pi[(size_t)i + 1] > pi[(size_t)i] ? 1 : 0);
      }
      lo = std::max<int64_t>(lo, 0);
      hi = std::min(hi, u.n_uids);
      return hi > lo ? u.h_nonempty[(size_t)hi] - u.h_nonempty[(size_t)lo] : (int64_t)0;
    };
    // compulsory HBM bytes of the permutation scoring of `count` joined paths: every paths0 row
    // once per uid, every paths1 row once per joined path, the masks and the maxima once, the join index
    auto alg_bytes = [&](int64_t first, int64_t count) {
      const double up = (double)uids_in(first, count);
      return 8.0 * g.W * g.method * (up + (double)count) + 8.0 * g.W * g.K + 4.0 * g.K + 24.0 * up;
    };
    // ---- inclusion-exclusion form (gcre_ie.hip): operands and their count planes, once per join ----
    const bool sparse_ok = sparse_enabled(c) && w_mt != nullptr;
    bool want_ie = sparse_ok && (c->null_kernel == 0 || c->null_kernel == 3);
    const int nkt_sp = (g.K + kSparseTile - 1) / kSparseTile;
    bool hinted = false;
    if (want_ie && u.red && u.d_red_index && u.n_red_index > u.max_loc) {
      auto it = c->live_sets.find(u.red_id);   // the caller may have freed the operand since
      hinted = it != c->live_sets.end() && it->second == u.red;
      if (replay && !u.insp_hinted) hinted = false;   // the cached run found the hint broken: it ended on paths1 itself
    }
    const gcre_pathset* red = nullptr;
    bool have_pz = false, have_p0 = false, res_planes = false, res_planes_ok = false, use_rec = false;
    const gcre_pathset *rec_a = nullptr, *rec_z = nullptr;
    gcre_recipe* rcp = nullptr;   // the recipe this join leaves with the rows it keeps (method 1)
    uint32_t join_max_tot = 0, join_max_len = 0;
    bool ie_ran = false, ie_stat_pending = false, recipe_started = false, recipe_broken = false;
    uint32_t over_next = 0;   // entries of the kept set's overflow area (gcre_recipe::over) handed out to this join's chunks so far
    auto collect_ie_stat = [&]() {
      if (launch_only) ie_stat_pending = false;   // (its counter sits behind its maxima and is read when the join is finished)
      if (!ie_stat_pending) return;
      uint32_t v = 0;
      if (hipMemcpyAsync(&v, flagblk + 3, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
        c->prof.ie_lookup_tiles += v;
      ie_stat_pending = false;
    };
    auto prepare_z = [&]() -> int {
      red = hinted ? u.red : jp.p1;
      // a kept set read as the added rows (level 5 adds rows of paths2) without planes of its own: they are rebuilt from
      // its bit lists now, and the join that writes its rows leaves them next time
      if (!planes_current(c, red) && red->rec) red->planes_wanted = true;
      if (int rc = ensure_lists(c, red)) return rc;
      if (int rc = ensure_planes(c, red)) return rc;
      have_pz = planes_current(c, red);
      return GCRE_OK;
    };
    const auto tp0 = std::chrono::steady_clock::now();
    // rows of paths0 this call reads: the uids of the joined paths it processes
    int64_t r_lo = 0, r_hi = 0;
    if (!segs.empty()) {
      int64_t first = P, last = 0;
      for (const Seg& sg : segs) {
        first = std::min(first, sg.b);
        last = std::max(last, sg.e);
      }
      const auto& pi = u.h_path_idx;
      r_lo = std::max<int64_t>(0, (int64_t)(std::upper_bound(pi.begin(), pi.end(), first) - pi.begin()) - 1);
      r_hi = std::min<int64_t>(u.n_uids, (int64_t)(std::lower_bound(pi.begin(), pi.end(), last) - pi.begin()));
    }
    if (want_ie) {
     if (inspect_only) {
      // count planes are the permutation kernel's business: the join proper looks after them (and decides, from what it
      // then finds, where its rows' planes go); the inspector only needs to know which rows the join adds
      red = hinted ? u.red : jp.p1;
      have_pz = have_p0 = true;
     } else {
      if (int rc = prepare_z()) return rc;
      have_p0 = planes_cover(c, jp.p0, r_lo, r_hi);        // a kept join left them behind
      if (!have_p0 && recipe_operands(c, jp.p0, &rec_a, &rec_z)) {
        // ... or it left the recipe: the kernel rebuilds a row's planes from the planes of the recipe's operands
        // (of which a multi-device run may hold a range only: the rows the recipe of [r_lo, r_hi) names -- the
        // producing join's paths0 rows, ascending)
        int64_t a_lo = 0, a_hi = rec_a->nrows;
        if (!planes_current(c, rec_a) && r_hi > r_lo && (size_t)r_hi <= jp.p0->rec->row0.cap) {
          uint32_t ends[2] = {0, 0};
          HIP_TRY(c, hipMemcpyAsync(&ends[0], jp.p0->rec->row0.p + r_lo, 4, hipMemcpyDeviceToHost, st));
          HIP_TRY(c, hipMemcpyAsync(&ends[1], jp.p0->rec->row0.p + (r_hi - 1), 4, hipMemcpyDeviceToHost, st));
          HIP_TRY(c, hipStreamSynchronize(st));
          a_lo = ends[0];
          a_hi = (int64_t)ends[1] + 1;
        }
        if (!planes_cover(c, rec_a, a_lo, a_hi)) {
          rec_a->planes_wanted = true;
          if (int rc = ensure_planes(c, rec_a)) return rc;
        }
        if (int rc = ensure_planes(c, rec_z)) return rc;
        use_rec = planes_cover(c, rec_a, a_lo, a_hi) && planes_current(c, rec_z);
        have_p0 = use_rec;
      }
      if (!have_p0) {
        if (int rc = ensure_planes(c, jp.p0)) return rc;   // from its bit lists (all rows)
        have_p0 = planes_current(c, jp.p0);
      }
      if (!have_p0 || !have_pz) want_ie = false;   // the planes do not fit in device memory: delta streaming (gcre_sparse.hip)
     }
      // (an ahead inspection has its own excess rows: the join in flight on the main stream may still be writing its own)
      auto& exbuf = inspect_only ? c->d_excess_b : c->d_excess;
      if (want_ie && hinted) {
        // the hint is checked without reading paths1 per joined path: once per distinct uid range here (the reduced
        // row lies inside paths1[loc]; what paths1[loc] has beyond it is collected per range), and per joined path in
        // k_stats_ie (that excess lies inside paths0[idx])
        if (int rc = ensure_ranges(c, u)) return rc;
        const size_t ewords = (size_t)std::max<int64_t>(u.n_ranges, 1) * g.S;
        HIP_TRY(c, exbuf.reserve(ewords));
        HIP_TRY(c, hipMemsetAsync(exbuf.p, 0, ewords * 8, st));
        HIP_TRY(c, hipMemsetAsync(flagblk, 0, 32, st));
        // its verdict lands in word 6 of the flag block, which the chunks leave alone: it is read with the first
        // chunk's flags (no round trip of its own); a reduced row that is not even part of the row it stands for sends
        // that chunk, and the join, back to paths1 itself like any other broken hint
        HIP_TRY(c, launch_range_union(jp.p1->d_rows, red->d_rows, u.d_red_index, u.d_pair_range, u.d_pair_loc, u.n_pairs, g.S,
                                      g.Wp, g.method, exbuf.p, flagblk + 6, st));
      }
      if (want_ie && keep) {
        // carriers of a joined row <= carriers(paths0 row) + carriers(added row); <= padded patient count
        const uint32_t bound = std::min<uint32_t>((uint32_t)(64 * g.Wp), row_max(c, jp.p0) + row_max(c, red));
        const int out_groups = plane_groups_for(bound);
        bool want_out = true;
        {
          // the kept rows leave with the recipe of this join (its inspector output: 60 B per row, 104 B for the signed
          // method's two lists).  Their planes are only written when they are small, or when the next join could not use the
          // recipe (it needs stored planes of paths0)
          if (!jp.res->rec) jp.res->rec = new gcre_recipe();
          rcp = jp.res->rec;
          if (!replay) rcp->valid = false;
          const size_t rows = (size_t)P, lists = rows * (size_t)g.method;
          hipError_t e = rcp->row0.reserve(rows + 64);
          if (e == hipSuccess) e = rcp->rowz.reserve(rows + 64);
          if (e == hipSuccess) e = rcp->linfo.reserve(lists + 64);
          if (e == hipSuccess) e = rcp->lover.reserve(lists + 64);
          if (e == hipSuccess) e = rcp->tot.reserve(lists + 64);
          if (e == hipSuccess) e = rcp->slot.reserve(lists * 8 + 64);
          if (e == hipSuccess) e = rcp->over.reserve(std::max<size_t>(rcp->over.cap, lists * 2 + ((size_t)1 << 26)));
          if (e != hipSuccess) {
            (void)hipGetLastError();
            rcp->release();
            rcp = nullptr;
          } else {
            // (sized for the nominal window: the same answer in every window of a run)
            const size_t nkt_nom = (size_t)((std::max(c->win_K, c->win_K_nominal) + kSparseTile - 1) / kSparseTile);
            const size_t nominal = (size_t)std::max<int64_t>(jp.res->nrows, 1) * g.method * nkt_nom * (size_t)out_groups * 1024;
            want_out = nominal <= c->planes_out_max || use_rec || jp.res->planes_wanted;
          }
        }
        if (inspect_only) {
          // (the join proper allocates -- or drops -- the kept rows' planes)
        } else if (want_out) {
          res_planes = alloc_planes(c, jp.res, out_groups);
          res_planes_ok = res_planes;
        } else {
          drop_planes(jp.res);   // nothing stale stays behind
        }
      }
    }
    c->prof.prepare_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count();
    auto queue_winners = [&](int64_t s0, uint32_t nsel, Winners& w, hipStream_t qs) -> int {
      w.n = nsel;
      if (nsel == 0) return GCRE_OK;
      HIP_TRY(c, c->d_wkey.reserve(nsel));
      HIP_TRY(c, c->d_wcases.reserve(nsel));
      HIP_TRY(c, c->d_wctrls.reserve(nsel));
      HIP_TRY(c, c->d_wrow0.reserve(nsel));
      HIP_TRY(c, c->d_wrow1.reserve(nsel));
      HIP_TRY(c, launch_gather_winners(c->d_sel.p, nsel, c->d_key.p + s0, c->d_cases.p + s0, c->d_ctrls.p + s0,
                                       c->d_row0.p + s0, c->d_row1.p + s0, c->d_wkey.p, c->d_wcases.p, c->d_wctrls.p, c->d_wrow0.p,
                                       c->d_wrow1.p, qs));
      w.sel.resize(nsel); w.cases.resize(nsel); w.ctrls.resize(nsel); w.r0.resize(nsel); w.r1.resize(nsel); w.key.resize(nsel);
      HIP_TRY(c, hipMemcpyAsync(w.sel.data(), c->d_sel.p, nsel * 4, hipMemcpyDeviceToHost, qs));
      HIP_TRY(c, hipMemcpyAsync(w.key.data(), c->d_wkey.p, nsel * 8, hipMemcpyDeviceToHost, qs));
      HIP_TRY(c, hipMemcpyAsync(w.cases.data(), c->d_wcases.p, nsel * 4, hipMemcpyDeviceToHost, qs));
      HIP_TRY(c, hipMemcpyAsync(w.ctrls.data(), c->d_wctrls.p, nsel * 4, hipMemcpyDeviceToHost, qs));
      HIP_TRY(c, hipMemcpyAsync(w.r0.data(), c->d_wrow0.p, nsel * 4, hipMemcpyDeviceToHost, qs));
      HIP_TRY(c, hipMemcpyAsync(w.r1.data(), c->d_wrow1.p, nsel * 4, hipMemcpyDeviceToHost, qs));
      return GCRE_OK;
    };
    // only the inclusion-exclusion kernels score part of a chunk; every other form gets chunks cut at the shard's ends
    bool split = !(want_ie && g.K > 0);
    for (const Seg& sg : segs) {
      int64_t next = sg.b;
      while (next < sg.e) {
        const int64_t cb = next;
        int64_t ce = std::min(cb + chunk_cap, sg.e);
        if (split && cb < sg.sb) ce = std::min(ce, sg.sb);
        else if (split && cb < sg.se) ce = std::min(ce, sg.se);
        next = ce;
        const int64_t n = ce - cb;
        const int64_t s0 = std::min(std::max<int64_t>(sg.sb - cb, 0), n);       // scored paths of this chunk: [s0, s1)
        const int64_t s1 = std::max(s0, std::min(std::max<int64_t>(sg.se - cb, 0), n));
        const bool scored = s1 > s0;
        const bool partial = scored && (s0 > 0 || s1 < n);
        SelectState& sel_state = c->h_sel;
        bool sel_begun = false, sel_done = false;
        hipStream_t sel_on = st;   // the stream this chunk's selection runs on
        Winners win;
        const int64_t npt = (n + tile - 1) / tile;
        const int64_t padded = npt * tile;
        // inspection cache: this chunk's buffers stand in for the context's scratch while the chunk is worked on
        ChunkInsp* ci = nullptr;
        if (c->insp_cache) {
          for (auto& e : u.insp)
            if (e.cb == cb && e.n == n) ci = &e;
          if (!ci) {
            for (auto& e : u.insp)   // an entry of an earlier join on this index: its buffers serve this chunk
              if (e.cb < 0) { ci = &e; break; }
            if (!ci) { u.insp.emplace_back(); ci = &u.insp.back(); }
            ci->inspected = ci->with_lists = ci->flags_valid = ci->win_valid = false;
            ci->cb = cb;
            ci->n = n;
          }
          if (ci->s0 != s0 || ci->s1 != s1) ci->win_valid = false;
          ci->s0 = s0;
          ci->s1 = s1;
        }
        InspSwap swapped(c, ci);
        const bool use_ie_chunk = want_ie && (scored || res_planes || rcp != nullptr);
        // replayed: the inspector's output is in place (rows, statistics, keys, kept rows, lists when this chunk wants them)
        const bool hit = ci && replay && ci->inspected &&
                         (!use_ie_chunk || (ci->with_lists && ci->flags_valid && ci->in_recipe == (rcp != nullptr)));
        if (ci && !hit) ci->inspected = ci->with_lists = ci->flags_valid = ci->win_valid = false;
        if (hit) c->prof.inspect_replays++;
        // room for this chunk.  A replayed chunk's buffers hold what its inspector wrote: they may only grow with their
        // contents (the path tile, and with it `cap` and the padding below, depends on the window's permutation count)
        auto hold = [&](auto& buf, size_t want) -> hipError_t { return hit ? buf.grow_keep(want, buf.cap, st) : buf.reserve(want); };
        HIP_TRY(c, hold(c->d_row0, cap));
        HIP_TRY(c, hold(c->d_row1, cap));
        HIP_TRY(c, hold(c->d_tot, cap * g.method));
        HIP_TRY(c, hold(c->d_cases, cap));
        HIP_TRY(c, hold(c->d_ctrls, cap));
        HIP_TRY(c, hold(c->d_key, cap));
        // the null kernel reads whole tiles: rows / totals beyond n must be valid (row 0, zero carriers)
        const bool pad_now = padded > n && (!hit || padded > ci->padded);
        if (ci) ci->padded = std::max(hit ? ci->padded : (int64_t)0, padded);
        if (pad_now) {
          HIP_TRY(c, hipMemsetAsync(c->d_row0.p + n, 0, (size_t)(padded - n) * 4, st));
          HIP_TRY(c, hipMemsetAsync(c->d_row1.p + n, 0, (size_t)(padded - n) * 4, st));
          HIP_TRY(c, hipMemsetAsync(c->d_tot.p + (size_t)n * g.method, 0, (size_t)(padded - n) * 4 * g.method, st));
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (!hit) {
          e0 = get_event(c);
          e1 = get_event(c);
          HIP_TRY(c, hipEventRecord(e0, st));
          HIP_TRY(c, launch_expand(u.d_path_idx, u.d_location, u.n_uids, u.d_signs, u.path_length, g.method,
                                   cb, n, c->d_row0.p, c->d_row1.p, st));
        }
        StatsArgs sa{};
        sa.p0 = jp.p0->d_rows;
        sa.p1 = jp.p1->d_rows;
        sa.row0 = c->d_row0.p;
        sa.row1 = c->d_row1.p;
        sa.case_mask = c->d_case_mask;
        sa.dvt = c->d_dvt;
        sa.key = c->d_key.p;
        sa.tot = c->d_tot.p;
        sa.cases = c->d_cases.p;
        sa.ctrls = c->d_ctrls.p;
        sa.res = keep ? jp.res->d_rows : nullptr;
        sa.first = cb;
        sa.count = n;
        sa.S = g.S;
        sa.Wp = g.Wp;
        const bool use_ie = use_ie_chunk;
        const bool use_sparse = scored && sparse_ok && !want_ie;
        if (partial && !use_ie && g.K > 0) {   // the join left the inclusion-exclusion form on an earlier chunk
          split = true;
          next = cb;
          continue;
        }
        if (use_sparse || use_ie) {
          collect_ie_stat();
          // flags of this chunk; the long-list counter (word 4) runs on across the chunks of a join that keeps a recipe
          HIP_TRY(c, hipMemsetAsync(flagblk, 0, (rcp && recipe_started) ? 16 : 24, st));   // words 6, 7: the join's
          // ... and word 4, the entries reserved so far in the recipe's overflow area, is the JOIN's, not the flag block's: a
          // chunk replayed from the inspection cache never touched this block, and an ahead inspection ran on the other one
          if (rcp) HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)(flagblk + 4), (int)over_next, 1, st));
          recipe_started = recipe_started || rcp != nullptr;
          sa.max_tot = flagblk;
          HIP_TRY(c, hold(c->d_dcnt, (size_t)n * g.method));
          sa.dcnt = c->d_dcnt.p;
        }
        if (use_ie) {
          // one pass: statistics, kept rows, the check of the reduced operand, and the lists the null kernel streams
          const size_t nl = (size_t)n * g.method;
          HIP_TRY(c, hold(c->d_rowz, cap));
          HIP_TRY(c, hold(c->d_linfo, nl));
          HIP_TRY(c, hold(c->d_lover, nl));
          HIP_TRY(c, hold(c->d_dlist, nl * 8 + 16));
          // + the waves' chunk slack: every wave of the inspector may leave most of a 2048-entry reservation unused
          const size_t over_slack = (std::min<size_t>(16384, ((size_t)n + 15) / 16 * 4) + 2) * 2048;
          HIP_TRY(c, hold(c->d_dover, std::max<size_t>(c->d_dover.cap, nl * 2 + over_slack)));
          sa.pz = red->d_rows;
          sa.zindex = hinted ? u.d_red_index : nullptr;
          sa.excess = hinted ? (inspect_only ? c->d_excess_b.p : c->d_excess.p) : nullptr;
          sa.range_of = hinted ? u.d_range_of : nullptr;
          sa.bad = flagblk + 1;
          sa.ie_bias = 8;
          sa.ie_rule = g.method == 1 ? 1 : 0;   // the bound filter and the quad kernel want overlap lists
          if (rcp) {   // straight into the recipe of the kept set (absolute row = cb + i)
            sa.rowz = rcp->rowz.p + cb;
            sa.linfo = rcp->linfo.p + (size_t)cb * g.method;
            sa.lover = rcp->lover.p + (size_t)cb * g.method;
            sa.slot = rcp->slot.p + (size_t)cb * g.method * 8;
            sa.over = rcp->over.p;
            sa.over_cap = (uint32_t)std::min<size_t>(rcp->over.cap - 16, 0xfffffff0u);
            if (!hit) HIP_TRY(c, hipMemcpyAsync(rcp->row0.p + cb, c->d_row0.p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
          } else {
            sa.rowz = c->d_rowz.p;
            sa.linfo = c->d_linfo.p;
            sa.lover = c->d_lover.p;
            sa.slot = c->d_dlist.p;
            sa.over = c->d_dover.p;
            sa.over_cap = (uint32_t)std::min<size_t>(c->d_dover.cap - 16, 0xfffffff0u);
          }
          sa.ov_count = flagblk + 4;
          sa.zoff = (uint32_t)(64 * g.Wp) << 8;
          if (g.method == 2 && one_sided(red)) sa.lz_off = red->d_loff;   // one round per path instead of one per half
          if (!hit) HIP_TRY(c, launch_stats_ie(sa, g.method, st));
          // a kept row's carrier total bounds every count of it: the next level loads only the plane groups that can be non-zero
          if (rcp && !hit) HIP_TRY(c, hipMemcpyAsync(rcp->tot.p + (size_t)cb * g.method, c->d_tot.p, (size_t)n * g.method * 4, hipMemcpyDeviceToDevice, st));
        } else if (!hit) {
          HIP_TRY(c, launch_stats(sa, g.method, st));
        }
        if (!hit) {
          HIP_TRY(c, hipEventRecord(e1, st));
          if (!inspect_only) HIP_TRY(c, hipEventRecord(c->ev_insp_main, st));
          c->ev_stats.emplace_back(e0, e1);
          if (ci) {
            ci->inspected = true;
            ci->with_lists = use_ie;
            ci->in_recipe = use_ie && rcp != nullptr;
          }
        }
        bool win_from_cache = false;
        if (hit && ci->win_valid && scored) {   // the chunk's top-k does not depend on the masks either
          win = ci->win;
          sel_done = true;
          win_from_cache = true;
        }
        // the top-k selection only needs the keys the inspector just wrote: its digit passes run now, their state
        // comes back with the inspector's flags, its winners are collected before the null kernel starts
        if (use_ie && g.K > 0 && scored && !sel_done) {
          if (c->sel_async) {   // beside the warm-up slice and the null kernel, behind the inspector
            HIP_TRY(c, hipEventRecord(c->ev_sel, st));
            HIP_TRY(c, hipStreamWaitEvent(c->sel_stream, c->ev_sel, 0));
            sel_on = c->sel_stream;
          }
          if (int rc = select_begin(c, s0, s1 - s0, c->top_k, &sel_state, sel_on)) return rc;
          sel_begun = true;
        }
        if (!scored && !(use_ie && g.K > 0)) continue;

        bool ran_sparse = false, redo = false;
        if (g.K > 0 && use_ie) do {
          const auto ti0 = std::chrono::steady_clock::now();
          const uint32_t zoff = (uint32_t)(64 * g.Wp) << 8;
          const int64_t nl = n * g.method;
          uint32_t flags[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // max carriers, hint broken, overlap lists, -, long-list entries
          if (hit) {
            std::memcpy(flags, ci->flags, sizeof flags);   // as the inspector left them (no round trip)
          } else {
            HIP_TRY(c, hipMemcpyAsync(flags, flagblk, 32, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
          }
          if ((size_t)flags[4] + 16 > (rcp ? rcp->over.cap : c->d_dover.cap)) {
            // more long lists than the area holds: size it from what the pass asked for, run the chunk again (a recipe
            // keeps what the earlier chunks wrote; the failed attempt's reservation is simply left unused)
            const size_t want = (size_t)flags[4] + (size_t)flags[4] / 4 + ((size_t)1 << 24);
            if (rcp) HIP_TRY(c, rcp->over.grow_keep(want, rcp->over.cap, st));
            else HIP_TRY(c, c->d_dover.reserve(want));
            redo = true;
            if (ci) ci->inspected = false;
            break;
          }
          const uint64_t n_list = (uint64_t)nl * 8 + flags[4];
          if (rcp) over_next = std::max(over_next, flags[4]);
          const uint32_t max_tot = flags[0];
          join_max_tot = std::max(join_max_tot, max_tot);
          join_max_len = std::max(join_max_len, flags[5]);
          if (flags[1] != 0 || (hinted && flags[6] != 0)) {
            // the hint does not describe this join: run it on paths1 itself (identity map) from this chunk on
            if (!hinted) return fail(c, GCRE_ERR_DEVICE, "internal: joined path differs from paths0 | paths1");
            hinted = false;
            if (int rc = prepare_z()) return rc;
            if (!have_pz) want_ie = false;
            // the kept rows' planes were sized for the reduced rows: rows of paths1 may carry more
            if (res_planes && plane_groups_for(std::min<uint32_t>((uint32_t)(64 * g.Wp), row_max(c, jp.p0) + row_max(c, red))) >
                                  jp.res->plane_groups) {
              res_planes = false;
              res_planes_ok = false;
            }
            if (cb > sg.b || &sg != &segs.front()) recipe_broken = true;   // earlier chunks added rows of another set
            if (cb > sg.b || &sg != &segs.front()) hint_broke_late = true;
            redo = true;
            if (ci) ci->inspected = false;
            break;
          }
          if (ci && !hit) {
            std::memcpy(ci->flags, flags, sizeof flags);
            ci->flags_valid = true;
          }
          if (inspect_only) {   // nothing of the permutation kernel's; the chunk's winners are collected below
            join_max_tot = std::max(join_max_tot, flags[0]);
            if (sel_begun && scored && !sel_done) {
              HIP_TRY(c, hipStreamSynchronize(sel_on));
              uint32_t nsel = 0;
              if (int rc = select_finish(c, s0, s1 - s0, c->top_k, sel_state, &nsel, sel_on)) return rc;
              if (int rc = queue_winners(s0, nsel, win, sel_on)) return rc;
              sel_done = true;
            }
            ran_sparse = true;
            break;
          }
          const int64_t nseg_est = std::max<int64_t>(uids_in(cb, n), 1);
          if (c->null_kernel == 0 && scored) {
            // auto: dense bit vectors (or tiny K) are cheaper on the AND+BCNT kernel (DESIGN.md "Kernel choice")
            const double base = have_p0 ? 0.0 : (double)row_max(c, jp.p0) * g.method * (double)nseg_est / (double)n;
            const double entries = (double)n_list / (double)n + base + 10.0 * g.method;
            const double ie_cost = entries * nkt_sp * 15.0;
            const double dense_cost = 2.0 * g.Wp * g.method * (double)g.K * 2.0 / 65.0;
            if (ie_cost >= dense_cost) { res_planes_ok = false; break; }
          }
          if (!scored && !(res_planes && cb < pl_e && cb + n > pl_b)) {   // rows of another shard that only needed their recipe entries
            ran_sparse = true;
            break;
          }
          int planes = 5;
          while (planes < 16 && (max_tot >> planes) != 0) planes++;
          IeArgs ia{};
          int64_t nseg_scored = 0;
          gcre_uids::SegCache* seg_entry = nullptr;
          if (int rc = sparse_segments(c, u, cb, n, cb + s0, cb + s1, res_planes ? pl_b : cb + s0, res_planes ? pl_e : cb + s1,
                                       &ia.segs, &ia.nsegs, &nseg_scored, &seg_entry))
            return rc;
          ia.mt = w_mt;
          ia.tot = c->d_tot.p;
          ia.rowz = rcp ? rcp->rowz.p + cb : c->d_rowz.p;
          ia.planes0 = (have_p0 && !use_rec) ? jp.p0->d_planes : nullptr;
          ia.g0 = (have_p0 && !use_rec) ? jp.p0->plane_groups : 0;
          ia.planesz = have_pz ? red->d_planes : nullptr;
          ia.gz = have_pz ? red->plane_groups : 0;
          ia.rows0 = (uint32_t)(jp.p0->nrows * g.method);
          ia.rowsz = (uint32_t)(red->nrows * g.method);
          ia.rows_out = res_planes ? (uint32_t)(jp.res->nrows * g.method) : 0u;
          ia.loff0 = jp.p0->d_loff;
          ia.lidx0 = jp.p0->d_lidx;
          ia.linfo = rcp ? rcp->linfo.p + (size_t)cb * g.method : c->d_linfo.p;
          ia.lover = rcp ? rcp->lover.p + (size_t)cb * g.method : c->d_lover.p;
          ia.dlist = rcp ? rcp->slot.p + (size_t)cb * g.method * 8 : c->d_dlist.p;
          ia.dover = rcp ? rcp->over.p : c->d_dover.p;
          if (use_rec) {
            const gcre_recipe* r0 = jp.p0->rec;
            ia.rec_row0 = r0->row0.p;
            ia.rec_rowz = r0->rowz.p;
            ia.rec_linfo = r0->linfo.p;
            ia.rec_lover = r0->lover.p;
            ia.rec_slot = r0->slot.p;
            ia.rec_over = r0->over.p;
            ia.rec_planes_a = rec_a->d_planes;
            ia.rec_planes_z = rec_z->d_planes;
            ia.rec_rows_a = (uint32_t)(rec_a->nrows * g.method);   // row-halves
            ia.rec_rows_z = (uint32_t)(rec_z->nrows * g.method);
            ia.rec_ga = rec_a->plane_groups;
            ia.rec_gz = rec_z->plane_groups;
            if (g.method == 1) {
              // the recipe entries of every segment's row, next to the segment table (no load depends on row0 any more)
              HIP_TRY(c, c->d_rec_segs.reserve((size_t)std::max<int64_t>(ia.nsegs, 1) * kRecSegWords));
              HIP_TRY(c, launch_fill_rec_segs(ia.segs, ia.nsegs, r0->row0.p, r0->rowz.p, r0->linfo.p, r0->lover.p, r0->slot.p, r0->tot.p,
                                              c->d_rec_segs.p, st));
              ia.rec_segs = c->d_rec_segs.p;
            }
          }
          ia.t32 = c->d_t32;
          ia.d64 = c->d_dmax;
          ia.ladder = c->d_ladder;
          ia.ladder_stride = g.TD;
          ia.lad_mode = !scored ? 1 : (c->ie_prune ? 0 : 2);
          ia.g00_rows = c->g00_rows;
          ia.null_bits = w_null;
          ia.planes_out = res_planes ? jp.res->d_planes : nullptr;
          ia.go = res_planes ? jp.res->plane_groups : 0;
          ia.out_first = cb;
          ia.score_begin = (uint32_t)s0;
          ia.score_end = (uint32_t)s1;
          ia.score_segs = (uint32_t)nseg_scored;
          ia.nkt = nkt_sp;
          ia.K = g.K;
          ia.mt_rows = (uint32_t)(64 * g.Wp + 1);
          ia.zoff = zoff;
          c->prof.null_row_loads += ((have_p0 ? 0.0 : (double)row_max(c, jp.p0) * g.method * (double)nseg_est) + (double)n_list) * nkt_sp;
          int dev_cus = 256;
          (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
          const int wpc = std::min(c->sparse_waves_per_cu, ie_max_waves_per_cu(g.method, planes, ia.gz, ia.planes_out != nullptr, ia.rec_slot != nullptr));
          ia.waves_per_xcd = std::max(4, (dev_cus * wpc / 8 / 4) * 4);
          // small joins: a wave's fixed costs (cold TLB and caches, LDS set-up, threshold exchange) are per tile it
          // visits, so give every wave at least ~32 joined paths of a tile -- counting the tiles a queue can hold whole
          // (a wave walks a contiguous piece of the tile-major sequence)
          const int64_t ie_tile_factor = std::min(std::max(nkt_sp * c->ie_small_join_tiles / 8, 1), nkt_sp);
          while (ia.waves_per_xcd > 4 && n * ie_tile_factor < (int64_t)8 * ia.waves_per_xcd * 32)
            ia.waves_per_xcd = std::max(4, (ia.waves_per_xcd / 2 / 4) * 4);
          c->prof.inspect_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti0).count();
          ia.stats = launch_only ? w_null + Kpad : flagblk + 3;   // 4th word of the flag block: zeroed with it before k_stats ran
          static uint64_t* d_timing = nullptr;   // diagnostics builds only (-DGCRE_IE_TIMING), GCRE_IE_TIMING=1
          const bool timing = std::getenv("GCRE_IE_TIMING") != nullptr;
          if (timing) {
            if (!d_timing) HIP_TRY(c, hipMalloc((void**)&d_timing, 64));
            HIP_TRY(c, hipMemsetAsync(d_timing, 0, 64, st));
            ia.timing = d_timing;
          }
          if (sel_begun && scored && sel_on == st) {   // (its state came back with the flags)
            uint32_t nsel = 0;
            if (int rc = select_finish(c, s0, s1 - s0, c->top_k, sel_state, &nsel, st)) return rc;
            if (int rc = queue_winners(s0, nsel, win, st)) return rc;
            sel_done = true;
          }
          hipEvent_t n0 = get_event(c), n1 = get_event(c);
          bool ie_quad_ran = false;
          HIP_TRY(c, hipEventRecord(n0, st));
          ia.seg_begin = 0;
          ia.seg_end = ia.nsegs;
          if (c->d_ladder && ia.lad_mode == 0) {
            // warm-up: the first segments are scored without pruning (every count looked up, general kernel); what
            // they find seeds the thresholds the pruned kernel starts from.  Joins of a few thousand paths run here whole.
            // (the scored segments lead the table.)  The general kernel is several times slower per path: a short
            // shard gives it an eighth of its segments, not all of them.
            // every tile warms up on its own permutations: with many tiles the slice gets shorter (its cost is per tile)
            int64_t n_warm = warm_segments(c, nseg_scored, nkt_sp);
            // maxima shared with the other devices right after the warm-up: every device warms its share of the slice
            if (jp.exchange && exchanges_done < jp.exchanges && P > 0)
              n_warm = std::min(n_warm, std::max<int64_t>(64, (int64_t)((double)n_warm * (double)(se - sb) / (double)P) + 1));
            if (c->ie_warm_segs == 0) n_warm = 0;   // GCRE_IE_WARM=0 (tests): everything through the pruned kernel, thresholds from 0
            IeArgs wa = ia;
            wa.seg_end = n_warm;
            // a wave walks its tiles one after the other: keep enough waves that each gets about four (segment, tile) items
            while (wa.waves_per_xcd > 4 && n_warm * nkt_sp < (int64_t)8 * wa.waves_per_xcd * c->ie_warm_items)
              wa.waves_per_xcd = std::max(4, (wa.waves_per_xcd / 2 / 4) * 4);
            if (n_warm > 0) HIP_TRY(c, launch_null_ie(wa, g.method, planes, true, st));
            ia.seg_begin = n_warm;
          }
          if (ia.seg_begin < ia.seg_end) {
            ia.queue = c->d_queue;
            ia.batch = c->ie_batch;
            // method 1, no plane output: the quad form (gcre_ieq.hip) wherever segments come in groups that join the same
            // paths1 rows -- every level but the one whose uids are the genes themselves (one uid per pivot)
            bool quad = false;
            if (g.method == 1 && c->d_ladder && c->ie_quad && !ia.planes_out && seg_entry &&
                ensure_quads(c, u, *seg_entry, ia.seg_begin) == GCRE_OK) {
              const int64_t nq = seg_entry->nquads - seg_entry->quad_begin;
              // worth it from 1 + (qmax - 1) / 4 segments per quad on average: 1.25 with two segments per quad (a quad of one
              // segment does what k_null_ie_m1 does with a longer prologue); the 1.5 of the four-segment form could never be
              // reached by joins that average 1.3-1.5
              const int64_t qmax = ieq_quad_segs();
              quad = nq > 0 && ((ia.seg_end - ia.seg_begin) * 4 >= nq * (4 + (qmax - 1)) || c->ie_quad == 2);
              if (std::getenv("GCRE_HOST_TIMING"))
                std::fprintf(stderr, "[host] quads: %lld segments in %lld quads (%.2f per quad), %lld joined paths, quad form %s\n",
                             (long long)(ia.seg_end - ia.seg_begin), (long long)nq, (double)(ia.seg_end - ia.seg_begin) / (double)std::max<int64_t>(nq, 1),
                             (long long)n, quad ? "on" : "off");
              if (quad) {
                ia.quads = seg_entry->d_quads;
                ia.quad_begin = seg_entry->quad_begin;
                ia.quad_end = seg_entry->nquads;
                ia.batch = c->ieq_batch > 0 ? c->ieq_batch : std::max(1, c->ie_batch * 2);   // quads per ticket: the headers of a ticket's quads are fetched one ahead
                const int wq = std::min(c->sparse_waves_per_cu, ieq_max_waves_per_cu(planes, ia.gz, ia.rec_slot != nullptr));
                ia.waves_per_xcd = std::max(4, (dev_cus * wq / 8 / 4) * 4);
                while (ia.waves_per_xcd > 4 && n * ie_tile_factor < (int64_t)8 * ia.waves_per_xcd * 128)
                  ia.waves_per_xcd = std::max(4, (ia.waves_per_xcd / 2 / 4) * 4);
              }
            }
            // tickets are 32-bit: (batches per tile) x tiles must stay below 2^32
            while (((ia.nsegs - ia.seg_begin) / ia.batch + 1) * (int64_t)ia.nkt > (int64_t)0xf0000000ll) ia.batch *= 2;
            // One launch -- or, when the maxima are shared with other devices as the join goes (gcre_join_opts.exchange),
            // E slices with an exchange before each, so that the thresholds of a shard follow the whole level's maxima: the
            // first slices double (1/2^k of the head), the last m = exchange_tail are equal steps 1/(m+1) of the range.
            // Measured at 8 ranks on configs[3]: equal steps at the end and up to 16 exchanges halve the
            // look-ups once more but every slice is a launch that starts cold -- doubling slices alone (m = 0) stay best.
            const bool pruned = c->d_ladder && ia.lad_mode == 0;
            const int n_slices = (pruned && jp.exchange) ? std::max(1, jp.exchanges - exchanges_done) : 1;
            const int64_t r_b = quad ? ia.quad_begin : ia.seg_begin, r_e = quad ? ia.quad_end : ia.seg_end;
            int64_t lo = r_b;
            for (int sl = 0; sl < n_slices && lo < r_e; sl++) {
              int64_t hi = r_e;
              if (sl + 1 < n_slices) {
                const int m = std::min(n_slices - 1, c->exchange_tail < 0 ? n_slices / 2 : c->exchange_tail), head = n_slices - m;
                const double end = sl >= head ? (double)(sl - head + 2) / (double)(m + 1) : std::ldexp(1.0 / (double)(m + 1), -(head - 1 - sl));
                hi = std::min(r_e, std::max(lo + 1, r_b + (int64_t)((double)(r_e - r_b) * end)));
              }
              if (pruned && jp.exchange)
                if (int rc = exchange_now()) return rc;
              IeArgs sa2 = ia;
              if (quad) { sa2.quad_begin = lo; sa2.quad_end = hi; }
              else { sa2.seg_begin = lo; sa2.seg_end = hi; }
              HIP_TRY(c, hipMemsetAsync(sa2.queue, 0, 8 * 16 * 4, st));
              if (quad) HIP_TRY(c, launch_null_ie_quad(sa2, planes, st));
              else HIP_TRY(c, launch_null_ie(sa2, g.method, planes, c->d_ladder == nullptr, st));
              lo = hi;
            }
            if (quad) c->prof.ie_quad_launches++;
            ie_quad_ran = quad;
          }
          HIP_TRY(c, hipEventRecord(n1, st));
          if (sel_begun && scored && !sel_done) {   // own stream: the digit passes ran beside the warm-up slice
            const auto tw0 = std::chrono::steady_clock::now();
            HIP_TRY(c, hipStreamSynchronize(sel_on));
            select_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
            uint32_t nsel = 0;
            if (int rc = select_finish(c, s0, s1 - s0, c->top_k, sel_state, &nsel, sel_on)) return rc;
            if (int rc = queue_winners(s0, nsel, win, sel_on)) return rc;
            sel_done = true;
          }
          if (timing && !launch_only) {
            uint64_t tmv[8] = {0};
            HIP_TRY(c, hipMemcpyAsync(tmv, d_timing, 64, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            const double waves = 8.0 * ia.waves_per_xcd;
            if (g.method == 2)
              std::fprintf(stderr, "[ie2 classes] paths %lld x %d tiles: path-tiles with the other half empty %llu, one test %llu, all steps %llu, both halves %llu; "
                           "lists of 5-8 rows %llu, long lists %llu; permutations looked up %llu\n", (long long)n, ia.nkt, (unsigned long long)tmv[0],
                           (unsigned long long)tmv[1], (unsigned long long)tmv[2], (unsigned long long)tmv[3], (unsigned long long)tmv[4],
                           (unsigned long long)tmv[6], (unsigned long long)tmv[5]);
            else if (ie_quad_ran)
              std::fprintf(stderr, "[ieq timing] paths %lld waves %.0f quads/wave %.0f: per-wave Mcycles header+loads %.2f base counters %.2f intervals %.2f filter pass %.2f exact pass %.2f exchange %.2f total %.2f\n",
                           (long long)n, waves, tmv[7] / waves, tmv[0] / waves / 1e6, tmv[1] / waves / 1e6, tmv[2] / waves / 1e6, tmv[3] / waves / 1e6,
                           tmv[4] / waves / 1e6, tmv[5] / waves / 1e6, tmv[6] / waves / 1e6);
            else
            std::fprintf(stderr, "[ie filter] uncertain path-tiles %llu of %lld x %d tiles, lanes that fetched rows %llu\n",
                         (unsigned long long)tmv[1], (long long)n, ia.nkt, (unsigned long long)tmv[3]);
            std::fprintf(stderr, "[ie timing] paths %lld out %d waves %.0f: per-wave Mcycles seg %.2f load %.2f comp(incl load) %.2f lookup %.2f exch %.2f total %.2f, slowest wave %.2f\n",
                         (long long)n, ia.planes_out != nullptr, waves, tmv[0] / waves / 1e6, tmv[1] / waves / 1e6,
                         tmv[2] / waves / 1e6, tmv[3] / waves / 1e6, tmv[4] / waves / 1e6, tmv[5] / waves / 1e6, tmv[6] / 1e6);
          }
          c->ev_null.emplace_back(n0, n1);
          if (scored) {
            c->prof.null_kernel_launches++;
            c->prof.null_alg_bytes += alg_bytes(cb + s0, s1 - s0);
          }
          c->prof.ie_launches++;
          c->prof.ie_overlap_lists += flags[2];
          ie_stat_pending = true;
          ie_ran = true;
          ran_sparse = true;
        } while (false);
        if (redo) {
          // the abandoned digit passes may still be reading the keys the relaunched inspector is about to rewrite
          if (sel_begun && sel_on != st) {
            HIP_TRY(c, hipEventRecord(c->ev_sel_done, sel_on));
            HIP_TRY(c, hipStreamWaitEvent(st, c->ev_sel_done, 0));
          }
          next = cb;   // same chunk again, now against paths1 itself
          continue;
        }
        if (partial && g.K > 0 && !ran_sparse) {   // priced out of the inclusion-exclusion form: the other kernels score whole chunks
          split = true;
          next = cb;
          continue;
        }
        if (!scored) continue;

        if (g.K > 0 && use_sparse && inspect_only && ci) {
          // the delta-streaming road sizes its counters from the inspector's flag block: an ahead inspection reads it now and
          // leaves it with the chunk (the launch that replays the chunk finds the block cleared)
          uint32_t flags[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          HIP_TRY(c, hipMemcpyAsync(flags, flagblk, 32, hipMemcpyDeviceToHost, st));
          HIP_TRY(c, hipStreamSynchronize(st));
          std::memcpy(ci->flags, flags, sizeof flags);
          ci->flags_valid = true;
        }
        if (g.K > 0 && use_sparse && !inspect_only) do {
          // inspector (once per chunk, shared by all permutation tiles): per joined path the bits paths1 adds
          // on top of paths0 -> offsets by a device scan of the counts k_stats left, entries by k_delta_fill
          if (int rc = ensure_lists(c, jp.p0)) return rc;
          if (int rc = ensure_lists(c, jp.p1)) return rc;
          const uint32_t zoff = (uint32_t)(64 * g.Wp) << 8;
          const int64_t nl = n * g.method;   // delta lists: one per joined path and half
          HIP_TRY(c, c->d_doff.reserve((size_t)nl + 1));
          HIP_TRY(c, c->d_scan.reserve((size_t)(nl + 1023) / 1024 + 2));
          HIP_TRY(c, launch_scan_u32_u64(c->d_dcnt.p, nl, c->d_doff.p, c->d_scan.p, st));
          uint32_t max_tot = 0;
          uint64_t n_delta = 0;
          HIP_TRY(c, hipMemcpyAsync(&max_tot, flagblk, 4, hipMemcpyDeviceToHost, st));
          HIP_TRY(c, hipMemcpyAsync(&n_delta, c->d_doff.p + nl, 8, hipMemcpyDeviceToHost, st));
          HIP_TRY(c, hipStreamSynchronize(st));
          if (hit) max_tot = ci->flags[0];   // the flag block was cleared for this window; the inspector's value was kept
          else if (ci) { ci->flags[0] = max_tot; ci->flags_valid = true; }
          if (c->null_kernel == 0) {
            // auto: price both forms for this chunk (DESIGN.md "Kernel choice").  Sparse: one mask-row load per list
            // entry per 2048-permutation tile at ~15 CU-cycles each; dense: 2 VALU ops per dword per permutation at
            // ~65 lane-ops/clk/CU.  Base lists are bounded by the largest carrier total, once per segment.
            const int64_t nseg_est = std::max<int64_t>(uids_in(cb, n), 1);
            const double entries = (double)n_delta / (double)n + (double)max_tot * g.method * (double)nseg_est / (double)n;
            const double sparse_cost = entries * ((g.K + kSparseTile - 1) / kSparseTile) * 15.0;
            const double dense_cost = 2.0 * g.Wp * g.method * (double)g.K * 2.0 / 65.0;
            if (sparse_cost >= dense_cost) break;   // dense kernel below
          }
          if (ci) ci->with_lists = false;   // the delta lists below take the place of the inspector's
          HIP_TRY(c, c->d_dlist.reserve((size_t)n_delta + 16));
          HIP_TRY(c, launch_delta_fill((const uint32_t*)jp.p0->d_rows, 2 * g.S, 2 * g.Wp, g.method, c->d_row0.p,
                                       c->d_row1.p, n, jp.p1->d_loff, jp.p1->d_lidx, c->d_doff.p, zoff,
                                       c->d_dlist.p, st));
          int planes = 5;
          while (planes < 16 && (max_tot >> planes) != 0) planes++;
          SparseArgs sp{};
          int64_t nseg_scored = 0;
          if (int rc = sparse_segments(c, u, cb, n, cb, cb + n, cb, cb + n, &sp.segs, &sp.nsegs, &nseg_scored)) return rc;
          sp.mt = w_mt;
          sp.tot = c->d_tot.p;
          sp.loff0 = jp.p0->d_loff;
          sp.lidx0 = jp.p0->d_lidx;
          sp.doff = c->d_doff.p;
          sp.dlist = c->d_dlist.p;
          sp.t32 = c->d_t32;
          sp.d64 = c->d_dmax;
          sp.null_bits = w_null;
          sp.nkt = (g.K + kSparseTile - 1) / kSparseTile;
          sp.mt_rows = (uint32_t)(64 * g.Wp + 1);
          sp.zoff = zoff;
          {
            // mask-row loads of this launch: every segment walks its paths0 list(s) once, every joined path its
            // delta list(s), for each permutation tile
            const auto& pi = u.h_path_idx;
            const auto& lo = jp.p0->h_loff;
            double base_entries = 0;
            int64_t i = std::upper_bound(pi.begin(), pi.end(), cb) - pi.begin() - 1;
            for (; i < u.n_uids && pi[(size_t)i] < cb + n; i++) {
              const int64_t a0 = std::max(pi[(size_t)i], cb), a1 = std::min(pi[(size_t)i + 1], cb + n);
              if (a1 <= a0) continue;
              const double segs_here = (double)((a1 - a0 + kSparseSegMax - 1) / kSparseSegMax);
              base_entries += segs_here * (double)(lo[(size_t)(i + 1) * g.method] - lo[(size_t)i * g.method]);
            }
            c->prof.null_row_loads += (base_entries + (double)n_delta) * sp.nkt;
          }
          int dev_cus = 256;
          (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
          if (const char* e = std::getenv("GCRE_SPARSE_ABLATE")) sp.ablate = std::atoi(e);
          const int wpc = std::min(c->sparse_waves_per_cu, sparse_max_waves_per_cu(g.method, planes));
          sp.waves_per_xcd = std::max(4, (dev_cus * wpc / 8 / 4) * 4);
          hipEvent_t n0 = get_event(c), n1 = get_event(c);
          HIP_TRY(c, hipEventRecord(n0, st));
          HIP_TRY(c, launch_null_sparse(sp, g.method, planes, st));
          HIP_TRY(c, hipEventRecord(n1, st));
          c->ev_null.emplace_back(n0, n1);
          c->prof.null_kernel_launches++;
          c->prof.null_alg_bytes += alg_bytes(cb, n);
          ran_sparse = true;
        } while (false);
        if (g.K > 0 && !ran_sparse && !inspect_only) {
          NullArgs na{};
          na.p0 = (const uint32_t*)jp.p0->d_rows;
          na.p1 = (const uint32_t*)jp.p1->d_rows;
          na.masks = w_masks;
          na.row0 = c->d_row0.p;
          na.row1 = c->d_row1.p;
          na.tot = c->d_tot.p;
          na.t32 = c->d_t32;
          na.d64 = c->d_dmax;
          na.null_bits = w_null;
          na.npaths = n;
          na.npt = npt;
          na.S32 = 2 * g.S;
          na.W32p = 2 * g.Wp;
          na.Kpad = Kstride;
          na.nkt = (g.K + cfg.perm_tile - 1) / cfg.perm_tile;
          int dev_cus = 256;
          (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device);
          const int64_t want = (int64_t)dev_cus * c->null_blocks_per_cu;
          int64_t groups = std::max<int64_t>(1, want / na.nkt);
          groups = std::min<int64_t>(groups, npt);
          na.pgroups = (int)groups;
          hipEvent_t n0 = get_event(c), n1 = get_event(c);
          HIP_TRY(c, hipEventRecord(n0, st));
          HIP_TRY(c, launch_null(na, g.method, cfg, st));
          HIP_TRY(c, hipEventRecord(n1, st));
          c->ev_null.emplace_back(n0, n1);
          c->prof.null_kernel_launches++;
          c->prof.null_alg_bytes += alg_bytes(cb, n);
        }

        // ---- top-k of this chunk ----
        const auto ts0 = std::chrono::steady_clock::now();
        if (!sel_done) {
          uint32_t nsel = 0;
          if (sel_begun) {   // begun, then the chunk left the inclusion-exclusion road: its state is still good
            HIP_TRY(c, hipStreamSynchronize(sel_on));
            if (int rc = select_finish(c, s0, s1 - s0, c->top_k, sel_state, &nsel, sel_on)) return rc;
          } else {
            if (c->sel_stream) HIP_TRY(c, hipStreamSynchronize(c->sel_stream));   // an abandoned selection shares the scratch
            int rc = select_chunk(c, s0, s1 - s0, c->top_k, &nsel, st);
            if (rc != GCRE_OK) return rc;
          }
          if (int rc2 = queue_winners(s0, nsel, win, sel_on)) return rc2;
          // whatever still runs on the selection stream reads this chunk's keys and rows: the next chunk's inspector (on the
          // main stream) rewrites them, so it queues behind an event -- not behind "the results are discarded anyway"
          if (sel_on != st) {
            HIP_TRY(c, hipEventRecord(c->ev_sel_done, sel_on));
            HIP_TRY(c, hipStreamWaitEvent(st, c->ev_sel_done, 0));
          }
        }
        if (win.n > 0) {
          const auto tw0 = std::chrono::steady_clock::now();
          if (!win_from_cache) {   // (cached winners are host data: nothing to wait for)
            if (sel_on != st) HIP_TRY(c, hipStreamSynchronize(sel_on));
            HIP_TRY(c, hipStreamSynchronize(st));   // the null kernel of this chunk: its time is not the selection's
          }
          select_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
          for (uint32_t i = 0; i < win.n; i++)
            cands.push_back(Candidate{key_to_score(win.key[i]), cb + s0 + (int64_t)win.sel[i], (int32_t)win.r0[i], (int32_t)win.r1[i],
                                      (int32_t)win.cases[i], (int32_t)win.ctrls[i]});
        }
        if (ci && !ci->win_valid) {   // (its copies have arrived: the stream was waited for above)
          ci->win = win;
          ci->win_valid = true;
        }
        select_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count();
        if (!inspect_only) c->prof.paths += s1 - s0;   // (a launch books them on its own profile: prof_swap)
      }
    }
    if (mode == kFull) collect_ie_stat();
    if (rcp && want_ie && !recipe_broken) {
      // the recipe names its operands by id and row version: the set paths0 was, and the rows the join really added
      rcp->a_id = jp.p0->id;
      rcp->a_ver = jp.p0->version;
      rcp->z_id = red->id;
      rcp->z_ver = red->version;
      rcp->valid = true;
      rcp->max_len = join_max_len;
      jp.res->max_bits = (join_max_tot + 3u) & ~3u;
      jp.res->max_known = true;
    }
    keep_planes_done = res_planes && res_planes_ok && g.K > 0;
    keep_planes_lo = pl_b;
    keep_planes_hi = pl_e;
    keep_max_tot = join_max_tot;
    if (c->insp_cache) {
      u.insp_hinted = hinted;
      u.insp_res_ver = keep ? jp.res->version : 0;
      u.insp_valid = !hint_broke_late;
    }
    if (ie_ran && hinted) c->prof.ie_hinted_joins++;
    if (ie_ran && have_p0) c->prof.ie_plane_joins++;
  }

  if (inspect_only) {
    // everything this inspection queued is behind this event: the join proper waits for it on the main stream
    HIP_TRY(c, hipEventRecord(c->ev_insp_done, st));
    return GCRE_OK;
  }
  if (keep_planes_done) {
    // the kept rows leave with their count planes: the next level's N0 (gcre_ie.hip)
    jp.res->planes_epoch = c->mask_epoch;
    jp.res->planes_valid = true;
    jp.res->planes_lo = keep_planes_lo;
    jp.res->planes_hi = keep_planes_hi;
    jp.res->max_bits = (keep_max_tot + 3u) & ~3u;
    jp.res->max_known = true;
  }

  // a join that took another road (small, unpruned, another kernel form) still makes the calls the other devices expect
  while (jp.exchange && exchanges_done < jp.exchanges)
    if (int rc = exchange_now()) return rc;

  if (launch_only) {
    // launched: the kernels are queued, nothing is waited for.  What the join's own call will need stays with the index
    gcre_uids::Launched& L = u.launch;
    HIP_TRY(c, hipEventRecord(L.done, st));
    L.cands = std::move(cands);
    L.key = ikey;
    L.res_ver = keep ? jp.res->version : 0;
    L.mask_epoch = c->mask_epoch;
    L.win_k0 = c->win_k0;
    L.win_K = c->win_K;
    L.active = true;
    return GCRE_OK;
  }

  // ---- null maxima: first K entries, f32 (methods.h:101-102; format_result, join_base.cpp:144-146) ----
  out->n_perm = g.K;
  out->null_max = (float*)std::calloc((size_t)std::max(g.K, 1), sizeof(float));
  if (g.K > 0) {
    HIP_TRY(c, hipMemcpyAsync(out->null_max, w_null, (size_t)g.K * 4, hipMemcpyDeviceToHost, st));
    if (jp.d_null_out) HIP_TRY(c, hipMemcpyAsync(jp.d_null_out, w_null, (size_t)g.K * 4, hipMemcpyDeviceToDevice, st));
  }
  if (ahead_plan) {
    // this join's copies are queued: the chain goes behind them, and only they are waited for
    HIP_TRY(c, hipEventRecord(c->ev_tail, st));
    if (int rc = run_chain()) return rc;
    HIP_TRY(c, hipEventSynchronize(c->ev_tail));
  } else {
    HIP_TRY(c, hipStreamSynchronize(st));
  }

  merge_candidates(cands, c->top_k, out);

  c->prof.null_kernel_ms = drain_events(c, c->ev_null);
  c->prof.stats_kernel_ms = drain_events(c, c->ev_stats);
  c->prof.select_ms = std::max(0.0, select_ms - select_wait_ms);
  c->prof.scores = c->prof.paths * (int64_t)g.K;
  c->prof.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return GCRE_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int gcre_abi_version(void) { return GCRE_ABI_VERSION; }
#ifndef GCRE_BUILD_FLAGS
#define GCRE_BUILD_FLAGS ""
#endif
const char* gcre_build_flags(void) { return GCRE_BUILD_FLAGS; }
int gcre_device_count(void) {
  int n = 0;
  return (hipGetDeviceCount(&n) == hipSuccess && n > 0) ? n : 0;
}

const char* gcre_last_error(const gcre_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

gcre_ctx* gcre_create(int method, int n_cases, int n_ctrls, int iterations, int device) {
  // check_true(num_cases > 0 && num_ctrls > 0 && iters >= 0), join_base.cpp:47
  if ((method != 1 && method != 2) || n_cases <= 0 || n_ctrls <= 0 || iterations < 0) {
    g_create_error = "assertion: need method in {1,2}, num_cases > 0, num_ctrls > 0, iterations >= 0";
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_error = "no HIP device: libgcre_hip has no CPU fallback";
    return nullptr;
  }
  if (device < 0 || device >= ndev) {
    g_create_error = "device index out of range";
    return nullptr;
  }
  hipDeviceProp_t prop;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
    g_create_error = "cannot select HIP device";
    return nullptr;
  }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName;
    return nullptr;
  }
  auto* c = new gcre_ctx();
  c->device = device;
  Geometry& g = c->g;
  g.method = method;
  g.n = n_cases + n_ctrls;
  g.n_cases = n_cases;
  g.W = (g.n + 63) / 64;
  g.Wp = ((g.W + 3) / 4) * 4;
  g.S = g.Wp * method;
  g.K = iterations;
  g.Kpad = ((iterations + kPermTileMax - 1) / kPermTileMax) * kPermTileMax;
  g.TD = 64 * g.Wp + 1;
  c->win_k0 = 0;
  c->win_K = g.K;
  if (const char* e = std::getenv("GCRE_QUIET")) c->quiet = std::atoi(e) != 0;
  if (const char* e = std::getenv("GCRE_CHUNK_PATHS")) c->chunk_paths = std::max<long long>(64, std::atoll(e));
  if (const char* e = std::getenv("GCRE_NULL_BLOCKS_PER_CU")) c->null_blocks_per_cu = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("GCRE_NULL_KERNEL"))
    c->null_kernel = !std::strcmp(e, "dense") ? 1 : !std::strcmp(e, "sparse") ? 2 : !std::strcmp(e, "ie") ? 3 : 0;
  if (const char* e = std::getenv("GCRE_IE_PRUNE")) c->ie_prune = std::atoi(e) != 0;
  if (const char* e = std::getenv("GCRE_EXCHANGE_TAIL")) c->exchange_tail = std::min(std::max(std::atoi(e), -1), 64);
  if (const char* e = std::getenv("GCRE_IE_WARM")) c->ie_warm_segs = std::min(std::max(std::atoi(e), 0), 1 << 20);
  if (const char* e = std::getenv("GCRE_IE_WARM_ITEMS")) c->ie_warm_items = std::min(std::max(std::atoi(e), 1), 64);
  if (const char* e = std::getenv("GCRE_IE_SJT")) c->ie_small_join_tiles = std::atoi(e);
  if (const char* e = std::getenv("GCRE_IE_BATCH")) c->ie_batch = std::min(std::max(std::atoi(e), 1), 4096);
  if (const char* e = std::getenv("GCRE_IEQ_BATCH")) c->ieq_batch = std::max(0, std::atoi(e));
  if (const char* e = std::getenv("GCRE_IE_QUAD")) c->ie_quad = std::min(std::max(std::atoi(e), 0), 2);   // 2: wherever it can run
  if (const char* e = std::getenv("GCRE_PLANES_OUT_MAX_MB")) c->planes_out_max = (size_t)std::max(0ll, std::atoll(e)) << 20;
  if (const char* e = std::getenv("GCRE_SPARSE_WAVES_PER_CU")) c->sparse_waves_per_cu = std::max(1, std::atoi(e));

  bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&c->sel_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&c->insp_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_insp_done, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_insp_main, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) == hipSuccess;
  if (const char* e = std::getenv("GCRE_AHEAD")) c->ahead_on = std::atoi(e) != 0;
  ok = ok && hipEventCreateWithFlags(&c->ev_sel, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_sel_done, hipEventDisableTiming) == hipSuccess;
  if (const char* e = std::getenv("GCRE_SELECT_STREAM")) c->sel_async = std::atoi(e) != 0;
  ok = ok && hipMalloc((void**)&c->d_case_mask, (size_t)g.Wp * 8) == hipSuccess;
  ok = ok && hipMalloc((void**)&c->d_max_tot, 32) == hipSuccess;
  ok = ok && hipMalloc((void**)&c->d_max_tot_b, 32) == hipSuccess;
  ok = ok && hipMalloc((void**)&c->d_queue, 8 * 16 * 4) == hipSuccess;
  if (ok && g.Kpad > 0) {
    ok = hipMalloc((void**)&c->d_masks, (size_t)2 * g.Wp * g.Kpad * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_null, (size_t)g.Kpad * 4) == hipSuccess;
    ok = ok && hipMemsetAsync(c->d_masks, 0, (size_t)2 * g.Wp * g.Kpad * 4, c->stream) == hipSuccess;
  }
  if (ok) {
    // cases are patient columns 0..n_cases-1, join_base.cpp:50-54
    std::vector<uint64_t> cm((size_t)g.Wp, 0);
    for (int k = 0; k < n_cases; k++) cm[(size_t)k / 64] |= uint64_t(1) << (k % 64);
    ok = hipMemcpy(c->d_case_mask, cm.data(), cm.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
  }
  if (!ok) {
    g_create_error = "device allocation failed in gcre_create";
    gcre_destroy(c);
    return nullptr;
  }
  return c;
}

void gcre_destroy(gcre_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->insp_stream) (void)hipStreamSynchronize(c->insp_stream);
  drop_ahead(c);
  // path sets and join indices the caller did not free: their rows, lists, count planes, recipes and segment tables go
  // with the context (their handles are invalid from here on, include/gcre_hip.h)
  while (!c->live_uids.empty()) free_uids(c->live_uids.back());
  {
    std::vector<const gcre_pathset*> sets;
    for (const auto& kv : c->live_sets) sets.push_back(kv.second);
    for (const gcre_pathset* ps : sets) gcre_pathset_free(const_cast<gcre_pathset*>(ps));
  }
  for (void* p : {(void*)c->d_case_mask, (void*)c->d_masks, (void*)c->d_t32, (void*)c->d_dvt, (void*)c->d_dmax,
                  (void*)c->d_null, (void*)c->d_mt, (void*)c->d_max_tot, (void*)c->d_max_tot_b, (void*)c->d_queue, (void*)c->d_ladder})
    if (p) (void)hipFree(p);
  for (auto* b : {&c->d_row0, &c->d_row1, &c->d_tot, &c->d_cases, &c->d_ctrls, &c->d_sel, &c->d_small, &c->d_chunk,
                  &c->d_rec_segs, &c->d_wcases, &c->d_wctrls, &c->d_wrow0, &c->d_wrow1})
    b->release();
  c->d_key.release();
  c->d_hub_null.release();
  c->d_wkey.release();
  c->d_doff.release();
  c->d_scan.release();
  c->d_excess.release();
  c->d_excess_b.release();
  c->d_dcnt.release();
  c->d_dlist.release();
  c->d_rowz.release();
  c->d_linfo.release();
  c->d_lover.release();
  c->d_dover.release();
  for (auto& pb : c->plane_pool) (void)hipFree(pb.p);
  c->plane_pool.clear();
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->sel_stream) (void)hipStreamSynchronize(c->sel_stream);
  if (c->sel_stream) (void)hipStreamDestroy(c->sel_stream);
  if (c->insp_stream) (void)hipStreamDestroy(c->insp_stream);
  if (c->ev_insp_done) (void)hipEventDestroy(c->ev_insp_done);
  if (c->ev_insp_main) (void)hipEventDestroy(c->ev_insp_main);
  if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
  if (c->ev_sel) (void)hipEventDestroy(c->ev_sel);
  if (c->ev_sel_done) (void)hipEventDestroy(c->ev_sel_done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int gcre_set_top_k(gcre_ctx* c, int top_k) {
  if (!c) return GCRE_ERR_ARG;
  if (top_k < 1) return fail(c, GCRE_ERR_ARG, "top_k must be >= 1");
  c->top_k = top_k;
  return GCRE_OK;
}

int gcre_width_ul(const gcre_ctx* c) { return c ? c->g.W : GCRE_ERR_ARG; }
int gcre_vlen(const gcre_ctx* c) { return c ? c->g.W * c->g.method : GCRE_ERR_ARG; }

int gcre_set_value_table(gcre_ctx* c, const double* table, int nrow, int ncol, int col_major) {
  HostTimer ht("set_value_table");
  if (!c || (!table && nrow > 0 && ncol > 0) || nrow < 0 || ncol < 0) return fail(c, GCRE_ERR_ARG, "bad value table");
  (void)hipSetDevice(c->device);
  const Geometry& g = c->g;
  const int n = g.n;
  // the reference pads the table to (n+1)x(n+1) with -1 (join_base.cpp:67-78); the device keeps it diagonal-major
  // (and covers carrier counts up to the padded word width with -1).  The repacking runs on the device: the raw table
  // goes up as it is (k_table_to_diag, gcre_frontend.hip) -- on the host it is 10 s of cache misses at 50,000 patients.
  const size_t TD = (size_t)g.TD, NT = tri(TD);
  for (void** p : {(void**)&c->d_dvt, (void**)&c->d_t32, (void**)&c->d_dmax})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  double* d_raw = nullptr;
  const size_t raw = (size_t)nrow * (size_t)ncol;
  HIP_TRY(c, hipMalloc((void**)&c->d_dvt, NT * 8));
  if (g.method == 1) HIP_TRY(c, hipMalloc((void**)&c->d_t32, NT * 4));
  else HIP_TRY(c, hipMalloc((void**)&c->d_dmax, NT * 8));
  HIP_TRY(c, hipMalloc((void**)&d_raw, std::max<size_t>(raw, 1) * 8));
  hipError_t e = hipSuccess;
  if (raw) e = hipMemcpyAsync(d_raw, table, raw * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess)
    e = launch_table_to_diag(d_raw, nrow, ncol, col_major, n, (int)TD, c->d_dvt, c->d_t32, c->d_dmax, c->stream);
  if (e == hipSuccess && TD <= 65536) {   // counts fit the 16-bit bounds of a ladder entry
    if (!c->d_ladder) e = hipMalloc((void**)&c->d_ladder, (size_t)(kLadder2Levels + 2) * TD * 4);
    if (e == hipSuccess)
      e = g.method == 1 ? launch_build_ladder(c->d_t32, (int)TD, c->d_ladder, c->stream)
                        : launch_build_ladder2(c->d_dmax, (int)TD, c->d_ladder, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_raw);
  c->g00_rows = 0xffffffffu;
  if (e == hipSuccess && g.method == 2) {
    // what an empty half adds to a path's null score: vtmax[0][0], as the device table holds it, rounded UP to ladder rows
    double g00 = 0;
    e = hipMemcpy(&g00, c->d_dmax, 8, hipMemcpyDeviceToHost);
    if (!(g00 != g00)) {
      const double x = g00 > 0 ? std::ceil(g00 * (double)(2 * kLadderPerUnit)) : 0.0;   // (times 16: exact)
      if (x < 1e9) c->g00_rows = (uint32_t)x;
    }
  }
  if (e != hipSuccess) return fail(c, GCRE_ERR_DEVICE, std::string("set_value_table: ") + hipGetErrorString(e));
  c->have_table = true;
  c->obs_epoch++;
  return GCRE_OK;
}

// GCRE_INSPECT_CACHE=0: every join inspects, whatever the caller asked for (cross-check)
static bool inspect_cache_allowed() {
  static const bool off = std::getenv("GCRE_INSPECT_CACHE") && std::atoi(std::getenv("GCRE_INSPECT_CACHE")) == 0;
  return !off;
}

int gcre_set_inspect_cache(gcre_ctx* c, int on) {
  if (!c) return GCRE_ERR_ARG;
  c->insp_cache = on != 0 && inspect_cache_allowed();
  if (!c->insp_cache) return gcre_drop_inspections(c, 1);
  return GCRE_OK;
}

int gcre_drop_inspections(gcre_ctx* c, int release_memory) {
  if (!c) return GCRE_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (release_memory && c->stream) (void)hipStreamSynchronize(c->stream);
  drop_ahead(c);
  for (gcre_uids* u : c->live_uids) {
    drop_launch(u);
    u->insp_valid = false;
    if (release_memory) {
      for (auto& ci : u->insp) ci.release();
      u->insp.clear();
    }
  }
  return GCRE_OK;
}

int gcre_set_perm_cases(gcre_ctx* c, const int32_t* perms, int nrow, int ncol, int col_major) {
  if (!c || nrow < 0 || ncol < 0) return fail(c, GCRE_ERR_ARG, "bad permutation matrix");
  (void)hipSetDevice(c->device);
  const Geometry& g = c->g;
  if (g.K == 0) { c->have_perms = true; return GCRE_OK; }
  // too few rows are reused cyclically (join_base.cpp:116-123) -- with zero rows the reference divides by zero
  if (nrow == 0 || !perms) return fail(c, GCRE_ERR_ASSERT, "assertion: iterations > 0 but no permuted cases");
  if (ncol != g.n) return fail(c, GCRE_ERR_ASSERT, "assertion: permuted cases must have num_cases + num_ctrls columns");
  if (nrow > g.K && !c->quiet) std::printf("  ** WARN more permuted cases than iterations, input will be truncated\n");   // :89-90
  if (nrow < g.K && !c->quiet)
    std::printf("  ** WARN not enough permuted cases, some will be reused to match iterations - hope this is for testing!\n");
  const int used = std::min(nrow, g.K);
  static const bool device_pack = std::getenv("GCRE_DEVICE_PACK") && std::atoi(std::getenv("GCRE_DEVICE_PACK")) != 0;
  if (!device_pack) {
    // setPermutedCases (join_base.cpp:85-125) on the host: mask_r = case_mask XOR flipped_r, flipped where the input is
    // not 1 (:103-104); cases are the first n_cases columns (:50-54).  Only the rows that are used are packed (a column-major
    // matrix keeps its row stride); the packed masks take the road of gcre_set_perm_masks (row reuse included).
    HostTimer hp("pack permutations (host)");
    std::vector<uint64_t> packed((size_t)used * g.W, 0);
    const int n_cases = g.n_cases;
    if (col_major) {
      // columns of the full matrix are nrow long: pack all of its rows column block by column block, keep the first `used`
      if (used == nrow) {
        pack_bits_host(perms, nrow, ncol, true, packed.data(), (size_t)g.W, [n_cases](int32_t v, int q) { return (v != 1) != (q < n_cases); });
      } else {
        std::vector<uint64_t> all((size_t)nrow * g.W, 0);
        pack_bits_host(perms, nrow, ncol, true, all.data(), (size_t)g.W, [n_cases](int32_t v, int q) { return (v != 1) != (q < n_cases); });
        std::memcpy(packed.data(), all.data(), packed.size() * 8);
      }
    } else {
      pack_bits_host(perms, used, ncol, false, packed.data(), (size_t)g.W, [n_cases](int32_t v, int q) { return (v != 1) != (q < n_cases); });
    }
    uint64_t* d_in = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d_in, packed.size() * 8));
    hipError_t e = hipMemcpyAsync(d_in, packed.data(), packed.size() * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_masks_from_words(d_in, used, g, c->d_masks, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_in);
    if (e != hipSuccess) return fail(c, GCRE_ERR_DEVICE, std::string("set_perm_cases: ") + hipGetErrorString(e));
    if (int rc = build_transposed_masks(c)) return rc;
    c->have_perms = true;
    return GCRE_OK;
  }
  std::vector<int32_t> rows;
  const int32_t* src = perms;   // row-major: the first `used` rows are contiguous
  if (col_major) {              // R matrix: gather the rows we need
    rows.resize((size_t)used * ncol);
    for (int r = 0; r < used; r++)
      for (int q = 0; q < ncol; q++) rows[(size_t)r * ncol + q] = perms[(size_t)q * nrow + r];
    src = rows.data();
  }
  int32_t* d_in = nullptr;
  const size_t bytes = (size_t)used * ncol * 4;
  HIP_TRY(c, hipMalloc((void**)&d_in, bytes));
  hipError_t e = hipMemcpyAsync(d_in, src, bytes, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = launch_masks_from_ints(d_in, used, ncol, 0, g, c->d_masks, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_in);
  if (e != hipSuccess) return fail(c, GCRE_ERR_DEVICE, std::string("set_perm_cases: ") + hipGetErrorString(e));
  if (int rc = build_transposed_masks(c)) return rc;
  c->have_perms = true;
  return GCRE_OK;
}

int gcre_set_perm_masks(gcre_ctx* c, const uint64_t* masks, int nrow) {
  if (!c || nrow < 0) return fail(c, GCRE_ERR_ARG, "bad mask matrix");
  (void)hipSetDevice(c->device);
  const Geometry& g = c->g;
  if (g.K == 0) { c->have_perms = true; return GCRE_OK; }
  if (nrow == 0 || !masks) return fail(c, GCRE_ERR_ASSERT, "assertion: iterations > 0 but no permuted cases");
  const int used = std::min(nrow, g.K);
  uint64_t* d_in = nullptr;
  HIP_TRY(c, hipMalloc((void**)&d_in, (size_t)used * g.W * 8));
  hipError_t e = hipMemcpyAsync(d_in, masks, (size_t)used * g.W * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = launch_masks_from_words(d_in, used, g, c->d_masks, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_in);
  if (e != hipSuccess) return fail(c, GCRE_ERR_DEVICE, std::string("set_perm_masks: ") + hipGetErrorString(e));
  if (int rc = build_transposed_masks(c)) return rc;
  c->have_perms = true;
  return GCRE_OK;
}

int gcre_generate_perm_masks(gcre_ctx* c, uint64_t seed, const int32_t* stratum, int n_strata) {
  if (!c || (stratum && n_strata < 1)) return fail(c, GCRE_ERR_ARG, "bad strata");
  (void)hipSetDevice(c->device);
  const Geometry& g = c->g;
  if (g.K == 0) { c->have_perms = true; return GCRE_OK; }
  if (!stratum) n_strata = 1;
  // cases are the first n_cases patient columns (join_base.cpp:50-54): cases and size per stratum
  std::vector<uint32_t> cases_in((size_t)n_strata, 0), size_of((size_t)n_strata, 0);
  for (int q = 0; q < g.n; q++) {
    const int s = stratum ? stratum[q] : 0;
    if (s < 0 || s >= n_strata) return fail(c, GCRE_ERR_RANGE, "stratum id out of range");
    size_of[(size_t)s]++;
    if (q < g.n_cases) cases_in[(size_t)s]++;
  }
  uint32_t *d_cases = nullptr, *d_size = nullptr, *d_work = nullptr;
  int32_t* d_str = nullptr;
  hipError_t e = hipMalloc((void**)&d_cases, (size_t)n_strata * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_size, (size_t)n_strata * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_work, (size_t)g.K * n_strata * 2 * 4);
  if (e == hipSuccess && stratum) e = hipMalloc((void**)&d_str, (size_t)g.n * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(d_cases, cases_in.data(), (size_t)n_strata * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_size, size_of.data(), (size_t)n_strata * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && stratum) e = hipMemcpyAsync(d_str, stratum, (size_t)g.n * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess)
    e = launch_generate_masks(seed, g.K, g.n, n_strata, d_str, d_cases, d_size, d_work, 2 * g.Wp, g.Kpad, c->d_masks, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  for (void* p : {(void*)d_cases, (void*)d_size, (void*)d_work, (void*)d_str})
    if (p) (void)hipFree(p);
  if (e != hipSuccess) return fail(c, GCRE_ERR_DEVICE, std::string("generate_perm_masks: ") + hipGetErrorString(e));
  if (int rc = build_transposed_masks(c)) return rc;
  c->have_perms = true;
  return GCRE_OK;
}

int gcre_set_perm_window(gcre_ctx* c, int k0, int k1) {
  if (!c) return GCRE_ERR_ARG;
  const int K = c->g.K;
  if (k0 < 0 || k1 < k0 || k1 > K || (k0 % kSparseTile) != 0 || (k1 != K && (k1 % kSparseTile) != 0))
    return fail(c, GCRE_ERR_ARG, "permutation window must be tile-aligned (2048) and inside [0, iterations]");
  if (k0 != c->win_k0 || k1 - k0 != c->win_K) {
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->win_k0 = k0;
    c->win_K = k1 - k0;
    c->mask_epoch++;   // count planes hold the tiles of one window
  }
  c->win_K_nominal = std::max(c->win_K_nominal, k1 - k0);
  return GCRE_OK;
}

int gcre_plan_perm_window(gcre_ctx* c, const int64_t* set_rows, int n_sets) {
  if (!c || n_sets < 0 || (n_sets > 0 && !set_rows)) return GCRE_ERR_ARG;
  const int K = c->g.K;
  const int nkt = (K + kSparseTile - 1) / kSparseTile;
  if (nkt <= 1 || !sparse_enabled(c)) return K;
  (void)hipSetDevice(c->device);
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return K;
  // sets that store their planes: the ones under the recipe limit (larger kept sets leave a recipe instead, and operands
  // that are not kept are small)
  double rows = 1;
  for (int i = 0; i < n_sets; i++) {
    const double full = (double)std::max<int64_t>(set_rows[i], 0) * c->g.method * 4096.0 * nkt;
    if (full <= (double)c->planes_out_max) rows += (double)std::max<int64_t>(set_rows[i], 0);
  }
  const double per_tile = rows * c->g.method * 4096.0;
  int64_t tiles = (int64_t)((double)free_b * 0.5 / per_tile);
  if (const char* e = std::getenv("GCRE_WINDOW_TILES")) tiles = std::max(1, std::atoi(e));   // tests
  if (tiles >= nkt) return K;
  return (int)std::max<int64_t>(tiles, 1) * kSparseTile;
}

int gcre_get_perm_mask(gcre_ctx* c, int r, uint64_t* out) {
  if (!c || !out) return GCRE_ERR_ARG;
  const Geometry& g = c->g;
  if (r < 0 || r >= g.K) return fail(c, GCRE_ERR_RANGE, "permutation index out of range");
  (void)hipSetDevice(c->device);
  std::vector<uint32_t> col((size_t)2 * g.Wp);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy2D(col.data(), 4, c->d_masks + r, (size_t)g.Kpad * 4, 4, (size_t)2 * g.Wp, hipMemcpyDeviceToHost));
  for (int k = 0; k < g.W; k++) out[k] = (uint64_t)col[(size_t)2 * k] | ((uint64_t)col[(size_t)2 * k + 1] << 32);
  return GCRE_OK;
}

// ---- path sets ----
gcre_pathset* gcre_pathset_zeros(gcre_ctx* c, int64_t nrows) {
  if (!c) return nullptr;
  (void)hipSetDevice(c->device);
  return new_pathset(c, nrows, true);
}

gcre_pathset* gcre_pathset_from_dense(gcre_ctx* c, const int32_t* data, int64_t nrow, int ncol, int col_major) {
  if (!c) return nullptr;
  (void)hipSetDevice(c->device);
  if (nrow < 0 || ncol < 0 || (!data && nrow > 0 && ncol > 0)) {
    fail(c, GCRE_ERR_ARG, "bad dense matrix");
    return nullptr;
  }
  // check_index(data[r].size(), width_ul*64), gcre_paths.h:63: every column must fit the mask words
  if (ncol > c->g.W * 64) {
    fail(c, GCRE_ERR_RANGE, "assertion: more data columns than mask bits");
    return nullptr;
  }
  gcre_pathset* ps = new_pathset(c, nrow, true);
  if (!ps || nrow == 0 || ncol == 0) return ps;
  hipError_t e = hipSuccess;
  static const bool device_pack = std::getenv("GCRE_DEVICE_PACK") && std::atoi(std::getenv("GCRE_DEVICE_PACK")) != 0;
  if (!device_pack) {
    // PathSet::load (gcre_paths.h:56-70) on the host: bit c of row r set iff data[r][c] != 0, (+) half only -- then the
    // packed rows go up (1/32 of the ints; GCRE_DEVICE_PACK=1: the ints go up and k_pack_dense packs them, as before round 4)
    HostTimer hp("pack genotypes (host)");
    std::vector<uint64_t> packed((size_t)nrow * c->g.S, 0);
    pack_bits_host(data, nrow, ncol, col_major != 0, packed.data(), (size_t)c->g.S, [](int32_t v, int) { return v != 0; });
    e = hipMemcpyAsync(ps->d_rows, packed.data(), packed.size() * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // `packed` is a local
  } else {
    int32_t* d_in = nullptr;
    const size_t bytes = (size_t)nrow * ncol * 4;
    e = hipMalloc((void**)&d_in, bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, data, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_pack_dense(d_in, nrow, ncol, col_major, ps->d_rows, c->g.S, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d_in) (void)hipFree(d_in);
  }
  if (e != hipSuccess) {
    fail(c, GCRE_ERR_DEVICE, std::string("pathset_from_dense: ") + hipGetErrorString(e));
    gcre_pathset_free(ps);
    return nullptr;
  }
  return ps;
}

gcre_pathset* gcre_pathset_from_words(gcre_ctx* c, const uint64_t* rows, int64_t nrows) {
  if (!c) return nullptr;
  (void)hipSetDevice(c->device);
  gcre_pathset* ps = new_pathset(c, nrows, true);
  if (!ps || nrows == 0) return ps;
  const Geometry& g = c->g;
  // host rows are [half][W]; device rows are [half][Wp]
  std::vector<uint64_t> dev((size_t)nrows * g.S, 0);
  for (int64_t r = 0; r < nrows; r++)
    for (int h = 0; h < g.method; h++)
      std::memcpy(&dev[(size_t)r * g.S + (size_t)h * g.Wp], &rows[(size_t)r * g.W * g.method + (size_t)h * g.W],
                  (size_t)g.W * 8);
  // same (non-blocking) stream as the zero-fill in new_pathset: a null-stream copy would race with it
  hipError_t e = hipMemcpyAsync(ps->d_rows, dev.data(), dev.size() * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    fail(c, GCRE_ERR_DEVICE, "pathset_from_words: copy failed");
    gcre_pathset_free(ps);
    return nullptr;
  }
  return ps;
}

gcre_pathset* gcre_pathset_select(gcre_ctx* c, const gcre_pathset* from, const int32_t* idx, int64_t n) {
  if (!c || !from || from->ctx != c || n < 0 || (!idx && n > 0)) {
    fail(c, GCRE_ERR_ARG, "bad select arguments");
    return nullptr;
  }
  (void)hipSetDevice(c->device);
  for (int64_t i = 0; i < n; i++)   // check_index(indices[k], size), gcre_paths.h:85
    if (idx[i] < 0 || idx[i] >= from->nrows) {
      fail(c, GCRE_ERR_RANGE, "assertion: select index out of range");
      return nullptr;
    }
  gcre_pathset* ps = new_pathset(c, n, false);
  if (!ps || n == 0) return ps;
  int32_t* d_idx = nullptr;
  hipError_t e = hipMalloc((void**)&d_idx, (size_t)n * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = launch_select(from->d_rows, d_idx, n, c->g.S, ps->d_rows, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (d_idx) (void)hipFree(d_idx);
  if (e != hipSuccess) {
    fail(c, GCRE_ERR_DEVICE, std::string("pathset_select: ") + hipGetErrorString(e));
    gcre_pathset_free(ps);
    return nullptr;
  }
  return ps;
}

int64_t gcre_pathset_size(const gcre_pathset* ps) { return ps ? ps->nrows : GCRE_ERR_ARG; }

int gcre_pathset_read(gcre_ctx* c, const gcre_pathset* ps, uint64_t* out_rows) {
  if (!c || !ps || ps->ctx != c || (!out_rows && ps->nrows > 0)) return fail(c, GCRE_ERR_ARG, "bad read arguments");
  (void)hipSetDevice(c->device);
  const Geometry& g = c->g;
  if (ps->nrows == 0) return GCRE_OK;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  std::vector<uint64_t> dev((size_t)ps->nrows * g.S);
  HIP_TRY(c, hipMemcpy(dev.data(), ps->d_rows, dev.size() * 8, hipMemcpyDeviceToHost));
  for (int64_t r = 0; r < ps->nrows; r++)
    for (int h = 0; h < g.method; h++)
      std::memcpy(&out_rows[(size_t)r * g.W * g.method + (size_t)h * g.W], &dev[(size_t)r * g.S + (size_t)h * g.Wp],
                  (size_t)g.W * 8);
  return GCRE_OK;
}

void gcre_pathset_free(gcre_pathset* ps) {
  HostTimer ht("pathset_free");
  if (!ps) return;
  if (ps->ctx) {
    (void)hipSetDevice(ps->ctx->device);
    if (ps->ctx->stream) (void)hipStreamSynchronize(ps->ctx->stream);
    if (ps->ctx->insp_stream) (void)hipStreamSynchronize(ps->ctx->insp_stream);
    if (ps->ctx->ahead)
      for (const JoinPlan& a : *ps->ctx->ahead)
        if (a.p0 == ps || a.p1 == ps || a.res == ps) { drop_ahead(ps->ctx); break; }
    for (gcre_uids* lu : ps->ctx->live_uids) drop_launch(lu);   // (a launched join may read or write this set)
  }
  if (ps->d_rows) (void)hipFree(ps->d_rows);
  drop_lists(ps);   // lists, and the plane buffer goes to the context's pool
  if (ps->rec) {
    ps->rec->release();
    delete ps->rec;
  }
  if (ps->ctx) ps->ctx->live_sets.erase(ps->id);
  delete ps;
}

int gcre_join(gcre_ctx* c, int path_length, const int32_t* uid_count, const int64_t* uid_location, int64_t n_uids,
              const int32_t* signs, int64_t n_signs, const gcre_pathset* paths0, const gcre_pathset* paths1,
              gcre_pathset* res, const gcre_join_opts* opts, gcre_result* out) {
  if (!c || !out || n_uids < 0 || (n_uids > 0 && (!uid_count || !uid_location)) || n_signs < 0 ||
      (n_signs > 0 && !signs))
    return fail(c, GCRE_ERR_ARG, "bad join arguments");
  (void)hipSetDevice(c->device);
  gcre_uids* u = make_uids(c, path_length, uid_count, uid_location, n_uids, signs, n_signs);
  if (!u) return c->last_code;
  JoinPlan jp{u, paths0, paths1, res, opts && opts->sharded, opts ? opts->shard_begin : 0,
              opts ? opts->shard_end : 0, opts ? opts->d_null_out : nullptr};
  jp.take(opts);
  int rc = run_join(c, jp, out);
  free_uids(u);
  if (rc != GCRE_OK) gcre_result_free(out);
  return rc;
}

gcre_uids* gcre_uids_create(gcre_ctx* c, int path_length, const int32_t* uid_count, const int64_t* uid_location,
                            int64_t n_uids, const int32_t* signs, int64_t n_signs) {
  if (!c || n_uids < 0 || (n_uids > 0 && (!uid_count || !uid_location)) || n_signs < 0 || (n_signs > 0 && !signs)) {
    fail(c, GCRE_ERR_ARG, "bad uids arguments");
    return nullptr;
  }
  (void)hipSetDevice(c->device);
  return make_uids(c, path_length, uid_count, uid_location, n_uids, signs, n_signs);
}

int64_t gcre_uids_total_paths(const gcre_uids* u) { return u ? u->total : GCRE_ERR_ARG; }

int gcre_uids_set_reduced(gcre_uids* u, const gcre_pathset* reduced, const int32_t* index, int64_t n) {
  if (!u) return GCRE_ERR_ARG;
  gcre_ctx* c = u->ctx;
  (void)hipSetDevice(c->device);
  if (u->d_red_index) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(u->d_red_index);
  }
  u->d_red_index = nullptr;
  u->red = nullptr;
  u->n_red_index = 0;
  // what the inspector left for this index was computed under the old hint (the cache key names the reduced SET, not the
  // index contents): a replay would skip the device-side check of the new one
  u->insp_valid = false;
  u->insp.clear();
  if (!reduced) return GCRE_OK;
  if (reduced->ctx != c || !index || n < 0) return fail(c, GCRE_ERR_ARG, "bad reduced operand");
  for (int64_t i = 0; i < n; i++)
    if ((int64_t)((uint32_t)index[i] & 0x7fffffffu) >= reduced->nrows) return fail(c, GCRE_ERR_RANGE, "reduced index out of range");
  if (n == 0) return GCRE_OK;
  HIP_TRY(c, hipMalloc((void**)&u->d_red_index, (size_t)n * 4));
  HIP_TRY(c, hipMemcpyAsync(u->d_red_index, index, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  u->red = reduced;
  u->red_id = reduced->id;
  u->n_red_index = n;
  return GCRE_OK;
}

void gcre_uids_free(gcre_uids* u) {
  if (u && u->ctx) (void)hipSetDevice(u->ctx->device);
  free_uids(u);
}

int gcre_join_uids(gcre_ctx* c, const gcre_uids* uids, const gcre_pathset* paths0, const gcre_pathset* paths1,
                   gcre_pathset* res, const gcre_join_opts* opts, gcre_result* out) {
  if (!c || !uids || !out) return fail(c, GCRE_ERR_ARG, "bad join arguments");
  (void)hipSetDevice(c->device);
  JoinPlan jp{uids, paths0, paths1, res, opts && opts->sharded, opts ? opts->shard_begin : 0,
              opts ? opts->shard_end : 0, opts ? opts->d_null_out : nullptr};
  jp.take(opts);
  int rc = run_join(c, jp, out);
  if (rc != GCRE_OK) gcre_result_free(out);
  return rc;
}

int gcre_join_ahead(gcre_ctx* c, const gcre_uids* uids, const gcre_pathset* paths0, const gcre_pathset* paths1,
                    gcre_pathset* res, const gcre_join_opts* opts) {
  if (!c) return GCRE_ERR_ARG;
  if (!uids) { drop_ahead(c); c->ahead_closed = false; return GCRE_OK; }   // (cancels the registrations)
  if (!c->ahead_on || !c->insp_cache || c->ahead_closed) return GCRE_OK;   // nothing to inspect ahead into: the joins simply run whole
  if (uids->ctx != c || !paths0 || !paths1 || paths0->ctx != c || paths1->ctx != c || (res && res->ctx != c))
    return fail(c, GCRE_ERR_ARG, "uids / path set do not belong to this context");
  if (opts && opts->exchange && opts->exchanges > 0) {
    // a join that exchanges thresholds with other devices runs whole, in its own call -- and so does everything behind it in
    // the sequence (it reads what this one writes)
    c->ahead_closed = true;
    return GCRE_OK;
  }
  if (!c->ahead) c->ahead = new std::vector<JoinPlan>();
  JoinPlan a{uids, paths0, paths1, res, opts && opts->sharded, opts ? opts->shard_begin : 0, opts ? opts->shard_end : 0, nullptr};
  a.take(opts);
  c->ahead->push_back(a);
  return GCRE_OK;
}

void gcre_result_free(gcre_result* r) {
  if (!r) return;
  std::free(r->scores);
  std::free(r->src);
  std::free(r->trg);
  std::free(r->cases);
  std::free(r->ctrls);
  std::free(r->null_max);
  std::memset(r, 0, sizeof *r);
}

int gcre_get_profile(const gcre_ctx* c, gcre_profile* out) {
  if (!c || !out) return GCRE_ERR_ARG;
  *out = c->prof;
  return GCRE_OK;
}

int gcre_resolve_count_locs(const int32_t* trg_uids, int64_t n_uids, const int32_t* keys, const int32_t* counts,
                            const int32_t* locations, int64_t n_keys, int32_t* out_count, int64_t* out_location) {
  if (n_uids < 0 || n_keys < 0 || (n_uids > 0 && (!trg_uids || !out_count || !out_location)) ||
      (n_keys > 0 && (!keys || !counts || !locations)))
    return GCRE_ERR_ARG;
  std::unordered_map<int32_t, std::pair<int32_t, int32_t>> map;   // count_locs, wrapper.cpp:106-112
  map.reserve((size_t)n_keys * 2);
  for (int64_t i = 0; i < n_keys; i++) map[keys[i]] = {counts[i], locations[i]};
  for (int64_t k = 0; k < n_uids; k++) {
    auto it = map.find(trg_uids[k]);   // a missing key reads as {0, 0}, wrapper.cpp:124-126
    out_count[k] = (it == map.end()) ? 0 : it->second.first;
    out_location[k] = (it == map.end()) ? 0 : it->second.second;
  }
  return GCRE_OK;
}

int gcre_process_paths(gcre_ctx* c, const gcre_pp_input* in, gcre_result out[5]) {
  if (!c || !in || !out) return fail(c, GCRE_ERR_ARG, "bad process_paths arguments");
  (void)hipSetDevice(c->device);
  for (int i = 0; i < 5; i++) {
    std::memset(&out[i], 0, sizeof out[i]);
    out[i].n = -1;   // NULL list entry, wrapper.cpp:223
  }
  const int L = in->path_length;
  gcre_profile total{};
  auto add_prof = [&]() {
    total.null_kernel_ms += c->prof.null_kernel_ms;
    total.null_kernel_launches += c->prof.null_kernel_launches;
    total.stats_kernel_ms += c->prof.stats_kernel_ms;
    total.select_ms += c->prof.select_ms;
    total.total_ms += c->prof.total_ms;
    total.paths += c->prof.paths;
    total.scores += c->prof.scores;
    total.null_alg_bytes += c->prof.null_alg_bytes;
    total.null_row_loads += c->prof.null_row_loads;
    total.ie_launches += c->prof.ie_launches;
    total.ie_overlap_lists += c->prof.ie_overlap_lists;
    total.ie_hinted_joins += c->prof.ie_hinted_joins;
    total.ie_plane_joins += c->prof.ie_plane_joins;
    total.ie_lookup_tiles += c->prof.ie_lookup_tiles;
    total.prepare_ms += c->prof.prepare_ms;
    total.inspect_ms += c->prof.inspect_ms;
  };

  int rc = gcre_set_value_table(c, in->value_table, in->vt_rows, in->vt_cols, in->vt_col_major);   // wrapper.cpp:213
  if (rc != GCRE_OK) return rc;
  // a context whose masks were generated on the device (gcre_generate_perm_masks) or uploaded packed keeps them
  // when the caller passes no matrix; otherwise wrapper.cpp:214
  if (!(in->perm_cases == nullptr && in->perm_rows == 0 && c->have_perms))
    rc = gcre_set_perm_cases(c, in->perm_cases, in->perm_rows, c->g.n, in->perm_col_major);
  if (rc != GCRE_OK) return rc;

  gcre_pathset *parsed1 = nullptr, *parsed2 = nullptr, *paths1 = nullptr, *paths2 = nullptr, *paths3 = nullptr;
  std::vector<gcre_pathset*> temps;
  gcre_uids** uids_to_free = nullptr;
  bool multi_window = false, cache_before = c->insp_cache;
  auto cleanup = [&]() {
    if (multi_window) c->insp_cache = cache_before;
    if (uids_to_free)
      for (int i = 0; i < 6; i++) { free_uids(uids_to_free[i]); uids_to_free[i] = nullptr; }
    for (auto* p : temps) gcre_pathset_free(p);
    for (auto* p : {parsed1, parsed2, paths1, paths2, paths3}) gcre_pathset_free(p);
  };
  auto total_paths = [](const gcre_level& lv) {
    int64_t t = 0;
    for (int64_t i = 0; i < lv.n_uids; i++) t += std::max(lv.uid_count[i], 0);
    return t;
  };
  // red/red_index: what the join really adds to paths0 (gcre_uids_set_reduced) -- checked per join, never trusted.
  // The join index of a level is uploaded once and serves every permutation window.
  gcre_uids* level_uids[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uids_to_free = level_uids;
  int pp_window = 0;   // permutations per window, once it is known (below)
  auto join = [&](int plen, int lvi, const gcre_pathset* p0, const gcre_pathset* p1, gcre_pathset* res,
                  gcre_result* o, const gcre_pathset* red = nullptr, const int32_t* red_index = nullptr,
                  int64_t n_red = 0) -> int {
    const gcre_level& lv = in->level[lvi];
    gcre_uids*& u = level_uids[lvi];
    if (!u) {
      u = make_uids(c, plen, lv.uid_count, lv.uid_location, lv.n_uids, lv.signs, lv.n_signs);
      if (!u) return c->last_code;
    }
    if (!u->red && red && red_index && n_red >= p1->nrows) {   // once: the operand outlives the windows
      bool in_range = true;
      for (int64_t i = 0; i < n_red && in_range; i++) in_range = (int64_t)((uint32_t)red_index[i] & 0x7fffffffu) < red->nrows;
      if (in_range && gcre_uids_set_reduced(u, red, red_index, n_red) != GCRE_OK) return c->last_code;
    }
    JoinPlan jp{u, p0, p1, res, false, 0, 0, nullptr};
    if (in->shard_world > 1) {   // one device of several: its slice of the joined paths, every kept row
      jp.sharded = true;
      jp.shard_begin = u->total * in->shard_rank / in->shard_world;
      jp.shard_end = u->total * (in->shard_rank + 1) / in->shard_world;
      if (c->hub && c->win_K > 0) {
        // the devices of this call share their running maxima during the join: one exchange per doubling of a device's
        // work beyond GCRE_EXCHANGE_UNIT path-tiles (as ResidentPlan.exchange_count; the same on every device)
        double unit = 2e6;
        if (const char* e = std::getenv("GCRE_EXCHANGE_UNIT")) unit = std::atof(e);
        const double work = (double)u->total / in->shard_world * (double)((pp_window + kSparseTile - 1) / kSparseTile);
        if (unit > 0 && work >= 2 * unit) {
          // the number of exchanges must be the same on every device BY CONSTRUCTION: a device that cannot get its buffer
          // fails the call (its thread then fails the hub) instead of silently joining with none -- the others would wait
          // for it for ever, or pair their round with its next join
          if (c->d_hub_null.reserve((size_t)c->g.Kpad + 64) != hipSuccess) return fail(c, GCRE_ERR_DEVICE, "no memory for the exchange buffer");
          double most = 8.0;
          if (const char* e = std::getenv("GCRE_EXCHANGE_MAX")) most = std::min(std::max(std::atof(e), 1.0), 32.0);
          jp.exchanges = (int)std::min(most, std::floor(std::log2(work / unit)));
          jp.d_null_out = c->d_hub_null.p;
          jp.exchange_user = c;
          c->hub_level = lvi;          // 0..5 = levels 1a, 1b, 2, 3, 4, 5
          c->hub_round = 0;
          jp.exchange = [](void* user, void* d_null, int32_t k0, int32_t k1) -> int {
            gcre_ctx* cc = (gcre_ctx*)user;
            std::vector<float> v((size_t)std::max(k1 - k0, 0));
            if (v.empty()) return 0;
            const uint64_t tag = ((uint64_t)(uint32_t)cc->hub_level << 48) ^ ((uint64_t)(uint32_t)k0 << 16) ^ (uint64_t)(cc->hub_round++ & 0xffff);
            if (cc->comm) {
              // RCCL: the maxima never leave the devices.  The threads meet first (same level, window and exchange on
              // every device, under the hub's deadline), then each issues the in-place MAX all-reduce on its stream
              if (cc->hub->meet(tag) != 0) return 1;
              const ncclResult_t nr = RcclApi::get().all_reduce(d_null, d_null, v.size(), ncclFloat32, ncclMax, cc->comm, cc->stream);
              cc->rccl_calls++;
              g_rccl_collectives++;
              if (nr != ncclSuccess || hipStreamSynchronize(cc->stream) != hipSuccess) { cc->hub->fail(); return 1; }
              return 0;
            }
            if (hipMemcpy(v.data(), d_null, v.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { cc->hub->fail(); return 1; }
            if (cc->hub->reduce(v, tag) != 0) return 1;
            return hipMemcpy(d_null, v.data(), v.size() * 4, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
          };
        }
      }
    }
    gcre_result tmp;
    int r = run_join(c, jp, &tmp);
    if (r != GCRE_OK) { gcre_result_free(&tmp); return r; }
    if (o && c->comm && c->hub && c->win_K > 0 && tmp.null_max) {   // (the 1a join's result is discarded: nothing to merge)
      // merge_scores across devices (src/methods.h:34-37) over RCCL: element-wise MAX all-reduce of this window's f32 null
      // maxima, in place on the device -- exact for any sharding (f32 max is associative, the values are f32-rounded).
      // Every device then returns the merged maxima; the caller's host-side MAX over devices changes nothing any more.
      const uint64_t tag = ((uint64_t)(uint32_t)lvi << 48) ^ ((uint64_t)(uint32_t)c->win_k0 << 16) ^ 0xffffull;
      if (c->hub->meet(tag) != 0) { gcre_result_free(&tmp); return fail(c, GCRE_ERR_DEVICE, "the devices did not meet for the level's RCCL merge"); }
      float* d = (float*)(c->d_null + c->win_k0);
      const ncclResult_t nr = RcclApi::get().all_reduce(d, d, (size_t)c->win_K, ncclFloat32, ncclMax, c->comm, c->stream);
      c->rccl_calls++;
      g_rccl_collectives++;
      hipError_t he = nr == ncclSuccess ? hipMemcpyAsync(tmp.null_max, d, (size_t)c->win_K * 4, hipMemcpyDeviceToHost, c->stream) : hipErrorUnknown;
      if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
      if (he != hipSuccess) {
        c->hub->fail();
        gcre_result_free(&tmp);
        return fail(c, GCRE_ERR_DEVICE, std::string("RCCL all-reduce of the null maxima failed: ") + (nr != ncclSuccess ? RcclApi::get().error_string(nr) : hipGetErrorString(he)));
      }
    }
    add_prof();
    if (o) *o = tmp; else gcre_result_free(&tmp);
    return GCRE_OK;
  };
#define PP_REQUIRE(ptr)            \
  do {                             \
    if (!(ptr)) { cleanup(); return c->last_code != GCRE_OK ? c->last_code : GCRE_ERR_DEVICE; } \
  } while (0)
#define PP_TRY(expr)               \
  do {                             \
    int r__ = (expr);              \
    if (r__ != GCRE_OK) { cleanup(); return r__; } \
  } while (0)

  const int ncol = c->g.n;
  // Large permutation counts run in windows of whole 2048-permutation tiles: the count planes of the kept sets are per
  // tile, and a window is as many tiles as fit next to the rows.  Scores and top-k lists do not depend on the window (the
  // first window's are returned); the null maxima of the windows are concatenated.
  std::vector<int64_t> set_rows = {in->data1_rows, in->data2_rows};
  for (int lv = 0; lv < 4 && lv <= L; lv++)
    if (lv != 1) set_rows.push_back(total_paths(in->level[lv]));
  const int Kall = c->g.K;
  int win = std::max(1, gcre_plan_perm_window(c, set_rows.data(), (int)set_rows.size()));
  if (in->window_perms > 0) {
    win = in->window_perms >= Kall ? std::max(Kall, 1) : std::max(kSparseTile, in->window_perms / kSparseTile * kSparseTile);
  } else if (Kall > kSparseTile && sparse_enabled(c)) {
    // A context without a pooled plane buffer of the full size (the R shim makes a fresh context per call, as the
    // reference does) has to hipMalloc the planes: ~40 ms per GB here, 3 KB per kept row and tile.  Against ~25 ms of
    // repeated inspector work per extra window, few tiles per window win: w* = sqrt(25 ms * tiles / (ms per tile)).
    // (A set above the recipe limit stores no planes.)
    const int nkt_all = (Kall + kSparseTile - 1) / kSparseTile;
    int64_t biggest = 0;
    for (size_t i = 2; i < set_rows.size(); i++) {
      const double full = (double)set_rows[i] * c->g.method * 3072.0 * nkt_all;
      if (full <= (double)c->planes_out_max) biggest = std::max(biggest, set_rows[i]);
    }
    const double ms_per_tile = (double)biggest * c->g.method * 3072.0 / 1e9 * 40.0;
    const size_t full_bytes = (size_t)biggest * c->g.method * 3072 * (size_t)((win + kSparseTile - 1) / kSparseTile);
    bool pooled = false;
    for (const auto& pb : c->plane_pool) pooled = pooled || pb.bytes >= full_bytes;
    if (ms_per_tile > 1.0 && !pooled) {
      const int w = std::max(1, (int)std::sqrt(25.0 * nkt_all / ms_per_tile));
      win = std::min(win, w * kSparseTile);
    }
  }
  pp_window = std::max(1, std::min(win, std::max(Kall, 1)));
  // The last level's segment table and quads (tens of milliseconds of host work at 2.5 M uids) depend on its join index
  // only: a helper thread builds them while the first levels run (GCRE_PREFETCH_TABLES=0: built when the join asks)
  if (L >= 3 && Kall > 0 && in->shard_world <= 1 && sparse_enabled(c) && (c->null_kernel == 0 || c->null_kernel == 3) &&
      !(std::getenv("GCRE_PREFETCH_TABLES") && std::atoi(std::getenv("GCRE_PREFETCH_TABLES")) == 0)) {
    const gcre_level& lv = in->level[L];
    const int64_t P = total_paths(lv);
    if (lv.n_uids > 100000 && P > 0 && P <= c->chunk_paths && !level_uids[L]) {
      gcre_uids* u = make_uids(c, L, lv.uid_count, lv.uid_location, lv.n_uids, lv.signs, lv.n_signs);
      if (!u) { cleanup(); return c->last_code; }
      level_uids[L] = u;
      u->prefetch.reset(new gcre_uids::Prefetch());
      gcre_uids::Prefetch* pf = u->prefetch.get();
      pf->first = 0; pf->count = P; pf->score_b = 0; pf->score_e = P; pf->plane_b = 0; pf->plane_e = P;
      const int nkt = (pp_window + kSparseTile - 1) / kSparseTile;
      const bool with_quads = c->g.method == 1 && c->ie_quad && c->d_ladder;
      const gcre_ctx* cc = c;
      pf->th = std::thread([u, pf, P, nkt, with_quads, cc]() {
        HostTimer hb("prefetched tables: build (helper thread)");
        build_segments_host(*u, 0, P, 0, P, 0, P, pf->segs, &pf->nscored);
        pf->ready = true;
        if (with_quads && (int64_t)pf->segs.size() < ((int64_t)1 << 30)) {
          pf->q_warm = warm_segments(cc, pf->nscored, nkt);
          build_quads_host(*u, pf->segs, 0, pf->nscored, pf->q_warm, pf->quads, &pf->quad_begin);
          pf->quads_ready = true;
        }
      });
    }
  }
  std::vector<float> null_all[5];
  // several windows: the joins of the 2nd..nth window start at their null kernels (inspection cache; the operands below
  // are made once, so that every window joins the same sets)
  multi_window = Kall > win;
  if (multi_window) c->insp_cache = inspect_cache_allowed();
  gcre_pathset *zero1 = nullptr, *input1 = nullptr, *zero2 = nullptr, *input2 = nullptr, *input_l2 = nullptr, *input_l3 = nullptr;
  for (int k0 = 0; k0 < std::max(Kall, 1); k0 += win) {
    const bool first_window = k0 == 0;
    if (Kall > 0) PP_TRY(gcre_set_perm_window(c, k0, std::min(Kall, k0 + win)));
    gcre_result wout[5];
    for (int i = 0; i < 5; i++) { std::memset(&wout[i], 0, sizeof wout[i]); wout[i].n = -1; }
    gcre_result* const out_w = first_window ? out : wout;
    auto fold = [&]() {   // this window's maxima behind the ones we have; later windows only contribute maxima
      for (int i = 0; i < 5; i++) {
        if (out_w[i].n < 0) continue;
        null_all[i].insert(null_all[i].end(), out_w[i].null_max, out_w[i].null_max + out_w[i].n_perm);
        if (!first_window) gcre_result_free(&out_w[i]);
      }
    };
    (void)fold;
    if (!parsed1) parsed1 = gcre_pathset_from_dense(c, in->data1, in->data1_rows, ncol, in->data_col_major);   // wrapper.cpp:216-217
    PP_REQUIRE(parsed1);

    if (L >= 1) {   // wrapper.cpp:225-244
      if (!c->quiet && first_window) std::printf("Processing Path Length: %d\n", 1);
      if (!paths1) paths1 = new_pathset(c, total_paths(in->level[0]), true);
      PP_REQUIRE(paths1);
      if (!zero1) temps.push_back(zero1 = gcre_pathset_zeros(c, in->n_data_inds[0]));
      PP_REQUIRE(zero1);
      if (!input1) temps.push_back(input1 = gcre_pathset_select(c, parsed1, in->data_inds[0], in->n_data_inds[0]));
      PP_REQUIRE(input1);
      PP_TRY(join(1, 0, zero1, input1, paths1, nullptr, parsed1, in->data_inds[0], in->n_data_inds[0]));   // result discarded, wrapper.cpp:233

      if (!c->quiet && first_window) std::printf("Processing Path Length: %d\n", 1);
      if (!zero2) temps.push_back(zero2 = gcre_pathset_zeros(c, in->n_data_inds[1]));
      PP_REQUIRE(zero2);
      if (!parsed2) parsed2 = gcre_pathset_from_dense(c, in->data2, in->data2_rows, ncol, in->data_col_major);
      PP_REQUIRE(parsed2);
      if (!input2) temps.push_back(input2 = gcre_pathset_select(c, parsed2, in->data_inds[1], in->n_data_inds[1]));
      PP_REQUIRE(input2);
      PP_TRY(join(1, 1, zero2, input2, nullptr, &out_w[0], parsed2, in->data_inds[1], in->n_data_inds[1]));
    }
    if (L >= 2) {   // wrapper.cpp:246-253
      if (!c->quiet && first_window) std::printf("Processing Path Length: %d\n", 2);
      if (!paths2) paths2 = new_pathset(c, total_paths(in->level[2]), true);
      PP_REQUIRE(paths2);
      // the sequence is known here: when level 3's rows are too many to leave with count planes they leave with a recipe
      // whose operand is this set -- which then must not be recipe-only itself (it would be rebuilt from bit lists)
      if (L >= 4 && plane_bytes(c, total_paths(in->level[3]), 2) > c->planes_out_max) paths2->planes_wanted = true;
      if (L >= 5) paths2->planes_wanted = true;   // level 5 adds rows of this set: their planes are read per joined path
      // the reference reads data_idx2 from r_data_inds3 (wrapper.cpp:207); R passes identical vectors
      if (!input_l2) temps.push_back(input_l2 = gcre_pathset_select(c, parsed1, in->data_inds[3], in->n_data_inds[3]));
      gcre_pathset* const input = input_l2;
      PP_REQUIRE(input);
      PP_TRY(join(2, 2, paths1, input, paths2, &out_w[1], parsed1, in->data_inds[3], in->n_data_inds[3]));
    }
    if (L >= 3) {   // wrapper.cpp:255-262
      if (!c->quiet && first_window) std::printf("Processing Path Length: %d\n", 3);
      if (!paths3) paths3 = new_pathset(c, total_paths(in->level[3]), true);
      PP_REQUIRE(paths3);
      if (!input_l3) temps.push_back(input_l3 = gcre_pathset_select(c, parsed1, in->data_inds[3], in->n_data_inds[3]));
      gcre_pathset* const input = input_l3;
      PP_REQUIRE(input);
      PP_TRY(join(3, 3, paths2, input, paths3, &out_w[2], parsed1, in->data_inds[3], in->n_data_inds[3]));
    }
    if (L >= 4) {   // wrapper.cpp:264-269
      if (!c->quiet && first_window) std::printf("Processing Path Length: %d\n", 4);
      // paths2[loc] = (c, d) with c already on paths3[idx]: the join adds gene d = the data row level 2 joined at loc
      // (signed method: level 2 put d into the (-) half when the relation's sign is not 1, gcre.h:71-81 -> bit 31)
      std::vector<int32_t> added((size_t)in->n_data_inds[3]);
      for (int64_t i = 0; i < in->n_data_inds[3]; i++) {
        const bool neg = c->g.method == 2 && i < in->level[2].n_signs && in->level[2].signs[i] != 1;
        added[(size_t)i] = (int32_t)((uint32_t)in->data_inds[3][i] | (neg ? 0x80000000u : 0u));
      }
      PP_TRY(join(4, 4, paths3, paths2, nullptr, &out_w[3], parsed1, added.data(), (int64_t)added.size()));
    }
    if (L >= 5) {   // wrapper.cpp:271-276
      if (!c->quiet && first_window) std::printf("Processing Path Length: %d\n", 5);
      // paths3[loc] = (c, d, e) with c on paths3[idx]: the join adds the 2-gene path (d, e) = paths2[second relation],
      // and the second relation of joined path q of level 3 is the paths1 row it joined: location3[uid] + offset
      std::vector<int32_t> second;
      {
        const gcre_level& l3 = in->level[3];
        for (int64_t i = 0; i < l3.n_uids; i++) {
          // signed method: (d, e) sits in paths3[loc] with both halves swapped when the relation (c, d) is not positive
          const bool neg = c->g.method == 2 && i < l3.n_signs && l3.signs[i] != 1;
          for (int32_t t = 0; t < l3.uid_count[i]; t++)
            second.push_back((int32_t)((uint32_t)(l3.uid_location[i] + t) | (neg ? 0x80000000u : 0u)));
        }
      }
      PP_TRY(join(5, 5, paths3, paths3, nullptr, &out_w[4], paths2, second.data(), (int64_t)second.size()));
    }
    fold();
    if (Kall == 0) break;
  }
  for (auto* p : temps) gcre_pathset_free(p);   // the operand copies (the same rows served every window)
  temps.clear();
  if (multi_window) {
    c->insp_cache = cache_before;
    if (!cache_before) (void)gcre_drop_inspections(c, 1);
  }
  if (Kall > 0) PP_TRY(gcre_set_perm_window(c, 0, Kall));
  for (int i = 0; i < 5; i++) {
    if (out[i].n < 0 || Kall == 0) continue;
    std::free(out[i].null_max);
    out[i].n_perm = Kall;
    out[i].null_max = (float*)std::calloc((size_t)Kall, sizeof(float));
    std::memcpy(out[i].null_max, null_all[i].data(), (size_t)Kall * sizeof(float));
  }
  if (!c->quiet) std::printf("[success]\n");   // wrapper.cpp:278
  cleanup();
  c->prof = total;
  return GCRE_OK;
#undef PP_REQUIRE
#undef PP_TRY
}

// ---- several devices, one process (include/gcre_hip.h) ----
int gcre_process_paths_devices(int method, int n_cases, int n_ctrls, int iterations, int top_k, const int* devices,
                               int n_devices, const gcre_pp_input* in, gcre_result out[5], char* err, size_t errlen) {
  auto say = [&](const std::string& m) {
    if (err && errlen) std::snprintf(err, errlen, "%s", m.c_str());
  };
  if (!in || !out) { say("bad arguments"); return GCRE_ERR_ARG; }
  for (int i = 0; i < 5; i++) { std::memset(&out[i], 0, sizeof out[i]); out[i].n = -1; }
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) { say("no HIP device: libgcre_hip has no CPU fallback"); return GCRE_ERR_DEVICE; }
  std::vector<int> dev;
  if (n_devices <= 0) n_devices = visible;
  for (int i = 0; i < n_devices; i++) dev.push_back(devices ? devices[i] : i % visible);
  const int N = (int)dev.size();
  // contexts first (one per device), so that every device can be asked for its window before anybody starts
  std::vector<gcre_ctx*> ctx((size_t)N, nullptr);
  int rc = GCRE_OK;
  for (int r = 0; r < N && rc == GCRE_OK; r++) {
    ctx[(size_t)r] = gcre_create(method, n_cases, n_ctrls, iterations, dev[(size_t)r]);
    if (!ctx[(size_t)r]) { say(gcre_last_error(nullptr)); rc = GCRE_ERR_DEVICE; break; }
    ctx[(size_t)r]->quiet = ctx[(size_t)r]->quiet || r > 0;   // the reference's progress lines once
    rc = gcre_set_top_k(ctx[(size_t)r], top_k);
    if (rc != GCRE_OK) say(gcre_last_error(ctx[(size_t)r]));
  }
  int window = 0;
  if (rc == GCRE_OK && N > 1 && iterations > 0) {
    // the smallest window any device plans: all of them walk the same windows (their maxima are merged per window)
    std::vector<int64_t> set_rows = {in->data1_rows, in->data2_rows};
    for (int lv = 0; lv < 4 && lv <= in->path_length; lv++)
      if (lv != 1) {
        int64_t t = 0;
        for (int64_t i = 0; i < in->level[lv].n_uids; i++) t += std::max(in->level[lv].uid_count[i], 0);
        set_rows.push_back(t);
      }
    window = iterations;
    for (int r = 0; r < N; r++) {
      (void)hipSetDevice(dev[(size_t)r]);
      window = std::min(window, std::max(1, gcre_plan_perm_window(ctx[(size_t)r], set_rows.data(), (int)set_rows.size())));
    }
  }
  std::vector<std::array<gcre_result, 5>> part((size_t)N);
  std::vector<int> rcs((size_t)N, GCRE_OK);
  ExchangeHub hub;   // where the device threads MAX-merge their running maxima during the large joins
  hub.n = N;
  if (const char* e = std::getenv("GCRE_HUB_TIMEOUT_S")) hub.timeout_s = std::max(0.01, std::atof(e));
  // RCCL communicators, one per device, when every device is listed once (RCCL refuses a device twice: the one-GPU
  // rehearsal stays on the host hub) and the library loads.  GCRE_RCCL=0: never; GCRE_RCCL=force: also for one device
  // (one-rank collectives: what a one-GPU box can exercise of this path).
  std::vector<ncclComm_t> comms;
  {
    const char* mode = std::getenv("GCRE_RCCL");
    const bool force = mode && std::strcmp(mode, "force") == 0;
    bool distinct = true;
    for (int a = 0; a < N; a++)
      for (int b = a + 1; b < N; b++) distinct = distinct && dev[(size_t)a] != dev[(size_t)b];
    if (rc == GCRE_OK && distinct && (N > 1 || force) && iterations > 0 && RcclApi::get().ok) {
      comms.assign((size_t)N, nullptr);
      if (RcclApi::get().comm_init_all(comms.data(), N, dev.data()) != ncclSuccess) {
        comms.clear();   // (no communicator: the host hub serves the call)
        (void)hipGetLastError();
      }
    }
  }
  if (rc == GCRE_OK) {
    auto work = [&](int r) {
      gcre_pp_input mine = *in;
      mine.shard_rank = N > 1 ? r : 0;
      mine.shard_world = N > 1 ? N : 0;
      mine.window_perms = N > 1 ? window : in->window_perms;
      ctx[(size_t)r]->hub = (N > 1 || !comms.empty()) ? &hub : nullptr;
      ctx[(size_t)r]->comm = comms.empty() ? nullptr : comms[(size_t)r];
      rcs[(size_t)r] = gcre_process_paths(ctx[(size_t)r], &mine, part[(size_t)r].data());
      ctx[(size_t)r]->hub = nullptr;
      ctx[(size_t)r]->comm = nullptr;
      if (rcs[(size_t)r] != GCRE_OK) hub.fail();   // nobody waits for a device that has given up
    };
    std::vector<std::thread> th;
    for (int r = 1; r < N; r++) th.emplace_back(work, r);
    work(0);
    for (auto& t : th) t.join();
    for (int r = 0; r < N; r++)
      if (rcs[(size_t)r] != GCRE_OK && rc == GCRE_OK) {
        rc = rcs[(size_t)r];
        say("device " + std::to_string(dev[(size_t)r]) + ": " + gcre_last_error(ctx[(size_t)r]) +
            (hub.timed_out ? " (a device waited longer than GCRE_HUB_TIMEOUT_S for the others at a threshold exchange)" : ""));
      }
  }
  if (rc == GCRE_OK) {
    // merge_scores / format_result (methods.h:25-39, join_base.cpp:138-154) across devices: f32 MAX of the maxima (exact
    // for any sharding: f32 max is associative, every device holds f32-rounded values), best top_k of all tables with the
    // canonical tie rule (smaller joined-path ordinal = smaller (idx, loc)), the sentinel when fewer than top_k exist
    for (int lv = 0; lv < 5; lv++) {
      if (part[0][(size_t)lv].n < 0) continue;
      gcre_result& o = out[lv];
      const int K = part[0][(size_t)lv].n_perm;
      o.n_perm = K;
      o.null_max = (float*)std::calloc((size_t)std::max(K, 1), sizeof(float));
      std::vector<Candidate> cands;
      for (int r = 0; r < N; r++) {
        const gcre_result& p = part[(size_t)r][(size_t)lv];
        for (int k = 0; k < K; k++) o.null_max[k] = std::max(o.null_max[k], p.null_max[k]);
        for (int i = 0; i < p.n; i++)
          if (p.src[i] >= 0) cands.push_back(Candidate{p.scores[i], ((int64_t)p.src[i] << 32) | (uint32_t)p.trg[i], p.src[i], p.trg[i], p.cases[i], p.ctrls[i]});
      }
      std::sort(cands.begin(), cands.end(), [](const Candidate& a, const Candidate& b) {
        if (a.score != b.score) return a.score > b.score;
        return a.path < b.path;
      });
      const size_t keepn = std::min(cands.size(), (size_t)top_k);
      const bool with_sentinel = cands.size() < (size_t)top_k;
      const size_t n_out = keepn + (with_sentinel ? 1 : 0);
      o.n = (int32_t)n_out;
      o.scores = (double*)std::calloc(std::max<size_t>(n_out, 1), sizeof(double));
      o.src = (int32_t*)std::calloc(std::max<size_t>(n_out, 1), sizeof(int32_t));
      o.trg = (int32_t*)std::calloc(std::max<size_t>(n_out, 1), sizeof(int32_t));
      o.cases = (int32_t*)std::calloc(std::max<size_t>(n_out, 1), sizeof(int32_t));
      o.ctrls = (int32_t*)std::calloc(std::max<size_t>(n_out, 1), sizeof(int32_t));
      size_t w = 0;
      if (with_sentinel) {
        o.scores[w] = -std::numeric_limits<double>::infinity();
        o.src[w] = o.trg[w] = -1;
        w++;
      }
      for (size_t i = keepn; i-- > 0;) {
        o.scores[w] = cands[i].score;
        o.src[w] = cands[i].src;
        o.trg[w] = cands[i].trg;
        o.cases[w] = cands[i].cases;
        o.ctrls[w] = cands[i].ctrls;
        w++;
      }
    }
  }
  for (ncclComm_t cm : comms)
    if (cm) (void)RcclApi::get().comm_destroy(cm);
  for (int r = 0; r < N; r++) {
    if (rcs[(size_t)r] == GCRE_OK)
      for (int lv = 0; lv < 5; lv++)
        if (part[(size_t)r][(size_t)lv].n >= 0 || part[(size_t)r][(size_t)lv].null_max) gcre_result_free(&part[(size_t)r][(size_t)lv]);
    if (ctx[(size_t)r]) {
      (void)hipSetDevice(dev[(size_t)r]);
      gcre_destroy(ctx[(size_t)r]);
    }
  }
  return rc;
}

int64_t gcre_rccl_collectives(void) { return g_rccl_collectives.load(); }

int gcre_rccl_selftest(int device, char* err, size_t errlen) {
  auto say = [&](const std::string& m) {
    if (err && errlen) std::snprintf(err, errlen, "%s", m.c_str());
  };
  RcclApi& api = RcclApi::get();
  if (!api.ok) { say("librccl.so not found (or GCRE_RCCL=0)"); return GCRE_ERR_DEVICE; }
  if (hipSetDevice(device) != hipSuccess) { say("no such device"); return GCRE_ERR_DEVICE; }
  ncclComm_t comm = nullptr;
  ncclResult_t nr = api.comm_init_all(&comm, 1, &device);
  if (nr != ncclSuccess) { say(std::string("ncclCommInitAll: ") + api.error_string(nr)); return GCRE_ERR_DEVICE; }
  const size_t n = 4096;
  std::vector<float> h(n), back(n);
  for (size_t i = 0; i < n; i++) h[i] = (float)((i * 2654435761u) % 1000u) / 8.0f;
  float* d = nullptr;
  hipStream_t st = nullptr;
  int rc = GCRE_OK;
  if (hipMalloc((void**)&d, n * 4) != hipSuccess || hipStreamCreate(&st) != hipSuccess) rc = GCRE_ERR_DEVICE;
  if (rc == GCRE_OK && hipMemcpyAsync(d, h.data(), n * 4, hipMemcpyHostToDevice, st) != hipSuccess) rc = GCRE_ERR_DEVICE;
  if (rc == GCRE_OK) {
    nr = api.all_reduce(d, d, n, ncclFloat32, ncclMax, comm, st);   // one rank: MAX over {x} = x
    g_rccl_collectives++;
    if (nr != ncclSuccess) { say(std::string("ncclAllReduce: ") + api.error_string(nr)); rc = GCRE_ERR_DEVICE; }
  }
  if (rc == GCRE_OK && (hipMemcpyAsync(back.data(), d, n * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
    rc = GCRE_ERR_DEVICE;
  if (rc == GCRE_OK && std::memcmp(h.data(), back.data(), n * 4) != 0) { say("one-rank MAX all-reduce changed the data"); rc = GCRE_ERR_DEVICE; }
  if (d) (void)hipFree(d);
  if (st) (void)hipStreamDestroy(st);
  (void)api.comm_destroy(comm);
  return rc;
}

}  // extern "C"
